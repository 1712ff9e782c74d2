#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2l
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_trainer_modes.py tests/test_gpu_round2.py -q -m gpu --timeout 600 -k "bf16 or fp32x6 or data_parallel or chunked" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest.log >> $out/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/summary.txt
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-extras > $out/bench_bf16.json 2> $out/bench_bf16.err
echo "bench bf16 rc=$?" >> $out/summary.txt
# two ranks sharing the one GPU (gloo): the multi-process path of bench.py, weak and strong mode
CLIFT_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_2rank_weak.json 2> $out/bench_2rank_weak.err
echo "bench 2-rank weak rc=$?" >> $out/summary.txt
CLIFT_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --global-rays 8192 --classes 2 --no-cpu-baseline --no-extras > $out/bench_2rank_strong.json 2> $out/bench_2rank_strong.err
echo "bench 2-rank strong rc=$?" >> $out/summary.txt
for r in 1024; do timeout 300 python bench.py --rays $r --inst-rays 1024 --classes 2 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $out/bench_rays$r.json 2> $out/bench_rays$r.err; done
cat $out/summary.txt
