#!/bin/bash
# Counter passes (one counter set per pass, with --kernel-trace only) over the bench command itself: HBM bytes of every kernel of the step.
# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950: FETCH_SIZE x 2 for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-pmc_bench}
mkdir -p $out
export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}.log" 2>&1 )
  echo "pmc bench $set rc=$?" >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_bench_* 2>/dev/null | grep -v "rocclr\|at::native" > $out/pmc_bench_table.txt
rm -rf $out/pmc_bench_FETCH_SIZE $out/pmc_bench_WRITE_SIZE
cat $out/summary.txt; cat $out/pmc_bench_table.txt
