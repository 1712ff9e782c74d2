#!/bin/bash
# the driver's round-end check, as a job: whole GPU suite + smoke + default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-full}
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 --durations=25 > $out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a $out/summary.txt
tail -6 $out/pytest_gpu.log >> $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $out/summary.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
