#!/bin/bash
# round 5: the ORIGINAL report's layout on the final library: two processes sharing the device, every render bit-compared with the process's own first
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_two_proc}
mkdir -p $out
export TMPDIR=/tmp
for mode in fp32 fp32x6; do
  n=$([ $mode = fp32 ] && echo ${2:-3000} || echo ${3:-2000})
  timeout 1500 python tools/determinism_soak.py A $mode $n 0 > $out/two_A_$mode.log 2>&1 &
  pa=$!
  timeout 1500 python tools/determinism_soak.py B $mode $n 0 > $out/two_B_$mode.log 2>&1 &
  pb=$!
  wait $pa; wait $pb
  echo "two processes, $mode:"; grep -E "RESULT|iter" $out/two_A_$mode.log $out/two_B_$mode.log | tail -8 | cut -c1-400
done
