#!/bin/bash
# Timing probes of the appearance front end: variants of libclift.so whose k_app_front_fwd leaves phases out (AF_ABL bit mask in csrc/heads_io.hip;
# results garbage by construction) into tools/_scratch/abl/, timed by tools/app_probe.py.
#   bash tools/jobs/app_probe.sh build   (here: hipcc cross-compiles)      bash tools/jobs/app_probe.sh run   (on the GPU box)
cd "$(dirname "$0")/../.." || exit 1
C=contrastive_lift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function"
VARIANTS=${VARIANTS:-"1 2 4 8 15"}
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null || exit 1
  mkdir -p tools/_scratch/abl
  rm -f tools/_scratch/abl/*
  for v in $VARIANTS; do
    ( /opt/rocm/bin/hipcc $FLAGS -D${ABLMACRO:-AF_ABL}=$v -I$C -Iinclude -c $C/heads_io.hip -o tools/_scratch/abl/heads_io_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v heads_io.o) tools/_scratch/abl/heads_io_$v.o -o tools/_scratch/abl/libclift_af$v.so ) &
  done
  wait
  rm -f tools/_scratch/abl/*.o
  ls -la tools/_scratch/abl
else
  for rep in 1 2; do
    timeout 120 python tools/app_probe.py
    for v in $VARIANTS; do echo "AF_ABL=$v"; timeout 120 python tools/app_probe.py tools/_scratch/abl/libclift_af$v.so | head -1; done
  done
fi
