#!/bin/bash
# Timing probes of the eight-wave fp32x6 layer kernel: builds variants of libclift.so whose k_layer_x6 leaves work out (X6_ABL bit mask in
# csrc/layer_x6.hip; results garbage by construction) into tools/_scratch/, for tools/x6_ablation.py to time on the GPU box.
#   bash tools/jobs/x6_ablation.sh build      (here: hipcc cross-compiles)        bash tools/jobs/x6_ablation.sh run   (on the GPU box)
cd "$(dirname "$0")/.." || exit 1
C=contrastive_lift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize"
VARIANTS=${VARIANTS:-"0 1 2 8 16 32 3"}      # (variants with bit 4 are meaningless: without the exchange the MFMAs are dead code)
STAGGER=0
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null || exit 1
  mkdir -p tools/_scratch/abl
  rm -f tools/_scratch/abl/*
  for sg in $STAGGER; do for v in $VARIANTS; do
    ( /opt/rocm/bin/hipcc $FLAGS -DX6_ABL=$v -I$C -Iinclude -c $C/layer_x6.hip -o tools/_scratch/abl/layer_x6_${v}_$sg.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v layer_x6.o) tools/_scratch/abl/layer_x6_${v}_$sg.o -o tools/_scratch/abl/libclift_abl${v}_s$sg.so ) &
  done; done
  wait
  rm -f tools/_scratch/abl/*.o
  ls -la tools/_scratch/abl
elif [ "$1" = fixed ]; then
  for v in 0 $VARIANTS; do timeout 200 python tools/x6_fixed_probe.py $v; done
else
  for rep in 1 2; do for v in $VARIANTS; do for sg in $STAGGER; do
    timeout 120 python tools/x6_ablation.py $v $sg
  done; done; done
fi
