#!/bin/bash
# round 6: clift_out_layer_fwd -- its tests, then its time at the bench step's row count and at a frame chunk's, for the build and for any probe
# library tools/_scratch/abl/libclift_of<tag>.so (the ablations of docs/history/round6.md section 3 were built from a -DOF_ABL=<bits> form of the
# two-barrier kernel: 1 = DMA re-reads one tile, 2 = no MFMAs, 4 = no slice reads / softmax, 8 = no stores)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu -k "out_layer" 2>&1 | tail -2
for v in base $(ls tools/_scratch/abl/ 2>/dev/null | sed -n 's/^libclift_of\([0-9]*\)\.so$/\1/p' | sort -n); do
lib=""; [ $v != base ] && lib="$GRAFT_REPO_ROOT/tools/_scratch/abl/libclift_of$v.so"
CLIFT_LIB_PATH=$lib python - "$v" <<'PY'
import sys
import torch
from contrastive_lift_amd import engine
res = []
for M in (265000, 5500000):
    H = torch.relu(torch.randn(M, 256, device="cuda")); W = torch.randn(22, 256, device="cuda") / 8; b = torch.randn(22, device="cuda")
    out = torch.empty(M, 22, device="cuda")
    for _ in range(5):
        engine.out_layer_fwd(M, H, W, b, out, 22, 0, 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        engine.out_layer_fwd(M, H, W, b, out, 22, 0, 2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    res.append(f"M {M}: {us:.1f} us, {M * 1112e-6 / us:.2f} TB/s")
print(f"library {sys.argv[1]}: " + "  |  ".join(res))
PY
done
