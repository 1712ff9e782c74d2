"""CPU-only checks of the drop-in boundary: libclift.so loads (hipcc cross-compiled, no GPU needed to load it) and
exports every symbol include/clift.h declares; the ctypes table covers exactly that set; no compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import REPO

HEADER = os.path.join(REPO, "include", "clift.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clift_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("clift_gen_rays", "clift_density_fwd", "clift_march_fwd", "clift_march_bwd", "clift_density_bwd", "clift_gemm",
                 "clift_composite_fwd", "clift_composite_bwd", "clift_contrastive", "clift_slow_fast", "clift_tv_fwd_bwd",
                 "clift_adam", "clift_ema", "clift_last_error", "clift_version"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from contrastive_lift_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/clift.h but not exported by libclift.so"
    assert sorted(_lib.exported_symbols()) == declared_symbols()
    assert _lib.load().clift_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from contrastive_lift_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CliftError):
        _lib.load()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under contrastive_lift_amd/ may import it."""
    pkg = os.path.join(REPO, "contrastive_lift_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"


def test_in_flight_register_guard():
    """tools/isa_asm_load_check.py (run by csrc/Makefile on every object's device listing; the build fails on a violation): the checker itself
    flags a consumer scheduled above the wait that publishes a hand-issued load, and every listing of the current build is clean."""
    import glob
    import subprocess
    import sys
    import tempfile
    tool = os.path.join(REPO, "tools", "isa_asm_load_check.py")
    bad = "\n".join(["_Z1kv:", "\t;;#ASMSTART", "\tds_read_b128 v[4:7], v1", "\t;;#ASMEND", "\tv_fmac_f32_e32 v9, v2, v5",
                     "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(0)", "\t;;#ASMEND", "\tv_fmac_f32_e32 v9, v2, v6", ""])
    good = bad.replace("\tv_fmac_f32_e32 v9, v2, v5\n", "")
    with tempfile.TemporaryDirectory() as d:
        for name, text, rc in (("bad.s", bad, 1), ("good.s", good, 0)):
            p = os.path.join(d, name)
            open(p, "w").write(text)
            r = subprocess.run([sys.executable, tool, p], capture_output=True, text=True)
            assert r.returncode == rc, (name, r.stdout)
    for s in glob.glob(os.path.join(REPO, "contrastive_lift_amd", "csrc", ".isa", "*.s")):
        r = subprocess.run([sys.executable, tool, s], capture_output=True, text=True)
        assert r.returncode == 0, (s, r.stdout[-2000:])


def test_counted_wait_guard():
    """The second check of tools/isa_asm_load_check.py (round 6): an LDS-DMA has no destination register, so a hand-counted ``s_waitcnt
    vmcnt(N)`` that is one too large is invisible to the register guard.  The kernels mark the DMA (``; @dma tag``) and the first read of the
    slot it fills (``; @use tag allow``); the build replays every marked loop's vector-memory stream in order.  Here: a synthetic loop with
    the right and a wrong count, and the REAL thing -- csrc/layer_x6.hip compiled with its hand-counted table entry X8_VM[0][0] = 6 instead
    of 5 must fail the check (the listing alone: a few seconds, no link)."""
    import shutil
    import subprocess
    import sys
    import tempfile
    tool = os.path.join(REPO, "tools", "isa_asm_load_check.py")

    def loop(n):            # per trip: wait, use of the slot DMA'd in the previous trip, then its refill and two stores younger than it
        return "\n".join(["_Z1kv:", ".LBB0_1:", f"\ts_waitcnt vmcnt({n})", "\t; @use slot 0", "\tds_read_b128 v[4:7], v1", "\t; @dma slot",
                          "\tglobal_load_lds_dwordx4 v[2:3], off", "\tglobal_store_dwordx4 v[2:3], v[8:11], off", "\tglobal_store_dwordx4 v[2:3], v[8:11], off",
                          "\ts_cbranch_scc1 .LBB0_1", ".Lfunc_end0:", ""])
    with tempfile.TemporaryDirectory() as d:
        for name, text, rc in (("good.s", loop(2), 0), ("bad.s", loop(3), 1), ("orphan.s", loop(2).replace("global_load_lds_dwordx4 v[2:3], off", "global_load_dwordx4 v[12:15], v[2:3], off"), 1)):
            p = os.path.join(d, name)
            open(p, "w").write(text)
            r = subprocess.run([sys.executable, tool, p], capture_output=True, text=True)
            assert r.returncode == rc, (name, r.stdout)
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            return
        csrc = os.path.join(REPO, "contrastive_lift_amd", "csrc")
        for define, rc in (("-DX6_VM00=6", 1), ("-DX6_VM00=5", 0)):
            out = os.path.join(d, "x6.s")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-fno-slp-vectorize", define,
                                "-I" + csrc, "-I" + os.path.join(REPO, "include"), "-S", "--cuda-device-only", os.path.join(csrc, "layer_x6.hip"), "-o", out],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            r = subprocess.run([sys.executable, tool, out], capture_output=True, text=True)
            assert r.returncode == rc, (define, r.stdout[-1500:])
            if rc:
                assert "@use piece0" in r.stdout


def test_head_acts_keep_sign_bytes_with_their_activation():
    """engine.HeadActs: the sign bytes of a persistent fp32x6 forward travel with the activation list, keyed by the activation's index,
    and are refused when they do not belong to it (wrong row count) -- host logic, no GPU."""
    import pytest
    import torch
    from contrastive_lift_amd import _lib, engine
    M = 100
    acts = engine.HeadActs([None, torch.zeros(M, 256)])
    assert acts.sign_bits_of(1, M) is None and acts[1].shape == (M, 256) and len(acts) == 2
    acts.signs[1] = torch.zeros((M + 31) // 32 * 1024, dtype=torch.uint8)
    assert acts.sign_bits_of(1, M) is acts.signs[1]
    with pytest.raises(_lib.CliftError, match="same forward"):
        acts.sign_bits_of(1, M + 64)
    acts.append(torch.zeros(M, 256))
    assert acts.sign_bits_of(2, M) is None
