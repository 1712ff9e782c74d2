"""ctypes binding of libclift.so (C ABI declared in include/clift.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is raised.
``build()`` compiles the library in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLIFT_LIB_PATH") or os.path.join(_HERE, "libclift.so")      # (the override: timing probes of variant builds, tools/jobs)
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 18

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class VM(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("line", C.c_void_p * 3), ("res", C.c_int * 3), ("comps", C.c_int)]


class VMGrad(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("line", C.c_void_p * 3), ("xcd_stride", C.c_long)]


class March(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("inv_ext2", C.c_float * 3), ("step_size", C.c_float),
                ("n_samples", C.c_int), ("distance_scale", C.c_float), ("density_shift", C.c_float),
                ("weight_thres", C.c_float)]


class TVSet(C.Structure):
    _fields_ = [("n", C.c_int), ("plane", C.c_void_p * 8), ("grad", C.c_void_p * 8), ("H", C.c_int * 8), ("W", C.c_int * 8),
                ("C", C.c_int * 8), ("weight", C.c_float * 8)]


class Gemm(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("a_trans", C.c_int),
                ("B", C.c_void_p), ("ldb", C.c_int), ("b_trans", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int),
                ("bias", C.c_void_p), ("act", C.c_int),
                ("mask", C.c_void_p), ("ldmask", C.c_int),
                ("accumulate", C.c_int), ("split_k", C.c_int), ("c_trans", C.c_int), ("colsum", C.c_void_p),
                ("precision", C.c_int), ("workspace", C.c_void_p), ("workspace_bytes", C.c_long),
                ("a_bf16", C.c_int), ("b_bf16", C.c_int), ("c_bf16", C.c_int), ("mask_bf16", C.c_int), ("sign_bits", C.c_void_p)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long
_SIGNATURES = {
    "clift_version": ([], C.c_int),
    "clift_set_cu_reserve": ([_I], C.c_int),
    "clift_segment_loss": ([_P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _I, _P], C.c_int),
    "clift_gemm_workspace_bytes": ([_I, _I], C.c_long),
    "clift_out_layer_fwd": ([_P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P], C.c_int),
    "clift_out_layer_bwd": ([_P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_out_layer_bwd_nh": ([_P, _I, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_last_error": ([], C.c_char_p),
    "clift_gen_rays": ([_I, _I, _P, _P, _F, _P, _P, _P], C.c_int),
    "clift_density_fwd": ([_P, _P, _P, _P, _I, _P, _P], C.c_int),
    "clift_density_points": ([_P, _P, _I, _L, _F, _I, _P, _P], C.c_int),
    "clift_xyz_head_first2_fwd": ([_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_xyz_head_first2_x6_fwd": ([_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_sign_bits_bytes": ([_I], C.c_long),
    "clift_xyz_head_first2_x6_bwd": ([_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_xyz_head_first2_x6_wgrad": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_xyz_head_last2_x6_workspace_bytes": ([_I], C.c_long),
    "clift_xyz_head_last2_x6_fwd": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _L, _P], C.c_int),
    "clift_xyz_head_first2_bf16_bwd": ([_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_xyz_head_first2_bwd": ([_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_xyz_head_first2_wgrad": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_xyz_head_last2_fwd": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_app_head_last2_fwd": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P], C.c_int),
    "clift_xyz_head_bf16_fwd": ([_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P], C.c_int),
    "clift_bind_rows_limit": ([_P], C.c_int),
    "clift_bind_grad_shards": ([_P], C.c_int),
    "clift_grad_shards_begin": ([_P, _P, _P, C.c_long, _P], C.c_int),
    "clift_grad_shards_fold": ([_P, C.c_long, C.c_long, _I, _P], C.c_int),
    "clift_scan_counts_capped": ([_P, _I, _P, _I, _P, _P, _P], C.c_int),
    "clift_compact_fill_capped": ([_P, _P, _I, _I, _F, _P, _I, _P], C.c_int),
    "clift_alpha_bbox": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P], C.c_int),
    "clift_vm_products_points": ([_P, _P, _I, _L, _P, _P], C.c_int),
    "clift_app_encode_points": ([_P, _I, _I, _I, _I, _P, _I, _L, _P, _I, _P], C.c_int),
    "clift_march_fwd": ([_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "clift_march_bwd": ([_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "clift_density_bwd": ([_P, _P, _P, _P, _P, _I, _P, _P, _P], C.c_int),
    "clift_xcd_reduce": ([_P, _L, _L, _P, _P], C.c_int),
    "clift_scan_counts": ([_P, _I, _P, _P], C.c_int),
    "clift_compact_fill": ([_P, _P, _I, _I, _F, _P, _P], C.c_int),
    "clift_app_gather_fwd": ([_P, _P, _P, _P, _P, _I, _P, _P, _P], C.c_int),
    "clift_active_xyz": ([_P, _P, _P, _P, _I, _P, _P], C.c_int),
    "clift_app_gather_bwd": ([_P, _P, _P, _P, _P, _P, _I, _P, _P, _P], C.c_int),
    "clift_app_encode_fwd": ([_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P], C.c_int),
    "clift_app_encode_bwd": ([_P, _I, _I, _I, _P, _I, _I, _P, _I, _P], C.c_int),
    "clift_app_front_fwd": ([_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_app_front_fwd_x": ([_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_app_head_last2_bf16_fwd": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P], C.c_int),
    "clift_app_head_last2_x6_fwd": ([_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P], C.c_int),
    "clift_out_layer_bwd_n128_bf16": ([_P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P], C.c_int),
    "clift_gemm": ([_P, _P], C.c_int),
    "clift_linear_k3_fwd": ([_P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P], C.c_int),
    "clift_linear_k3_bwd": ([_P, _P, _I, _I, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_wgrad_narrow": ([_P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P], C.c_int),
    "clift_colsum": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "clift_rows_act_fwd": ([_P, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "clift_rows_act_bwd": ([_P, _I, _P, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "clift_composite_fwd": ([_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P], C.c_int),
    "clift_composite_bwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                             _P, _P, _P], C.c_int),
    "clift_composite_bwd_act": ([_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _P, _P],
                                C.c_int),
    "clift_tv_fwd_bwd": ([_P, _I, _I, _I, _F, _P, _P, _P], C.c_int),
    "clift_tv_fwd_bwd_multi": ([_P, _P, _P], C.c_int),
    "clift_pixel_losses": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P], C.c_int),
    "clift_pixel_losses_sce": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P], C.c_int),
    "clift_semantic_loss_rows": ([_P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P], C.c_int),
    "clift_contrastive": ([_P, _P, _I, _I, _F, _P, _P, _P, _P], C.c_int),
    "clift_slow_fast": ([_P, _P, _P, _I, _I, _P, _P, _P, _P], C.c_int),
    "clift_nearest_centroid": ([_P, _I, _I, _P, _I, _P, _L, _P, _P], C.c_int),
    "clift_adam": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P], C.c_int),
    "clift_ema": ([_P, _P, _L, _F, _P], C.c_int),
}

_lib = None


class CliftError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> libclift.so for gfx950 (in-tree, so the .so travels with the repo snapshot)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise CliftError("building libclift.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def load():
    """Load the library and attach signatures.  Raises CliftError if it is missing (no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CliftError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(or `make -C {CSRC}`).  There is no CPU/PyTorch fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.clift_version() != ABI_VERSION:
        raise CliftError(f"libclift.so ABI {lib.clift_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES)


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


_launch_stream = None      # side-stream override (engine.Branches); torch's current stream keeps owning all allocations


def set_launch_stream(s):
    global _launch_stream
    _launch_stream = s


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the raw handle of torch's current stream without building a Stream object


def stream():
    """hipStream_t of the stream launches go to: the side-stream override, else torch's current stream of the current device.
    (torch.cuda.current_stream() costs ~9 us a call through its device-index helpers -- 0.5 ms per training step at ~60 launches;
    the raw accessor is ~0.3 us.)"""
    if _launch_stream is not None:
        return C.c_void_p(_launch_stream.cuda_stream)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CliftError(f"{name} failed (rc={rc}): {lib.clift_last_error().decode()}")


def f32(t, what="tensor"):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise CliftError(f"{what}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")
    return t
