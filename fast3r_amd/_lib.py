"""ctypes binding of libf3r_hip.so (the C ABI declared in include/f3r.h).

This is the drop-in boundary: the Python host hands raw device pointers (torch tensors are only the
allocator) to hand-written HIP kernels.  There is NO fallback: if the shared library is missing or a
kernel reports an error, the call raises -- the product path never silently computes with torch ops.
"""
import ctypes
import os

import torch

F3R_F16, F3R_BF16 = 0, 1
F3R_A_PLAIN, F3R_A_CONV3X3 = 0, 1
F3R_EPI_GENERIC, F3R_EPI_QKV, F3R_EPI_CONVT = 0, 1, 2
F3R_ACT_NONE, F3R_ACT_GELU, F3R_ACT_RELU = 0, 1, 2
F3R_SPLIT_NONE, F3R_SPLIT_W2, F3R_SPLIT_X3, F3R_SPLIT_W2F8, F3R_SPLIT_X3F8 = 0, 1, 2, 3, 4
F3R_MAX_SEG = 8

_c_i64, _c_i32, _c_f32, _c_vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p


class GemmArgs(ctypes.Structure):
    """struct f3r_gemm_args (include/f3r.h) -- field order and types must match exactly."""
    _fields_ = [
        ("A", _c_vp), ("W", _c_vp), ("bias", _c_vp),
        ("M", _c_i64), ("N", _c_i32), ("K", _c_i32), ("Kpad", _c_i32), ("lda", _c_i64),
        ("a_mode", _c_i32), ("a_relu", _c_i32),
        ("conv_H", _c_i32), ("conv_W", _c_i32), ("conv_C", _c_i32), ("conv_stride", _c_i32),
        ("conv_OH", _c_i32), ("conv_OW", _c_i32),
        ("epi", _c_i32), ("act", _c_i32),
        ("rowadd", _c_vp), ("rowadd_div", _c_i64),
        ("res_f32", _c_vp), ("ldr_f32", _c_i64),
        ("res_lp", _c_vp), ("ldr_lp", _c_i64),
        ("res_lp2", _c_vp), ("ldr_lp2", _c_i64),
        ("out_f32", _c_vp), ("ldo_f32", _c_i64),
        ("out_lp", _c_vp), ("ldo_lp", _c_i64),
        ("q", _c_vp), ("k", _c_vp), ("vt", _c_vp), ("seq_len", _c_i64), ("ldvt", _c_i64),
        ("rope_cos", _c_vp), ("rope_sin", _c_vp), ("rope_w", _c_i32), ("q_scale", _c_f32),
        ("ct_s", _c_i32), ("ct_h", _c_i32), ("ct_w", _c_i32), ("ct_cout", _c_i32),
        ("dtype", _c_i32), ("rope_mode", _c_i32),
        ("split", _c_i32), ("kernel_sel", _c_i32), ("A_lo", _c_vp),
        ("out_lp_lo", _c_vp), ("res_lp_lo", _c_vp), ("res_lp2_lo", _c_vp), ("out_relu", _c_vp), ("out_relu_lo", _c_vp),
        ("qkv_dq", _c_i32), ("out_lp_f8", _c_i32),
        ("w_scale", _c_vp), ("W_aux", _c_vp),
        ("out_f8", _c_vp), ("out_relu_f8", _c_vp),
        ("fin_w", _c_vp), ("fin_b", _c_vp), ("fin_pts", _c_vp), ("fin_conf", _c_vp),
        ("fin_n_out", _c_i32), ("fin_depth_mode", _c_i32), ("fin_conf_mode", _c_i32), ("fin_vmin", _c_f32), ("fin_vmax", _c_f32), ("reserved0", _c_i32),
    ]


class AttnArgs(ctypes.Structure):
    """struct f3r_attn_args (include/f3r.h)."""
    _fields_ = [
        ("q", _c_vp), ("o", _c_vp),
        ("ldq", _c_i64), ("ldo", _c_i64),
        ("q_batch_stride", _c_i64), ("o_batch_stride", _c_i64),
        ("tq", _c_i64),
        ("batch", _c_i32), ("n_heads", _c_i32), ("n_seg", _c_i32), ("dtype", _c_i32),
        ("k_seg", _c_vp * F3R_MAX_SEG), ("vt_seg", _c_vp * F3R_MAX_SEG),
        ("seg_len", _c_i64 * F3R_MAX_SEG), ("ldvt", _c_i64 * F3R_MAX_SEG),
        ("ldk", _c_i64),
        ("k_batch_stride", _c_i64 * F3R_MAX_SEG), ("vt_batch_stride", _c_i64 * F3R_MAX_SEG),
        ("scale", _c_f32), ("q_prescaled", _c_i32),
        ("st_o", _c_vp), ("st_ml", _c_vp), ("state_in", _c_i32), ("state_out", _c_i32),
        ("kv_group", _c_i32), ("causal", _c_i32), ("q_pos0", _c_i64), ("seg_pos0", _c_i64 * F3R_MAX_SEG),
        ("kernel_sel", _c_i32), ("head_dim", _c_i32),
        ("dbg_counters", _c_vp),
        ("sched_counter", _c_vp),
        ("qk_planes", _c_i32), ("reserve_cus", _c_i32),
    ]


class AttnF32Args(ctypes.Structure):
    """struct f3r_attn_f32_args (include/f3r.h)."""
    _fields_ = [
        ("q", _c_vp), ("k", _c_vp), ("v", _c_vp), ("ldq", _c_i64), ("ldkv", _c_i64),
        ("o_hi", _c_vp), ("o_lo", _c_vp), ("o_f32", _c_vp), ("ldo", _c_i64),
        ("n_seq", _c_i64), ("tq", _c_i64), ("tk", _c_i64), ("q_pos0", _c_i64), ("k_pos0", _c_i64),
        ("n_heads", _c_i32), ("kv_group", _c_i32), ("causal", _c_i32), ("dtype", _c_i32), ("head_dim", _c_i32), ("scale", _c_f32),
    ]


# every symbol include/f3r.h declares: (name, restype, argtypes)
SYMBOLS = {
    "f3r_version": (ctypes.c_int, []),
    "f3r_last_error_string": (ctypes.c_char_p, []),
    "f3r_sizeof": (ctypes.c_size_t, [ctypes.c_int]),
    "f3r_wall_clock_khz": (ctypes.c_int, []),
    "f3r_patchify": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_interp_bilinear": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp] + [ctypes.c_int] * 9 + [_c_vp]),
    "f3r_interp_bilinear_f8": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp] + [ctypes.c_int] * 9 + [_c_vp]),
    "f3r_layernorm": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, _c_f32, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_layernorm_f8": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_f32, ctypes.c_int, _c_vp]),
    "f3r_gemm": (ctypes.c_int, [ctypes.POINTER(GemmArgs), _c_vp]),
    "f3r_attn_fwd": (ctypes.c_int, [ctypes.POINTER(AttnArgs), _c_vp]),
    "f3r_attn_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(AttnArgs)]),
    "f3r_block_workspace_bytes": (ctypes.c_size_t, [_c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.POINTER(ctypes.c_size_t)]),
    "f3r_block_workspace_bytes_ex": (ctypes.c_size_t, [_c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "f3r_upsample2x": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_dpt_final": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, _c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     _c_f32, _c_f32, ctypes.c_int, _c_vp]),
    "f3r_cast_f32_to_lp": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, _c_vp]),
    "f3r_align_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "f3r_align_local_to_global": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_size_t,
                                                 ctypes.c_int, _c_i64, _c_f32, _c_vp]),
    "f3r_focal_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "f3r_estimate_focal": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          _c_f32, _c_f32, _c_f32, ctypes.c_int, _c_f32, _c_f32, _c_vp]),
    "f3r_resample_u8": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int, _c_vp]),
    "f3r_imgnorm_u8": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_silu_mul": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_rows_add_f32": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_int, _c_vp]),
    "f3r_rope2d_f32": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_vp]),
    "f3r_attn_f32": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, ctypes.c_int, _c_f32, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_rope_f32": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, _c_vp]),
    "f3r_silu_mul_f32": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_attn_f32_ex": (ctypes.c_int, [ctypes.POINTER(AttnF32Args), _c_vp]),
    "f3r_attn_f32_mfma_workspace": (ctypes.c_int64, [ctypes.POINTER(AttnF32Args)]),
    "f3r_attn_f32_mfma": (ctypes.c_int, [ctypes.POINTER(AttnF32Args), _c_vp, ctypes.c_int64, _c_vp]),
    "f3r_qkv_planes": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, _c_i64, ctypes.c_int, ctypes.c_int, _c_f32, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int, _c_vp]),
    "f3r_attn_state_finish": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, _c_vp]),
    "f3r_estimate_poses": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f32, _c_f32,
                                          _c_f32, ctypes.c_int, ctypes.c_int, _c_vp]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libf3r_hip.so")
if os.environ.get("F3R_LAB_LIB"):  # measurement builds only (tools/lab: kernels with ablation bits); never set by the product or the tests
    LIB_PATH = os.environ["F3R_LAB_LIB"]
_lib = None


class F3RError(RuntimeError):
    pass


ABI_VERSION = 350  # f3r_version() of include/f3r.h this file mirrors


def lib():
    """Load (once) and return the shared library.  Raises loudly if it is missing: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise F3RError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(fast3r_amd has no CPU / torch fallback)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        if l.f3r_sizeof(0) != ctypes.sizeof(GemmArgs) or l.f3r_sizeof(1) != ctypes.sizeof(AttnArgs) or l.f3r_sizeof(2) != ctypes.sizeof(AttnF32Args):
            raise F3RError("fast3r_amd/_lib.py struct layout does not match include/f3r.h "
                           f"(gemm {l.f3r_sizeof(0)} vs {ctypes.sizeof(GemmArgs)}, attn {l.f3r_sizeof(1)} vs {ctypes.sizeof(AttnArgs)}, "
                           f"attn_f32 {l.f3r_sizeof(2)} vs {ctypes.sizeof(AttnF32Args)}): rebuild the library (fast3r_amd/csrc/build.sh)")
        if l.f3r_version() < ABI_VERSION:
            raise F3RError(f"{LIB_PATH} is version {l.f3r_version()}, this host code needs >= {ABI_VERSION}: rebuild it (fast3r_amd/csrc/build.sh)")
        _lib = l
    return _lib


def check(status: int, what: str = ""):
    """Map a negative f3r_status onto a Python exception (ValueError for argument errors, like the reference's
    asserts / ValueErrors on bad shapes; RuntimeError for launch failures)."""
    if status == 0:
        return
    msg = lib().f3r_last_error_string().decode(errors="replace")
    if status in (-1, -2):
        raise ValueError(f"{what}: {msg} (f3r_status {status})")
    raise F3RError(f"{what}: {msg} (f3r_status {status})")


def dtype_id(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return F3R_F16
    if dt == torch.bfloat16:
        return F3R_BF16
    raise ValueError(f"fast3r_amd: MFMA operand dtype must be torch.float16 or torch.bfloat16, got {dt}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(t: torch.Tensor, name="tensor"):
    if not t.is_cuda:
        raise F3RError(f"fast3r_amd: {name} lives on {t.device}; the HIP kernels need a ROCm device "
                       "(there is no CPU fallback -- the CPU path is oracle/, test infrastructure only)")


def work_device(t: torch.Tensor, what="tensor"):
    """Where the kernels run for `t`: its own ROCm device, or -- for a CPU tensor, e.g. the preds `inference()` hands back after its
    `to_cpu` (inference_multiview.py:92) -- the current ROCm device (the caller uploads, runs the kernels, and returns results on
    `t.device`).  Still no CPU *compute* path: without a GPU this raises."""
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise F3RError(f"fast3r_amd: {what} lives on {t.device} and no ROCm device is available; the HIP kernels need one "
                       "(there is no CPU fallback -- the CPU path is oracle/, test infrastructure only)")
    return torch.device("cuda", torch.cuda.current_device())
