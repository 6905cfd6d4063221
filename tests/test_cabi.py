"""The C-ABI library: loads, exports every symbol include/f3r.h declares, struct layouts agree, and argument
validation returns error codes (no kernel is launched in this file -> runs without a GPU)."""
import ctypes
import os
import re

import pytest

from fast3r_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "f3r.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(f3r_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(built_lib):
    names = declared_functions()
    assert "f3r_gemm" in names and "f3r_attn_fwd" in names and len(names) >= 10
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/f3r.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in fast3r_amd/_lib.py"
    assert built_lib.f3r_version() >= _lib.ABI_VERSION == 350


def test_struct_layouts_match(built_lib):
    assert built_lib.f3r_sizeof(0) == ctypes.sizeof(_lib.GemmArgs)
    assert built_lib.f3r_sizeof(1) == ctypes.sizeof(_lib.AttnArgs)
    assert built_lib.f3r_sizeof(99) == 0


def test_argument_errors_are_codes_not_crashes(built_lib):
    assert built_lib.f3r_gemm(None, None) == -1
    assert b"null args" in built_lib.f3r_last_error_string()
    g = _lib.GemmArgs()
    g.A, g.W = 0x1000, 0x2000
    g.M, g.N, g.K, g.Kpad, g.lda = 128, 128, 64, 100, 64  # Kpad not a multiple of 64
    assert built_lib.f3r_gemm(ctypes.byref(g), None) == -1
    assert b"Kpad" in built_lib.f3r_last_error_string()
    g.Kpad, g.N = 64, 130  # N % 4 != 0
    assert built_lib.f3r_gemm(ctypes.byref(g), None) == -1
    g.N, g.out_lp, g.ldo_lp = 128, 0x3000, 1 << 26  # the epilogues address rows with 32-bit byte offsets: strides stay below 2^26 elements
    assert built_lib.f3r_gemm(ctypes.byref(g), None) == -1
    assert b"row strides" in built_lib.f3r_last_error_string()
    # ABI 350: fp8 correction planes / fp8 output planes / the fused DPT tail are argument-checked before anything is launched
    c = _lib.GemmArgs()
    c.A, c.W, c.A_lo, c.out_lp = 0x1000, 0x2000, 0x3000, 0x4000
    c.M, c.N, c.Kpad, c.ldo_lp = 64 * 64, 128, 2 * 9 * 128, 128
    c.a_mode, c.conv_H, c.conv_W, c.conv_C, c.conv_stride, c.conv_OH, c.conv_OW = _lib.F3R_A_CONV3X3, 64, 64, 128, 1, 64, 64
    c.split = _lib.F3R_SPLIT_X3F8
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -1 and b"w_scale" in built_lib.f3r_last_error_string()   # no scale words
    c.w_scale, c.conv_C, c.Kpad = 0x5000, 64, 2 * 9 * 64
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -1 and b"128" in built_lib.f3r_last_error_string()       # Cin % 128
    c.conv_C, c.Kpad, c.dtype = 128, 2 * 9 * 128, _lib.F3R_BF16
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -1                                                       # fp16 planes only
    c.dtype, c.split, c.A_lo, c.w_scale, c.Kpad = _lib.F3R_F16, _lib.F3R_SPLIT_NONE, None, None, 9 * 128
    c.fin_w, c.fin_b, c.fin_pts, c.fin_n_out = 0x6000, 0x7000, 0x8000, 5
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -1 and b"fin_n_out" in built_lib.f3r_last_error_string()
    c.fin_n_out, c.out_f8 = 4, 0x9000
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -1 and b"fin_pts / fin_conf only" in built_lib.f3r_last_error_string()
    c.out_f8, c.N, c.ldo_lp = None, 256, 256
    assert built_lib.f3r_gemm(ctypes.byref(c), None) == -2 and b"not eligible" in built_lib.f3r_last_error_string()  # N != 128: no kernel takes it
    q = _lib.GemmArgs()
    q.A, q.W, q.M, q.N, q.K, q.Kpad, q.lda, q.epi, q.out_f8 = 0x1000, 0x2000, 128, 192, 64, 64, 64, _lib.F3R_EPI_QKV, 0x9000
    assert built_lib.f3r_gemm(ctypes.byref(q), None) == -1 and b"generic epilogue" in built_lib.f3r_last_error_string()
    a = _lib.AttnArgs()
    a.q, a.o, a.n_heads, a.batch, a.tq, a.n_seg = 0x1000, 0x2000, 2, 1, 64, 9
    assert built_lib.f3r_attn_fwd(ctypes.byref(a), None) == -1
    assert b"n_seg" in built_lib.f3r_last_error_string()
    assert built_lib.f3r_patchify(0x1000, 0x2000, 1, 30, 32, 16, 0, 0, None) == -1  # H not a multiple of the patch size
    assert built_lib.f3r_layernorm(0x1000, 0x1000, None, 0x1000, None, 4, 30, 1e-6, 0, 0, None) == -1  # D % 4
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(_lib.F3RError):
        _lib.check(-3, "x")


def test_ctypes_prototypes_have_the_header_arity(built_lib):
    """Every ctypes prototype in fast3r_amd/_lib.py takes as many arguments as the declaration in include/f3r.h (a drift here is a silent
    stack mismatch, not an exception)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    decls = re.findall(r"\b(f3r_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    seen = set()
    for name, params in decls:
        params = " ".join(params.split())
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert name in _lib.SYMBOLS, name
        assert len(_lib.SYMBOLS[name][1]) == n, (name, n, len(_lib.SYMBOLS[name][1]))
        seen.add(name)
    assert seen == set(_lib.SYMBOLS), set(_lib.SYMBOLS) ^ seen
