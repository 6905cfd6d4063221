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
