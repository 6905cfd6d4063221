"""INTEGRATION.md section 2 shows the ctypes binding a reference maintainer would add (the pattern of fast3r/croco/models/curope/curope.cpp:49-69 +
curope2d.py:18-47).  That text drifted once (ABI 330 grew f3r_attn_args by `sched_counter`, the document kept 592 bytes: VERDICT round 5, row b), so
the document is now EXECUTED: every fenced Python block under "## 2." runs against the built library, its struct must have the library's size, and every
field of it -- and of fast3r_amd/_lib.py's own structures -- must sit at the offset a C compiler gives the field of the same name in include/f3r.h.
On a GPU box the document's `sdpa_f3r` is also called and compared with a float64 softmax."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from fast3r_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER_DIR = os.path.join(ROOT, "include")


def _section2_blocks():
    text = open(DOC).read()
    m = re.search(r"^## 2\..*?(?=^## 3\.)", text, flags=re.S | re.M)
    assert m, "INTEGRATION.md lost its section 2"
    blocks = re.findall(r"```python\n(.*?)```", m.group(0), flags=re.S)
    assert blocks, "INTEGRATION.md section 2 has no fenced python block"
    return blocks


def _exec_doc(built_lib):
    """Run the document's code with `ctypes.CDLL("libf3r_hip.so")` resolved to the in-tree library (a maintainer would install it on the loader path)."""
    real = ctypes.CDLL

    def cdll(name, *a, **k):
        return real(_lib.LIB_PATH if os.path.basename(str(name)) == "libf3r_hip.so" else name, *a, **k)
    ns = {"__name__": "integration_md"}
    ctypes.CDLL = cdll
    try:
        for b in _section2_blocks():
            exec(compile(b, DOC, "exec"), ns)
    finally:
        ctypes.CDLL = real
    return ns


def _c_offsets(tmp_path, struct, fields):
    """offsetof(struct, field) for every field + sizeof, from a C99 program compiled against include/f3r.h"""
    src = tmp_path / f"probe_{struct}.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "f3r.h"', "int main(void) {"]
    lines += [f'  printf("{f} %zu\\n", offsetof({struct}, {f}));' for f in fields]
    lines += [f'  printf("sizeof %zu\\n", sizeof({struct}));', "  return 0; }"]
    src.write_text("\n".join(lines))
    exe = tmp_path / f"probe_{struct}"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", HEADER_DIR, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (ln.split() for ln in out.strip().splitlines())}


def _check_struct(tmp_path, cstruct, pystruct):
    names = [f[0] for f in pystruct._fields_]
    offs = _c_offsets(tmp_path, cstruct, names)   # a field the header does not have fails to compile
    for n in names:
        assert getattr(pystruct, n).offset == offs[n], (cstruct, n, getattr(pystruct, n).offset, offs[n])
    assert ctypes.sizeof(pystruct) == offs["sizeof"], (cstruct, ctypes.sizeof(pystruct), offs["sizeof"])
    return offs


def test_integration_md_binding_runs_and_matches_the_library(built_lib, tmp_path):
    ns = _exec_doc(built_lib)   # the document's own `assert _lib.f3r_sizeof(1) == ctypes.sizeof(F3RAttnArgs)` runs here
    doc_struct = ns["F3RAttnArgs"]
    assert built_lib.f3r_sizeof(1) == ctypes.sizeof(doc_struct) == ctypes.sizeof(_lib.AttnArgs)
    offs = _check_struct(tmp_path, "f3r_attn_args", doc_struct)
    assert offs["sched_counter"] == 592 and offs["qk_planes"] == 600 and offs["sizeof"] == 608   # ABI 340
    # same field list, same order as the product's own binding
    assert [f[0] for f in doc_struct._fields_] == [f[0] for f in _lib.AttnArgs._fields_]
    assert callable(ns["sdpa_f3r"])


def test_product_bindings_match_header_offsets(built_lib, tmp_path):
    _check_struct(tmp_path, "f3r_attn_args", _lib.AttnArgs)
    _check_struct(tmp_path, "f3r_gemm_args", _lib.GemmArgs)
    _check_struct(tmp_path, "f3r_attn_f32_args", _lib.AttnF32Args)
    assert built_lib.f3r_sizeof(0) == ctypes.sizeof(_lib.GemmArgs) and built_lib.f3r_sizeof(2) == ctypes.sizeof(_lib.AttnF32Args)


@pytest.mark.gpu
def test_integration_md_sdpa_f3r_computes_attention(built_lib):
    """the document's function, called as a maintainer would call it from Attention.forward (blocks.py:158-190)"""
    import torch
    ns = _exec_doc(built_lib)
    torch.manual_seed(0)
    T, H = 2048, 2
    dev = "cuda"
    q = torch.randn(T, H * 64, device=dev).half()
    k = torch.randn(T, H * 64, device=dev).half()
    v = torch.randn(T, H * 64, device=dev).half()
    vt = v.t().contiguous()
    scale = 0.125
    o = ns["sdpa_f3r"](q, k, vt, scale)
    torch.cuda.synchronize()
    qd, kd, vd = (t.double().view(T, H, 64).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(1, 2) * scale, dim=-1) @ vd).transpose(0, 1).reshape(T, H * 64)
    err = ((o.double() - ref).norm() / ref.norm()).item()
    assert err < 2e-3, err
