"""The hand-scheduled GEMM kernels without a GPU: the instruction lists that fast3r_amd/csrc/asm/gemm_gen.py prints for the assembler are
executed lane-exactly by tools/gfx950_emu.py (a consumer placed before its s_waitcnt reads a NaN pattern; LDS-DMA data lands at the issuing
wave's vmcnt wait; waves of a workgroup run one after the other between barriers, so a read that no barrier orders sees stale data) and
compared with float64 on the same rounded operands; the static hazard walk must be clean; the text must assemble for gfx950.
Replaces the nn.Linear layers of fast3r/croco/models/blocks.py:94-105,125-131 (see gemm_gen.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "fast3r_amd", "csrc", "asm"))


@pytest.mark.parametrize("nk1", [4, 5, 6, 7, 8, 11])
def test_emulated_f32_role_over_the_ring(nk1):
    """K-tile counts that end an output tile in every one of the five unrolled copies of the ring: x += A W^T + b in place"""
    import emu_gemm
    assert emu_gemm.run_case("f16", "f32", nk1=nk1) < 1e-6


@pytest.mark.parametrize("kw", [dict(nk1=4, ntm=3, grid=1), dict(nk1=5, ntm=3, grid=1), dict(nk1=6, ntm=3, grid=1), dict(nk1=7, ntm=3, grid=1),
                                dict(nk1=8, ntm=3, grid=1), dict(nk1=4, ntm=2, ntn=3, grid=2), dict(nk1=2, segs=2, ntm=2, ntn=2, grid=1),
                                dict(nk1=5, ntm=3, grid=1, bias=False), dict(nk1=5, ntm=3, grid=1, res=False)])
def test_emulated_persistent_workgroups(kw):
    """fewer workgroups than output tiles: a workgroup walks tiles b, b + grid, ...; the operand streams cross into the next tile's panels
    up to three K-tiles ahead of the MFMAs (at every phase of the ring), the accumulators are written out and re-initialised with the next
    tile's bias between two k-steps, the last tile ends the kernel"""
    import emu_gemm
    assert emu_gemm.run_case("f16", "f32", **kw) < 1e-6


@pytest.mark.parametrize("kw", [dict(nk1=3, segs=2), dict(nk1=4, segs=2, res=False), dict(nk1=4, bias=False), dict(nk1=4, bias=False, res=False),
                                dict(nk1=4, dtype="bf16"), dict(nk1=2, dtype="bf16", segs=2), dict(nk1=4, lda_pad=64)])
def test_emulated_f32_role_variants(kw):
    """split-precision weights as K segments (the A stream wraps to k = 0, the W stream runs on into the lo plane), no bias / no residual,
    bf16 operands (three-term bias split by truncation), a padded A row stride"""
    import emu_gemm
    kw = dict(kw)
    assert emu_gemm.run_case(kw.pop("dtype", "f16"), "f32", **kw) < 1e-6


@pytest.mark.parametrize("dtype,act,segs,tol", [("f16", "none", 1, 5e-4), ("f16", "gelu", 1, 5e-4), ("f16", "relu", 2, 5e-4), ("bf16", "gelu", 2, 4e-3),
                                                ("bf16", "none", 1, 4e-3)])
def test_emulated_lowp_role(dtype, act, segs, tol):
    """out_lp = act(A W^T + b) with the permuted weight rows (a lane half owns 16 consecutive columns); tolerance = one rounding of the output"""
    import emu_gemm
    assert emu_gemm.run_case(dtype, "lp", nk1=3 if segs == 2 else 5, segs=segs, act=act, ntm=2, grid=1) < tol


@pytest.mark.parametrize("tiles,wgs", [((2, 2), None), ((3, 2), None), ((5, 1), None), ((12, 3), (0, 7, 8, 13, 35)), ((16, 4), (0, 1, 9, 31, 63))])
def test_emulated_tile_map(tiles, wgs):
    """several workgroups: every emulated workgroup writes exactly one 256 x 256 tile, no two the same one (the XCD-aware order is a
    bijection), with the right operand panels"""
    import emu_gemm
    assert emu_gemm.run_case("f16", "f32", ntm=tiles[0], ntn=tiles[1], nk1=4, wgs=wgs, grid=tiles[0] * tiles[1]) < 1e-6


@pytest.mark.parametrize("kw,tol", [(dict(), 5e-4), (dict(n_seq=1, seq_tiles=2, d_tiles=2, nk1=2, segs=2, grid=2), 5e-4),
                                    (dict(dtype="bf16", n_seq=3, nk1=5, grid=1), 4e-3),
                                    # round 5: RoPE-2D in the q | k launch's epilogue (ACT_ROPE) -- pairs through v_permlane32_swap, positions by magic division
                                    (dict(n_seq=2, d_tiles=2, nk1=4, rope_w=12, grid=1), 8e-4), (dict(n_seq=3, nk1=2, segs=2, rope_w=1), 8e-4)])
def test_emulated_qkv_as_two_launches(kw, tol):
    """the fusion decoder's QKV projection (no rotary embedding): q | k through output segments with the q scale, V^T through the swapped
    operand roles (weights with their lo plane as the kernel's A operand, activations wrapping per K segment as its W operand, bias by
    output row, one output segment per sequence, padding columns of V^T untouched)"""
    import emu_gemm
    assert emu_gemm.run_qkv_case(**kw) < tol


def test_tile_map_is_a_bijection_at_the_model_shapes():
    """the kernel's tile order restated in Python (gemm_gen.tile_of) over whole grids -- N = 320 (1280 x 4 / x 16 tiles), N = 100, odd view
    counts (ntm % 8 == 4), grids smaller than 8 (test_emulated_tile_map checks the kernel's own scalar code against written tiles)"""
    import gemm_gen
    for ntm, ntn in ((1280, 4), (1280, 16), (400, 4), (12, 16), (1, 1), (3, 1), (7, 5), (44, 12), (1500 * 4, 4)):
        seen = set(gemm_gen.tile_of(wg, ntm, ntn) for wg in range(ntm * ntn))
        assert len(seen) == ntm * ntn and all(0 <= tm < ntm and 0 <= tn < ntn for tm, tn in seen), (ntm, ntn)


def test_generated_text_assembles_and_has_no_hazards(tmp_path):
    import gemm_gen
    gens = gemm_gen.product_generators()
    for g in gens:
        assert g.p.check_hazards() == []
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler on this machine")
    src = tmp_path / "gemm.s"
    src.write_text(gemm_gen.module_text(gens))
    subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "gemm.o")], check=True)


@pytest.mark.parametrize("kw,tol", [(dict(role="f32", nk16=4), 1e-6), (dict(role="f32", ntm=3, nk16=6, grid=1, outliers=False), 1e-6),
                                    (dict(role="lp", nk16=4, act="gelu", out8=True), 5e-4), (dict(role="lp", ntn=2, nk16=4, act="none", out8=True, grid=1, bias=False), 8e-4)])
def test_emulated_fp8_low_plane(kw, tol):
    """F3R_SPLIT_W2F8 (round 5): after the fp16 K-tiles the same two operand streams run on into fp8 K-tiles [256 rows][128 k] consumed by the
    block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (emulated with the lane layout and scale semantics measured on an MI355X,
    tools/ubench/mfma_scale_probe.py) -- the four window kinds (fp16 -> fp16, fp16 -> fp8, fp8 -> fp8, fp8 -> next output tile), the per-channel
    scale words re-loaded per output tile by persistent workgroups, activations beyond the fp8 range (clamped in their fp8 copy), and the GELU
    epilogue's fp8 copy of its own output (rows [N fp16 | N fp8] for the next GEMM; not written without GELU) -- against float64 on the planes"""
    import emu_gemm
    assert emu_gemm.run_case_f8(**kw) < tol
