"""The hand-scheduled attention kernel without a GPU: the instruction list that fast3r_amd/csrc/asm/attn_gen.py prints for the
assembler is executed lane-exactly by tools/gfx950_emu.py (pessimistic asynchrony: a consumer placed before its s_waitcnt reads a
NaN pattern, LDS-DMA data lands at the issuing wave's vmcnt wait) and compared with a float64 softmax on the same rounded operands;
the static hazard walk (isa.Program.check_hazards: MFMA result -> VALU reader distance, VALU result -> MFMA operand, M0 -> LDS-DMA)
must be clean; the text must assemble for gfx950."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "fast3r_amd", "csrc", "asm"))


@pytest.mark.parametrize("dtype,tiles,spike,tol", [("f16", 1, False, 6e-4), ("f16", 3, False, 6e-4), ("f16", 6, True, 6e-4),
                                                   ("bf16", 2, False, 5e-3), ("bf16", 5, True, 5e-3)])
def test_emulated_kernel_matches_fp64(dtype, tiles, spike, tol):
    import emu_attn
    err = emu_attn.run_case(dtype, tiles, n_heads=2, wgs=((0, 1, 0),), spike=spike)
    assert err < tol


def test_emulated_kernel_batch_gqa_and_second_query_block():
    import emu_attn
    err = emu_attn.run_case("f16", 2, n_heads=4, wgs=((1, 3, 1), (0, 0, 0)), batch=2, kv_shift=1, q_blocks=2)
    assert err < 6e-4


@pytest.mark.parametrize("tiles,split", [((1, 2), False), ((2, 1, 3), False), ((1, 1, 1, 1, 1, 1, 1, 1), False), ((2, 3), True), ((1, 2, 1), True)])
def test_emulated_kernel_walks_segments_and_carries_the_softmax_state(tiles, split):
    """K/V as several segments (the view-sharded layout; a segment hop in the LDS-DMA stream) in one launch, or as two launches that
    park / resume the online-softmax state (f3r_attn_args.state_out / state_in)"""
    import emu_attn
    assert emu_attn.run_case("f16", list(tiles), n_heads=2, wgs=((0, 1, 0),), spike=True, split_state=split) < 6e-4


@pytest.mark.parametrize("kw", [dict(tiles=1), dict(tiles=6, spike=True), dict(tiles=[2, 1, 3], spike=True), dict(tiles=[2, 3], split=True),
                                dict(tiles=7, spike=True, gen=dict(pf=3, nslot=4)), dict(tiles=5, dtype="bf16", spike=True, tol=5e-3)])
def test_emulated_generator_settings(kw):
    """further tile counts, segments, the two-launch state form, a deeper prefetch setting and bf16 on the same generator"""
    import emu_attn
    err = emu_attn.run_case(kw.get("dtype", "f16"), kw["tiles"], n_heads=2, wgs=((0, 1, 0),), spike=kw.get("spike", False),
                            split_state=kw.get("split", False), layout=2, gen_kwargs=kw.get("gen"))
    assert err < kw.get("tol", 6e-4)


def test_emulated_kernel_counts_its_rebases():
    """f3r_attn_args.dbg_counters: {entries into the re-base block, waves, tiles walked} summed over waves.  Every wave enters once (the
    forced first re-base) and at most once per half-tile stage; these logits (std ~2.3) move the reference a few more times."""
    import emu_attn
    c = []
    assert emu_attn.run_case("f16", 4, n_heads=2, wgs=((0, 1, 0),), layout=2, counters=c) < 6e-4
    assert c[1:3] == [4, 16] and 4 <= c[0] <= 2 * 16, c
    # the clock sums (ABI 330): four waves each bracket their life with s_memtime / s_memrealtime (the emulator's stand-ins count instructions)
    assert c[3] > 4 * 1000 and 0 < c[4] < c[3], c
    base = c[0]
    assert emu_attn.run_case("f16", 4, n_heads=2, wgs=((0, 1, 0),), spike=True, layout=2, counters=c) < 6e-4
    assert c[1:3] == [4, 16] and c[0] > base, c   # same keys + one spiked key in the last tile: at least one more re-base


@pytest.mark.parametrize("kw", [dict(tq=200, wg=(0, 1, 0), tiles=2), dict(tq=1000, wg=(1, 0, 0), tiles=[1, 2], split=True, spike=True),
                                dict(tq=640, wg=(1, 1, 0), tiles=3, dtype="bf16", tol=5e-3)])
def test_emulated_partial_last_workgroup(kw):
    """a query count that is not a multiple of 512 (layout 2): the waves that would run past the end work on the last 128 rows and store
    only the rows they own -- the output / state buffers hold exactly tq rows, so a store past the end is an emulator error, and a row
    nobody wrote stays NaN"""
    import emu_attn
    err = emu_attn.run_case(kw.get("dtype", "f16"), kw["tiles"], n_heads=2, wgs=(kw["wg"],), spike=kw.get("spike", False),
                            split_state=kw.get("split", False), layout=2, tq=kw["tq"])
    assert err < kw.get("tol", 6e-4)


@pytest.mark.parametrize("hd", [80, 128])
@pytest.mark.parametrize("kw", [dict(n_tiles=1), dict(n_tiles=3), dict(n_tiles=5, spike=True), dict(dtype="bf16", n_tiles=4, spike=True, tol=5e-3),
                                dict(n_tiles=[2, 1, 3], spike=True), dict(n_tiles=[2, 3], split_state=True), dict(tq=300, q_blocks=2, wgs=((1, 0, 0), (0, 1, 0))),
                                dict(tq=64, wgs=((0, 0, 0),), n_tiles=2), dict(kv_shift=1, n_heads=4, batch=2, wgs=((0, 3, 1),))])
def test_emulated_other_head_dims(hd, kw):
    """AttnGen(head_dim=80 / 128): two query blocks per wave, D / 16 k-steps, ceil(D / 32) O^T blocks (80: the 16-column K remainder group, the mixed
    LDS-DMA piece, the zeroed V^T padding rows); tiles, segments, the two-launch state form, partial workgroups, grouped heads + batch"""
    import emu_attn
    kw = dict(kw)
    tol = kw.pop("tol", 6e-4)
    assert emu_attn.run_case(head_dim=hd, **kw) < tol


def test_head_dim_64_stream_is_what_it_was_before_the_other_widths():
    """the generalisation must not move one instruction of the benchmarked kernel: a digest of the head_dim-64 program text (the stream of
    round 3 + the three EXEC resets of the state-out epilogue that test_emulated_moved_wave_parks_the_rows_it_owns asked for + round 5's clock
    bracket: five scalar instructions in the prologue, the 64-bit sums in the optional counter block of the epilogue; the work-stealing entry / FETCH block around the unchanged prologue; the main loop is untouched)"""
    import hashlib
    import attn_gen
    gens = []
    for dt in ("f16", "bf16"):
        g = attn_gen.AttnGen(dt)
        g.build()
        gens.append(g)
    assert hashlib.md5(attn_gen.module_text(gens).encode()).hexdigest() == "8d0cba09028fe0b883145107f2f1dc3b"


@pytest.mark.parametrize("hd,tq,wg", [(64, 1000, (1, 0, 0)), (64, 600, (1, 1, 0)), (80, 700, (2, 0, 0)), (128, 696, (2, 1, 0))])
def test_emulated_moved_wave_parks_the_rows_it_owns(hd, tq, wg):
    """two launches with the softmax state parked in between, a query count that leaves the last wave PART of its rows: the wave is moved
    back to end at the last query and parks only the rows it owns -- per 32-query block a lane mask that must be computed with all lanes
    active (v_cmp writes 0 for inactive lanes; round 4 found the state-out epilogue computing block qb's mask under block qb-1's, which left
    rows of the later blocks unparked whenever tq is not a multiple of 32 x the blocks per wave)"""
    import emu_attn
    assert emu_attn.run_case(head_dim=hd, n_tiles=[2, 3], split_state=True, tq=tq, q_blocks=3, wgs=(wg,)) < 6e-4


def test_hazard_walk_flags_a_lane_mask_computed_under_a_narrowed_exec():
    """the static walk (isa.Program.check_hazards) knows the bug the moved-wave test above found: remove the EXEC resets of the state-out
    epilogue again and it names the three compares that would run with inactive lanes"""
    import attn_gen
    import isa
    g = attn_gen.AttnGen("f16")
    p = g.build()
    assert p.check_hazards() == []
    p.items = [it for it in p.items if not (isinstance(it, isa.Ins) and it.op == "s_mov_b64" and "mask of block qb" in (it.comment or ""))]
    found = p.check_hazards()
    assert len(found) == 3 and all("EXEC is narrowed" in f for f in found), found


def test_generated_text_assembles_and_has_no_hazards(tmp_path):
    import attn_gen
    gens = attn_gen.product_generators()
    for g in gens:
        assert g.p.check_hazards() == []
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler on this machine")
    src = tmp_path / "attn.s"
    src.write_text(attn_gen.module_text(gens))
    subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "attn.o")], check=True)


@pytest.mark.parametrize("kw", [dict(tq=1024, n_heads=2, tiles=2, wgs=((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)), steal=2),
                                dict(tq=700, n_heads=1, tiles=[1, 1], split=True, wgs=((0, 0, 0), (1, 0, 0)), steal=3),
                                dict(tq=256, n_heads=2, tiles=2, wgs=((0, 0, 0), (0, 1, 0)), steal=2, head_dim=80)])
def test_emulated_work_stealing_form_computes_every_item_and_leaves_the_counter_zero(kw):
    """f3r_attn_args.sched_counter (round 5): persistent workgroups fetch (query block, head, batch) items from a shared counter -- wave 0's
    atomic, the LDS word behind the ring, two barriers -- decode them with the magic-number divisions, re-enter the prologue per item (partial
    last query block, two-launch state form included) and the last workgroup to leave zeroes {next, done} for the next launch"""
    import emu_attn
    err = emu_attn.run_case(kw.get("dtype", "f16"), kw["tiles"], n_heads=kw["n_heads"], wgs=kw["wgs"], tq=kw["tq"], split_state=kw.get("split", False),
                            steal=kw["steal"], head_dim=kw.get("head_dim", 64))
    assert err < 6e-4


@pytest.mark.parametrize("kw", [dict(n_tiles=1), dict(n_tiles=3), dict(n_tiles=6, spike=True), dict(n_tiles=[2, 1, 3], spike=True), dict(n_tiles=[2, 3], split_state=True),
                                dict(tq=300, q_blocks=2, wgs=((1, 0, 0), (0, 1, 0))), dict(tq=64, wgs=((0, 0, 0),), n_tiles=2),
                                dict(kv_shift=1, n_heads=4, batch=2, wgs=((0, 3, 1),)), dict(tq=700, n_heads=1, n_tiles=[1, 1], split_state=True, wgs=((0, 0, 0), (2, 0, 0)), steal=3)])
def test_emulated_three_product_form(kw):
    """AttnGen(qk_planes=2) (round 6, precision "robust"; f3r_attn_args.qk_planes): Q and K rows hold [hi | lo] fp16 planes per head and every score
    block is q_hi k_hi + q_lo k_hi + q_hi k_lo (twelve MFMA k-steps, fp32 accumulate); P V is the head_dim-64 form.  Two query blocks per wave like
    the head_dim-80 / 128 kernels: tiles, segments, the two-launch state form, partial workgroups, grouped heads + batch, work stealing.  The
    reference is float64 on hi + lo."""
    import emu_attn
    assert emu_attn.run_case(qk_planes=2, **kw) < 6e-4


def test_emulated_three_product_form_uses_its_low_planes():
    """the kernel really multiplies its low planes: its output is closer to the float64 softmax of hi + lo than to the one computed from the hi planes
    alone (a kernel that ignored them would sit at the latter, one rounding of P away), and the two references are further apart than the kernel
    is from the right one"""
    import emu_attn
    errs = []
    emu_attn.run_case(qk_planes=2, n_tiles=4, n_heads=2, wgs=((0, 1, 0),), spike=True, errs=errs)
    err, one_product_distance, err_vs_one_product = errs[0]
    assert err < 6e-4 and one_product_distance > 1.5 * err and err_vs_one_product > 1.5 * err, errs


@pytest.mark.parametrize("kw", [dict(n_tiles=1), dict(n_tiles=3), dict(n_tiles=6, spike=True), dict(n_tiles=[2, 1, 3], spike=True), dict(n_tiles=[2, 3], split_state=True),
                                dict(tq=300, q_blocks=2, wgs=((1, 0, 0), (0, 1, 0))), dict(kv_shift=1, n_heads=4, batch=2, wgs=((0, 3, 1),)),
                                dict(tq=700, n_heads=1, n_tiles=[1, 1], split_state=True, wgs=((0, 0, 0), (2, 0, 0)), steal=3)])
def test_emulated_three_product_form_with_fp8_corrections(kw):
    """AttnGen(qk_planes=2, corr="f8") (f3r_attn_args.qk_planes = 3): rows [hi fp16 | e4m3(hi) | e4m3(lo 2^12)] per head; a score block is four fp16
    MFMAs (q_hi k_hi) + two block-scaled fp8 MFMAs over the whole head (q_lo8 k_hi8 with the Q scale 2^-12, q_hi8 k_lo8 with the K scale 2^-12); the
    8-register fp8 fragments are the SAME LDS reads / global loads as the fp16 lo plane's (two consecutive 16-byte chunks = the hardware's native
    k order).  Reference: float64 on exactly those planes."""
    import emu_attn
    assert emu_attn.run_case(qk_planes=2, corr="f8", **kw) < 6e-4


def test_emulated_fp8_corrections_stay_close_to_the_exact_planes():
    """what the fp8 copies cost: the softmax computed from the kernel's planes (fp16 hi product + fp8 corrections) against the one from hi + lo
    (~22 bits) -- an order of magnitude inside the kernel's own rounding of P"""
    import emu_attn
    errs = []
    assert emu_attn.run_case(qk_planes=2, corr="f8", n_tiles=4, n_heads=2, wgs=((0, 1, 0),), spike=True, errs=errs) < 6e-4
    assert errs and errs[0][1] < 1e-4, errs


@pytest.mark.parametrize("corr", ["f16", "f8"])
def test_emulated_three_product_form_parks_its_state_per_sequence(corr):
    """what precision "robust" launches for the encoder: ONE launch over a batch of sequences with state_out; sequence z owns state rows [z tq, (z + 1)
    tq) (the other kernels carry state at batch 1 only); the output is computed from the parked state as f3r_attn_state_finish does; tq = 300: a
    partial last workgroup per sequence must not write another sequence's rows"""
    import emu_attn
    assert emu_attn.run_case(qk_planes=2, corr=corr, n_tiles=2, n_heads=2, batch=3, wgs=((0, 1, 2), (1, 0, 1), (0, 0, 0)), q_blocks=2, tq=300, finish_state=True) < 6e-4


@pytest.mark.parametrize("kw", [dict(n_tiles=1), dict(n_tiles=3), dict(n_tiles=6, spike=True), dict(dtype="bf16", n_tiles=4, spike=True, tol=5e-3),
                                dict(n_tiles=[2, 1, 3], spike=True), dict(n_tiles=[2, 3], split_state=True), dict(tq=300, q_blocks=2, wgs=((1, 0, 0), (0, 1, 0))),
                                dict(kv_shift=1, n_heads=4, batch=2, wgs=((0, 3, 1),)), dict(tq=700, n_heads=1, n_tiles=[1, 1], split_state=True, wgs=((0, 0, 0), (2, 0, 0)), steal=3)])
def test_emulated_head_dim_64_with_256_query_work_items(kw):
    """AttnGen(head_dim=64, qpw=2) (round 6, kernels f3r_attn_asm_q256_*): the head_dim-64 kernel with two query blocks per wave -- 256-query work items
    for launches whose 512-query items do not fill the chip evenly (f3r_attn_asm.hip use_q256).  Same state layout as every other kernel."""
    import emu_attn
    kw = dict(kw)
    tol = kw.pop("tol", 6e-4)
    assert emu_attn.run_case(qpw=2, **kw) < tol
