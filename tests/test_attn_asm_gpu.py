"""The hand-scheduled attention kernel (fast3r_amd/csrc/asm/attn_gen.py, f3r_attn_args.kernel_sel = 2) on a real MI355X, through the
C ABI: against fp64 on the same rounded operands, against the general HIP kernel, and on the inputs that exercise its rare path (the
lazy softmax reference moving mid-stream).  Tolerances as in test_kernels_gpu.py: 2 x 2^-9 (fp16) / 2 x 2^-6 (bf16) of the output scale.
"""
import math

import pytest
import torch

from fast3r_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float16, torch.bfloat16]
LOG2E = 1.4426950408889634


def lp_tol(dt):
    return 2.0 ** -9 if dt == torch.float16 else 2.0 ** -6


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt)


def assert_close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.double()
    scale = float(ref.abs().max().clamp_min(1e-6))
    err = float((got - ref).abs().max())
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol:.1e})"


def ref_prescaled(qs, k, v, H, Hkv=None, hd=64):
    """qs already holds q * scale * log2(e) (rounded to lowp): softmax in base 2 over the rounded operands, fp64."""
    Hkv = Hkv or H
    Tq = qs.shape[0]
    qh = qs.double().reshape(Tq, H, hd).transpose(0, 1)
    kh = k.double().reshape(-1, Hkv, hd).transpose(0, 1).repeat_interleave(H // Hkv, 0)
    vh = v.double().reshape(-1, Hkv, hd).transpose(0, 1).repeat_interleave(H // Hkv, 0)
    s = (qh @ kh.transpose(1, 2)) * math.log(2.0)
    return (s.softmax(-1) @ vh).transpose(0, 1).reshape(Tq, H * hd)


def vt_of(v, Hkv, hd=64):
    T = v.shape[0]
    vt = torch.zeros((Hkv * hd, ops.vt_ld(T)), dtype=v.dtype)
    vt[:, :T] = v.t()
    return vt


def run(qs, k, v, H, Hkv=None, sel=2, hd=64):
    Hkv = Hkv or H
    o = torch.full((qs.shape[0], H * hd), float("nan"), dtype=qs.dtype, device=DEV)
    ops.attention(qs.to(DEV), o, H, 1.0, [(k.to(DEV), vt_of(v, Hkv, hd).to(DEV), k.shape[0], 0, 0)], q_prescaled=True, kv_group=H // Hkv, kernel_sel=sel,
                  head_dim=hd)
    return o


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Tq,Tk,H", [(512, 64, 1), (512, 128, 2), (1024, 192, 2), (512, 1024, 3), (1536, 4096, 2),
                                     (128, 256, 1), (200, 192, 2), (768, 768, 2), (1000, 320, 1), (2304 + 60, 1024, 2)])
def test_asm_kernel_vs_fp64(built_lib, dt, Tq, Tk, H):
    """incl. query counts that are not multiples of 512 (3 views of 768 tokens; a ragged 1000): the waves of the partial last workgroup work
    on the LAST 128 rows and store only the rows they own"""
    qs = rnd((Tq, H * 64), dt, 1, 0.125 * LOG2E * 1.5)
    k, v = rnd((Tk, H * 64), dt, 2, 1.5), rnd((Tk, H * 64), dt, 3)
    o = run(qs, k, v, H)
    assert_close(o.float(), ref_prescaled(qs, k, v, H), 2 * lp_tol(dt), f"asm attn {Tq}x{Tk}")
    hip = run(qs, k, v, H, sel=1)
    assert_close(o.float(), hip.float().cpu(), 2 * lp_tol(dt), "asm vs HIP kernel")


@pytest.mark.parametrize("dt", DTYPES)
def test_asm_kernel_forced_rebase(built_lib, dt):
    """keys far above the rest in late tiles (and one inside the first tile): the lazy reference must move and everything accumulated
    so far -- O, l, the already computed scores of the next half tile -- be rescaled exactly once"""
    Tq, Tk, H = 512, 1280, 1
    q = rnd((Tq, 64), dt, 10).float()
    k, v = rnd((Tk, 64), dt, 11, 0.3), rnd((Tk, 64), dt, 12)
    k[900] = (q[7] * 3.0).to(dt)
    k[130] = (q[300] * 2.0).to(dt)
    k[1279] = (q[511] * 4.0).to(dt)     # in the very last half tile
    k[33] = (q[100] * 2.5).to(dt)       # second half of the first tile
    qs = (q * (0.125 * LOG2E)).to(dt)
    o = run(qs, k, v, H)
    assert torch.isfinite(o.float()).all()
    assert_close(o.float(), ref_prescaled(qs, k, v, H), 2 * lp_tol(dt), "forced rebase")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pattern", ["drift_up", "stairs", "outlier_first", "all_equal"])
def test_asm_kernel_lazy_reference_patterns(built_lib, dt, pattern):
    """score sequences built to sit on the re-base trigger (see test_kernels_gpu.py::test_attention_lazy_reference_patterns)"""
    Tq, Tk, H = 512, 640, 1
    g = torch.Generator().manual_seed(90)
    u = torch.randn(64, generator=g)
    u = u / u.norm()
    a = torch.linspace(0.5, 8.0, Tq)
    tile = torch.arange(Tk) // 64
    if pattern == "drift_up":
        b = 0.35 * tile.float() + 0.01 * torch.randn(Tk, generator=g)
    elif pattern == "stairs":
        b = torch.where(torch.arange(Tk) < 200, 0.0, torch.where(torch.arange(Tk) < 500, 6.0, 15.0)) + 0.05 * torch.randn(Tk, generator=g)
    elif pattern == "outlier_first":
        b = -4.0 + 0.5 * torch.randn(Tk, generator=g)
        b[3] = 12.0
    else:
        b = torch.full((Tk,), 1.5)
    qs = (a[:, None] * u[None, :] * (8.0 * 0.125 * LOG2E)).to(dt)
    k = (b[:, None] * u[None, :]).to(dt)
    v = rnd((Tk, 64), dt, 91)
    o = run(qs, k, v, H)
    assert torch.isfinite(o.float()).all()
    assert_close(o.float(), ref_prescaled(qs, k, v, H), 2 * lp_tol(dt), f"lazy reference {pattern}")


@pytest.mark.parametrize("dt", DTYPES)
def test_asm_kernel_grouped_query_and_batch(built_lib, dt):
    """kv_group 2 (4 query heads on 2 K/V heads) and a batch of 2 independent sequences with strided batch access"""
    nb, Tq, Tk, H, Hkv = 2, 512, 320, 4, 2
    D, Dk = H * 64, Hkv * 64
    qs = rnd((nb * Tq, D), dt, 20, 0.125 * LOG2E * 1.5)
    k, v = rnd((nb * Tk, Dk), dt, 21, 1.5), rnd((nb * Tk, Dk), dt, 22)
    ld = ops.vt_ld(Tk)
    vt = torch.zeros((nb, Dk, ld), dtype=dt)
    for b in range(nb):
        vt[b, :, :Tk] = v[b * Tk:(b + 1) * Tk].t()
    o = torch.full((nb * Tq, D), float("nan"), dtype=dt, device=DEV)
    ops.attention(qs.to(DEV), o, H, 1.0, [(k.to(DEV), vt.to(DEV), Tk, Tk * Dk, Dk * ld)], tq=Tq, batch=nb, q_batch_stride=Tq * D, o_batch_stride=Tq * D,
                  q_prescaled=True, kv_group=2, kernel_sel=2)
    ref = torch.cat([ref_prescaled(qs[b * Tq:(b + 1) * Tq], k[b * Tk:(b + 1) * Tk], v[b * Tk:(b + 1) * Tk], H, Hkv) for b in range(nb)])
    assert_close(o.float(), ref, 2 * lp_tol(dt), "gqa + batch")


@pytest.mark.parametrize("Tq", [1024, 1024 + 768, 1024 + 200])
@pytest.mark.parametrize("dt", DTYPES)
def test_asm_kernel_segments_and_carried_state(built_lib, dt, Tq):
    """The view-sharded layout: K/V as segments (one launch over all of them == one launch over their concatenation), and the
    two-launch form (local segment with state_out, remote segments with state_in) == one launch -- for the hand-scheduled kernel alone
    and with the general HIP kernel taking either launch (one state layout for both).  Tq = 1792: a partial last workgroup, whose moved
    waves read state rows that another wave owns (and must not write them); Tq = 1224: a moved wave that owns PART of its rows."""
    H = 2
    lens = [192, 64, 320, 128]
    qs = rnd((Tq, H * 64), dt, 70, 0.125 * LOG2E * 1.5)
    ks = [rnd((n, H * 64), dt, 71 + i, 1.5) for i, n in enumerate(lens)]
    vs = [rnd((n, H * 64), dt, 81 + i) for i, n in enumerate(lens)]
    ld = ops.vt_ld(max(lens))

    def vt_pad(v):
        vt = torch.zeros((H * 64, ld), dtype=dt)
        vt[:, :v.shape[0]] = v.t()
        return vt
    segs = [(kk.to(DEV), vt_pad(vv).to(DEV), n, 0, 0) for kk, vv, n in zip(ks, vs, lens)]
    ref = ref_prescaled(qs, torch.cat(ks), torch.cat(vs), H)
    q = qs.to(DEV)
    one = torch.full((Tq, H * 64), float("nan"), dtype=dt, device=DEV)
    ops.attention(q, one, H, 1.0, segs, q_prescaled=True, kernel_sel=2)
    assert_close(one.float(), ref, 2 * lp_tol(dt), "segments, one launch")
    for sel_local, sel_remote in ((2, 2), (2, 1), (1, 2)):
        two = torch.full((Tq, H * 64), float("nan"), dtype=dt, device=DEV)
        state = ops.attention_state(Tq, H, DEV)
        ops.attention(q, two, H, 1.0, segs[:1], q_prescaled=True, state=state, state_out=True, kernel_sel=sel_local)
        assert torch.isnan(two.float()).all()  # the first launch must not write the output
        ops.attention(q, two, H, 1.0, segs[1:], q_prescaled=True, state=state, state_in=True, kernel_sel=sel_remote)
        if (sel_local, sel_remote) == (2, 2):
            # same tiles in the same order, state kept in fp32; the resumed launch takes its first half tile through the re-base path,
            # whose row sum is an fp32 dot instead of the packed partial sums of the steady state: equal up to that rounding
            assert_close(two.float(), one.float().cpu(), lp_tol(dt) / 2, "two launches vs one")
        assert_close(two.float(), ref, 2 * lp_tol(dt), f"state carry, kernels {sel_local} -> {sel_remote}")


def test_asm_kernel_is_the_automatic_choice_and_refuses_what_it_cannot_do(built_lib):
    dt = torch.float16
    H, Tq, Tk = 1, 512, 2048
    qs, k, v = rnd((Tq, 64), dt, 30, 0.2), rnd((Tk, 64), dt, 31), rnd((Tk, 64), dt, 32)
    auto, forced = run(qs, k, v, H, sel=0), run(qs, k, v, H, sel=2)
    assert torch.equal(auto, forced)  # >= F3R_ATTN_ASM_MIN_KEYS keys, eligible: kernel_sel 0 takes the hand-scheduled kernel
    with pytest.raises(ValueError, match="not eligible"):
        run(qs[:100], k, v, H, sel=2)  # fewer than 128 query rows
    with pytest.raises(ValueError, match="not eligible"):
        run(qs, k[:100], v[:100], H, sel=2)  # keys not a multiple of 64
    o = torch.empty((Tq, 64), dtype=dt, device=DEV)
    with pytest.raises(ValueError, match="not eligible"):
        ops.attention(qs.to(DEV), o, H, 0.125, [(k.to(DEV), vt_of(v, H).to(DEV), Tk, 0, 0)], kernel_sel=2)  # q not pre-scaled


def test_asm_kernel_full_size_rows_sum_to_one(built_lib):
    """BASELINE size (327 680 tokens would take a second per head; 2 heads x 65 536 keys here): V = const => O = const exactly up to the
    rounding of l, for every query row -- a size-independent property that any dropped / duplicated tile or stale LDS slot breaks."""
    dt = torch.float16
    H, Tq, Tk = 2, 1024, 65536
    qs, k = rnd((Tq, H * 64), dt, 40, 0.3), rnd((Tk, H * 64), dt, 41)
    v = torch.full((Tk, H * 64), 0.75, dtype=dt)
    o = run(qs, k, v, H)
    assert_close(o.float(), torch.full((Tq, H * 64), 0.75), 2.0 ** -10, "rows sum to one")


# ---- head_dim 80 / 128: the same generator with two query blocks per wave (256-query workgroups; AttnGen(head_dim=...))
HDS = [80, 128]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd", HDS)
@pytest.mark.parametrize("Tq,Tk,H", [(256, 64, 1), (256, 128, 2), (512, 192, 2), (256, 1024, 3), (768, 4096, 2), (64, 256, 1), (200, 192, 2), (1000, 320, 1),
                                     (2304 + 60, 1024, 2)])
def test_asm_kernel_other_head_dims_vs_fp64(built_lib, dt, hd, Tq, Tk, H):
    """head_dim 80 (5 k-steps of Q K^T, a third O^T block of which 16 rows exist, the mixed LDS-DMA piece) and 128 (two 64-column K groups, four
    O^T blocks): against fp64 on the same rounded operands and against the generic HIP kernel, incl. partial last workgroups (tq not a
    multiple of 256) and a single-wave launch (tq = 64)"""
    sc = hd ** -0.5 * LOG2E * 1.5
    qs = rnd((Tq, H * hd), dt, 1, sc)
    k, v = rnd((Tk, H * hd), dt, 2, 1.5), rnd((Tk, H * hd), dt, 3)
    o = run(qs, k, v, H, hd=hd)
    assert_close(o.float(), ref_prescaled(qs, k, v, H, hd=hd), 2 * lp_tol(dt), f"asm attn hd {hd} {Tq}x{Tk}")
    gen = run(qs, k, v, H, sel=1, hd=hd)
    assert_close(o.float(), gen.float().cpu(), 2 * lp_tol(dt), "asm vs generic kernel")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd", HDS)
def test_asm_kernel_other_head_dims_forced_rebase(built_lib, dt, hd):
    Tq, Tk, H = 256, 1280, 1
    q = rnd((Tq, hd), dt, 10).float()
    k, v = rnd((Tk, hd), dt, 11, 0.3), rnd((Tk, hd), dt, 12)
    k[900] = (q[7] * 3.0).to(dt)
    k[130] = (q[200] * 2.0).to(dt)
    k[1279] = (q[255] * 4.0).to(dt)     # in the very last half tile
    k[33] = (q[100] * 2.5).to(dt)       # second half of the first tile
    qs = (q * (hd ** -0.5 * LOG2E)).to(dt)
    o = run(qs, k, v, H, hd=hd)
    assert torch.isfinite(o.float()).all()
    assert_close(o.float(), ref_prescaled(qs, k, v, H, hd=hd), 2 * lp_tol(dt), "forced rebase")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd", HDS)
def test_asm_kernel_other_head_dims_grouped_query_and_batch(built_lib, dt, hd):
    nb, Tq, Tk, H, Hkv = 2, 512, 320, 4, 2
    D, Dk = H * hd, Hkv * hd
    qs = rnd((nb * Tq, D), dt, 20, hd ** -0.5 * LOG2E * 1.5)
    k, v = rnd((nb * Tk, Dk), dt, 21, 1.5), rnd((nb * Tk, Dk), dt, 22)
    ld = ops.vt_ld(Tk)
    vt = torch.zeros((nb, Dk, ld), dtype=dt)
    for b in range(nb):
        vt[b, :, :Tk] = v[b * Tk:(b + 1) * Tk].t()
    o = torch.full((nb * Tq, D), float("nan"), dtype=dt, device=DEV)
    ops.attention(qs.to(DEV), o, H, 1.0, [(k.to(DEV), vt.to(DEV), Tk, Tk * Dk, Dk * ld)], tq=Tq, batch=nb, q_batch_stride=Tq * D, o_batch_stride=Tq * D,
                  q_prescaled=True, kv_group=2, kernel_sel=2, head_dim=hd)
    ref = torch.cat([ref_prescaled(qs[b * Tq:(b + 1) * Tq], k[b * Tk:(b + 1) * Tk], v[b * Tk:(b + 1) * Tk], H, Hkv, hd=hd) for b in range(nb)])
    assert_close(o.float(), ref, 2 * lp_tol(dt), "gqa + batch")


@pytest.mark.parametrize("Tq", [512, 512 + 200])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd", HDS)
def test_asm_kernel_other_head_dims_segments_and_carried_state(built_lib, dt, hd, Tq):
    """K/V segments in one launch, and the two-launch form with the state parked in between -- by this kernel alone and with the generic HIP
    kernel taking either launch (one state layout: st_o rows of n_heads * head_dim)"""
    H = 2
    lens = [192, 64, 320, 128]
    qs = rnd((Tq, H * hd), dt, 70, hd ** -0.5 * LOG2E * 1.5)
    ks = [rnd((n, H * hd), dt, 71 + i, 1.5) for i, n in enumerate(lens)]
    vs = [rnd((n, H * hd), dt, 81 + i) for i, n in enumerate(lens)]
    ld = ops.vt_ld(max(lens))

    def vt_pad(v):
        vt = torch.zeros((H * hd, ld), dtype=dt)
        vt[:, :v.shape[0]] = v.t()
        return vt
    segs = [(kk.to(DEV), vt_pad(vv).to(DEV), n, 0, 0) for kk, vv, n in zip(ks, vs, lens)]
    ref = ref_prescaled(qs, torch.cat(ks), torch.cat(vs), H, hd=hd)
    q = qs.to(DEV)
    one = torch.full((Tq, H * hd), float("nan"), dtype=dt, device=DEV)
    ops.attention(q, one, H, 1.0, segs, q_prescaled=True, kernel_sel=2, head_dim=hd)
    assert_close(one.float(), ref, 2 * lp_tol(dt), "segments, one launch")
    for sel_local, sel_remote in ((2, 2), (2, 1), (1, 2)):
        two = torch.full((Tq, H * hd), float("nan"), dtype=dt, device=DEV)
        state = ops.attention_state(Tq, H, DEV, head_dim=hd)
        ops.attention(q, two, H, 1.0, segs[:1], q_prescaled=True, state=state, state_out=True, kernel_sel=sel_local, head_dim=hd)
        assert torch.isnan(two.float()).all()  # the first launch must not write the output
        ops.attention(q, two, H, 1.0, segs[1:], q_prescaled=True, state=state, state_in=True, kernel_sel=sel_remote, head_dim=hd)
        assert_close(two.float(), ref, 2 * lp_tol(dt), f"state carry, kernels {sel_local} -> {sel_remote}")


@pytest.mark.parametrize("hd", HDS)
def test_asm_kernel_other_head_dims_automatic_choice_and_full_size_rows(built_lib, hd):
    """kernel_sel 0 takes the generated kernel from F3R_ATTN_ASM_MIN_KEYS keys on (bit-identical to kernel_sel 2), the generic kernel below and
    for what the generated kernel cannot do; 65 536 keys with V = const: every output is that constant (any dropped / duplicated tile, a
    stale LDS slot or an unzeroed V^T padding row breaks it)"""
    dt = torch.float16
    H, Tq, Tk = 2, 512, 65536
    qs, k = rnd((Tq, H * hd), dt, 40, 0.3), rnd((Tk, H * hd), dt, 41)
    v = torch.full((Tk, H * hd), 0.75, dtype=dt)
    o = run(qs, k, v, H, hd=hd)
    assert_close(o.float(), torch.full((Tq, H * hd), 0.75), 2.0 ** -10, "rows sum to one")
    assert torch.equal(run(qs, k[:2048], v[:2048], H, sel=0, hd=hd), run(qs, k[:2048], v[:2048], H, sel=2, hd=hd))
    with pytest.raises(ValueError, match="not eligible"):
        run(qs[:40], k[:2048], v[:2048], H, sel=2, hd=hd)  # fewer than 64 query rows
    small = run(qs[:40], k[:2048], v[:2048], H, sel=0, hd=hd)  # ... which the generic kernel takes
    assert_close(small.float(), torch.full((40, H * hd), 0.75), 2.0 ** -10, "generic kernel")
    with pytest.raises(ValueError, match="not eligible"):
        o96 = torch.empty((Tq, H * 96), dtype=dt, device=DEV)
        ops.attention(rnd((Tq, H * 96), dt, 1).to(DEV), o96, H, 1.0, [(rnd((64, H * 96), dt, 2).to(DEV), torch.zeros((H * 96, 64), dtype=dt, device=DEV), 64, 0, 0)],
                      q_prescaled=True, kernel_sel=2, head_dim=96)   # no generated kernel for 96


# (head_dim 64 sizes chosen so that the launch is ONE launch: 45 x 16 = 720 items leave 208 for the last round, 40 x 13 = 520 leave 8 but 13 heads make
# whole rounds 256-block multiples -- the last-round split of tail_split_rows does not apply; test_small_launches_take_256_query_work_items covers it)
@pytest.mark.parametrize("dt,hd,Tq,H", [(torch.float16, 64, 45 * 512 - 200, 16), (torch.bfloat16, 64, 20480, 13), (torch.float16, 80, 8192 + 64, 16),
                                        (torch.float16, 128, 8192, 17)])
def test_work_stealing_form_is_bit_identical_and_leaves_its_counter_zero(built_lib, dt, hd, Tq, H):
    """f3r_attn_args.sched_counter (round 5): launches of at least two rounds of workgroups run as one persistent workgroup per CU taking
    (query block, head) items from a shared counter.  Every item is computed by the same instruction stream as in the one-item-per-workgroup
    form, so the outputs must be BIT-identical, item numbers that are no multiples of anything included; the kernel must leave {next, done}
    at zero (the next launch starts from it), three launches in a row; and the per-XCD wave counts of the debug block must add up."""
    Tk = 512
    qs = rnd((Tq, H * hd), dt, 1, hd ** -0.5 * LOG2E * 1.5)
    k, v = rnd((Tk, H * hd), dt, 2, 1.5), rnd((Tk, H * hd), dt, 3)
    saved = ops.ATTN_WORK_STEALING
    try:
        ops.ATTN_WORK_STEALING = False
        plain = run(qs, k, v, H, hd=hd)
        ops.ATTN_WORK_STEALING = True
        ctr = ops._sched_counter(torch.device(DEV, torch.cuda.current_device()))
        assert ctr.tolist() == [0, 0]
        ops.ATTN_COUNTERS = torch.zeros(56, dtype=torch.int32, device=DEV)
        for _ in range(3):
            stolen = run(qs, k, v, H, hd=hd)
            torch.cuda.synchronize()
            assert ctr.tolist() == [0, 0]
            assert torch.equal(stolen.view(torch.int16), plain.view(torch.int16))
        # leftovers of an aborted launch in the pair: the library clears it on the launch stream before the kernel starts (round 6)
        saved_ctrs, ops.ATTN_COUNTERS = ops.ATTN_COUNTERS, None
        ctr.copy_(torch.tensor([7, 3], dtype=torch.int32))
        again = run(qs, k, v, H, hd=hd)
        torch.cuda.synchronize()
        assert ctr.tolist() == [0, 0] and torch.equal(again.view(torch.int16), plain.view(torch.int16))
        # a scope hands the launches another pair (what Fast3R's graph cache does per captured graph)
        mine = torch.full((2,), 5, dtype=torch.int32, device=DEV)
        with ops.sched_scope(mine):
            assert ops._sched_counter(mine.device) is mine
            scoped = run(qs, k, v, H, hd=hd)
        torch.cuda.synchronize()
        assert mine.tolist() == [0, 0] and torch.equal(scoped.view(torch.int16), plain.view(torch.int16))
        # f3r_attn_args.reserve_cus (round 6): the persistent form on fewer workgroups (a rank leaves CUs to the exchange kernels) computes the same
        # items with the same instruction stream -- also when the launch is below the two-rounds threshold of the plain work-stealing form
        for rsv in (16, 200):
            o_r = torch.full((qs.shape[0], H * hd), float("nan"), dtype=dt, device=DEV)
            ops.attention(qs.to(DEV), o_r, H, 1.0, [(k.to(DEV), vt_of(v, H, hd).to(DEV), k.shape[0], 0, 0)], q_prescaled=True, kernel_sel=2, head_dim=hd,
                          reserve_cus=rsv)
            torch.cuda.synchronize()
            assert ctr.tolist() == [0, 0] and torch.equal(o_r.view(torch.int16), plain.view(torch.int16)), rsv
        ops.ATTN_COUNTERS = saved_ctrs
        c = [int(x) & 0xFFFFFFFF for x in ops.ATTN_COUNTERS.tolist()]
        ops.ATTN_TIMER = []   # which form did these launches take?  (head_dim 64: 512-query items, or 256-query ones for launches that fill the chip unevenly)
        run(qs, k, v, H, hd=hd)
        q256 = "q256" in ops.ATTN_TIMER[0][5]
        ops.ATTN_TIMER = None
        wq = 512 if (hd == 64 and not q256) else 256
        waves = 3 * 4 * (-(-Tq // wq)) * H
        per_xcd = [c[8 + 6 * x + 4] | (c[8 + 6 * x + 5] << 32) for x in range(8)]
        assert c[1] == waves and sum(per_xcd) == waves and all(w > 0 for w in per_xcd), (c[:4], per_xcd)
    finally:
        ops.ATTN_WORK_STEALING = saved
        ops.ATTN_COUNTERS = None
    assert_close(stolen.float()[:1024], ref_prescaled(qs[:1024], k, v, H, hd=hd), 2 * lp_tol(dt), "work-stealing attention vs fp64")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Tq,H,expect", [(3072, 16, "q256"), (20480, 16, "split"), (8192, 16, "asm_"), (40960, 16, "asm_"), (3000, 5, "q256"), (4096, 16, "q256"), (6144, 16, "asm_"),
                                         (20480 - 300, 16, "split"), (20480 + 300, 16, "asm_"), (24576, 16, "asm_"), (102400, 16, "asm_"), (50 * 512, 16, "split"), (9 * 512, 32, "split")])
def test_small_launches_take_256_query_work_items(built_lib, dt, Tq, H, expect):
    """f3r_attn_asm_q256_* (round 6): at head_dim 64 a launch of less than one round of 512-query items -- N = 3 views: 96 items on 256 CUs -- runs the
    same kernel with two query blocks per wave (twice the items at 0.55 - 0.64 of an item's time); a full round (N = 8: 256 items) or more keeps the
    512-query form (N = 20: 2.5 rounds would win on paper and on random operands, and loses inside the model: f3r_attn_asm.hip use_q256).  Which form a launch takes is the library's choice (f3r_attn_kernel_name says which); both are held
    to the same float64 reference.
    "split" (second half of round 6, f3r_attn_asm.hip tail_split_rows): a launch whose last round fills at most half the chip -- N = 20: 640 items = 2.5
    rounds, N = 100: 12.5 -- runs its whole rounds on 512-query items and the query blocks of the remainder as a second launch on 256-query items
    (24 576 rows x 16 heads = 768 items = 3 full rounds, and 40 960 = 5, stay one launch; so do 41 blocks = 2 rounds + 144 items, and launches of more than
    8 whole rounds like N = 100, where two launches lose more to the XCDs' clock spread than the last round costs: measured, no gain)."""
    Tk = 2048
    qs = rnd((Tq, H * 64), dt, 5, 0.125 * LOG2E * 1.5)
    k, v = rnd((Tk, H * 64), dt, 6, 1.5), rnd((Tk, H * 64), dt, 7)
    ops.ATTN_TIMER = []
    try:
        o = run(qs, k, v, H, sel=2)
        name = ops.ATTN_TIMER[0][5]
    finally:
        ops.ATTN_TIMER = None
    assert name.startswith("f3r_attn_asm_q256_") == (expect == "q256") and (" + f3r_attn_asm_q256_" in name) == (expect == "split") and "f3r_attn_asm_" in name, name
    rows = slice(0, 2048)
    assert_close(o.float()[rows], ref_prescaled(qs[rows], k, v, H), 2 * lp_tol(dt), f"{name} vs fp64")
    tail = slice(Tq - 300, Tq)
    assert_close(o.float()[tail], ref_prescaled(qs[tail], k, v, H), 2 * lp_tol(dt), f"{name} vs fp64, last rows")
    if expect == "split":   # rows on both sides of the cut between the two launches (a multiple of 512 rows: whole rounds of 512-query items)
        cut = (Tq + 511) // 512 // (256 // math.gcd(256, H)) * (256 // math.gcd(256, H)) * 512
        mid = slice(cut - 600, min(Tq, cut + 600))
        assert_close(o.float()[mid], ref_prescaled(qs[mid], k, v, H), 2 * lp_tol(dt), f"{name} vs fp64, around the cut at row {cut}")
        assert not torch.isnan(o.float()).any()
