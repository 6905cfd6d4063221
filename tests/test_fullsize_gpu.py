"""Parity at BASELINE.json's FULL sizes (N = 320 views -> 327 680 tokens, ViT-L widths), where neither the oracle nor an fp64
reference can run: size-independent properties of the operators, checked on the real kernels.

  attention   (a) V = const  =>  O = const exactly (softmax rows sum to one)
              (b) linearity in V:  att(a V1 + b V2) = a att(V1) + b att(V2)
              (c) key-order invariance:  permuting (K rows, V^T columns) together leaves O unchanged
              (d) the multi-GPU split -- 8 K/V segments, local launch parking (m, l, O), remote launch resuming -- equals ONE launch
                  over the concatenation, bit for bit (same tiles, same order)
              (e) a sampled block of query rows against the fp64 reference (the only place a reference is affordable)
  GEMM        checksum of checksums (ABFT): column sums of A W^T equal (column sums of A) W^T
"""
import pytest
import torch

from fast3r_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, H, D = 327680, 16, 1024
SCALE = 0.160192  # fast3r.py attention scale with the inference-time bias


def _mk(shape, seed, scale=1.0, dt=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * scale).to(dt)


@pytest.fixture(scope="module")
def qkv():
    q, k, vt = _mk((T, D), 1), _mk((T, D), 2), _mk((D, T), 3)
    return q, k, vt


def _att(q, k, vt, **kw):
    o = torch.empty_like(q)
    ops.attention(q, o, H, SCALE, [(k, vt, k.shape[0], 0, 0)], **kw)
    return o


def test_attention_full_size_rows_sum_to_one(built_lib, qkv):
    q, k, _ = qkv
    ones = torch.full((D, T), 0.75, dtype=torch.bfloat16, device=DEV)
    o = _att(q, k, ones)
    assert torch.equal(o, torch.full_like(o, 0.75))


def test_attention_full_size_linear_in_v(built_lib, qkv):
    q, k, vt = qkv
    vt2 = _mk((D, T), 4)
    a, b = 0.5, -2.0  # exactly representable scalings
    lhs = _att(q, k, (a * vt.float() + b * vt2.float()).to(torch.bfloat16))
    rhs = a * _att(q, k, vt).float() + b * _att(q, k, vt2).float()
    err = float((lhs.float() - rhs).abs().max())
    ref = float(rhs.abs().max())
    assert err <= 2.0 ** -6 * ref, (err, ref)  # three bf16 roundings of outputs of that magnitude


def test_attention_full_size_key_permutation_invariance(built_lib, qkv):
    q, k, vt = qkv
    perm = torch.randperm(T, generator=torch.Generator(device=DEV).manual_seed(5), device=DEV)
    o1 = _att(q, k, vt)
    o2 = _att(q, k[perm].contiguous(), vt[:, perm].contiguous())
    err = float((o1.float() - o2.float()).abs().max())
    ref = float(o1.float().abs().max())
    assert err <= 2.0 ** -6 * ref, (err, ref)


def test_attention_full_size_sharded_split_is_bit_identical(built_lib, qkv):
    """The 8-GPU decomposition of BASELINE configs[3] on one device: rank r = 3 owns 40 views (queries 122 880 .. 163 840); its local
    launch parks the state, the remote launch walks the 7 other shards; must equal one launch over [local, remote...] in that order."""
    q, k, vt = qkv
    R, per = 8, T // 8
    r = 3
    qs = q[r * per:(r + 1) * per]
    segs = [(k[i * per:(i + 1) * per], vt[:, i * per:(i + 1) * per].contiguous(), per, 0, 0) for i in range(R)]
    order = [segs[r]] + [segs[i] for i in range(R) if i != r]
    one = torch.empty_like(qs)
    ops.attention(qs, one, H, SCALE, order)
    two = torch.empty_like(qs)
    state = ops.attention_state(per, H, DEV)
    ops.attention(qs, two, H, SCALE, order[:1], state=state, state_out=True)
    ops.attention(qs, two, H, SCALE, order[1:], state=state, state_in=True)
    assert torch.equal(one, two)
    # and a different segment order changes nothing beyond rounding
    full = _att(qs, k, vt)
    err = float((full.float() - one.float()).abs().max())
    assert err <= 2.0 ** -6 * float(full.float().abs().max())


def test_attention_full_size_sampled_rows_vs_fp64(built_lib, qkv):
    q, k, vt = qkv
    rows = torch.tensor([0, 1, 77, 4095, 163840, 327679], device=DEV)
    o = _att(q, k, vt)[rows].float().cpu()
    for h in (0, 7, 15):
        qh = q[rows, h * 64:(h + 1) * 64].double()
        kh = k[:, h * 64:(h + 1) * 64].double()
        p = torch.softmax(qh @ kh.t() * SCALE, dim=-1)
        ref = (p @ vt[h * 64:(h + 1) * 64].double().t()).cpu()
        got = o[:, h * 64:(h + 1) * 64].double()
        assert float((got - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max().clamp_min(1e-3)) + 2e-4, h


def test_gemm_full_size_checksum_of_checksums(built_lib):
    """fc2 shape of the fusion MLP at N = 320: (327 680 x 4096) x (4096 -> 1024), fp32 output."""
    M, K, N = T, 4096, 1024
    a = _mk((M, K), 10, 0.5)
    w = _mk((N, K), 11, 0.03)
    wp = ops.pack_linear_weight(w.float().cpu(), torch.bfloat16).to(DEV)
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(a, wp, out_f32=y)
    col = y.sum(0, dtype=torch.float64)                                   # checksum of the result
    ref = a.sum(0, dtype=torch.float64) @ w.double().t()                  # result of the checksum
    scale = float((a.abs().sum(0, dtype=torch.float64) @ w.double().abs().t()).max())  # magnitude of the summed terms
    assert float((col - ref).abs().max()) <= 1e-6 * scale
    # rows too: a sampled row block against fp64
    rows = torch.tensor([0, 5, 131071, 327679], device=DEV)
    ref_rows = a[rows].double() @ w.double().t()
    assert float((y[rows].double() - ref_rows).abs().max()) <= 2e-4 * float(ref_rows.abs().max())
