// xp body of f3r_attn_fwd: the lean softmax of attn_kernel (reference max inside the MFMA, lazy re-base, dot2 row sums,
// LDS-DMA staging) run as a software pipeline over HALF tiles (32 keys) with the issue order written out by hand.
//
// Why (tools/ubench/mfma_overlap.hip, MI355X): a wave issues in order; a 32x32x16 MFMA holds the matrix pipe for 32 cycles,
// during which the SAME wave can issue ~6 plain VALU instructions or ~2 v_exp_f32 + 2 plain ones for free.  In the phased body
// (Q K^T -> softmax -> P V) a wave spends ~1350 cycles per tile in its two MFMA clusters and ~1650 cycles in everything else,
// strictly one after the other (profiles/: variant 61), so two waves per SIMD keep the matrix pipe ~70 % busy at best.
// The lean softmax needs 4 VALU per MFMA (2 exp, 1 pack, 1 dot2) -- it fits into the MFMA shadows.  So iteration h
//   * issues the 8 + 8 MFMAs (+ 2 small bias steps) that do not depend on it:  P V of half h-1  and  Q K^T of half h+1,
//   * and puts one (exp, exp, pack, sum) group of the softmax of half h behind each of them,
//   * with the LDS fragment of slot i+2 requested in slot i,
// every slot closed by __builtin_amdgcn_sched_barrier(0) (the compiler's own schedule clusters the MFMAs).
//
// 64 queries per wave (two 32-query blocks share every LDS fragment: at one MFMA per fragment the LDS port would be the
// bound), 4 waves per workgroup, LDS ring of 4 [K | V^T] tile slots (64 KB, two workgroups per CU); during tile t the
// pipeline reads V^T(t-1), K/V^T(t), K(t+1) while tile t+2 lands.  One barrier per tile.
// Rare path (wave-uniform, taken when a half tile's row sum says P outgrew the reference, and once at the start):
// recompute S(h) from LDS, move the reference there, rescale O / l, shift S(h+1), redo P(h).
#pragma once

#define F3R_SB() __builtin_amdgcn_sched_barrier(0)

template <class T, int NW, int PROF, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel_xp(const f3r_attn_args p) {
  constexpr int QB = NW * 64;
  constexpr int DPW = 8 / NW;
  static_assert(NW == 4 || NW == 8, "xp body: 4 or 8 waves");
  typedef typename T::vec8 V8;
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * 2 * AT_TILE];  // 4 slots x [K | V^T] x 8 KB = 64 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 64;

  // ---- Q fragments, pre-multiplied by scale * log2(e) (the bias step needs scores in exp2 units)
  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow[2];
  bool q_ok[2];
  V8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = q0 + qb * 32 + lq;
    q_ok[qb] = qrow[qb] < p.tq;
    if (!q_ok[qb]) qrow[qb] = p.tq - 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + qrow[qb] * p.ldq + head * 64 + ds * 16 + g * 8);
      if (!p.q_prescaled) {
        const float cq = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * cq, hi_f<T>(raw[j]) * cq);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
  }

  int n_tiles = 0;
  for (int sg = 0; sg < p.n_seg; ++sg) n_tiles += (int)((p.seg_len[sg] + AT_KB - 1) / AT_KB);

  // ---- DMA tile walker (global_load_lds: wave-uniform tile base + lane-constant 32-bit offsets, see attn_kernel bit 6)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const int d_lrow = lane >> 3;
  const int64_t kstep_b = (int64_t)AT_KB * p.ldk * 2;
  int seg_ld = -1, rem_keys = 0, tiles_issued = 0;
  uint32_t seg_ldvt = 0;
  uint32_t dk_off[DPW], dv_off[DPW];
  const char* dKb = nullptr;
  const char* dVb = nullptr;
  auto dma_segment = [&]() {
    rem_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        rem_keys = (int)p.seg_len[seg_ld];
        seg_ldvt = (uint32_t)p.ldvt[seg_ld];
        dKb = (const char*)((const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64);
        dVb = (const char*)((const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt);
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
          const int row = (wid * DPW + i) * 8 + d_lrow;
          const uint32_t lch = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
          dk_off[i] = (uint32_t)row * (uint32_t)p.ldk * 2u + lch;
          dv_off[i] = (uint32_t)row * seg_ldvt * 2u + lch;
        }
        break;
      }
  };
  dma_segment();
  auto dma_tile = [&]() {  // next tile of the walk -> slot (tiles_issued & 3)
    if (tiles_issued >= n_tiles) return;
    const int valid = rem_keys < AT_KB ? rem_keys : AT_KB;
    uint16_t* kt = lds + (tiles_issued & 3) * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int blk = wid * DPW + i;
      uint32_t ko = dk_off[i];
      if (valid < AT_KB) {  // rows past the segment end: re-read the last valid row (masked in the softmax)
        const int row = blk * 8 + d_lrow;
        const int krow = row < valid ? row : valid - 1;
        ko = (uint32_t)krow * (uint32_t)p.ldk * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(dKb + ko), (lds_ptr_t)(kt + blk * 8 * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(dVb + dv_off[i]), (lds_ptr_t)(vt + blk * 8 * 64), 16, 0, 0);
    }
    dKb += kstep_b;
    dVb += AT_KB * 2;
    rem_keys -= AT_KB;
    ++tiles_issued;
    if (rem_keys <= 0) dma_segment();
  };
  // valid-key count of the tiles in consumption order
  int vs_seg = -1, vs_rem = 0;
  auto next_valid = [&]() -> int {
    if (vs_rem <= 0) {
      for (++vs_seg; vs_seg < p.n_seg && p.seg_len[vs_seg] <= 0; ++vs_seg) {}
      vs_rem = vs_seg < p.n_seg ? (int)p.seg_len[vs_seg] : 0;
    }
    const int v = vs_rem < AT_KB ? vs_rem : AT_KB;
    vs_rem -= AT_KB;
    return v;
  };

  // ---- LDS fragment offsets (elements) inside a tile image: K rows go through pi (swap bits 2, 3), see attn_kernel
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
  // aswz(row, 2 c + g) = aswz(row, g) ^ (c << 4) in elements: two lane constants serve all eight fragment addresses
  const int koff0 = aswz(krow_pi, g);  // K fragment of k-step c: koff0 ^ (c << 4);  + 32 * 64 for the second key half
  const int voff0 = aswz(lq, g);       // V^T fragment c = 2 * (key half) + ks: voff0 ^ (c << 4);  + 32 * 64 for d >= 32
  auto koff = [&](int c) -> int { return koff0 ^ (c << 4); };
  auto voff = [&](int c) -> int { return voff0 ^ (c << 4); };

  // ---- running state
  float16v o[2][2];
  float m_run[2], l_run[2];
  u32x2 mfrag[2];
  const u32x2 onesfrag = {g == 0 ? pack2<T>(1.0f, 1.0f) : 0u, 0u};
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
    m_run[qb] = 0.f;
    l_run[qb] = 0.f;
    mfrag[qb] = u32x2{0u, 0u};
    if (p.state_in) {
      const int64_t row = (int64_t)b * p.tq + qrow[qb];
      const float* so = p.st_o + row * ((int64_t)p.n_heads * 64) + head * 64;
      const float* sm = p.st_ml + (row * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4v v = *(const float4v*)(so + db * 32 + 8 * rq + 4 * g);
          o[qb][db][rq * 4 + 0] = v[0]; o[qb][db][rq * 4 + 1] = v[1]; o[qb][db][rq * 4 + 2] = v[2]; o[qb][db][rq * 4 + 3] = v[3];
        }
      m_run[qb] = sm[0];
      l_run[qb] = sm[1 + g];
      const float h = from_lp<T>(to_lp<T>(m_run[qb]));
      if (g == 0) mfrag[qb][0] = pack2<T>(-h, -(m_run[qb] - h));
    }
  }

  constexpr float REBASE_SUM = 64.f;  // a lane's 16-key partial sum of a half tile: below it every P < 64

  // scores of one half: bias step + 4 k-steps (used by the prologue and by the rare path)
  auto qk_half = [&](const uint16_t* kt_half, float16v (&s)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float16v z;
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = 0.f;
      s[qb] = T::mfma32k8(onesfrag, mfrag[qb], z);
    }
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const V8 a = as_vec8<T>(*(const u32x4*)(kt_half + koff(ds)));
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s[qb] = T::mfma32(a, qf[qb][ds], s[qb]);
    }
  };
  // register r of a half's score block is key  KB*32 + 16*(r>>3) + 8*g + (r&7)  of the tile
  auto mask_half = [&](float16v (&s)[2], int valid, int kb) {
    const int vg = valid - kb * 32 - 8 * g;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = 16 * (r >> 3) + (r & 7);
        if (kc >= vg) s[qb][r] = -1e30f;
      }
  };

  // ---- one pipeline stage: softmax of half h = 2 t + KB;  MFMAs of P V (h-1) and Q K^T (h+1)
  // sc: S(h) (consumed), sn: receives S(h+1), pp: P(h-1) as [qb][ks], pc: receives P(h)
  auto stage = [&](auto pv_tag, auto qk_tag, auto kb_tag, float16v (&sc)[2], float16v (&sn)[2], V8 (&pp)[2][2], V8 (&pc)[2][2], const int t,
                   const int valid, const bool forced) {
    constexpr bool HAS_PV = decltype(pv_tag)::value;
    constexpr bool HAS_QK = decltype(qk_tag)::value;
    constexpr int KB = decltype(kb_tag)::value;
    if (valid < AT_KB) mask_half(sc, valid, KB);
    const uint16_t* kt = lds + ((KB ? t + 1 : t) & 3) * 2 * AT_TILE + (KB ^ 1) * 32 * 64;  // K rows of half h+1
    const uint16_t* vt = lds + ((KB ? t : t + 3) & 3) * 2 * AT_TILE + AT_TILE;            // V^T tile of half h-1 (its key half: KB^1)
    // slot i < 4: P V fragment (db = i & 1, ks = i >> 1);  slot i >= 4: K fragment of k-step i - 4.  P V first: S(h+1) only comes
    // to life when half of S(h) is already consumed and P(h-1) is dead (24 fewer live registers than interleaving the two)
    auto slot_on = [&](int i) -> bool { return i >= 4 ? HAS_QK : HAS_PV; };
    auto slot_ptr = [&](int i) -> const u32x4* {
      if (i >= 4) return (const u32x4*)(kt + koff(i - 4));
      return (const u32x4*)(vt + (i & 1) * 32 * 64 + voff((KB ^ 1) * 2 + (i >> 1)));
    };
    u32x4 fr[8];
    constexpr int PF = 1;  // fragment prefetch distance in slots (one slot = 2 MFMAs ~ 70-100 cycles ~ the LDS latency)
    if (slot_on(0)) fr[0] = *slot_ptr(0);
    if (PF > 1 && slot_on(1)) fr[1] = *slot_ptr(1);
    F3R_SB();
    u32x4 pw[2][2];
    float ps[2] = {0.f, 0.f};
    float e0p = 0.f, e1p = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + PF < 8 && slot_on(i + PF)) fr[i + PF] = *slot_ptr(i + PF);
      if (i == 4 && HAS_QK) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          float16v z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sn[qb] = T::mfma32k8(onesfrag, mfrag[qb], z);
        }
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (i >= 4) {
          if (HAS_QK) sn[qb] = T::mfma32(as_vec8<T>(fr[i]), qf[qb][i - 4], sn[qb]);
        } else {
          if (HAS_PV) o[qb][i & 1] = T::mfma32(as_vec8<T>(fr[i]), pp[qb][i >> 1], o[qb][i & 1]);
        }
        // shadow k: exps of pair k, pack + row sum of pair k - 1
        const int k = 2 * i + qb;
        const int sq = k >> 3, j = k & 7;
        const float e0 = __builtin_amdgcn_exp2f(sc[sq][2 * j]);
        const float e1 = __builtin_amdgcn_exp2f(sc[sq][2 * j + 1]);
        if (k > 0) {
          const int kp = k - 1, sp = kp >> 3, jp = kp & 7;
          const uint32_t w = pack2<T>(e0p, e1p);
          pw[sp][jp >> 2][jp & 3] = w;
          ps[sp] = T::sum2(w, ps[sp]);
          asm volatile("" : "+v"(pw[sp][jp >> 2][jp & 3]), "+v"(ps[sp]));
        }
        e0p = e0;
        e1p = e1;
        asm volatile("" : "+v"(e0p), "+v"(e1p));  // anchor the slice in its slot (IR-level sinking ignores sched_barrier)
        F3R_SB();
      }
    }
    {
      const uint32_t w = pack2<T>(e0p, e1p);
      pw[1][1][3] = w;
      ps[1] = T::sum2(w, ps[1]);
    }
    float psum[2] = {ps[0], ps[1]};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      pc[qb][0] = as_vec8<T>(pw[qb][0]);
      pc[qb][1] = as_vec8<T>(pw[qb][1]);
    }
    // ---- rare: move the reference (wave-uniform)
    if (forced || __any(psum[0] >= REBASE_SUM || psum[1] >= REBASE_SUM)) {
      float16v sr[2];
      qk_half(lds + (t & 3) * 2 * AT_TILE + KB * 32 * 64, sr);
      if (valid < AT_KB) mask_half(sr, valid, KB);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx = sr[qb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sr[qb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float target = m_run[qb] + (forced ? mx : fmaxf(mx, 0.f));
        const float nh = from_lp<T>(to_lp<T>(target));
        const float nl = from_lp<T>(to_lp<T>(target - nh));
        const float delta = (nh + nl) - m_run[qb];
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[qb] = nh + nl;
        if (g == 0) mfrag[qb][0] = pack2<T>(-nh, -nl);
        l_run[qb] *= alpha;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o[qb][0][i] *= alpha;
          o[qb][1][i] *= alpha;
          if (HAS_QK) sn[qb][i] -= delta;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pk[j] = pack2<T>(__builtin_amdgcn_exp2f(sr[qb][ks * 8 + 2 * j] - delta), __builtin_amdgcn_exp2f(sr[qb][ks * 8 + 2 * j + 1] - delta));
            if (j & 1) a1 = T::sum2(pk[j], a1); else a0 = T::sum2(pk[j], a0);
          }
          pc[qb][ks] = as_vec8<T>(pk);
        }
        psum[qb] = a0 + a1;
      }
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];
  };

  // ---- per tile: make tile t+1 visible to every wave, release the slot of tile t-2, start tile t+2
  auto top = [&]() {
    __syncthreads();  // drains this wave's DMA (vmcnt) before the barrier
    dma_tile();
  };

  const std::true_type yes{};
  const std::false_type no{};
  const std::integral_constant<int, 0> kb0{};
  const std::integral_constant<int, 1> kb1{};
  float16v sA[2], sB[2];
  V8 pA[2][2], pB[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      pA[qb][ks] = as_vec8<T>(z);
      pB[qb][ks] = as_vec8<T>(z);
    }

  unsigned long long t_loop0 = 0;
  dma_tile();
  dma_tile();
  top();  // tiles 0 (and 1) landed; tile 2 on its way
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing the loop waits for later may still be pending from the prologue (Q loads)
  if (PROF) t_loop0 = __builtin_readcyclecounter();
  qk_half(lds, sA);
  int valid = next_valid();
  const bool forced0 = !p.state_in;
  if (n_tiles == 1) {
    stage(no, yes, kb0, sA, sB, pB, pA, 0, valid, forced0);
    stage(yes, no, kb1, sB, sA, pA, pB, 0, valid, false);
  } else {
    stage(no, yes, kb0, sA, sB, pB, pA, 0, valid, forced0);
    stage(yes, yes, kb1, sB, sA, pA, pB, 0, valid, false);
    int t = 1;
    for (; t < n_tiles - 1; ++t) {
      top();
      valid = next_valid();
      stage(yes, yes, kb0, sA, sB, pB, pA, t, valid, false);
      stage(yes, yes, kb1, sB, sA, pA, pB, t, valid, false);
    }
    top();
    valid = next_valid();
    stage(yes, yes, kb0, sA, sB, pB, pA, t, valid, false);
    stage(yes, no, kb1, sB, sA, pA, pB, t, valid, false);
  }
  // ---- drain: P V of the last half (P in pB, second key half of the last tile)
  {
    const uint16_t* vt = lds + ((n_tiles - 1) & 3) * 2 * AT_TILE + AT_TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const V8 a = as_vec8<T>(*(const u32x4*)(vt + db * 32 * 64 + voff(2 + ks)));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[qb][db] = T::mfma32(a, pB[qb][ks], o[qb][db]);
      }
  }
  if (PROF && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    g_attn_prof[0] = 0; g_attn_prof[1] = 0; g_attn_prof[2] = 0; g_attn_prof[3] = 0; g_attn_prof[4] = (unsigned long long)n_tiles;
    g_attn_prof[5] = __builtin_readcyclecounter() - t_loop0;
  }

  // ---- epilogue (as attn_kernel): hand the state over, or normalise and store.  Row indices are recomputed here instead of
  // being carried through the loop in four registers.
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = (int64_t)blockIdx.x * QB + wid * 64 + qb * 32 + lq;
    q_ok[qb] = qrow[qb] < p.tq;
  }
  asm volatile("" : "+v"(qrow[0]), "+v"(qrow[1]));
  if (p.state_out) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (!q_ok[qb]) continue;
      const int64_t row = (int64_t)b * p.tq + qrow[qb];
      float* so = p.st_o + row * ((int64_t)p.n_heads * 64) + head * 64;
      float* sm = p.st_ml + (row * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float4v v = {o[qb][db][rq * 4 + 0], o[qb][db][rq * 4 + 1], o[qb][db][rq * 4 + 2], o[qb][db][rq * 4 + 3]};
          *(float4v*)(so + db * 32 + 8 * rq + 4 * g) = v;
        }
      if (g == 0) sm[0] = m_run[qb];
      sm[1 + g] = l_run[qb];
    }
    return;
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_ok[qb]) {
      uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow[qb] * p.ldo + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 w;
          w[0] = pack2<T>(o[qb][db][rq * 4 + 0] * inv, o[qb][db][rq * 4 + 1] * inv);
          w[1] = pack2<T>(o[qb][db][rq * 4 + 2] * inv, o[qb][db][rq * 4 + 3] * inv);
          *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
        }
    }
  }
}

template <class T, int NW, int PROF, int MINW>
int attn_launch_xp(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * 64;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  for (int sg = 0; sg < a.n_seg; ++sg) F3R_REQUIRE(a.seg_len[sg] < (1ll << 31), "f3r_attn_fwd: segment of %lld keys", (long long)a.seg_len[sg]);
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_xp<T, NW, PROF, MINW>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}
