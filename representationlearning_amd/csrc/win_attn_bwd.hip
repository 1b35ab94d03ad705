// Fused 7x7-window cross attention, backward (autograd of the forward in win_attn_fwd.hip; reference:
// torch autograd through InterlacedPoolAttention2.forward :164-188 and Mhca, modules/DAL.py:873-1020,
// incl. the three gradient paths into q/k — softmax, mean(M), argmax-routed max(M) — SURVEY App. C).
//
// One wavefront per window.  The forward is recomputed from (x, y, LN stats, omega) so nothing but the
// block inputs is saved; HBM traffic = read x, y, dout + write dxhat, dyhat.  All operands that are reused
// with a different contraction axis are staged in LDS twice (token-major for K = channels, channel-major
// for K = tokens); softmax / dS tiles stay in registers and are computed in both orientations instead of
// being transposed through LDS.  Weight gradients are accumulated in LDS per workgroup and flushed once.
#include "win_attn.cuh"
using namespace rssf;
using namespace rssf::wa;

namespace {

template <typename T, typename DM, bool ACC_LDS> struct BwdLayout {
  static constexpr int P = Pad<T>::X;
  static constexpr int LDX = DM::CP + P;     // XS/YS/GS   [token][in channel]
  static constexpr int LDV = DM::CV + P;     // QS/KS/VS   [token][virtual channel]
  static constexpr int LDT = LP + P;         // *T buffers [virtual channel][token]
  static constexpr int LDW = DM::CP + P;     // Wq/Wk/Wv/WoT rows = virtual channel, k = real channel
  static constexpr int LDM = DM::CV + P;     // WqT/WkT/WvT  rows = real channel,    k = virtual channel
  static constexpr int LDD = DM::DP + P;     // dM / dM^T scratch
  static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
  static constexpr int REGION = (max3(LP * LDX, LP * LDV, DM::CV * LDT) + 7) / 8 * 8;
  static constexpr int NREG = 11;
  static constexpr int SCRATCH = (2 * DM::DP * LDD + 7) / 8 * 8;
  static constexpr int W_ELEMS = 4 * DM::CV * LDW + 3 * DM::CP * LDM;
  static constexpr int F_ELEMS = 3 * DM::CV + 2 * DM::CP;                       // bq bk bv, gamma beta
  static constexpr int A_ELEMS = ACC_LDS ? 4 * DM::CV * DM::CP + 3 * DM::CV + DM::CP : 0;
  static constexpr size_t SHARED_OFF = (sizeof(T) * W_ELEMS + sizeof(float) * (F_ELEMS + A_ELEMS) + 15) / 16 * 16;
  static constexpr size_t WAVE_BYTES = sizeof(T) * (NREG * REGION + SCRATCH);
  static constexpr size_t LIM = 160 * 1024;
  static constexpr int WAVES = (SHARED_OFF + 4 * WAVE_BYTES <= LIM) ? 4 : (SHARED_OFF + 3 * WAVE_BYTES <= LIM) ? 3
                             : (SHARED_OFF + 2 * WAVE_BYTES <= LIM) ? 2 : 1;
  static constexpr size_t BYTES = SHARED_OFF + WAVE_BYTES * WAVES;
  static constexpr bool FITS = BYTES <= LIM;
};

// plain (no LN / gate) token-major tile load, zero for pad / dead slots
template <typename T, typename DM>
__device__ __forceinline__ void load_plain_tile(const Geom& g, const T* src, int64_t img, int qh, int qw, T* dst, int ldx,
                                                int lane) {
  constexpr int V = Vec<T>::N;
  if constexpr (DM::C % V == 0) {
    constexpr int cpr = DM::CP / V;
    for (int e = lane; e < LP * cpr; e += 64) {
      const int t = e / cpr, c0 = (e % cpr) * V;
      const int n = slot_token(g, qh, qw, t);
      Vec<T> o;
      o.raw = {0, 0, 0, 0};
      if (n >= 0 && c0 < DM::C) o.load(src + (img + n) * DM::C + c0);
      o.store(dst + t * ldx + c0);
    }
  } else {
    for (int e = lane; e < LP * DM::CP; e += 64) {
      const int t = e / DM::CP, c = e % DM::CP;
      const int n = slot_token(g, qh, qw, t);
      stf(dst + t * ldx + c, (n >= 0 && c < DM::C) ? ldf(src + (img + n) * DM::C + c) : 0.f);
    }
  }
}

// C-layout tile (rows = virtual channel mt*16+4g+r, col = token tt*16+l15) -> token-major buffer [token][ld]
template <typename T>
__device__ __forceinline__ void store_tok_major(T* buf, int ld, int mt, int tt, const f32x4& v, int l15, int grp) {
  T* p = buf + (tt * 16 + l15) * ld + mt * 16 + grp * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) stf(p + r, v[r]);
}
// ... -> channel-major buffer [channel][ldt]; dead tokens (>= L) written as zero
template <typename T>
__device__ __forceinline__ void store_ch_major(T* buf, int ldt, int mt, int tt, const f32x4& v, int l15, int grp, int L) {
  const int tok = tt * 16 + l15;
  const bool live = tok < L;
#pragma unroll
  for (int r = 0; r < 4; ++r) stf(buf + (mt * 16 + grp * 4 + r) * ldt + tok, live ? v[r] : 0.f);
}
// B-operand "chain" tile gathered from a token-major LDS buffer: rows (k-slots) = tokens t0+4g+r, col = channel c
template <typename T>
__device__ __forceinline__ f32x4 gather_rows(const T* buf, int ld, int t0, int c, int grp) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = ldf(buf + (t0 + grp * 4 + r) * ld + c);
  return v;
}

template <typename T, typename DM, bool ACC_LDS>
__global__ void __launch_bounds__((BwdLayout<T, DM, ACC_LDS>::WAVES * 64))
winattn_bwd_kernel(rssf_winattn_bwd_params bp, Geom g) {
  using LY = BwdLayout<T, DM, ACC_LDS>;
  constexpr int LDX = LY::LDX, LDV = LY::LDV, LDT = LY::LDT, LDW = LY::LDW, LDM = LY::LDM, LDD = LY::LDD;
  constexpr int C = DM::C, CP = DM::CP, CV = DM::CV, MT = DM::MT, CT = DM::CT, TPH = DM::TPH, D = DM::D, DP = DM::DP;
  const rssf_winattn_fwd_params& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* sWq = reinterpret_cast<T*>(smem_raw);       // [CV][LDW]
  T* sWk = sWq + CV * LDW;
  T* sWv = sWk + CV * LDW;
  T* sWoT = sWv + CV * LDW;                      // [CV][LDW]  WoT[m][c] = Wo[c][m]
  T* sWqT = sWoT + CV * LDW;                     // [CP][LDM]  WqT[c][m] = Wq[m][c]
  T* sWkT = sWqT + CP * LDM;
  T* sWvT = sWkT + CP * LDM;
  float* sB = reinterpret_cast<float*>(sWvT + CP * LDM);   // bq bk bv [CV]
  float* sLn = sB + 3 * CV;                      // gamma, beta [CP]
  float* aW = sLn + 2 * CP;                      // accumulators: dWq dWk dWv [CV][CP], dWo^T [CV][CP], dbq dbk dbv [CV], dbo [CP]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, grp = lane >> 4;
  T* base = reinterpret_cast<T*>(smem_raw + LY::SHARED_OFF) + (size_t)wave * (LY::NREG * LY::REGION + LY::SCRATCH);
  T* XS = base;                 T* YS = XS + LY::REGION;   T* GS = YS + LY::REGION;
  T* QS = GS + LY::REGION;      T* KS = QS + LY::REGION;   T* VS = KS + LY::REGION;
  T* QT = VS + LY::REGION;      T* KT = QT + LY::REGION;   T* VT = KT + LY::REGION;   // later dq^T, dk^T, dv^T
  T* OT = VT + LY::REGION;      T* DUT = OT + LY::REGION;
  T* dMs = DUT + LY::REGION;    T* dMTs = dMs + DP * LDD;

  for (int i = threadIdx.x; i < CV * LDW; i += blockDim.x) {
    const int m = i / LDW, k = i % LDW;
    const int rc = real_ch<DM>(m);
    const bool ok = rc >= 0 && k < C;
    stf(sWq + i, ok ? p.wq[rc * C + k] : 0.f);
    stf(sWk + i, ok ? p.wk[rc * C + k] : 0.f);
    stf(sWv + i, ok ? p.wv[rc * C + k] : 0.f);
    stf(sWoT + i, ok ? p.wo[k * C + rc] : 0.f);
  }
  for (int i = threadIdx.x; i < CP * LDM; i += blockDim.x) {
    const int c = i / LDM, m = i % LDM;
    const int rc = m < CV ? real_ch<DM>(m) : -1;
    const bool ok = rc >= 0 && c < C;
    stf(sWqT + i, ok ? p.wq[rc * C + c] : 0.f);
    stf(sWkT + i, ok ? p.wk[rc * C + c] : 0.f);
    stf(sWvT + i, ok ? p.wv[rc * C + c] : 0.f);
  }
  for (int i = threadIdx.x; i < CV; i += blockDim.x) {
    const int rc = real_ch<DM>(i);
    sB[i] = rc >= 0 ? p.bq[rc] : 0.f;
    sB[CV + i] = rc >= 0 ? p.bk[rc] : 0.f;
    sB[2 * CV + i] = rc >= 0 ? p.bv[rc] : 0.f;
  }
  for (int i = threadIdx.x; i < CP; i += blockDim.x) {
    sLn[i] = i < C ? p.ln_gamma[i] : 0.f;
    sLn[CP + i] = i < C ? p.ln_beta[i] : 0.f;
  }
  if (ACC_LDS) for (int i = threadIdx.x; i < LY::A_ELEMS; i += blockDim.x) aW[i] = 0.f;
  __syncthreads();

  // accumulate one weight-gradient element: which = 0 q, 1 k, 2 v (index [m][c]); 3 = o (index [m][c] meaning dWo[c][m])
  auto acc_w = [&](int which, int m, int c, float v) {
    if (ACC_LDS) { atomicAdd(&aW[which * CV * CP + m * CP + c], v); return; }
    const int rc = real_ch<DM>(m);
    if (rc < 0 || c >= C) return;
    float* dst = which == 0 ? bp.dwq : which == 1 ? bp.dwk : which == 2 ? bp.dwv : bp.dwo;
    atomicAdd(which == 3 ? dst + c * C + rc : dst + rc * C + c, v);
  };
  auto acc_b = [&](int which, int idx, float v) {   // 0..2: dbq/dbk/dbv[virtual m]; 3: dbo[c]
    if (ACC_LDS) { atomicAdd(&aW[4 * CV * CP + (which < 3 ? which * CV + idx : 3 * CV + idx)], v); return; }
    if (which == 3) { if (idx < C) atomicAdd(bp.dbo + idx, v); return; }
    const int rc = real_ch<DM>(idx);
    if (rc < 0) return;
    atomicAdd((which == 0 ? bp.dbq : which == 1 ? bp.dbk : bp.dbv) + rc, v);
  };

  const float scale = rsqrtf((float)D);
  const T* X = reinterpret_cast<const T*>(p.x);
  const T* Y = reinterpret_cast<const T*>(p.y);
  const T* DOUT = reinterpret_cast<const T*>(bp.dout);
  T* DXH = reinterpret_cast<T*>(bp.dxhat);
  T* DYH = reinterpret_cast<T*>(bp.dyhat);
  const int wpi = g.QH * g.QW;

  for (int wi = blockIdx.x * LY::WAVES + wave; wi < g.nWin; wi += gridDim.x * LY::WAVES) {
    const int b = wi / wpi, qh = (wi % wpi) / g.QW, qw = wi % g.QW;
    const int64_t img = (int64_t)b * g.N;
    const float* om0 = p.omega + (int64_t)b * 2 * g.N;
    float* dom0 = bp.domega + (int64_t)b * 2 * g.N;

    // ---- S1: gated LN'ed inputs and dout tile, token-major -------------------------------------------------
    wave_sync();
    load_gated_tiles<T, DM>(p, g, sLn, X, Y, om0, img, qh, qw, XS, YS, LDX, lane);
    load_plain_tile<T, DM>(g, DOUT, img, qh, qw, GS, LDX, lane);
    wave_sync();

    // ---- S2: projections; stage q,k,v both token-major (K = channels) and channel-major (K = tokens) --------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mrow = mt * 16 + grp * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
        aq = mma_tile<T>(sWq + mt * 16 * LDW, LDW, XS + tt * 16 * LDX, LDX, CP, aq);
        ak = mma_tile<T>(sWk + mt * 16 * LDW, LDW, YS + tt * 16 * LDX, LDX, CP, ak);
        av = mma_tile<T>(sWv + mt * 16 * LDW, LDW, YS + tt * 16 * LDX, LDX, CP, av);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          aq[r] = (aq[r] + sB[mrow + r]) * scale;
          ak[r] += sB[CV + mrow + r];
          av[r] += sB[2 * CV + mrow + r];
        }
        store_tok_major<T>(QS, LDV, mt, tt, aq, l15, grp);
        store_tok_major<T>(KS, LDV, mt, tt, ak, l15, grp);
        store_tok_major<T>(VS, LDV, mt, tt, av, l15, grp);
        store_ch_major<T>(QT, LDT, mt, tt, aq, l15, grp, g.L);
        store_ch_major<T>(KT, LDT, mt, tt, ak, l15, grp, g.L);
        store_ch_major<T>(VT, LDT, mt, tt, av, l15, grp, g.L);
      }
    }
    wave_sync();

    // dbo += sum_tokens dout   (columns of GS)
    for (int c = lane; c < CP; c += 64) {
      float s = 0.f;
      for (int t = 0; t < g.L; ++t) s += ldf(GS + t * LDX + c);
      acc_b(3, c, s);
    }

    f32x4 dxt[CT][NT], dyt[CT][NT];     // d(x~)^T, d(y~)^T accumulators: rows = real channel, col = token
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) { dxt[ct][tt] = {0.f, 0.f, 0.f, 0.f}; dyt[ct][tt] = {0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
    for (int h = 0; h < DM::HEADS; ++h) {
      const int hoff = h * DP;
      // ---- S3: alpha = sigmoid(mean(M)+max(M)), M = q_h^T k_h, with its argmax ---------------------------------
      float msum = 0.f, mmax = -INFINITY;
      int marg = 0;
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<T>(QT + (hoff + it * 16) * LDT, LDT, KT + (hoff + jt * 16) * LDT, LDT, LP, acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            if (i < D && j < D) {
              msum += acc[r];
              if (acc[r] > mmax) { mmax = acc[r]; marg = i * DP + j; }
            }
          }
        }
      msum = wave_sum(msum);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mmax, o, 64);
        const int oa = __shfl_xor(marg, o, 64);
        if (om > mmax || (om == mmax && oa < marg)) { mmax = om; marg = oa; }
      }
      const float alpha = sigmoidf(msum / (float)(D * D) + mmax);

      // dO^T (head rows) = WoT * dout^T ; dU = alpha * dO
      f32x4 dU[TPH][NT];
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          dU[mi][qt] = mma_tile<T>(sWoT + (hoff + mi * 16) * LDW, LDW, GS + qt * 16 * LDX, LDX, CP, acc);
        }

      // ---- S4: orientation 1 (rows = keys, col = query): P, U = P v, dalpha, dS -> dq -------------------------
      f32x4 dq[TPH][NT], dk[TPH][NT], dv[TPH][NT];
      float smx[NT], sinv[NT], srs[NT];
      float dalpha = 0.f;
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          s[kt] = mma_tile<T>(KS + kt * 16 * LDV + hoff, LDV, QS + qt * 16 * LDV + hoff, LDV, DP, acc);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kt * 16 + grp * 4 + r >= g.L) s[kt][r] = -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
        smx[qt] = mx; sinv[qt] = inv;
        // U^T = v^T P^T ; O = alpha U -> OT ; dalpha += <dO, U> ; dU = alpha dO
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) u = mma_lds_chain<T>(VT + (hoff + mi * 16) * LDT, LDT, kt * 16, s[kt], u);
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dalpha += dU[mi][qt][r] * u[r];
            o[r] = alpha * u[r];
            dU[mi][qt][r] *= alpha;
          }
          store_ch_major<T>(OT, LDT, h * TPH + mi, qt, o, l15, grp, g.L);
          store_ch_major<T>(DUT, LDT, h * TPH + mi, qt, dU[mi][qt], l15, grp, g.L);
        }
        // dP^T[key][query] = sum_m v[key][m] dU[query][m]
        f32x4 dp[NT];
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) acc = mma_lds_chain<T>(VS + kt * 16 * LDV, LDV, hoff + mi * 16, dU[mi][qt], acc);
          dp[kt] = acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) rs += s[kt][r] * acc[r];
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        srs[qt] = rs;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[kt][r] = s[kt][r] * (dp[kt][r] - rs);       // dS^T
        // dq^T[m][query] = sum_key k^T[m][key] dS^T[key][query]
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) acc = mma_lds_chain<T>(KT + (hoff + mi * 16) * LDT, LDT, kt * 16, dp[kt], acc);
          dq[mi][qt] = acc;
        }
      }
      dalpha = wave_sum(dalpha);            // each tile element lives in exactly one lane: plain sum
      wave_sync();                          // OT / DUT rows of this head complete

      // dWo[c][m] (m in this head) = sum_t dout[t][c] O[t][m]
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            acc = mma_lds_chain<T>(OT + (hoff + mi * 16) * LDT, LDT, tt * 16, gather_rows<T>(GS, LDX, tt * 16, ct * 16 + l15, grp), acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_w(3, hoff + mi * 16 + grp * 4 + r, ct * 16 + l15, acc[r]);
        }

      // ---- S5: orientation 2 (rows = queries, col = key): dS -> dk, P -> dv ---------------------------------------
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        f32x4 p2[NT], ds2[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<T>(QS + qt * 16 * LDV + hoff, LDV, KS + kt * 16 * LDV + hoff, LDV, DP, acc);
          f32x4 dpp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mi = 0; mi < TPH; ++mi) dpp = mma_chain_lds<T>(dU[mi][qt], VS + kt * 16 * LDV, LDV, hoff + mi * 16, dpp);
          const bool keylive = kt * 16 + l15 < g.L;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int src = grp * 4 + r;                      // lane (in group 0) that owns query qt*16 + 4g + r
            const float mxr = __shfl(smx[qt], src, 64), invr = __shfl(sinv[qt], src, 64), rsr = __shfl(srs[qt], src, 64);
            const float pv = keylive ? __expf(acc[r] - mxr) * invr : 0.f;
            p2[qt][r] = pv;
            ds2[qt][r] = pv * (dpp[r] - rsr);
          }
        }
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = ak;
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            ak = mma_lds_chain<T>(QT + (hoff + mi * 16) * LDT, LDT, qt * 16, ds2[qt], ak);
            av = mma_lds_chain<T>(DUT + (hoff + mi * 16) * LDT, LDT, qt * 16, p2[qt], av);
          }
          dk[mi][kt] = ak; dv[mi][kt] = av;
        }
      }

      // ---- S6: alpha path: dM = du * (1/d^2 + onehot(argmax)) -> dq += k dM^T, dk += q dM ---------------------------
      const float du = dalpha * alpha * (1.f - alpha);
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            float v = 0.f;
            if (i < D && j < D) v = du / (float)(D * D) + ((i * DP + j) == marg ? du : 0.f);
            stf(dMs + i * LDD + j, v);
            stf(dMTs + j * LDD + i, v);
          }
      wave_sync();
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int mi = 0; mi < TPH; ++mi) {
          f32x4 aq = dq[mi][tt], ak = dk[mi][tt];
#pragma unroll
          for (int mj = 0; mj < TPH; ++mj) {
            // k^T / q^T tiles (rows = channel mj*16+4g+r, col = token) gathered from the token-major copies
            f32x4 kt4, qt4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              kt4[r] = ldf(KS + (tt * 16 + l15) * LDV + hoff + mj * 16 + grp * 4 + r);
              qt4[r] = ldf(QS + (tt * 16 + l15) * LDV + hoff + mj * 16 + grp * 4 + r);
            }
            aq = mma_lds_chain<T>(dMs + mi * 16 * LDD, LDD, mj * 16, kt4, aq);
            ak = mma_lds_chain<T>(dMTs + mi * 16 * LDD, LDD, mj * 16, qt4, ak);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) aq[r] *= scale;       // q = (W x + b) * scale
          dq[mi][tt] = aq; dk[mi][tt] = ak;
        }
      wave_sync();   // every read of QT/KT/VT rows of this head is done -> reuse them for dq^T, dk^T, dv^T

      // ---- S7: stage gradients of the projection outputs; bias grads; input grads --------------------------------
#pragma unroll
      for (int mi = 0; mi < TPH; ++mi) {
        float sq[4] = {0.f, 0.f, 0.f, 0.f}, sk[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          store_ch_major<T>(QT, LDT, h * TPH + mi, tt, dq[mi][tt], l15, grp, g.L);
          store_ch_major<T>(KT, LDT, h * TPH + mi, tt, dk[mi][tt], l15, grp, g.L);
          store_ch_major<T>(VT, LDT, h * TPH + mi, tt, dv[mi][tt], l15, grp, g.L);
          const bool live = tt * 16 + l15 < g.L;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!live) { dq[mi][tt][r] = 0.f; dk[mi][tt][r] = 0.f; dv[mi][tt][r] = 0.f; }
            sq[r] += dq[mi][tt][r]; sk[r] += dk[mi][tt][r]; sv[r] += dv[mi][tt][r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            sq[r] += __shfl_xor(sq[r], o, 64); sk[r] += __shfl_xor(sk[r], o, 64); sv[r] += __shfl_xor(sv[r], o, 64);
          }
          if (l15 == 0) {
            const int m = hoff + mi * 16 + grp * 4 + r;
            acc_b(0, m, sq[r]); acc_b(1, m, sk[r]); acc_b(2, m, sv[r]);
          }
        }
        // d(x~)^T[c][t] += sum_m Wq[m][c] dq[m][t] ;  d(y~)^T += Wk^T dk + Wv^T dv
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            dxt[ct][tt] = mma_lds_chain<T>(sWqT + ct * 16 * LDM, LDM, hoff + mi * 16, dq[mi][tt], dxt[ct][tt]);
            dyt[ct][tt] = mma_lds_chain<T>(sWkT + ct * 16 * LDM, LDM, hoff + mi * 16, dk[mi][tt], dyt[ct][tt]);
            dyt[ct][tt] = mma_lds_chain<T>(sWvT + ct * 16 * LDM, LDM, hoff + mi * 16, dv[mi][tt], dyt[ct][tt]);
          }
      }
    }
    wave_sync();

    // ---- S8: weight gradients dW[m][c] = sum_t d(proj)^T[m][t] * in[t][c] -------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const f32x4 xg = gather_rows<T>(XS, LDX, tt * 16, ct * 16 + l15, grp);
          const f32x4 yg = gather_rows<T>(YS, LDX, tt * 16, ct * 16 + l15, grp);
          aq = mma_lds_chain<T>(QT + mt * 16 * LDT, LDT, tt * 16, xg, aq);
          ak = mma_lds_chain<T>(KT + mt * 16 * LDT, LDT, tt * 16, yg, ak);
          av = mma_lds_chain<T>(VT + mt * 16 * LDT, LDT, tt * 16, yg, av);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mt * 16 + grp * 4 + r, c = ct * 16 + l15;
          acc_w(0, m, c, aq[r]); acc_w(1, m, c, ak[r]); acc_w(2, m, c, av[r]);
        }
      }

    // ---- S9: dxhat = d(x~) * omega0, dyhat = d(y~) * omega1 ; domega += d(x~) * LN(x) ----------------------------------
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int n = slot_token(g, qh, qw, tt * 16 + l15);
      if (n < 0) continue;
      const float2 sx = *reinterpret_cast<const float2*>(p.stats_x + (img + n) * 2);
      const float2 sy = *reinterpret_cast<const float2*>(p.stats_y + (img + n) * 2);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = ct * 16 + grp * 4 + r;
          if (c >= C) continue;
          const int64_t f = (int64_t)n * C + c;
          const int pp = (int)(f % g.N);
          const int64_t off = img * C + f;
          const float xh = (ldf(X + off) - sx.x) * sx.y * sLn[c] + sLn[CP + c];
          const float yh = (ldf(Y + off) - sy.x) * sy.y * sLn[c] + sLn[CP + c];
          stf(DXH + off, dxt[ct][tt][r] * om0[pp]);
          stf(DYH + off, dyt[ct][tt][r] * om0[g.N + pp]);
          atomicAdd(dom0 + pp, dxt[ct][tt][r] * xh);
          atomicAdd(dom0 + g.N + pp, dyt[ct][tt][r] * yh);
        }
    }
  }

  if (ACC_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * CV * CP; i += blockDim.x) {
      const int which = i / (CV * CP), m = (i / CP) % CV, c = i % CP;
      const int rc = real_ch<DM>(m);
      if (rc < 0 || c >= C) continue;
      float* dst = which == 0 ? bp.dwq : which == 1 ? bp.dwk : which == 2 ? bp.dwv : bp.dwo;
      atomicAdd(which == 3 ? dst + c * C + rc : dst + rc * C + c, aW[i]);
    }
    for (int i = threadIdx.x; i < 3 * CV; i += blockDim.x) {
      const int which = i / CV, rc = real_ch<DM>(i % CV);
      if (rc >= 0) atomicAdd((which == 0 ? bp.dbq : which == 1 ? bp.dbk : bp.dbv) + rc, aW[4 * CV * CP + i]);
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(bp.dbo + i, aW[4 * CV * CP + 3 * CV + i]);
  }
}

template <typename T, typename DM, bool ACC_LDS>
int launch_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  using LY = BwdLayout<T, DM, ACC_LDS>;
  int blocks = (g.nWin + LY::WAVES - 1) / LY::WAVES;
  if (blocks > 512) blocks = 512;           // persistent-ish: fewer weight-gradient flushes
  auto kern = winattn_bwd_kernel<T, DM, ACC_LDS>;
  static bool attr_set = false;     // idempotent per instantiation; kept out of replayed hipGraph captures
  if (LY::BYTES > 64 * 1024 && !attr_set) {
    attr_set = true;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY::BYTES);
    if (e != hipSuccess) { set_error("winattn_bwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  }
  kern<<<blocks, LY::WAVES * 64, LY::BYTES, st>>>(*p, g);
  return check_launch("winattn_bwd");
}

template <typename T, typename DM>
int pick_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  if constexpr (BwdLayout<T, DM, true>::FITS) return launch_bwd<T, DM, true>(p, g, st);
  else if constexpr (BwdLayout<T, DM, false>::FITS) return launch_bwd<T, DM, false>(p, g, st);
  else {
    set_error("winattn_bwd: C=%d in this dtype exceeds the 160 KiB LDS of a CU (use bf16 activations)", DM::C);
    return RSSF_ERR_UNSUPPORTED;
  }
}

template <typename T>
int dispatch_bwd(const rssf_winattn_bwd_params* p, const Geom& g, hipStream_t st) {
  if (p->f.heads == 2 && p->f.C == 32) return pick_bwd<T, Dims<32, 2>>(p, g, st);
  if (p->f.heads == 2 && p->f.C == 18) return pick_bwd<T, Dims<18, 2>>(p, g, st);
  if (p->f.heads == 2 && p->f.C == 48) return pick_bwd<T, Dims<48, 2>>(p, g, st);
  set_error("winattn_bwd: no kernel instantiated for C=%d heads=%d (built: 18/32/48 x 2 heads)", p->f.C, p->f.heads);
  return RSSF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int rssf_winattn_bwd(const rssf_winattn_bwd_params* p, void* stream) {
  RSSF_REQUIRE(p, "winattn_bwd: null params");
  const rssf_winattn_fwd_params& f = p->f;
  RSSF_REQUIRE(f.x && f.y && f.stats_x && f.stats_y && f.omega && f.ln_gamma && f.ln_beta && f.wq && f.bq && f.wk && f.bk &&
                   f.wv && f.bv && f.wo && f.bo,
               "winattn_bwd: null forward tensor pointer");
  RSSF_REQUIRE(p->dout && p->dxhat && p->dyhat && p->domega && p->dwq && p->dbq && p->dwk && p->dbk && p->dwv && p->dbv &&
                   p->dwo && p->dbo,
               "winattn_bwd: null gradient pointer");
  RSSF_REQUIRE(f.B > 0 && f.H > 0 && f.W > 0 && f.C > 0 && f.heads > 0 && f.window == 7, "winattn_bwd: bad shape");
  RSSF_REQUIRE(f.C % f.heads == 0, "winattn_bwd: embed_dim must be divisible by num_heads");
  const Geom g = make_geom(f.B, f.H, f.W, f.window);
  hipStream_t st = (hipStream_t)stream;
  if (f.dtype == RSSF_F32) return dispatch_bwd<float>(p, g, st);
  if (f.dtype == RSSF_BF16) return dispatch_bwd<bf16_t>(p, g, st);
  set_error("winattn_bwd: unsupported dtype %d", f.dtype);
  return RSSF_ERR_UNSUPPORTED;
}
