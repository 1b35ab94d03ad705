// Fused 7x7-window cross attention, forward.
//
// Replaces, in one kernel and one HBM round trip (read low, read high, write out):
//   InterlacedPoolAttention2.forward :164-188  (gate multiply, pad, permute, attention, rev-permute, depad)
//   PadBlock / LocalPermuteModule              (modules/multihead_isa_attention.py:364-426)   -> index math only
//   Mhca.multi_head_attention_forward          (modules/DAL.py:873-1020)
//   the attention residual of GeneralTransformerBlock.forward (modules/MTFM.py:107)
//
// Mapping: one 64-lane wavefront owns one window (49 live tokens padded to 64 = 4 MFMA tiles); 4 waves per
// workgroup share the projection weights in LDS.  Everything between the single global read of the two
// 49xC tiles and the single global write stays in LDS / registers.  All matmuls run "transposed"
// (channels x tokens) so that every MFMA result lands in exactly the register layout the next MFMA wants
// as an operand (q,k -> S^T -> P^T -> O^T -> out^T); only q^T,k^T (for M = q^T k) and v^T go through LDS.
// The kernel is HBM-bound (0.27 GFLOP vs 3.1 MB per image-block); MFMA just keeps the
// arithmetic out of the way.
#include "win_attn.cuh"
using namespace rssf;
using namespace rssf::wa;

namespace {

template <typename T, typename DM> struct FwdLayout {
  static constexpr int LDX = DM::CP + Pad<T>::X;     // xs/ys rows  [token][channel]
  static constexpr int LDT = LP + Pad<T>::X;         // qT/kT/vT rows [virtual channel][token]
  static constexpr int LDW = DM::CP + Pad<T>::X;     // Wq/Wk/Wv rows [virtual channel][in channel]
  static constexpr int LDO = DM::CV + Pad<T>::X;     // Wo rows [out channel][virtual channel]
  static constexpr int REGION = ((LP * LDX > DM::CV * LDT ? LP * LDX : DM::CV * LDT) + 7) / 8 * 8;
  static constexpr int W_ELEMS = 3 * DM::CV * LDW + DM::CP * LDO;
  static constexpr int F_ELEMS = 3 * DM::CV + 3 * DM::CP;   // biases + LN affine (fp32)
  static constexpr size_t SHARED_OFF = (sizeof(T) * W_ELEMS + sizeof(float) * F_ELEMS + 15) / 16 * 16;
  static constexpr size_t WAVE_BYTES = sizeof(T) * 3 * REGION;
  // waves per workgroup: as many (<= 4) as fit the 160 KiB LDS of one CU
  static constexpr int WAVES = (SHARED_OFF + 4 * WAVE_BYTES <= 160 * 1024) ? 4 : (SHARED_OFF + 2 * WAVE_BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr size_t BYTES = SHARED_OFF + WAVE_BYTES * WAVES;
};

template <typename T, typename DM>
__global__ void __launch_bounds__((FwdLayout<T, DM>::WAVES * 64)) winattn_fwd_kernel(rssf_winattn_fwd_params p, Geom g) {
  using LY = FwdLayout<T, DM>;
  constexpr int LDX = LY::LDX, LDT = LY::LDT, LDW = LY::LDW, LDO = LY::LDO;
  constexpr int C = DM::C, CP = DM::CP, CV = DM::CV, MT = DM::MT, CT = DM::CT, TPH = DM::TPH, D = DM::D;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // ---- workgroup-shared: weights as T (A operands), biases / LN affine fp32 ------------------------------
  T* sWq = reinterpret_cast<T*>(smem_raw);           // [CV][LDW]  virtual rows, k = real input channel
  T* sWk = sWq + CV * LDW;
  T* sWv = sWk + CV * LDW;
  T* sWo = sWv + CV * LDW;                           // [CP][LDO]  rows = output channel, k = virtual channel
  float* sB = reinterpret_cast<float*>(sWo + CP * LDO);   // bq[CV] bk[CV] bv[CV] bo[CP]
  float* sLn = sB + 3 * CV + CP;                     // gamma[CP] beta[CP]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, grp = lane >> 4;
  T* regA = reinterpret_cast<T*>(smem_raw + LY::SHARED_OFF) + (size_t)wave * 3 * LY::REGION;   // xs -> qT
  T* regB = regA + LY::REGION;                                                                  // ys -> kT
  T* regC = regB + LY::REGION;                                                                  //       vT

  for (int i = threadIdx.x; i < CV * LDW; i += blockDim.x) {
    const int m = i / LDW, k = i % LDW;
    const int rc = real_ch<DM>(m);
    const bool ok = rc >= 0 && k < C;
    stf(sWq + i, ok ? p.wq[rc * C + k] : 0.f);
    stf(sWk + i, ok ? p.wk[rc * C + k] : 0.f);
    stf(sWv + i, ok ? p.wv[rc * C + k] : 0.f);
  }
  for (int i = threadIdx.x; i < CP * LDO; i += blockDim.x) {
    const int co = i / LDO, m = i % LDO;
    const int rc = m < CV ? real_ch<DM>(m) : -1;
    stf(sWo + i, (co < C && rc >= 0) ? p.wo[co * C + rc] : 0.f);
  }
  for (int i = threadIdx.x; i < CV; i += blockDim.x) {
    const int rc = real_ch<DM>(i);
    sB[i] = rc >= 0 ? p.bq[rc] : 0.f;
    sB[CV + i] = rc >= 0 ? p.bk[rc] : 0.f;
    sB[2 * CV + i] = rc >= 0 ? p.bv[rc] : 0.f;
  }
  for (int i = threadIdx.x; i < CP; i += blockDim.x) {
    sB[3 * CV + i] = i < C ? p.bo[i] : 0.f;
    sLn[i] = i < C ? p.ln_gamma[i] : 0.f;
    sLn[CP + i] = i < C ? p.ln_beta[i] : 0.f;
  }
  __syncthreads();

  const float scale = rsqrtf((float)D);
  const T* X = reinterpret_cast<const T*>(p.x);
  const T* Y = reinterpret_cast<const T*>(p.y);
  T* OUT = reinterpret_cast<T*>(p.out);
  const int wpi = g.QH * g.QW;

  for (int wi = blockIdx.x * LY::WAVES + wave; wi < g.nWin; wi += gridDim.x * LY::WAVES) {
    const int b = wi / wpi, qh = (wi % wpi) / g.QW, qw = wi % g.QW;
    const int64_t img = (int64_t)b * g.N;
    const float* om0 = p.omega + (int64_t)b * 2 * g.N;

    // ---- 1. load both 49xC tiles, LayerNorm (given stats) * gate weight -> LDS as T, zero padded ------
    wave_sync();
    load_gated_tiles<T, DM>(p, g, sLn, X, Y, om0, img, qh, qw, regA, regB, LDX, lane);
    wave_sync();

    // ---- 2. projections (transposed): q^T,k^T [virtual channel][token] kept in C-layout registers,
    //         v^T staged only (goes to LDS in step 3) -----------------------------------------------------
    f32x4 q[MT][NT], k[MT][NT], v[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mrow = mt * 16 + grp * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
        aq = mma_tile<T>(sWq + mt * 16 * LDW, LDW, regA + tt * 16 * LDX, LDX, CP, aq);
        ak = mma_tile<T>(sWk + mt * 16 * LDW, LDW, regB + tt * 16 * LDX, LDX, CP, ak);
        av = mma_tile<T>(sWv + mt * 16 * LDW, LDW, regB + tt * 16 * LDX, LDX, CP, av);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          aq[r] = (aq[r] + sB[mrow + r]) * scale;
          ak[r] += sB[CV + mrow + r];
          av[r] += sB[2 * CV + mrow + r];
        }
        q[mt][tt] = aq; k[mt][tt] = ak; v[mt][tt] = av;
      }
    }
    wave_sync();   // all reads of xs/ys done before the regions are reused

    // ---- 3. q^T, k^T (tokens >= 49 zeroed: they must not enter M) and v^T to LDS, channel-major ----------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int tok = tt * 16 + l15;
        const bool live = tok < g.L;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mt * 16 + grp * 4 + r;
          stf(regA + m * LDT + tok, live ? q[mt][tt][r] : 0.f);
          stf(regB + m * LDT + tok, live ? k[mt][tt][r] : 0.f);
          stf(regC + m * LDT + tok, live ? v[mt][tt][r] : 0.f);
        }
      }
    wave_sync();

    // ---- 4. per head: alpha from M = q^T k; per query tile: S^T = k q^T, softmax over keys, O^T = v^T P^T -
    f32x4 o[MT][NT];
#pragma unroll
    for (int h = 0; h < DM::HEADS; ++h) {
      // channel alpha = sigmoid(mean(M) + max(M)),  M = q_h^T k_h  (d x d)   (DAL.py:1003-1010)
      float msum = 0.f, mmax = -INFINITY;
#pragma unroll
      for (int it = 0; it < TPH; ++it)
#pragma unroll
        for (int jt = 0; jt < TPH; ++jt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<T>(regA + (h * DM::DP + it * 16) * LDT, LDT, regB + (h * DM::DP + jt * 16) * LDT, LDT, LP, acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + grp * 4 + r, j = jt * 16 + l15;
            if (i < D && j < D) { msum += acc[r]; mmax = fmaxf(mmax, acc[r]); }
          }
        }
      msum = wave_sum(msum);
      mmax = wave_max(mmax);
      const float alpha = sigmoidf(msum / (float)(D * D) + mmax);

#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        f32x4 s[NT];              // [key tile]: lane holds key kt*16+4*grp+r for query qt*16+l15
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mt = h * TPH; mt < (h + 1) * TPH; ++mt) acc = mma_chain<T>(k[mt][kt], q[mt][qt], acc);
          s[kt] = acc;
        }
        // softmax over the 49 live keys (no mask, no bias: DAL.py:959,996)
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
        const float inv = alpha / sum;      // fold alpha into the normalisation: o = alpha * (P v)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
        // O^T[dcol][query] = sum_key v^T[dcol][key] P^T[key][query]
#pragma unroll
        for (int mt = h * TPH; mt < (h + 1) * TPH; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) acc = mma_lds_chain<T>(regC + mt * 16 * LDT, LDT, kt * 16, s[kt], acc);
          o[mt][qt] = acc;
        }
      }
    }

    // ---- 5. out-projection (transposed) + bias + residual, store live tokens --------------------------------
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      const int n = slot_token(g, qh, qw, qt * 16 + l15);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc = mma_lds_chain<T>(sWo + ct * 16 * LDO, LDO, mt * 16, o[mt][qt], acc);
        if (n < 0) continue;
        const int c0 = ct * 16 + grp * 4;
        const int64_t off = (img + n) * C + c0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + r < C) stf(OUT + off + r, ldf(X + off + r) + acc[r] + sB[3 * CV + c0 + r]);
      }
    }
  }
}

template <typename T, typename DM>
int launch_fwd(const rssf_winattn_fwd_params* p, const Geom& g, hipStream_t st) {
  using LY = FwdLayout<T, DM>;
  static_assert(LY::BYTES <= 160 * 1024, "LDS budget");
  int blocks = (g.nWin + LY::WAVES - 1) / LY::WAVES;
  if (blocks > 4096) blocks = 4096;
  auto kern = winattn_fwd_kernel<T, DM>;
  if (LY::BYTES > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY::BYTES);
    if (e != hipSuccess) { set_error("winattn_fwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  }
  kern<<<blocks, LY::WAVES * 64, LY::BYTES, st>>>(*p, g);
  return check_launch("winattn_fwd");
}

template <typename T>
int dispatch_fwd(const rssf_winattn_fwd_params* p, const Geom& g, hipStream_t st) {
  if (p->heads == 2 && p->C == 32) return launch_fwd<T, Dims<32, 2>>(p, g, st);   // Base  (hrnetv2_w32)
  if (p->heads == 2 && p->C == 18) return launch_fwd<T, Dims<18, 2>>(p, g, st);   // Tiny  (hrnetv2_w32s table)
  if (p->heads == 2 && p->C == 48) return launch_fwd<T, Dims<48, 2>>(p, g, st);   // Large (hrnetv2_w48)
  set_error("winattn_fwd: no kernel instantiated for C=%d heads=%d (built: 18/32/48 x 2 heads)", p->C, p->heads);
  return RSSF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int rssf_winattn_fwd(const rssf_winattn_fwd_params* p, void* stream) {
  RSSF_REQUIRE(p, "winattn_fwd: null params");
  RSSF_REQUIRE(p->x && p->y && p->stats_x && p->stats_y && p->omega && p->ln_gamma && p->ln_beta && p->wq && p->bq &&
                   p->wk && p->bk && p->wv && p->bv && p->wo && p->bo && p->out,
               "winattn_fwd: null tensor pointer");
  RSSF_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->heads > 0 && p->window == 7,
               "winattn_fwd: bad shape B=%d H=%d W=%d C=%d heads=%d window=%d", p->B, p->H, p->W, p->C, p->heads, p->window);
  RSSF_REQUIRE(p->C % p->heads == 0, "winattn_fwd: embed_dim must be divisible by num_heads");   // DAL.py:700-702
  const Geom g = make_geom(p->B, p->H, p->W, p->window);
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == RSSF_F32) return dispatch_fwd<float>(p, g, st);
  if (p->dtype == RSSF_BF16) return dispatch_fwd<bf16_t>(p, g, st);
  set_error("winattn_fwd: unsupported dtype %d", p->dtype);
  return RSSF_ERR_UNSUPPORTED;
}
