// Weight gradient of the channels-last implicit-GEMM convolution (MFMA, split-K over pixels).
//
//   dW[tap][co][ci] = sum_p dout[p][co] * in[src(p, tap)][ci]          (autograd of conv_fwd.hip)
//
// GEMM view per tap: M = Cout, N = Cin, K = B*OH*OW pixels.  Both operands are stored pixel-major in HBM
// (channels contiguous), but the MFMA wants the contraction axis contiguous per lane, so each 64-pixel slab is
// transposed on its way into LDS ([channel][pixel] images; one lane per pixel => conflict-free 2-byte stores).
// A block owns one (tap, 64x64 co x ci tile, pixel range); partial sums are atomically added straight into the
// fp32 torch-layout gradient [Cout][Cin][kh][kw] (the fused 19-tap MLP conv scatters to its three source convs).
// Optional bias gradient (sum_p dout) for convolutions that are not followed by a BatchNorm.
#include "conv.cuh"
using namespace rssf;
using namespace rssf::cv;

namespace {

struct WgradArgs {
  const void* dout;      // [B, OH, OW, Cout]
  const void* in;        // [B, IH, IW, Cin]
  float* dw[3];          // torch-layout fp32 grads of up to 3 source convs
  int ks[3];
  float* dbias;          // [Cout] or null
  int src_of_tap[MAX_TAPS];
  int kpos_of_tap[MAX_TAPS];
  int B, IH, IW, Cin, OH, OW, Cout, stride, ksplit, ctiles_m, ctiles_n;
  Taps taps;
};

constexpr int TM = 64, TN = 64;

template <typename T>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs a) {
  using MK = MmaK<T>;
  constexpr int V = Vec<T>::N;
  constexpr int KP = 64;                                   // pixels per staged slab
  constexpr int LDT = KP + LdsPad<T>::X;
  constexpr int CH_CHUNKS = TM / V;                        // 16-byte chunks per pixel row of a 64-channel tile
  __shared__ __attribute__((aligned(16))) T DT[TM * LDT];  // dout^T [co][pixel]
  __shared__ __attribute__((aligned(16))) T XT[TN * LDT];  // in^T   [ci][pixel]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
  int bid = blockIdx.x;
  const int cit = bid % a.ctiles_n; bid /= a.ctiles_n;
  const int cot = bid % a.ctiles_m; bid /= a.ctiles_m;
  const int tap = bid;
  const int co0 = cot * TM, ci0 = cit * TN;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  const int64_t per = ((M + a.ksplit - 1) / a.ksplit + KP - 1) / KP * KP;
  const int64_t kbeg = (int64_t)blockIdx.y * per, kend = kbeg + per < M ? kbeg + per : M;
  const T* DO = reinterpret_cast<const T*>(a.dout);
  const T* IN = reinterpret_cast<const T*>(a.in);
  const int dy = a.taps.dy[tap], dx = a.taps.dx[tap];
  const bool ovec = (a.Cout % V) == 0, ivec = (a.Cin % V) == 0;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = a.dbias && tap == 0 && cit == 0;

  for (int64_t k0 = kbeg; k0 < kend; k0 += KP) {
    // ---- stage: lane = pixel, each wave takes a quarter of the channel chunks ------------------------------------
    const int64_t p = k0 + lane;
    const bool pvalid = p < kend;
    int b = 0, oy = 0, ox = 0;
    if (pvalid) {
      b = (int)(p / ((int64_t)a.OH * a.OW));
      const int rem = (int)(p % ((int64_t)a.OH * a.OW));
      oy = rem / a.OW; ox = rem % a.OW;
    }
    const int sy = oy * a.stride + dy, sx = ox * a.stride + dx;
    const bool svalid = pvalid && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW;
    const T* drow = DO + p * a.Cout;
    const T* xrow = IN + (((int64_t)b * a.IH + sy) * a.IW + sx) * a.Cin;
    for (int ch = wave; ch < CH_CHUNKS; ch += 4) {
      const int c = ch * V;
      Vec<T> vd, vx;
      vd.raw = {0, 0, 0, 0}; vx.raw = {0, 0, 0, 0};
      if (pvalid && co0 + c < a.Cout) {
        if (ovec) vd.load(drow + co0 + c);
        else for (int e = 0; e < V; ++e) if (co0 + c + e < a.Cout) vd.set(e, ldf(drow + co0 + c + e));
      }
      if (svalid && ci0 + c < a.Cin) {
        if (ivec) vx.load(xrow + ci0 + c);
        else for (int e = 0; e < V; ++e) if (ci0 + c + e < a.Cin) vx.set(e, ldf(xrow + ci0 + c + e));
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        stf(DT + (c + e) * LDT + lane, vd.get(e));
        stf(XT + (c + e) * LDT + lane, vx.get(e));
      }
    }
    __syncthreads();
    // ---- MFMA: each wave a 32 (co) x 32 (ci) sub-tile, K = 64 pixels -------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < KP; ks += MK::KSTEP) {
      typename MK::frag fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = MK::load(DT + (wm * 32 + i * 16 + l15) * LDT + ks + grp * MK::KPL);
        fb[i] = MK::load(XT + (wn * 32 + i * 16 + l15) * LDT + ks + grp * MK::KPL);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MK::mma(fa[i], fb[j], acc[i][j]);
    }
    if (do_bias && tid < TM) {
      float s = 0.f;
      for (int k = 0; k < KP; ++k) s += ldf(DT + tid * LDT + k);
      bsum += s;
    }
    __syncthreads();
  }

  // ---- flush: rows = co (4*grp + r), cols = ci (l15) -----------------------------------------------------------------
  const int s = a.src_of_tap[tap], kk = a.ks[s] * a.ks[s], kpos = a.kpos_of_tap[tap];
  float* dw = a.dw[s];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 32 + i * 16 + grp * 4 + r, ci = ci0 + wn * 32 + j * 16 + l15;
        if (co < a.Cout && ci < a.Cin) atomicAdd(dw + ((int64_t)co * a.Cin + ci) * kk + kpos, acc[i][j][r]);
      }
  if (do_bias && tid < TM && co0 + tid < a.Cout) atomicAdd(a.dbias + co0 + tid, bsum);
}

}  // namespace

extern "C" int rssf_conv_wgrad(const void* dout, const void* in, float* dw0, float* dw1, float* dw2, const int* ksizes,
                               int nsrc, const int* src_of_tap, const int* kpos_of_tap, float* dbias, int B, int IH, int IW,
                               int Cin, int OH, int OW, int Cout, int stride, int ntaps, const int* dy, const int* dx,
                               int dtype, void* stream) {
  RSSF_REQUIRE(dout && in && dw0 && ksizes && src_of_tap && kpos_of_tap && dy && dx && nsrc >= 1 && nsrc <= 3 && ntaps >= 1 &&
                   ntaps <= MAX_TAPS && B > 0 && IH > 0 && IW > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0 && stride >= 1,
               "conv_wgrad: bad arguments");
  WgradArgs a;
  a.dout = dout; a.in = in; a.dw[0] = dw0; a.dw[1] = dw1; a.dw[2] = dw2; a.dbias = dbias;
  for (int i = 0; i < 3; ++i) a.ks[i] = i < nsrc ? ksizes[i] : 1;
  a.taps.n = ntaps;
  for (int t = 0; t < ntaps; ++t) {
    a.src_of_tap[t] = src_of_tap[t]; a.kpos_of_tap[t] = kpos_of_tap[t];
    a.taps.dy[t] = dy[t]; a.taps.dx[t] = dx[t];
  }
  a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.stride = stride;
  a.ctiles_m = (Cout + TM - 1) / TM; a.ctiles_n = (Cin + TN - 1) / TN;
  const int tiles = ntaps * a.ctiles_m * a.ctiles_n;
  const int64_t M = (int64_t)B * OH * OW;
  int64_t ks = 2048 / tiles;                      // aim at ~8 blocks per CU
  const int64_t maxks = (M + 255) / 256;          // at least 4 slabs per block
  if (ks > maxks) ks = maxks;
  if (ks < 1) ks = 1;
  a.ksplit = (int)ks;
  dim3 grid((unsigned)tiles, (unsigned)ks);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) conv_wgrad_kernel<float><<<grid, 256, 0, st>>>(a);
  else if (dtype == RSSF_BF16) conv_wgrad_kernel<bf16_t><<<grid, 256, 0, st>>>(a);
  else { set_error("conv_wgrad: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("conv_wgrad");
}
