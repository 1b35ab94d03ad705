// Up-sampling on channels-last activations:
//   bilinear, align_corners=True  — F.interpolate(..., 'bilinear', align_corners=True) x3 in SimpleFusion8
//                                   (hrnet_aux.py:61-65) and nn.UpsamplingBilinear2d(x4) of the head (:80)
//   nearest (integer factor) fused with the running branch sum — nn.Upsample(mode='nearest') of the HRNet fuse
//                                   layers (_hrnet_rssformer.py:380) followed by `low = low + ...` (:424-427)
// Both backward passes are written as GATHERS (each input pixel sums the output pixels that read it), so there
// are no atomics: the ATen scatter-add backward costs 13 ms per call in bf16 on this shape, this one is HBM-bound.
#include "common.hip.h"
using namespace rssf;

namespace {

__device__ __forceinline__ float src_coord(int o, float scale) { return scale * (float)o; }

template <typename T, int VEC>
__global__ void __launch_bounds__(256) bilinear_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int IH, int IW, int OH,
                                                           int OW, int C, int ldw, float sy, float sx) {
  const int cols = C / VEC;
  const int64_t total = (int64_t)B * OH * OW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float fy = src_coord(oy, sy), fx = src_coord(ox, sx);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < IH ? y0 + 1 : IH - 1, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
    const float wy = fy - y0, wx = fx - x0;
    const T* base = in + (int64_t)b * IH * IW * C + cv * VEC;
    const T* p00 = base + ((int64_t)y0 * IW + x0) * C;
    const T* p01 = base + ((int64_t)y0 * IW + x1) * C;
    const T* p10 = base + ((int64_t)y1 * IW + x0) * C;
    const T* p11 = base + ((int64_t)y1 * IW + x1) * C;
    T* dst = out + (((int64_t)b * OH + oy) * OW + ox) * ldw + cv * VEC;     // ldw: pixel stride of the full-resolution tensor
    const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
    if constexpr (VEC > 1) {
      Vec<T> a, bq, c, d, o;
      a.load(p00); bq.load(p01); c.load(p10); d.load(p11);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.set(e, w00 * a.get(e) + w01 * bq.get(e) + w10 * c.get(e) + w11 * d.get(e));
      o.store(dst);
    } else {
      stf(dst, w00 * ldf(p00) + w01 * ldf(p01) + w10 * ldf(p10) + w11 * ldf(p11));
    }
  }
}

// din(iy,ix) = sum over output pixels (oy,ox) of coef_y(oy,iy) * coef_x(ox,ix) * dout(oy,ox)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int IH, int IW, int OH,
                                                           int OW, int C, int ldw, float sy, float sx) {
  const int cols = C / VEC;
  const int64_t total = (int64_t)B * IH * IW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    int oy_lo = 0, oy_hi = OH - 1, ox_lo = 0, ox_hi = OW - 1;
    if (sy > 0.f) {
      oy_lo = (int)floorf((iy - 1) / sy); oy_hi = (int)ceilf((iy + 1) / sy);
      if (oy_lo < 0) oy_lo = 0;
      if (oy_hi > OH - 1) oy_hi = OH - 1;
    }
    if (sx > 0.f) {
      ox_lo = (int)floorf((ix - 1) / sx); ox_hi = (int)ceilf((ix + 1) / sx);
      if (ox_lo < 0) ox_lo = 0;
      if (ox_hi > OW - 1) ox_hi = OW - 1;
    }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float fy = src_coord(oy, sy);
      const int y0 = (int)fy, y1 = y0 + 1 < IH ? y0 + 1 : IH - 1;
      const float wy = fy - y0;
      const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (cy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float fx = src_coord(ox, sx);
        const int x0 = (int)fx, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
        const float wx = fx - x0;
        const float cx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (cx == 0.f) continue;
        const T* src = dout + (((int64_t)b * OH + oy) * OW + ox) * ldw + cv * VEC;
        const float w = cy * cx;
        if constexpr (VEC > 1) {
          Vec<T> v;
          v.load(src);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += w * v.get(e);
        } else {
          acc[0] += w * ldf(src);
        }
      }
    }
    T* dst = din + (((int64_t)b * IH + iy) * IW + ix) * C + cv * VEC;
    if constexpr (VEC > 1) {
      Vec<T> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
      o.store(dst);
    } else {
      stf(dst, acc[0]);
    }
  }
}

// Large factors (x 4, x 8: the neck's 32^2 and 16^2 branches resized to 128^2, hrnet_aux.py:61-65): an input pixel gathers from up
// to 17 x 17 output pixels, and with one thread per (pixel, 16-byte channel chunk) that was a chain of ~290 loads on 512 blocks
// (139 us for the 134 MB of the x 8 gradient).  Here a BLOCK owns one input pixel: 256 / cols row parts of `cols` chunk lanes each;
// a part walks every R-th output row of the window (its lanes read one contiguous C-channel pixel per load), the parts' partial sums
// are folded through LDS.  Same sums in another order (fp32 accumulation of weighted bf16 values).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) bilinear_bwd_rows_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int IH, int IW, int OH,
                                                                int OW, int C, int ldw, float sy, float sx, int xcd_per) {
  __shared__ float sred[256 * VEC];
  const int cols = C / VEC, R = 256 / cols;                  // cols divides 256 (the launcher checks)
  const int cv = threadIdx.x % cols, r = threadIdx.x / cols;
  // XCD-major numbering: the windows of neighbouring input pixels overlap by (window - 1 / factor) in each direction; hardware block
  // order deals neighbours round-robin over the eight L2s, each of which then fetched its own copy of the overlap (round 4: 417 MB
  // across the fabric for 201 MB of gradient).  An XCD takes a contiguous run of input pixels instead.
  int64_t p = (int64_t)(blockIdx.x & 7u) * xcd_per + (blockIdx.x >> 3);
  if (p >= (int64_t)B * IH * IW) return;
  const int ix = (int)(p % IW); p /= IW;
  const int iy = (int)(p % IH);
  const int b = (int)(p / IH);
  int oy_lo = (int)floorf((iy - 1) / sy), oy_hi = (int)ceilf((iy + 1) / sy);
  int ox_lo = (int)floorf((ix - 1) / sx), ox_hi = (int)ceilf((ix + 1) / sx);
  oy_lo = oy_lo < 0 ? 0 : oy_lo; oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi;
  ox_lo = ox_lo < 0 ? 0 : ox_lo; ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int oy = oy_lo + r; oy <= oy_hi; oy += R) {
    const float fy = src_coord(oy, sy);
    const int y0 = (int)fy, y1 = y0 + 1 < IH ? y0 + 1 : IH - 1;
    const float wy = fy - y0;
    const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
    if (cy == 0.f) continue;
    const T* row = dout + (((int64_t)b * OH + oy) * OW) * ldw + cv * VEC;
#pragma unroll 4
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = src_coord(ox, sx);
      const int x0 = (int)fx, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
      const float wx = fx - x0;
      const float w = cy * ((x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f));
      Vec<T> v;
      v.load(row + (int64_t)ox * ldw);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += w * v.get(e);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) sred[threadIdx.x * VEC + e] = acc[e];
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < R; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += sred[(k * cols + cv) * VEC + e];
    Vec<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(din + (((int64_t)b * IH + iy) * IW + ix) * C + cv * VEC);
  }
}

// ---- few-channel tensors (the 6-class logits of the head, x 4 up-sampling of hrnet_aux.py:80): ONE thread per pixel -------------
// The vector kernels above need C % VEC == 0; their scalar instantiation spends a full set of coordinate / index arithmetic on
// every single element (16 x 512 x 512 x 6: 120 us forward, 94 us backward = 0.4 TB/s).  Here a thread computes the coordinates
// once and walks the pixel's channels in 4-byte units (two bf16 or one fp32 channel).
template <typename T> struct PxUnit;
template <> struct PxUnit<float> {
  static constexpr int CH = 1;
  typedef float raw;
  static __device__ __forceinline__ void get(raw r, float* v) { v[0] = r; }
  static __device__ __forceinline__ raw put(const float* v) { return v[0]; }
};
template <> struct PxUnit<bf16_t> {
  static constexpr int CH = 2;
  typedef uint32_t raw;
  static __device__ __forceinline__ void get(raw r, float* v) { v[0] = __uint_as_float(r << 16); v[1] = __uint_as_float(r & 0xffff0000u); }
  static __device__ __forceinline__ raw put(const float* v) { return f2bf2(v[0], v[1]); }
};
constexpr int PX_MAXU = 8;           // up to 8 units per pixel (8 fp32 / 16 bf16 channels)

template <typename T>
__global__ void __launch_bounds__(256) bilinear_fwd_px_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int IH, int IW, int OH, int OW,
                                                              int C, int ldw, float sy, float sx) {
  using U = PxUnit<T>;
  typedef typename U::raw raw;
  const int nu = C / U::CH;
  const int64_t total = (int64_t)B * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float fy = src_coord(oy, sy), fx = src_coord(ox, sx);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < IH ? y0 + 1 : IH - 1, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
    const float wy = fy - y0, wx = fx - x0;
    const T* base = in + (int64_t)b * IH * IW * C;
    const raw* p00 = reinterpret_cast<const raw*>(base + ((int64_t)y0 * IW + x0) * C);
    const raw* p01 = reinterpret_cast<const raw*>(base + ((int64_t)y0 * IW + x1) * C);
    const raw* p10 = reinterpret_cast<const raw*>(base + ((int64_t)y1 * IW + x0) * C);
    const raw* p11 = reinterpret_cast<const raw*>(base + ((int64_t)y1 * IW + x1) * C);
    raw* dst = reinterpret_cast<raw*>(out + (((int64_t)b * OH + oy) * OW + ox) * ldw);
    const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
#pragma unroll
    for (int u = 0; u < PX_MAXU; ++u) {
      if (u >= nu) break;
      float a[U::CH], bq[U::CH], c[U::CH], d[U::CH], o[U::CH];
      U::get(p00[u], a); U::get(p01[u], bq); U::get(p10[u], c); U::get(p11[u], d);
#pragma unroll
      for (int e = 0; e < U::CH; ++e) o[e] = w00 * a[e] + w01 * bq[e] + w10 * c[e] + w11 * d[e];
      dst[u] = U::put(o);
    }
  }
}

// (four lanes per input pixel, every fourth window row each, folded by lane exchange, were measured: 98 against 87 us - the pass is
// bound by its 16-byte pieces of 67 MB, not by the chain)
template <typename T>
__global__ void __launch_bounds__(256) bilinear_bwd_px_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int IH, int IW, int OH, int OW,
                                                              int C, int ldw, float sy, float sx) {
  using U = PxUnit<T>;
  typedef typename U::raw raw;
  const int nu = C / U::CH;
  const int64_t total = (int64_t)B * IH * IW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    int oy_lo = 0, oy_hi = OH - 1, ox_lo = 0, ox_hi = OW - 1;
    if (sy > 0.f) {
      oy_lo = (int)floorf((iy - 1) / sy); oy_hi = (int)ceilf((iy + 1) / sy);
      if (oy_lo < 0) oy_lo = 0;
      if (oy_hi > OH - 1) oy_hi = OH - 1;
    }
    if (sx > 0.f) {
      ox_lo = (int)floorf((ix - 1) / sx); ox_hi = (int)ceilf((ix + 1) / sx);
      if (ox_lo < 0) ox_lo = 0;
      if (ox_hi > OW - 1) ox_hi = OW - 1;
    }
    float acc[PX_MAXU * U::CH];
#pragma unroll
    for (int e = 0; e < PX_MAXU * U::CH; ++e) acc[e] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float fy = src_coord(oy, sy);
      const int y0 = (int)fy, y1 = y0 + 1 < IH ? y0 + 1 : IH - 1;
      const float wy = fy - y0;
      const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (cy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float fx = src_coord(ox, sx);
        const int x0 = (int)fx, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
        const float wx = fx - x0;
        const float cx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (cx == 0.f) continue;
        const raw* src = reinterpret_cast<const raw*>(dout + (((int64_t)b * OH + oy) * OW + ox) * ldw);
        const float w = cy * cx;
#pragma unroll
        for (int u = 0; u < PX_MAXU; ++u) {
          if (u >= nu) break;
          float v[U::CH];
          U::get(src[u], v);
#pragma unroll
          for (int e = 0; e < U::CH; ++e) acc[u * U::CH + e] += w * v[e];
        }
      }
    }
    raw* dst = reinterpret_cast<raw*>(din + (((int64_t)b * IH + iy) * IW + ix) * C);
#pragma unroll
    for (int u = 0; u < PX_MAXU; ++u) {
      if (u >= nu) break;
      dst[u] = U::put(acc + u * U::CH);
    }
  }
}

// The same gather for a TILE of input pixels per block, separable and out of LDS (round 5).  With a thread per input pixel every lane
// walks its own ~9 x 9 window of 12-byte pixels: a wave's load touches 64 different cache lines for 4 bytes each, every output pixel is
// fetched by up to four input pixels, and the 16 x 512 x 512 x 6 gradient of the head's x 4 up-sampling (50 MB) took 81 us.  Here a
// block owns TY x TX input pixels: the output rows / columns their windows cover are staged ONCE as flat dword runs (lanes on
// consecutive dwords), summed along x into [window row][input column] partials (weights cx) and then along y (weights cy).  NU = 4-byte
// units per pixel (PxUnit), pixel pitch of both tensors = NU units.
constexpr int PXT_TY = 8, PXT_TX = 16, PXT_NYM = 40, PXT_NXM = 72;
template <typename T, int NU>
__global__ void __launch_bounds__(256) bilinear_bwd_px_tile_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int IH, int IW, int OH, int OW,
                                                                   float sy, float sx, int tiles_y, int tiles_x) {
  using U = PxUnit<T>;
  typedef typename U::raw raw;
  constexpr int CH = U::CH, NF = NU * CH;
  __shared__ raw hi[PXT_NYM][PXT_NXM * NU];
  __shared__ float part[PXT_NYM][PXT_TX][NF];
  const int tid = threadIdx.x;
  const int tx = (int)(blockIdx.x % (unsigned)tiles_x), t1 = (int)(blockIdx.x / (unsigned)tiles_x);
  const int ty = t1 % tiles_y, b = t1 / tiles_y;
  const int iy0 = ty * PXT_TY, ix0 = tx * PXT_TX;
  // output rows / columns read by the tile's pixels: o with |o * s - i| < 1 for some i of the tile
  int oyA = (int)floorf((iy0 - 1) / sy), oyB = (int)ceilf((iy0 + PXT_TY) / sy);
  int oxA = (int)floorf((ix0 - 1) / sx), oxB = (int)ceilf((ix0 + PXT_TX) / sx);
  oyA = oyA < 0 ? 0 : oyA; oyB = oyB > OH - 1 ? OH - 1 : oyB;
  oxA = oxA < 0 ? 0 : oxA; oxB = oxB > OW - 1 ? OW - 1 : oxB;
  const int NY = oyB - oyA + 1, NX = oxB - oxA + 1;          // <= PXT_NYM, PXT_NXM: the launcher checked the scale factors
  const raw* src = reinterpret_cast<const raw*>(dout);
  const int run = NX * NU;
  for (int e = tid; e < NY * run; e += 256) {
    const int r = e / run, c = e - r * run;
    hi[r][c] = src[(((int64_t)b * OH + oyA + r) * OW + oxA) * NU + c];
  }
  __syncthreads();
  // along x: part[r][ixl] = sum over ox of cx(ox, ix) * dout[oyA + r][ox]
  for (int e = tid; e < NY * PXT_TX; e += 256) {
    const int r = e / PXT_TX, ixl = e % PXT_TX, ix = ix0 + ixl;
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = 0.f;
    if (ix < IW) {
      int lo = (int)floorf((ix - 1) / sx), hi_ = (int)ceilf((ix + 1) / sx);
      lo = lo < oxA ? oxA : lo; hi_ = hi_ > oxB ? oxB : hi_;
      for (int ox = lo; ox <= hi_; ++ox) {
        const float fx = src_coord(ox, sx);
        const int x0 = (int)fx, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
        const float wx = fx - x0;
        const float cx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          float v[CH];
          U::get(hi[r][(ox - oxA) * NU + u], v);
#pragma unroll
          for (int k = 0; k < CH; ++k) acc[u * CH + k] += cx * v[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NF; ++k) part[r][ixl][k] = acc[k];
  }
  __syncthreads();
  // along y
  for (int e = tid; e < PXT_TY * PXT_TX; e += 256) {
    const int iyl = e / PXT_TX, ixl = e % PXT_TX, iy = iy0 + iyl, ix = ix0 + ixl;
    if (iy >= IH || ix >= IW) continue;
    int lo = (int)floorf((iy - 1) / sy), hi_ = (int)ceilf((iy + 1) / sy);
    lo = lo < oyA ? oyA : lo; hi_ = hi_ > oyB ? oyB : hi_;
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = 0.f;
    for (int oy = lo; oy <= hi_; ++oy) {
      const float fy = src_coord(oy, sy);
      const int y0 = (int)fy, y1 = y0 + 1 < IH ? y0 + 1 : IH - 1;
      const float wy = fy - y0;
      const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
#pragma unroll
      for (int k = 0; k < NF; ++k) acc[k] += cy * part[oy - oyA][ixl][k];
    }
    raw* dst = reinterpret_cast<raw*>(din) + (((int64_t)b * IH + iy) * IW + ix) * NU;
#pragma unroll
    for (int u = 0; u < NU; ++u) dst[u] = U::put(acc + u * CH);
  }
}

// out = (acc ? acc : 0) + nearest_up(in, s)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) nearest_add_fwd_kernel(const T* __restrict__ acc, const T* __restrict__ in, T* __restrict__ out, int B,
                                                              int IH, int IW, int s, int C) {
  const int cols = C / VEC, OH = IH * s, OW = IW * s;
  const int64_t total = (int64_t)B * OH * OW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const T* src = in + (((int64_t)b * IH + oy / s) * IW + ox / s) * C + cv * VEC;
    const int64_t o = i * VEC;
    if constexpr (VEC > 1) {
      Vec<T> v, a, r;
      v.load(src);
      if (acc) a.load(acc + o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) r.set(e, v.get(e) + (acc ? a.get(e) : 0.f));
      r.store(out + o);
    } else {
      stf(out + o, ldf(src) + (acc ? ldf(acc + o) : 0.f));
    }
  }
}

// din(iy,ix) = sum of the s x s block of dout
template <typename T, int VEC>
__global__ void __launch_bounds__(256) nearest_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int IH, int IW, int s, int C) {
  const int cols = C / VEC, OW = IW * s, OH = IH * s;
  const int64_t total = (int64_t)B * IH * IW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    float a[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    for (int dy = 0; dy < s; ++dy)
      for (int dx = 0; dx < s; ++dx) {
        const T* src = dout + (((int64_t)b * OH + iy * s + dy) * OW + ix * s + dx) * C + cv * VEC;
        if constexpr (VEC > 1) {
          Vec<T> v;
          v.load(src);
#pragma unroll
          for (int e = 0; e < VEC; ++e) a[e] += v.get(e);
        } else {
          a[0] += ldf(src);
        }
      }
    if constexpr (VEC > 1) {
      Vec<T> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.set(e, a[e]);
      o.store(din + i * VEC);
    } else {
      stf(din + i, a[0]);
    }
  }
}

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
int bilinear_launch(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int ldw, int backward, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const bool vec = C % V == 0;
  const int64_t px = (int64_t)B * (backward ? IH * IW : OH * OW);
  constexpr int UCH = PxUnit<T>::CH;
  if (!vec && C % UCH == 0 && ldw % UCH == 0 && C / UCH <= PX_MAXU) {        // few channels: one thread per pixel
    const int gp = grid_for(px);
    if (!backward) bilinear_fwd_px_kernel<T><<<gp, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
    else if (C == 3 * UCH && ldw == C && sy > 0.f && sx > 0.f && (PXT_TY + 1) / sy + 3.f <= (float)PXT_NYM && (PXT_TX + 1) / sx + 3.f <= (float)PXT_NXM &&
             (int64_t)B * OH * OW * C < ((int64_t)1 << 31)) {
      // (the 6-class logits of the head in bf16, factor <= 4: a tile of input pixels per block, separable, out of LDS)
      const int tiles_y = (IH + PXT_TY - 1) / PXT_TY, tiles_x = (IW + PXT_TX - 1) / PXT_TX;
      bilinear_bwd_px_tile_kernel<T, 3><<<dim3((unsigned)(B * tiles_y * tiles_x)), 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, sy, sx, tiles_y, tiles_x);
    } else bilinear_bwd_px_kernel<T><<<gp, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
    return check_launch(backward ? "upsample_bilinear_bwd" : "upsample_bilinear_fwd");
  }
  const int g = grid_for(px * (vec ? C / V : C));
  if (!backward) {
    if (vec) bilinear_fwd_kernel<T, V><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
    else bilinear_fwd_kernel<T, 1><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
  } else {
    const int cols = vec ? C / V : 0;
    // factor >= 4 (a window of >= 9 x 9 output pixels per input pixel): a block per input pixel, the window's rows over its threads
    if (vec && sy > 0.f && sx > 0.f && sy <= 0.26f && sx <= 0.26f && cols >= 4 && cols <= 256 && 256 % cols == 0 && px < ((int64_t)1 << 31))
      bilinear_bwd_rows_kernel<T, V><<<dim3((unsigned)((px + 7) / 8 * 8)), 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx,
                                                                                         (int)((px + 7) / 8));
    else if (vec) bilinear_bwd_kernel<T, V><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
    else bilinear_bwd_kernel<T, 1><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C, ldw, sy, sx);
  }
  return check_launch(backward ? "upsample_bilinear_bwd" : "upsample_bilinear_fwd");
}

template <typename T>
int nearest_launch(const void* acc, const void* in, void* out, int B, int IH, int IW, int s, int C, int backward, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const bool vec = C % V == 0;
  const int64_t px = (int64_t)B * IH * IW * (backward ? 1 : s * s);
  const int g = grid_for(px * (vec ? C / V : C));
  if (!backward) {
    if (vec) nearest_add_fwd_kernel<T, V><<<g, 256, 0, st>>>((const T*)acc, (const T*)in, (T*)out, B, IH, IW, s, C);
    else nearest_add_fwd_kernel<T, 1><<<g, 256, 0, st>>>((const T*)acc, (const T*)in, (T*)out, B, IH, IW, s, C);
  } else {
    if (vec) nearest_bwd_kernel<T, V><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, s, C);
    else nearest_bwd_kernel<T, 1><<<g, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, s, C);
  }
  return check_launch(backward ? "upsample_nearest_bwd" : "upsample_nearest_add");
}
}  // namespace

namespace {
// Evaluation head (hrnet_aux.py:80, 103-104; eval.py:66-71 / predict.py:41-43 of the reference): nn.UpsamplingBilinear2d(x s)
// of the [B,IH,IW,K] class logits -> softmax over K -> (optional) argmax, one thread per output pixel, the K interpolated
// logits never leave registers.  Replaces a bilinear pass + an ATen softmax pass (+ an argmax pass) over the full-resolution
// [B,OH,OW,K] tensor by ONE write of the probabilities (fp32, channels-last) and/or the int32 class map.
constexpr int HEAD_MAXK = 32;
template <typename T>
__global__ void __launch_bounds__(256) head_softmax_kernel(const T* __restrict__ in, float* __restrict__ probs, int32_t* __restrict__ pred,
                                                           int B, int IH, int IW, int OH, int OW, int K, float sy, float sx) {
  const int64_t total = (int64_t)B * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float fy = src_coord(oy, sy), fx = src_coord(ox, sx);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < IH ? y0 + 1 : IH - 1, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
    const float wy = fy - y0, wx = fx - x0;
    const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
    const T* base = in + (int64_t)b * IH * IW * K;
    const T* p00 = base + ((int64_t)y0 * IW + x0) * K;
    const T* p01 = base + ((int64_t)y0 * IW + x1) * K;
    const T* p10 = base + ((int64_t)y1 * IW + x0) * K;
    const T* p11 = base + ((int64_t)y1 * IW + x1) * K;
    float v[HEAD_MAXK], mx = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      v[k] = w00 * ldf(p00 + k) + w01 * ldf(p01 + k) + w10 * ldf(p10 + k) + w11 * ldf(p11 + k);     // same order as bilinear_fwd_kernel
      if (v[k] > mx) { mx = v[k]; arg = k; }                                                          // first maximum wins (torch.argmax)
    }
    if (pred) pred[i] = arg;
    if (probs) {
      float se = 0.f;
      for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - mx); se += v[k]; }
      const float inv = 1.f / se;
      float* dst = probs + i * K;
      for (int k = 0; k < K; ++k) dst[k] = v[k] * inv;
    }
  }
}
}  // namespace

extern "C" int rssf_head_upsample_softmax(const void* logits, float* probs, int32_t* pred, int B, int IH, int IW, int OH, int OW, int K,
                                          int dtype, void* stream) {
  RSSF_REQUIRE(logits && (probs || pred) && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && K > 0 && K <= HEAD_MAXK,
               "head_upsample_softmax: bad arguments (K <= %d)", HEAD_MAXK);
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  const int64_t total = (int64_t)B * OH * OW;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) head_softmax_kernel<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)logits, probs, pred, B, IH, IW, OH, OW, K, sy, sx);
  else if (dtype == RSSF_BF16) head_softmax_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)logits, probs, pred, B, IH, IW, OH, OW, K, sy, sx);
  else { set_error("head_upsample_softmax: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("head_upsample_softmax");
}

extern "C" int rssf_upsample_bilinear_slice(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int ld_wide,
                                            int backward, int dtype, void* stream) {
  RSSF_REQUIRE(in && out && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && ld_wide >= C, "upsample_bilinear: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return bilinear_launch<float>(in, out, B, IH, IW, OH, OW, C, ld_wide, backward, st);
  if (dtype == RSSF_BF16) return bilinear_launch<bf16_t>(in, out, B, IH, IW, OH, OW, C, ld_wide, backward, st);
  set_error("upsample_bilinear: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_upsample_bilinear(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, int backward, int dtype,
                                      void* stream) {
  return rssf_upsample_bilinear_slice(in, out, B, IH, IW, OH, OW, C, C, backward, dtype, stream);
}

// ---- the fuse sum of a HighResolutionModule output in ONE pass (_hrnet_rssformer.py:424-435): out = sum_k up(term_k, s_k), s_k = 1
// for the terms that live at the output's resolution (the branch itself, the down-sampling chains), 2 / 4 / 8 for the 1x1 paths
// from the lower-resolution branches.  The chain of nearest_add launches read and wrote the running sum once per term; here every
// term is read once and the sum is written once (fp32 accumulation in the reference's order of j, rounded once).  Backward: the
// s x s block sums of the output gradient for every term with s > 1, all terms in one grid (the identity terms take the gradient
// itself).
namespace {
constexpr int NS_MAX = 4;
struct NearestSumArgs {
  const void* term[NS_MAX];      // forward: [B, OH / s, OW / s, C]; backward: the gradients to write (null for s == 1)
  int scale[NS_MAX];
  int start[NS_MAX + 1];         // backward: first block of each term
  int n, B, OH, OW, C;
};
template <typename T>
__global__ void __launch_bounds__(256) nearest_sum_fwd_kernel(NearestSumArgs a, T* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  const int cols = a.C / VEC;
  const int64_t total = (int64_t)a.B * a.OH * a.OW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ox = (int)(p % a.OW); p /= a.OW;
    const int oy = (int)(p % a.OH);
    const int b = (int)(p / a.OH);
    Vec<T> v[NS_MAX];
#pragma unroll
    for (int k = 0; k < NS_MAX; ++k)
      if (k < a.n) {
        const int s = a.scale[k], ih = a.OH / s, iw = a.OW / s;
        v[k].load((const T*)a.term[k] + (((int64_t)b * ih + oy / s) * iw + ox / s) * a.C + cv * VEC);
      }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NS_MAX; ++k)
      if (k < a.n) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[k].get(e);
      }
    Vec<T> r;
    r.set_all(acc);
    r.store(out + i * VEC);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) nearest_sum_bwd_kernel(NearestSumArgs a, const T* __restrict__ dout) {
  constexpr int VEC = Vec<T>::N;
  int k = 0;
#pragma unroll
  for (int j = 1; j < NS_MAX; ++j)
    if (j < a.n && blockIdx.x >= (unsigned)a.start[j]) k = j;
  const int s = a.scale[k], IH = a.OH / s, IW = a.OW / s, cols = a.C / VEC;
  const int64_t total = (int64_t)a.B * IH * IW * cols;
  const int64_t nblk = a.start[k + 1] - a.start[k];
  T* din = (T*)const_cast<void*>(a.term[k]);
  for (int64_t i = (int64_t)(blockIdx.x - a.start[k]) * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int dy = 0; dy < s; ++dy)
      for (int dx = 0; dx < s; ++dx) {
        Vec<T> v;
        v.load(dout + (((int64_t)b * a.OH + iy * s + dy) * a.OW + ix * s + dx) * a.C + cv * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
      }
    Vec<T> o;
    o.set_all(acc);
    o.store(din + i * VEC);
  }
}
template <typename T>
int nearest_sum_launch(const void* const* terms, const int* scales, int n, void* io, int B, int OH, int OW, int C, int backward, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  NearestSumArgs a = {};
  a.B = B; a.OH = OH; a.OW = OW; a.C = C;
  if (!backward) {
    a.n = n;
    for (int k = 0; k < n; ++k) { a.term[k] = terms[k]; a.scale[k] = scales[k]; }
    nearest_sum_fwd_kernel<T><<<grid_for((int64_t)B * OH * OW * (C / V)), 256, 0, st>>>(a, (T*)io);
    return check_launch("upsample_nearest_sum");
  }
  int m = 0, blocks = 0;
  for (int k = 0; k < n; ++k) {
    if (scales[k] == 1) continue;                              // identity terms take the gradient itself: nothing to write
    a.term[m] = terms[k]; a.scale[m] = scales[k];
    a.start[m] = blocks;
    blocks += grid_for((int64_t)B * (OH / scales[k]) * (OW / scales[k]) * (C / V));
    ++m;
  }
  if (m == 0) return RSSF_OK;
  a.n = m;
  for (int k = m; k <= NS_MAX; ++k) a.start[k] = blocks;
  nearest_sum_bwd_kernel<T><<<blocks, 256, 0, st>>>(a, (const T*)io);
  return check_launch("upsample_nearest_sum(bwd)");
}
}  // namespace

extern "C" int rssf_upsample_nearest_sum(const void* const* terms, const int* scales, int nterms, void* io, int B, int OH, int OW, int C,
                                         int backward, int dtype, void* stream) {
  RSSF_REQUIRE(terms && scales && io && nterms >= 1 && nterms <= NS_MAX && B > 0 && OH > 0 && OW > 0 && C > 0, "upsample_nearest_sum: bad arguments");
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  RSSF_REQUIRE(C % V == 0, "upsample_nearest_sum: %d channels are no multiple of the %d-element vector", C, V);
  for (int k = 0; k < nterms; ++k)
    RSSF_REQUIRE((terms[k] || (backward && scales[k] == 1)) && scales[k] >= 1 && OH % scales[k] == 0 && OW % scales[k] == 0,
                 "upsample_nearest_sum: bad term %d (scale %d)", k, scales[k]);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return nearest_sum_launch<float>(terms, scales, nterms, io, B, OH, OW, C, backward, st);
  if (dtype == RSSF_BF16) return nearest_sum_launch<bf16_t>(terms, scales, nterms, io, B, OH, OW, C, backward, st);
  set_error("upsample_nearest_sum: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}

extern "C" int rssf_upsample_nearest_add(const void* acc, const void* in, void* out, int B, int IH, int IW, int scale, int C, int backward,
                                         int dtype, void* stream) {
  RSSF_REQUIRE(in && out && B > 0 && IH > 0 && IW > 0 && scale >= 1 && C > 0, "upsample_nearest_add: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return nearest_launch<float>(acc, in, out, B, IH, IW, scale, C, backward, st);
  if (dtype == RSSF_BF16) return nearest_launch<bf16_t>(acc, in, out, B, IH, IW, scale, C, backward, st);
  set_error("upsample_nearest_add: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}
