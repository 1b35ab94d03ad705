// Small glue launches that replace chains of framework element-wise kernels on the training step's critical path:
//   * the three biases of MlpDWBN's summed convolutions (ffn_block.py:226-228, 250-257: dw(u) + dw6(u) + dw12(u), all with bias)
//     add up to ONE epilogue bias, and the bias gradient of the fused launch belongs to all three - was stack + sum (3 launches)
//     forward and three adds backward per block,
//   * the network input: fp32 NCHW (or channels-last) image -> channels-last activations of the compute dtype, channels zero-padded
//     to the 16-byte vector width (3 -> 8 bf16 / 4 fp32) - was dtype cast + layout copy + fill + pad copy, and the pad again in
//     the backward pass of the stem (HighResolutionNet.forward, _hrnet_rssformer.py:605-613).
#include "common.hip.h"
using namespace rssf;

namespace {

__global__ void __launch_bounds__(256) vec_sum3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                       float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (a[i] + b[i]) + (c ? c[i] : 0.f);
}

__global__ void __launch_bounds__(256) vec_add_to3_kernel(const float* __restrict__ src, float* __restrict__ d0, float* __restrict__ d1,
                                                          float* __restrict__ d2, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = src[i];
  d0[i] += v;
  if (d1) d1[i] += v;
  if (d2) d2[i] += v;
}

// out = a + b over n elements (out may be a or b): the few sums of the step that no convolution / BatchNorm epilogue carries - the
// running sums of a HighResolutionModule's fuse outputs (_hrnet_rssformer.py:424-435) and the gradient of a tensor that has both
// convolution and non-convolution consumers (nnf._Fanout) - as the library's own launch: the training step holds no framework kernel
template <typename T>
// (no __restrict__: out may alias a or b - every thread reads its elements before it writes them)
__global__ void __launch_bounds__(256) add_kernel(const T* a, const T* b, T* out, int64_t nvec, int64_t n) {
  constexpr int V = Vec<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    Vec<T> x, y, o;
    x.load(a + i * V); y.load(b + i * V);
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = x.get(e) + y.get(e);
    o.set_all(v);
    o.store(out + i * V);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - nvec * V)) {             // tail shorter than one vector
    const int64_t i = nvec * V + threadIdx.x;
    stf(out + i, ldf(a + i) + ldf(b + i));
  }
}

// out = a + b + c (one pass instead of two): the gradient of a branch output that meets an accumulated convolution gradient AND two
// autograd consumers (nnf._Fanout with aliases) - summed in fp32, rounded once
template <typename T>
__global__ void __launch_bounds__(256) add3_kernel(const T* a, const T* b, const T* c, T* out,      // may alias, as in add_kernel
                                                   int64_t nvec, int64_t n) {
  constexpr int V = Vec<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    Vec<T> x, y, z, o;
    x.load(a + i * V); y.load(b + i * V); z.load(c + i * V);
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = x.get(e) + y.get(e) + z.get(e);
    o.set_all(v);
    o.store(out + i * V);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - nvec * V)) {
    const int64_t i = nvec * V + threadIdx.x;
    stf(out + i, ldf(a + i) + ldf(b + i) + ldf(c + i));
  }
}

// dst[r][0..Cp) = src[r][0..C) followed by zeros (Cp a multiple of the 16-byte vector): one thread per output row
template <typename T>
__global__ void __launch_bounds__(256) pad_channels_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t rows, int C, int Cp) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (int c = 0; c < Cp; ++c) stf(dst + r * Cp + c, c < C ? ldf(src + r * C + c) : 0.f);
}

// dst[r][c] += src[r][c] for c < cols, row pitches ld_dst / ld_src (fp32): a gradient computed for a channel-padded problem added
// into the unpadded parameter gradient
__global__ void __launch_bounds__(256) add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows, int cols, int ld_dst, int ld_src) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  dst[(int64_t)r * ld_dst + c] += src[(int64_t)r * ld_src + c];
}

// one thread per output pixel: C strided reads (coalesced across the threads of a row for NCHW sources), one 16-byte store
template <typename T>
__global__ void __launch_bounds__(256) image_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t npix, int HW, int W,
                                                            int C, int64_t sb, int64_t sc, int64_t sh, int64_t sw) {
  constexpr int V = Vec<T>::N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int64_t b = i / HW;
  const int r = (int)(i - b * HW), y = r / W, x = r - y * W;
  const float* p = src + b * sb + y * sh + x * sw;
  float v[V];
#pragma unroll
  for (int c = 0; c < V; ++c) v[c] = c < C ? p[c * sc] : 0.f;
  Vec<T> o;
  o.set_all(v);
  o.store(dst + i * V);
}

}  // namespace

extern "C" int rssf_vec_sum3(const float* a, const float* b, const float* c, float* out, int n, void* stream) {
  RSSF_REQUIRE(a && b && out && n > 0, "vec_sum3: bad arguments");
  vec_sum3_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, b, c, out, n);
  return check_launch("vec_sum3");
}

extern "C" int rssf_vec_add_to3(const float* src, float* d0, float* d1, float* d2, int n, void* stream) {
  RSSF_REQUIRE(src && d0 && n > 0, "vec_add_to3: bad arguments");
  vec_add_to3_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, d0, d1, d2, n);
  return check_launch("vec_add_to3");
}

extern "C" int rssf_image_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                  int dtype, void* stream) {
  RSSF_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "image_to_nhwc: bad arguments");
  const int64_t npix = (int64_t)B * H * W;
  const unsigned blocks = (unsigned)((npix + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_BF16) {
    RSSF_REQUIRE(C >= 1 && C <= 8, "image_to_nhwc: 1..8 channels fit the bf16 vector (got %d)", C);
    image_to_nhwc_kernel<bf16_t><<<blocks, 256, 0, st>>>(src, reinterpret_cast<bf16_t*>(dst), npix, H * W, W, C, sb, sc, sh, sw);
  } else if (dtype == RSSF_F32) {
    RSSF_REQUIRE(C >= 1 && C <= 4, "image_to_nhwc: 1..4 channels fit the fp32 vector (got %d)", C);
    image_to_nhwc_kernel<float><<<blocks, 256, 0, st>>>(src, reinterpret_cast<float*>(dst), npix, H * W, W, C, sb, sc, sh, sw);
  } else {
    set_error("image_to_nhwc: unsupported dtype %d", dtype);
    return RSSF_ERR_UNSUPPORTED;
  }
  return check_launch("image_to_nhwc");
}

extern "C" int rssf_pad_channels(const void* src, void* dst, int64_t rows, int C, int Cp, int dtype, void* stream) {
  RSSF_REQUIRE(src && dst && rows > 0 && C > 0 && Cp >= C, "pad_channels: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  if (dtype == RSSF_BF16) pad_channels_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, rows, C, Cp);
  else if (dtype == RSSF_F32) pad_channels_kernel<float><<<blocks, 256, 0, st>>>((const float*)src, (float*)dst, rows, C, Cp);
  else { set_error("pad_channels: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("pad_channels");
}

extern "C" int rssf_add_rows(float* dst, const float* src, int rows, int cols, int ld_dst, int ld_src, void* stream) {
  RSSF_REQUIRE(dst && src && rows > 0 && cols > 0 && ld_dst >= cols && ld_src >= cols, "add_rows: bad arguments");
  const int64_t n = (int64_t)rows * cols;
  add_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dst, src, rows, cols, ld_dst, ld_src);
  return check_launch("add_rows");
}

extern "C" int rssf_add3(const void* a, const void* b, const void* c, void* out, int64_t n, int dtype, void* stream) {
  RSSF_REQUIRE(a && b && c && out && n > 0, "add3: bad arguments");
  RSSF_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0, "add3: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  const int64_t nvec = n / V;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == RSSF_BF16) add3_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)out, nvec, n);
  else if (dtype == RSSF_F32) add3_kernel<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)a, (const float*)b, (const float*)c, (float*)out, nvec, n);
  else { set_error("add3: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("add3");
}

extern "C" int rssf_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
  RSSF_REQUIRE(a && b && out && n > 0, "add: bad arguments");
  RSSF_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "add: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == RSSF_BF16 ? 8 : 4;
  const int64_t nvec = n / V;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == RSSF_BF16) add_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, nvec, n);
  else if (dtype == RSSF_F32) add_kernel<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)a, (const float*)b, (float*)out, nvec, n);
  else { set_error("add: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("add");
}
