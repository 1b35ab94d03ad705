// Flat-buffer optimizer kernels: global L2 gradient norm, and a fused {grad-scale, clip, weight-decay, momentum,
// update} SGD step (reference recipe: configs/base/loveda.py:68-99 — SGD lr 0.01 poly 0.9, momentum 0.9,
// wd 1e-4, clip_grad_norm 35; the external `ever` trainer applies torch.optim.SGD semantics).
// One launch over all 32 M parameters instead of ~1.1 k per-tensor launches; HBM-bound (16 B/param read+write).
#include "common.hip.h"
using namespace rssf;

namespace {

// Two levels, fixed order (no float atomics): block partials into out[1 .. blocks], then one block adds them in sequence into
// out[0].  Data-parallel replicas hold bit-identical gradients after the all-reduce and must take bit-identical clip
// coefficients from them, or the replicas drift apart by an ulp per step (tests/test_gpu_dp.py).
__global__ void __launch_bounds__(256) sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int64_t j = i; j < n; ++j) acc += g[j] * g[j];
    }
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[1 + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(256) sqnorm_fold_kernel(float* __restrict__ out, int blocks) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) acc += out[1 + i];
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += part[i];
    out[0] = t;
  }
}

// torch.optim.SGD semantics (dampening 0, nesterov off):  g' = s*g*clip + wd*p ; buf = mu*buf + g' ; p -= lr*buf
// clip = min(1, max_norm / (s*sqrt(sqnorm) + 1e-6))   (torch.nn.utils.clip_grad_norm_), s = grad_scale (1/world).
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, const float* __restrict__ sqnorm, float grad_scale,
                                                  float max_norm, const float* __restrict__ lr_ptr, float lr_host, float momentum,
                                                  float wd, int first_step) {
  const float lr = lr_ptr ? *lr_ptr : lr_host;     // device-resident LR: the launch can live in a replayed hipGraph
  float coef = grad_scale;
  if (max_norm > 0.f) {
    const float nrm = grad_scale * sqrtf(*sqnorm);
    coef *= fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 pv = *reinterpret_cast<float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 bv = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(buf + i);
      float* pp = &pv.x; const float* gp = &gv.x; float* bp = &bv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = gp[k] * coef + wd * pp[k];
        bp[k] = first_step ? d : momentum * bp[k] + d;
        pp[k] -= lr * bp[k];
      }
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(buf + i) = bv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        const float d = g[j] * coef + wd * p[j];
        const float b = first_step ? d : momentum * buf[j] + d;
        buf[j] = b;
        p[j] -= lr * b;
      }
    }
  }
}
}  // namespace

namespace {
__global__ void __launch_bounds__(256) zero_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}
}  // namespace
namespace rssf {
int zero_floats(float* p, int64_t n, hipStream_t st) {
  if (n <= 0) return RSSF_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  zero_kernel<<<(unsigned)blocks, 256, 0, st>>>(p, n);
  return check_launch("zero_floats");
}
}  // namespace rssf

extern "C" int rssf_zero_f32(float* p, int64_t n, void* stream) {
  RSSF_REQUIRE(p || n == 0, "zero_f32: null buffer");
  return zero_floats(p, n, (hipStream_t)stream);
}

extern "C" int rssf_grad_sqnorm(const float* g, int64_t n, float* out, void* stream) {
  RSSF_REQUIRE(g && out && n > 0, "grad_sqnorm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > RSSF_SQNORM_BLOCKS) blocks = RSSF_SQNORM_BLOCKS;
  if (blocks < 1) blocks = 1;
  sqnorm_kernel<<<(unsigned)blocks, 256, 0, st>>>(g, n, out);
  int rc = check_launch("grad_sqnorm");
  if (rc) return rc;
  sqnorm_fold_kernel<<<1, 256, 0, st>>>(out, (int)blocks);
  return check_launch("grad_sqnorm(fold)");
}

extern "C" int rssf_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, const float* sqnorm, float grad_scale,
                             float max_norm, const float* lr_dev, float lr, float momentum, float weight_decay, int first_step,
                             void* stream) {
  RSSF_REQUIRE(p && g && momentum_buf && n > 0, "sgd_step: bad arguments");
  RSSF_REQUIRE(max_norm <= 0.f || sqnorm, "sgd_step: clipping needs the squared-norm buffer");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  sgd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(p, g, momentum_buf, n, sqnorm, grad_scale, max_norm, lr_dev, lr,
                                                                  momentum, weight_decay, first_step);
  return check_launch("sgd_step");
}
