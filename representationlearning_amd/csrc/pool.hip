// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on channels-last activations - the stem pooling of the ResNet-50 CAM path
// (WaveCAM-TMM2023/net/resnet50.py:68, 88; BASELINE config 5's conv-only relative).  Inference only
// (the reference freezes the stem; the CAM path runs under no_grad).  HBM-bound: one read of the input (each pixel is touched by
// <= 4 windows, served by L2), one write of the quarter-size output; a thread owns one output pixel x one 16-byte channel group.
#include "common.hip.h"
using namespace rssf;

namespace {
template <typename T, int VEC>
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int IH, int IW, int OH,
                                                           int OW, int C) {
  const int cols = C / VEC;
  const int64_t total = (int64_t)B * OH * OW * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cols);
    int64_t p = i / cols;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;                       // padding never wins (torch pads with -inf)
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= IH) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= IW) continue;
        const T* src = in + (((int64_t)b * IH + iy) * IW + ix) * C + cv * VEC;
        if constexpr (VEC > 1) {
          Vec<T> v;
          v.load(src);
#pragma unroll
          for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], v.get(e));
        } else {
          m[0] = fmaxf(m[0], ldf(src));
        }
      }
    }
    T* dst = out + (((int64_t)b * OH + oy) * OW + ox) * C + cv * VEC;
    if constexpr (VEC > 1) {
      Vec<T> o;
      o.set_all(m);
      o.store(dst);
    } else {
      stf(dst, m[0]);
    }
  }
}

template <typename T>
int launch(const void* in, void* out, int B, int IH, int IW, int OH, int OW, int C, hipStream_t st) {
  constexpr int V = Vec<T>::N;
  const bool vec = C % V == 0;
  const int64_t total = (int64_t)B * OH * OW * (vec ? C / V : C);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec) maxpool3x3s2_kernel<T, V><<<(unsigned)blocks, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C);
  else maxpool3x3s2_kernel<T, 1><<<(unsigned)blocks, 256, 0, st>>>((const T*)in, (T*)out, B, IH, IW, OH, OW, C);
  return check_launch("maxpool3x3s2");
}
}  // namespace

extern "C" int rssf_maxpool3x3s2(const void* in, void* out, int B, int IH, int IW, int C, int dtype, void* stream) {
  RSSF_REQUIRE(in && out && B > 0 && IH > 0 && IW > 0 && C > 0, "maxpool3x3s2: bad arguments");
  const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32) return launch<float>(in, out, B, IH, IW, OH, OW, C, st);
  if (dtype == RSSF_BF16) return launch<bf16_t>(in, out, B, IH, IW, OH, OW, C, st);
  set_error("maxpool3x3s2: unsupported dtype %d", dtype);
  return RSSF_ERR_UNSUPPORTED;
}
