// Device input pipeline: the reference's training transform chain on LoveDA tiles
//   RandomCrop(512,512) -> OneOf([HorizontalFlip, VerticalFlip, RandomRotate90], p=0.75) -> Normalize(mean, std,
//   max_pixel_value=1) -> ToTensor, and `mask = imread(...).astype(long) - 1`   (configs/base/loveda.py:18-36,
//   data/loveda.py:82-91)
// as ONE gather kernel over a device-resident uint8 dataset: an output pixel reads its source pixel (crop offset, then the
// flip / rot90 index map), normalises the three channels and writes the channels-last image and the shifted int64 label.
// The random draws stay on the host (a handful of integers per image).  ShiftScaleRotate (p=0.2, an OpenCV affine warp with
// fixed-point bilinear taps) is NOT covered - see DESIGN.md §7.
// HBM-bound by construction: 4 B read (3 + 1), 3 x sizeof(T) + 8 B written per output pixel.
#include "common.hip.h"
using namespace rssf;

namespace {

template <typename T>
__global__ void __launch_bounds__(256) input_pipeline_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                             const int* __restrict__ params, T* __restrict__ out_img,
                                                             int64_t* __restrict__ out_mask, int B, int SH, int SW, int OH, int OW,
                                                             float m0, float m1, float m2, float r0, float r1, float r2) {
  const int64_t total = (int64_t)B * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), b = (int)(i / ((int64_t)OW * OH));
    const int src = params[4 * b], y0 = params[4 * b + 1], x0 = params[4 * b + 2], op = params[4 * b + 3];
    // index map of the geometric op on the OH x OW crop (np.rot90 is counter-clockwise; the crop is square for k odd)
    int cy = oy, cx = ox;
    switch (op) {
      case RSSF_AUG_HFLIP: cx = OW - 1 - ox; break;                       // img[:, ::-1]
      case RSSF_AUG_VFLIP: cy = OH - 1 - oy; break;                       // img[::-1]
      case RSSF_AUG_ROT90 + 1: cy = ox; cx = OW - 1 - oy; break;           // out[i][j] = m[j][W-1-i]
      case RSSF_AUG_ROT90 + 2: cy = OH - 1 - oy; cx = OW - 1 - ox; break;
      case RSSF_AUG_ROT90 + 3: cy = OH - 1 - ox; cx = oy; break;           // out[i][j] = m[H-1-j][i]
      default: break;                                                    // none, rot90 k = 0
    }
    const int64_t sp = ((int64_t)src * SH + (y0 + cy)) * SW + (x0 + cx);
    const uint8_t* p = img + sp * 3;
    // albumentations Normalize: float32(img) - mean, then * reciprocal(std)  (two roundings, no fused multiply-add)
    const float v0 = __fmul_rn(__fsub_rn((float)p[0], m0), r0);
    const float v1 = __fmul_rn(__fsub_rn((float)p[1], m1), r1);
    const float v2 = __fmul_rn(__fsub_rn((float)p[2], m2), r2);
    T* o = out_img + i * 3;
    stf(o, v0); stf(o + 1, v1); stf(o + 2, v2);
    if (out_mask) out_mask[i] = (int64_t)mask[sp] - 1;                   // data/loveda.py:84: no-data 0 -> ignore -1
  }
}

}  // namespace

extern "C" int rssf_input_pipeline(const uint8_t* img, const uint8_t* mask, const int* params, void* out_img, int64_t* out_mask,
                                   int B, int nsrc, int SH, int SW, int OH, int OW, const float* mean3, const float* std3,
                                   float max_pixel_value, int dtype, void* stream) {
  RSSF_REQUIRE(img && params && out_img && mean3 && std3 && B > 0 && nsrc > 0 && SH >= OH && SW >= OW && OH > 0 && OW > 0,
               "input_pipeline: bad arguments");
  RSSF_REQUIRE((mask == nullptr) == (out_mask == nullptr), "input_pipeline: mask and out_mask go together");
  float m[3], r[3];
  for (int c = 0; c < 3; ++c) {
    m[c] = mean3[c] * max_pixel_value;                    // float32 products, as numpy does them
    const float s = std3[c] * max_pixel_value;
    RSSF_REQUIRE(s != 0.f, "input_pipeline: zero std");
    r[c] = 1.0f / s;                                      // np.reciprocal(float32): correctly rounded
  }
  const int64_t total = (int64_t)B * OH * OW;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    input_pipeline_kernel<float><<<(unsigned)blocks, 256, 0, st>>>(img, mask, params, (float*)out_img, out_mask, B, SH, SW, OH, OW, m[0],
                                                                  m[1], m[2], r[0], r[1], r[2]);
  else if (dtype == RSSF_BF16)
    input_pipeline_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>(img, mask, params, (bf16_t*)out_img, out_mask, B, SH, SW, OH, OW,
                                                                   m[0], m[1], m[2], r[0], r[1], r[2]);
  else { set_error("input_pipeline: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("input_pipeline");
}
