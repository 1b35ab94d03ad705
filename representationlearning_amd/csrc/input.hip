// Device input pipeline: the reference's training transform chain on LoveDA tiles
//   RandomCrop(512,512) -> OneOf([HorizontalFlip, VerticalFlip, RandomRotate90], p=0.75) -> Normalize(mean, std,
//   max_pixel_value=1) -> ToTensor, and `mask = imread(...).astype(long) - 1`   (configs/base/loveda.py:18-36,
//   data/loveda.py:82-91)
// as ONE gather kernel over a device-resident uint8 dataset: an output pixel reads its source pixel (crop offset, then the
// flip / rot90 index map), normalises the three channels and writes the channels-last image and the shifted int64 label.
// The random draws stay on the host (a handful of numbers per image).  ShiftScaleRotate (p=0.2: cv2.warpAffine with fixed-point
// bilinear taps, nearest for the mask, BORDER_REFLECT_101) sits between the flips and Normalize, as in the reference's Compose.
// HBM-bound by construction: 4 B read (3 + 1), 3 x sizeof(T) + 8 B written per output pixel.
#include "common.hip.h"
using namespace rssf;

namespace {

// index map of the geometric op on the OH x OW crop (np.rot90 is counter-clockwise; the crop is square for k odd): position
// (oy, ox) of the transformed crop reads position (cy, cx) of the plain crop
__device__ __forceinline__ void op_map(int op, int oy, int ox, int OH, int OW, int& cy, int& cx) {
  cy = oy; cx = ox;
  switch (op) {
    case RSSF_AUG_HFLIP: cx = OW - 1 - ox; break;                       // img[:, ::-1]
    case RSSF_AUG_VFLIP: cy = OH - 1 - oy; break;                       // img[::-1]
    case RSSF_AUG_ROT90 + 1: cy = ox; cx = OW - 1 - oy; break;           // out[i][j] = m[j][W-1-i]
    case RSSF_AUG_ROT90 + 2: cy = OH - 1 - oy; cx = OW - 1 - ox; break;
    case RSSF_AUG_ROT90 + 3: cy = OH - 1 - ox; cx = oy; break;           // out[i][j] = m[H-1-j][i]
    default: break;                                                    // none, rot90 k = 0
  }
}
// cv::borderInterpolate(p, len, BORDER_REFLECT_101): ... 2 1 | 0 1 2 ... len-1 | len-2 len-3 ...
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// ShiftScaleRotate = cv2.warpAffine(crop, M, (W, H), INTER_LINEAR / INTER_NEAREST (mask), BORDER_REFLECT_101) restated (OpenCV
// imgwarp.cpp, WarpAffineInvoker + remapBilinear / remapNearest on 8-bit data): source coordinates in fixed point with
// AB_BITS = 10 (per-column term cvRound(iM[0]*x*1024), per-row term cvRound((iM[1]*y + iM[2])*1024) + rounding offset), 5
// fractional bits kept for the bilinear taps, tap weights = products of 1/32 steps scaled to 2^15 (exact integers), result
// (sum + 2^14) >> 15.  `aff` = the INVERSE matrix iM (6 doubles, computed on the host as cv::invertAffineTransform does).
template <typename T>
__global__ void __launch_bounds__(256) input_pipeline_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                             const int* __restrict__ params, const double* __restrict__ affine,
                                                             T* __restrict__ out_img, int64_t* __restrict__ out_mask, int B, int SH, int SW,
                                                             int OH, int OW, float m0, float m1, float m2, float r0, float r1, float r2) {
  const int64_t total = (int64_t)B * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), b = (int)(i / ((int64_t)OW * OH));
    const int src = params[4 * b], y0 = params[4 * b + 1], x0 = params[4 * b + 2], op = params[4 * b + 3];
    const uint8_t* tile = img + (int64_t)src * SH * SW * 3;
    const uint8_t* mtile = mask ? mask + (int64_t)src * SH * SW : nullptr;
    const double* A = affine ? affine + 6 * b : nullptr;
    float p0, p1, p2;
    int64_t lab = 0;
    if (A && A[0] == A[0]) {                                            // NaN in A[0]: this image is not warped
      const int adx = __double2int_rn(A[0] * ox * 1024.0), bdx = __double2int_rn(A[3] * ox * 1024.0);
      const int X0 = __double2int_rn((A[1] * oy + A[2]) * 1024.0), Y0 = __double2int_rn((A[4] * oy + A[5]) * 1024.0);
      // bilinear taps of the image
      const int X = (X0 + 16 + adx) >> 5, Y = (Y0 + 16 + bdx) >> 5;     // round_delta = AB_SCALE / INTER_TAB_SIZE / 2
      const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
      const int w[4] = {(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32};
      int acc[3] = {1 << 14, 1 << 14, 1 << 14};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ty = reflect101(sy + (t >> 1), OH), tx = reflect101(sx + (t & 1), OW);
        int cy, cx;
        op_map(op, ty, tx, OH, OW, cy, cx);
        const uint8_t* p = tile + ((int64_t)(y0 + cy) * SW + (x0 + cx)) * 3;
        acc[0] += p[0] * w[t]; acc[1] += p[1] * w[t]; acc[2] += p[2] * w[t];
      }
      p0 = (float)(acc[0] >> 15); p1 = (float)(acc[1] >> 15); p2 = (float)(acc[2] >> 15);       // <= 255 by construction
      if (out_mask) {                                                                            // nearest: round_delta = AB_SCALE / 2
        const int nx = reflect101((X0 + 512 + adx) >> 10, OW), ny = reflect101((Y0 + 512 + bdx) >> 10, OH);
        int cy, cx;
        op_map(op, ny, nx, OH, OW, cy, cx);
        lab = (int64_t)mtile[(int64_t)(y0 + cy) * SW + (x0 + cx)] - 1;
      }
    } else {
      int cy, cx;
      op_map(op, oy, ox, OH, OW, cy, cx);
      const int64_t sp = (int64_t)(y0 + cy) * SW + (x0 + cx);
      const uint8_t* p = tile + sp * 3;
      p0 = (float)p[0]; p1 = (float)p[1]; p2 = (float)p[2];
      if (out_mask) lab = (int64_t)mtile[sp] - 1;                         // data/loveda.py:84: no-data 0 -> ignore -1
    }
    // albumentations Normalize: float32(img) - mean, then * reciprocal(std)  (two roundings, no fused multiply-add)
    T* o = out_img + i * 3;
    stf(o, __fmul_rn(__fsub_rn(p0, m0), r0)); stf(o + 1, __fmul_rn(__fsub_rn(p1, m1), r1)); stf(o + 2, __fmul_rn(__fsub_rn(p2, m2), r2));
    if (out_mask) out_mask[i] = lab;
  }
}

}  // namespace

extern "C" int rssf_input_pipeline(const uint8_t* img, const uint8_t* mask, const int* params, const double* affine, void* out_img,
                                   int64_t* out_mask, int B, int nsrc, int SH, int SW, int OH, int OW, const float* mean3,
                                   const float* std3, float max_pixel_value, int dtype, void* stream) {
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
    input_pipeline_kernel<float><<<(unsigned)blocks, 256, 0, st>>>(img, mask, params, affine, (float*)out_img, out_mask, B, SH, SW, OH, OW, m[0],
                                                                  m[1], m[2], r[0], r[1], r[2]);
  else if (dtype == RSSF_BF16)
    input_pipeline_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>(img, mask, params, affine, (bf16_t*)out_img, out_mask, B, SH, SW, OH, OW,
                                                                   m[0], m[1], m[2], r[0], r[1], r[2]);
  else { set_error("input_pipeline: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("input_pipeline");
}
