// Evaluation metric kernel: argmax over classes + ignore mask + confusion matrix in one pass over channels-last scores.
// Reference: eval.py:66-71 (`pred.argmax(dim=1)`, `valid_inds = y_true != -1`, `er.metric.PixelMetric.forward`) and
// the in-tree confusion-matrix recipe SCD-AAAI2023/utils/evaluate.py:9-35 (bincount of true * K + pred).
// argmax(softmax(z)) == argmax(z), so the scores may be logits or probabilities.  First maximum wins (torch.argmax).
// HBM-bound: one read of [B, HW, K] scores + [B, HW] labels; K*K int64 counters accumulated through an LDS histogram.
#include "common.hip.h"
using namespace rssf;

namespace {
constexpr int MAXK = 32;

template <typename T>
__global__ void __launch_bounds__(256) argmax_confusion_kernel(const T* __restrict__ scores, const int64_t* __restrict__ labels,
                                                               int32_t* __restrict__ pred, unsigned long long* __restrict__ cm,
                                                               int64_t npix, int K, int ignore_index) {
  __shared__ unsigned int hist[MAXK * MAXK];
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const T* s = scores + p * K;
    float best = ldf(s);
    int arg = 0;
    for (int k = 1; k < K; ++k) {
      const float v = ldf(s + k);
      if (v > best) { best = v; arg = k; }
    }
    if (pred) pred[p] = arg;
    const int64_t y = labels ? labels[p] : (int64_t)ignore_index;
    if (labels && y != ignore_index && y >= 0 && y < K) atomicAdd(&hist[(int)y * K + arg], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += blockDim.x)
    if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}
}  // namespace

extern "C" int rssf_argmax_confusion(const void* scores, const int64_t* labels, int32_t* pred, int64_t* cm, int64_t npix, int K,
                                     int ignore_index, int dtype, void* stream) {
  RSSF_REQUIRE(scores && npix > 0 && K >= 1 && K <= MAXK, "argmax_confusion: bad arguments (K <= %d)", MAXK);
  RSSF_REQUIRE(pred || (labels && cm), "argmax_confusion: nothing to produce");
  RSSF_REQUIRE(!labels || cm, "argmax_confusion: labels without a confusion matrix");
  int64_t blocks = (npix + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RSSF_F32)
    argmax_confusion_kernel<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)scores, labels, pred, (unsigned long long*)cm, npix, K,
                                                                     ignore_index);
  else if (dtype == RSSF_BF16)
    argmax_confusion_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)scores, labels, pred, (unsigned long long*)cm, npix, K,
                                                                      ignore_index);
  else { set_error("argmax_confusion: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  return check_launch("argmax_confusion");
}
