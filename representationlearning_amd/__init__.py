"""representationlearning_amd — MI355X-native RSSFormer training-step hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed = plumbing); the
arithmetic of the hot path lives in `lib/librssf.so` (hand-written HIP for gfx950, C ABI in
include/rssf.h).  There is no CPU or eager fallback: importing `ops` without the built library, or
calling an op on a non-GPU tensor, raises.
"""
__version__ = "0.1.0"
