"""Procedural (formula-defined) weights and inputs shared by the golden generator,
the oracle tests and the GPU parity tests (SURVEY.md §8c "fixture recipe").

No RNG-order dependence and no weight blobs in the repo: every tensor is a closed
form of its index, so the reference (in the build container), the CPU oracle and the
HIP path can all regenerate identical parameters; fixtures store outputs only.
"""
import math

import torch


def proc_values(numel, tid, freq=0.731):
    i = torch.arange(numel, dtype=torch.float64)
    return torch.sin(freq * i + 1.37 * tid)


def procedural_state(template):
    """template: ordered mapping name -> tensor (shapes/dtypes only are used).

    Returns name -> tensor with
      >=2-D float  : sin(.)/sqrt(fan_in)
      1-D 'weight' : 1 + 0.1 sin(.)
      running_var  : 1 + 0.25 sin(.)^2
      other 1-D    : 0.05 sin(.)          (biases, running_mean)
      integer      : zeros                (num_batches_tracked)
    tid = rank of the name in sorted order.
    """
    out = {}
    for tid, name in enumerate(sorted(template.keys())):
        t = template[name]
        if not torch.is_floating_point(t):
            out[name] = torch.zeros_like(t)
            continue
        v = proc_values(t.numel(), tid).reshape(t.shape)
        if t.dim() >= 2:
            fan_in = t[0].numel()
            v = v / math.sqrt(fan_in)
        elif name.endswith("running_var"):
            v = 1.0 + 0.25 * v * v
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * v
        else:
            v = 0.05 * v
        out[name] = v.to(t.dtype)
    return out


def proc_input(shape, phase, freq=0.377):
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    return torch.sin(freq * i + phase).reshape(shape).float()


def proc_labels(B, H, W, classes=6, block=4, phase=0):
    """Blocky integer labels in [-1, classes) (-1 = ignore), deterministic."""
    bh, bw = (H + block - 1) // block, (W + block - 1) // block
    i = torch.arange(B * bh * bw, dtype=torch.int64).reshape(B, bh, bw)
    v = (i * 7 + (i // 3) * 5 + phase) % (classes + 1) - 1
    v = v.repeat_interleave(block, 1).repeat_interleave(block, 2)
    return v[:, :H, :W].contiguous()


# ------------------------------------------------------------------------------------------------------
# Seeded pseudo-random state.  The sin-based tensors above are fine for single ops, but a whole network built
# from them is numerically chaotic (train-mode BatchNorm over nearly-constant channels: the CPU oracle in fp32
# and in fp64 already disagree by 20-80 % at stage 3/4), so multi-layer fixtures (transformer block, MLP,
# full model) use weights drawn from a seeded CPU generator in sorted-key order instead: deterministic for a
# given torch build, PyTorch-default-like scales, fp32-vs-fp64 distance ~6e-5 at the logits.
def seeded_state(template, seed=1234):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(template.keys()):
        t = template[name]
        if not torch.is_floating_point(t):
            out[name] = torch.zeros_like(t)
            continue
        if t.dim() >= 2:
            bound = math.sqrt(3.0 / t[0].numel())
            v = (torch.rand(t.shape, generator=g) * 2 - 1) * bound
        elif name.endswith("running_var"):
            v = 1.0 + 0.2 * torch.rand(t.shape, generator=g)
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        else:
            v = 0.05 * torch.randn(t.shape, generator=g)
        out[name] = v.to(t.dtype)
    return out


def seeded_input(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
