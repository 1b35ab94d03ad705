"""Shared helpers for the parity tests (oracle + golden fixtures)."""
import os

import numpy as np
import torch

from oracle.procedural import procedural_state, seeded_state

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def proc_params(template, requires_grad=True):
    P = procedural_state(template)
    for k, v in P.items():
        if torch.is_floating_point(v) and requires_grad and "running" not in k:
            v.requires_grad_()
    return P


def seeded_params(template, requires_grad=True, seed=1234):
    P = seeded_state(template, seed)
    for k, v in P.items():
        if torch.is_floating_point(v) and requires_grad and "running" not in k:
            v.requires_grad_()
    return P


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
