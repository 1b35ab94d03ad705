"""Mhca — multi-head cross attention with the per-(window, head) sigmoid "channel alpha"
(reference: modules/DAL.py:676-1030).  Parameters and state_dict keys match the reference; the arithmetic
runs in the fused HIP window-attention kernel (csrc/win_attn_{fwd,bwd}.hip)."""
import torch
import torch.nn as nn

from ..... import autograd as AG


class Mhca(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        if (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)) or add_bias_kv or add_zero_attn or not bias:
            raise NotImplementedError("Mhca (HIP): only the configuration used by RSSFormer is built "
                                      "(kdim=vdim=embed_dim, bias=True, no bias_kv / zero_attn)")
        if dropout != 0.0:
            raise NotImplementedError("Mhca (HIP): attention dropout is 0.0 on the RSSFormer path")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)

    def proj_params(self):
        return (self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.k_proj.bias,
                self.v_proj.weight, self.v_proj.bias, self.out_proj.weight, self.out_proj.bias)

    def forward(self, query, key, value, **unused):
        """Seq-first [L=49, nWin, C] like the reference; every window is one 7x7 group (no gate, no LN)."""
        if key is not value:
            raise NotImplementedError("Mhca (HIP): key and value are the same tensor on the RSSFormer path")
        L, nW, C = query.shape
        if L != 49:
            raise NotImplementedError("Mhca (HIP): built for 7x7 local groups (L=49)")
        x = query.transpose(0, 1).contiguous()
        y = key.transpose(0, 1).contiguous()
        out = AG.PlainWindowCrossAttention.apply(x, y, *self.proj_params(), 7, 7, self.num_heads)
        return out.transpose(0, 1)
