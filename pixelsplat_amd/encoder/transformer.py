"""Attention / PreNorm / FeedForward / Transformer with the reference's parameter names and
call conventions (/root/reference/src/model/transformer/{attention,pre_norm,feed_forward,
transformer}.py, vendored there from stelzner/srt).  Plain PyTorch: the dense projections are
library GEMMs; the epipolar cross-attention is routed to the fused HIP kernel by
EpipolarTransformer, everything else (self-attention over 256 tokens / image) is out of the
hot path."""
from __future__ import annotations

import torch
from torch import nn


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, selfatt=True, kv_dim=None):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)   # visualisers hook this module
        if selfatt:
            self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        else:
            self.to_q = nn.Linear(dim, inner_dim, bias=False)
            self.to_kv = nn.Linear(kv_dim, inner_dim * 2, bias=False)
        self.to_out = (nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
                       if project_out else nn.Identity())

    def forward(self, x, z=None):
        if z is None:
            q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        else:
            q = self.to_q(x)
            k, v = self.to_kv(z).chunk(2, dim=-1)
        split = lambda t: t.reshape(t.shape[0], t.shape[1], self.heads, -1).transpose(1, 2)
        q, k, v = split(q), split(k), split(v)
        attn = self.attend(torch.matmul(q, k.transpose(-1, -2)) * self.scale)
        out = torch.matmul(attn, v).transpose(1, 2)
        return self.to_out(out.reshape(out.shape[0], out.shape[1], -1))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0, selfatt=True,
                 kv_dim=None, feed_forward_layer=FeedForward):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout,
                                       selfatt=selfatt, kv_dim=kv_dim)),
                PreNorm(dim, feed_forward_layer(dim, mlp_dim, dropout=dropout)),
            ]))

    def forward(self, x, z=None, **kwargs):
        for attn, ff in self.layers:
            x = attn(x, z=z) + x
            x = ff(x, **kwargs) + x
        return x
