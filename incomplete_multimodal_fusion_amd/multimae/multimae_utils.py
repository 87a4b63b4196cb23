"""Decoder-side building blocks and helpers with the reference's names and state-dict keys
(reference: pretraining/multimae/multimae_utils.py), computed by the gfx950 kernels in ../csrc.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .zorro_utils import compute_dtype, linear, wcast


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.):
    """Fixed 2-D sin-cos table, (1, D, h, w).  Channel quarters [sin_w, cos_w, sin_h, cos_h]; the reference builds the
    grids with meshgrid(grid_w, grid_h) in 'ij' order and then reads the flattened (w h) order as (h w)
    (multimae_utils.py:34-44) -- reproduced exactly, it only matters for non-square grids."""
    assert embed_dim % 4 == 0, 'Embed dimension must be divisible by 4 for 2D sin-cos position embedding'
    pos_dim = embed_dim // 4
    omega = 1. / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    gw = torch.arange(w, dtype=torch.float32)[:, None].expand(w, h).reshape(-1)
    gh = torch.arange(h, dtype=torch.float32)[None, :].expand(w, h).reshape(-1)
    out_w = gw[:, None] * omega[None, :]
    out_h = gh[:, None] * omega[None, :]
    pe = torch.cat([out_w.sin(), out_w.cos(), out_h.sin(), out_h.cos()], dim=1)
    return pe.reshape(1, h, w, embed_dim).permute(0, 3, 1, 2).contiguous()


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal init by inverse-CDF sampling (same scheme as multimae_utils.py:48-102)."""
    def cdf(x):
        return (1. + math.erf(x / math.sqrt(2.))) / 2.
    with torch.no_grad():
        lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.)).add_(mean).clamp_(min=a, max=b)
    return tensor


class Mlp(nn.Module):            # multimae_utils.py:138-155
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        assert drop == 0.0

    def forward(self, x):
        T = compute_dtype(self.fc1.weight)
        h = linear(wcast(x, T), self.fc1.weight, self.fc1.bias)
        return linear(ops.gelu(h), self.fc2.weight, self.fc2.bias)


class Attention(nn.Module):      # multimae_utils.py:158-182 (fused qkv with bias, scale applied to the scores)
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert attn_drop == 0.0 and proj_drop == 0.0
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward_rows(self, y, B, N, seg=None, once=False):
        """y (B*N, C) already normalised, compute dtype -> (B*N, C)."""
        C = y.shape[1]
        H = self.num_heads
        qkv = linear(y, self.qkv.weight, self.qkv.bias, once=once)            # columns [q | k | v], heads inside each (:172)
        if seg is None:
            seg = ops.Segments.dense(B, N, y.device)
        a = ops.mha_self(qkv, H, C // H, seg, self.scale)
        return linear(a, self.proj.weight, self.proj.bias, once=once)

    def forward(self, x):
        B, N, C = x.shape
        T = compute_dtype(self.qkv.weight)
        return self.forward_rows(wcast(x.reshape(B * N, C), T).contiguous(), B, N).reshape(B, N, C)


class Block(nn.Module):          # multimae_utils.py:217-232
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop == 0.0
        from .zorro_utils import DropPath                                  # one implementation (multimae_utils.py:105-135 = zorro_utils.py:69-96)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()         # :224
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)

    def forward_rows(self, x, delta, B, N, seg=None, once=False):
        """Residual stream x (B*N, C) fp32 plus a pending delta (compute dtype or None).  Returns (x, delta)."""
        T = compute_dtype(self.attn.qkv.weight)
        (x,), y = ops.parts_add_ln([x], delta, [0 if delta is not None else -1], self.norm1.weight, self.norm1.bias,
                                   eps1=self.norm1.eps, out_dtype=T)
        dp = self.drop_path.rows if not isinstance(self.drop_path, nn.Identity) else (lambda t, *_: t)
        a = dp(self.attn.forward_rows(y, B, N, seg, once), B, (N,))                                     # :230
        (x,), y = ops.parts_add_ln([x], a, [0], self.norm2.weight, self.norm2.bias, eps1=self.norm2.eps, out_dtype=T)
        h = linear(y, self.mlp.fc1.weight, self.mlp.fc1.bias, once=once)
        f = dp(linear(ops.gelu(h), self.mlp.fc2.weight, self.mlp.fc2.bias, once=once), B, (N,))        # :231
        return x, f

    def forward(self, x):
        B, N, C = x.shape
        x2, f = self.forward_rows(x.reshape(B * N, C).float().contiguous(), None, B, N)
        return (x2 + f.float()).reshape(B, N, C)
