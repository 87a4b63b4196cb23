"""Input adapters with the reference's constructor / attribute / state-dict surface
(reference: pretraining/multimae/input_adapters.py:27-206).

In the pretraining hot path MultiMAE does not call PatchedInputAdapter.forward: it embeds only the KEPT patches of
all modalities with one patchify-gather kernel + one GEMM (see multimae_crossattn.py).  The standalone forward below
(all patches, as the reference does) runs the same kernel in dense mode.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .multimae_utils import build_2d_sincos_posemb, pair, trunc_normal_
from .zorro_utils import compute_dtype, linear


def interp_posemb(pos_emb: torch.Tensor, nh: int, nw: int) -> torch.Tensor:
    """(1, D, h, w) -> (nh*nw, D).  Bicubic resize like the reference (input_adapters.py:113); at the native grid the
    resize is a bit-identity and is skipped."""
    if pos_emb.shape[-2:] != (nh, nw):
        pos_emb = F.interpolate(pos_emb, size=(nh, nw), mode='bicubic', align_corners=False)
    return pos_emb.flatten(2).transpose(1, 2)[0]


class _GridAdapter(nn.Module):
    """Shared geometry + positional-embedding surface of the two adapters (input_adapters.py:41-91, 135-179)."""

    has_projection = False

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224):
        super().__init__()
        ih, iw = pair(image_size)
        ph, pw = pair(patch_size_full)
        self.num_channels, self.stride_level = num_channels, stride_level
        self.patch_size_full, self.image_size = (ph, pw), (ih, iw)
        self.sincos_pos_emb, self.learnable_pos_emb = sincos_pos_emb, learnable_pos_emb
        self.num_patches = (ih // patch_size_full) * (iw // patch_size_full)
        self.P_H, self.P_W = max(1, ph // stride_level), max(1, pw // stride_level)
        self.dim_tokens = dim_tokens
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        """Called by MultiMAE.__init__ once the encoder width is known (multimae_crossattn.py:76-77)."""
        self.dim_tokens = dim_tokens
        gh = self.image_size[0] // (self.stride_level * self.P_H)
        gw = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            table = build_2d_sincos_posemb(h=gh, w=gw, embed_dim=dim_tokens)
            self.pos_emb = nn.Parameter(table, requires_grad=self.learnable_pos_emb)
        else:
            self.pos_emb = nn.Parameter(trunc_normal_(torch.zeros(1, dim_tokens, gh, gw), std=0.02))
        if self.has_projection:
            # nn.Conv2d only as the parameter container: checkpoint keys/shapes proj.weight (D,C,p,p), proj.bias
            self.proj = nn.Conv2d(self.num_channels, dim_tokens, kernel_size=(self.P_H, self.P_W),
                                  stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb'}

    def _grid(self, H, W):
        assert self.dim_tokens is not None, 'Need to call init(dim_tokens) function first'
        assert (H % self.P_H == 0) and (W % self.P_W == 0), \
            f'Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}'
        return H // self.P_H, W // self.P_W


class PatchedInputAdapter(_GridAdapter):
    has_projection = True

    def forward(self, x):
        B, C, H, W = x.shape
        nh, nw = self._grid(H, W)
        assert self.P_H == self.P_W
        T = compute_dtype(self.proj.weight)
        K = C * self.P_H * self.P_W
        patches = ops.patchify_gather([x], [0], -1, K, self.P_H, None, None, nh * nw, T)
        tok = linear(patches, self.proj.weight.reshape(self.dim_tokens, K), self.proj.bias)
        return tok.reshape(B, nh * nw, self.dim_tokens).float() + interp_posemb(self.pos_emb, nh, nw)[None]


class FusionInputAdapter(_GridAdapter):
    def posemb_rows(self):
        W, H = self.image_size[0], self.image_size[1]      # (sic) reference order, input_adapters.py:193
        nh, nw = self._grid(H, W)
        return interp_posemb(self.pos_emb, nh, nw)

    def forward(self, x):
        assert x.shape[1] == self.num_patches
        return x + self.posemb_rows()[None]
