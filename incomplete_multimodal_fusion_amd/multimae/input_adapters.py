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

    # -- what MultiMAE's packed patch embedding needs from a modality adapter ----------------------------------------------
    @property
    def packed_channels(self) -> int:
        return self.num_channels

    def packed_image(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, H, W) fp32 image whose kept patches are gathered into the shared embedding GEMM."""
        return x

    def packed_bias(self) -> torch.Tensor:
        return self.proj.bias

    def packed_weight(self) -> torch.Tensor:
        """(D, C*ph*pw) conv weight in (c ph pw) column order."""
        return self.proj.weight.reshape(self.dim_tokens, -1)

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


class SemSegInputAdapter(_GridAdapter):
    """Class-map modality (reference: input_adapters.py:209-328; the `dnw` land-cover input of the 4-modality driver,
    pretrain_mmae_my.py:67-74): x (B, H, W) int64 -> class_emb lookup (num_classes, dim_class_emb) -> Conv2d(k = s = patch).
    State-dict keys kept: pos_emb, class_emb.weight, proj.{weight,bias}.

    Computed here as ONE GEMM over the one-hot class image: proj(class_emb[x]) = W_c @ onehot(x) with the composed weight
    W_c[d, (k, i, j)] = sum_c proj.weight[d, c, i, j] * class_emb[k, c] -- exact algebra (each pixel selects one embedding
    row), and it lets the modality share MultiMAE's single kept-patch embedding GEMM.  Gradients reach class_emb and proj
    through the composition (autograd on the small einsum)."""
    has_projection = False

    def __init__(self, num_classes: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224, dim_class_emb: int = 64, interpolate_class_emb: bool = False,
                 emb_padding_idx: int = None):
        self.num_classes = num_classes + (1 if emb_padding_idx is not None else 0)
        self.dim_class_emb, self.interpolate_class_emb, self.emb_padding_idx = dim_class_emb, bool(interpolate_class_emb), emb_padding_idx
        super().__init__(self.num_classes, stride_level, patch_size_full, dim_tokens, sincos_pos_emb, learnable_pos_emb,
                         image_size)

    def init(self, dim_tokens: int = 768):
        super().init(dim_tokens)
        self.class_emb = nn.Embedding(self.num_classes, self.dim_class_emb, padding_idx=self.emb_padding_idx)
        trunc_normal_(self.class_emb.weight, std=0.02)
        if self.interpolate_class_emb:
            # reference :288-294: bilinear down-sampling of the embedding map by the patch size, then a 1x1 convolution.
            # Modules kept as the parameter container (state-dict keys proj.1.weight / proj.1.bias); the arithmetic is the
            # composed patch weight below: with align_corners=False the sample point of patch row i is 16 i + 7.5, i.e. the
            # mean of the two centre pixels per axis (one centre pixel for an odd patch size).
            self.proj = nn.Sequential(nn.Upsample(scale_factor=(1 / self.P_H, 1 / self.P_W), mode='bilinear'),
                                      nn.Conv2d(self.dim_class_emb, dim_tokens, kernel_size=1, stride=1))
        else:
            self.proj = nn.Conv2d(self.dim_class_emb, dim_tokens, kernel_size=(self.P_H, self.P_W), stride=(self.P_H, self.P_W))

    @property
    def _conv(self) -> nn.Conv2d:
        return self.proj[1] if self.interpolate_class_emb else self.proj

    def packed_bias(self) -> torch.Tensor:
        return self._conv.bias

    @staticmethod
    def _centre_taps(p: int) -> torch.Tensor:
        k = torch.zeros(p)
        if p % 2:
            k[(p - 1) // 2] = 1.0
        else:
            k[p // 2 - 1] = k[p // 2] = 0.5
        return k

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb', 'class_emb'}

    @property
    def packed_channels(self) -> int:
        return self.num_classes

    def packed_image(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 3 and not x.dtype.is_floating_point, "class map must be (B, H, W) integer"
        return F.one_hot(x.long(), self.num_classes).permute(0, 3, 1, 2).to(torch.float32).contiguous()

    def packed_weight(self) -> torch.Tensor:
        emb = self.class_emb.weight.float()
        if self.emb_padding_idx is not None:              # nn.Embedding(padding_idx): that row receives no gradient
            keep = torch.ones(self.num_classes, 1, device=emb.device)
            keep[self.emb_padding_idx] = 0.0
            emb = emb * keep + (emb * (1.0 - keep)).detach()
        cw = self._conv.weight.float()
        if self.interpolate_class_emb:
            taps = torch.outer(self._centre_taps(self.P_H), self._centre_taps(self.P_W)).to(cw.device)
            cw = cw * taps                                 # (D, C, 1, 1) x (P_H, P_W) -> (D, C, P_H, P_W)
        w = torch.einsum('dcij,kc->dkij', cw, emb)
        return w.reshape(self.dim_tokens, -1)

    def forward(self, x):
        B, H, W = x.shape
        nh, nw = self._grid(H, W)
        assert self.P_H == self.P_W
        T = compute_dtype(self._conv.weight)
        K = self.num_classes * self.P_H * self.P_W
        patches = ops.patchify_gather([self.packed_image(x)], [0], -1, K, self.P_H, None, None, nh * nw, T)
        tok = linear(patches, self.packed_weight(), self._conv.bias)
        pe = self.pos_emb
        if pe.shape[-2:] != (nh, nw):
            pe = F.interpolate(pe, size=(nh, nw), mode='bilinear')                 # reference :322 (bilinear here)
        return tok.reshape(B, nh * nw, self.dim_tokens).float() + pe.flatten(2).transpose(1, 2)
