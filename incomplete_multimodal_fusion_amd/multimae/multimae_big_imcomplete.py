"""Downstream backbone on the packed MI355X path.

Reference: downstream/instance_segmentation/modeling/multimae/multimae_big_imcomplete.py -- class ViTBaseline :418-754
(forward_features :534-664, forward :666-680, necks :432-440), factory ViTMAE :756-797.  Same encoder weights and
state-dict keys as the pretraining model (its checkpoints are loaded with strict=False, :456-460), so the encoder is the
same packed pipeline (multimae_crossattn.MultiMAE._encode): per-forward modality subset, 90 % token keep while training,
feature taps after the layers in `flags`.  What differs from pretraining and is reproduced here: a modality that is
absent from the forward gets NO slot in the modality attention (:635-648 builds M'+1 slots), whereas a masked patch of a
present modality still reads mask_embedding.  The four necks (ConvTranspose2d / GroupNorm / GELU / MaxPool2d) are
ordinary torch modules -- library ops outside the hot path.
"""
import os
import random
from typing import Dict, List, Optional, Union

import torch
from torch import nn

from .multimae_crossattn import MultiMAE
from .zorro_utils import LayerNorm, TokenTypes


class ViTBaseline(MultiMAE):
    def __init__(self, pretrained=None, pretrain_size=224, frozen_stages=12, freeze_attn=False, freeze_ffn=False,
                 *args, in_domains: list = None, **kwargs):
        kwargs.setdefault("output_adapters", None)
        super().__init__(*args, **kwargs)
        self.in_domains = list(in_domains) if in_domains is not None else list(self.domains)
        for d in self.domains:                      # the downstream class has no contrastive return tokens (:25-117)
            delattr(self, 'return_token_' + d)
        self.frozen_stages, self.freeze_attn, self.freeze_ffn = frozen_stages, freeze_attn, freeze_ffn
        self.cls_token = None
        self.num_block = len(self.blocks)
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.flags = [i for i in range(-1, self.num_block, self.num_block // 4)][1:]                 # :427
        D = self.dim_tokens
        self.up1 = nn.Sequential(nn.ConvTranspose2d(D, D, (2, 2), (2, 2)), nn.GroupNorm(32, D), nn.GELU(),
                                 nn.ConvTranspose2d(D, D, (2, 2), (2, 2)))
        self.up2 = nn.ConvTranspose2d(D, D, (2, 2), (2, 2))
        self.up3 = nn.Identity()
        self.up4 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.incomplete_domains = list(self.in_domains)
        if pretrained and os.path.exists(pretrained):
            self.init_weights(pretrained)

    def never_used_parameters(self):
        return [self.return_tokens]

    def init_weights(self, pretrained: str = None):
        ckpt = torch.load(pretrained, map_location="cpu", weights_only=False)
        return self.load_state_dict(ckpt['model'], strict=False)

    def forward_features(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], mask_inputs: bool = False,
                         task_masks: Dict[str, torch.Tensor] = None, num_encoded_tokens: int = None,
                         alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False):
        if self.training:                                                                            # :541-548
            self.incomplete_domains = random.sample(self.in_domains, random.randint(1, 3))
        else:
            self.incomplete_domains = list(self.in_domains)
        one_mod = self.incomplete_domains[0]
        x = {one_mod: x} if isinstance(x, torch.Tensor) else x
        present = [d for d in x if d in self.input_adapters and d in self.incomplete_domains]       # dict order (:560-564)
        B, _, H, W = x[present[0]].shape
        device = x[present[0]].device
        ps = self.input_adapters[present[0]].P_H
        N_H, N_W = H // ps, W // ps
        P = N_H * N_W
        Mp = len(present)
        if mask_inputs:
            N = num_encoded_tokens
        else:
            N = int(Mp * P * 0.9) if self.training else Mp * P                                       # :577-582
        if task_masks is None:
            placeholders = {d: torch.empty(B, P, 0, device=device) for d in present}
            task_masks, _, _ = self.generate_random_masks(placeholders, N, alphas=alphas,
                                                          sample_tasks_uniformly=sample_tasks_uniformly)
            mask_all = torch.cat([task_masks[d][:(B if self.per_sample_masks else 1)] for d in present], dim=1)
            explicit = False
        else:
            mask_full = torch.cat([task_masks[d] for d in present], dim=1).to(torch.int64)
            mask_all = mask_full if self.per_sample_masks else mask_full[:1]
            explicit = True
        _, _, _, _, outs = self._encode(x, present, mask_all.contiguous(), N, explicit, taps=tuple(self.flags))
        return outs, N_H, N_W

    def forward(self, input_dict):
        outs, H, W = self.forward_features(input_dict)
        f1, f2, f3, f4 = [self.norm(f).transpose(1, 2).reshape(f.shape[0], f.shape[2], H, W) for f in outs]
        return [self.up1(f1).contiguous(), self.up2(f2).contiguous(), self.up3(f3).contiguous(), self.up4(f4).contiguous()]


def ViTMAE(args, *argss, **kwargs):
    """Factory with the reference's fixed tiny preset (:756-797); args carries MultiMAE.{patch_size,input_size,in_domains,
    extra_fusion_token,drop_path} and MODEL.BACKBONE.PRETRAINED_WEIGHTS like the reference's yacs config."""
    from .input_adapters import FusionInputAdapter, PatchedInputAdapter
    ch = {'s1': 1, 's2': 3, 'dem': 1}
    mm = args.MultiMAE
    input_adapters = {d: PatchedInputAdapter(num_channels=ch[d], stride_level=1, patch_size_full=mm.patch_size,
                                             image_size=mm.input_size) for d in mm.in_domains}
    if mm.extra_fusion_token:
        input_adapters['fusion'] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=mm.patch_size,
                                                      image_size=mm.input_size)
    return ViTBaseline(input_adapters=input_adapters, output_adapters=None, num_fusion_tokens=256,
                       return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                       drop_path_rate=mm.drop_path, dim_tokens=192, depth=12, dim_head=64, heads=3, ff_mult=4,
                       norm_layer=LayerNorm, in_domains=mm.in_domains, frozen_stages=11,
                       pretrained=args.MODEL.BACKBONE.PRETRAINED_WEIGHTS, *argss, **kwargs)
