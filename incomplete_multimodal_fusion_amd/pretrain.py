"""Host-side restatement of the reference's pretraining step on top of the native MultiMAE path.

Reference: pretraining/pretrain_mmae.py -- DOMAIN_CONF :45-72, get_model :188-248, and the step of train_one_epoch
:447-517 (H2D of the tile stack, autocast forward, per-task masked losses :479-486, 3x DINO-style contrastive loss
:489-493, weighted sum :499-500, backward, optimizer step).  Logging / checkpointing / dataset code is out of scope
(SURVEY.md section 2).  Device tensors only.
"""
from functools import partial
from typing import Dict, Iterable, Optional

import torch

from . import ops
from .multimae import (FusionInputAdapter, MaskedL1Loss, MaskedMSELoss, PatchedInputAdapter, SpatialOutputAdapter,
                       TokenTypes, dino_loss_func)
from .multimae import multimae_crossattn as mc

DOMAIN_CONF = {                                   # pretrain_mmae.py:45-72
    's1': {'channels': 1, 'stride_level': 1, 'loss': MaskedMSELoss},
    's2': {'channels': 3, 'stride_level': 1, 'loss': MaskedMSELoss},
    'dem': {'channels': 1, 'stride_level': 1, 'loss': MaskedL1Loss},
    'fusion': {'channels': 1, 'stride_level': 1},
}

PRESETS = {'tiny': (192, 12, 3), 'small': (384, 12, 8), 'base': (768, 12, 8), 'large': (1024, 24, 8)}


def get_model(model: str = 'base', in_domains=('s1', 's2', 'dem'), out_domains=('s1', 's2', 'dem'), patch_size: int = 16,
              input_size: int = 256, decoder_dim: int = 256, decoder_depth: int = 2, decoder_num_heads: int = 8,
              dim_head: int = 64):
    """Adapters + model as get_model builds them (pretrain_mmae.py:193-246); `model` picks the size preset (the
    reference ignores --model and always builds the tiny factory, :239 -- SURVEY.md 0.5)."""
    input_adapters = {
        d: PatchedInputAdapter(num_channels=DOMAIN_CONF[d]['channels'], stride_level=DOMAIN_CONF[d]['stride_level'],
                               patch_size_full=patch_size, image_size=input_size) for d in in_domains}
    output_adapters = {
        d: SpatialOutputAdapter(num_channels=DOMAIN_CONF[d]['channels'], stride_level=DOMAIN_CONF[d]['stride_level'],
                                patch_size_full=patch_size, dim_tokens=decoder_dim, depth=decoder_depth,
                                num_heads=decoder_num_heads, use_task_queries=True, task=d,
                                context_tasks=list(in_domains), use_xattn=True) for d in out_domains}
    input_adapters['fusion'] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=patch_size,
                                                  image_size=input_size)
    D, depth, heads = PRESETS[model]
    P = (input_size // patch_size) ** 2
    return mc.MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, num_global_tokens=1,
                       dim_tokens=D, depth=depth, dim_head=dim_head, heads=heads, ff_mult=4, num_fusion_tokens=P,
                       return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                       drop_path_rate=0.0)


def make_loss_fns(out_domains=('s1', 's2', 'dem'), patch_size: int = 16):
    return {d: DOMAIN_CONF[d]['loss'](patch_size=patch_size, stride=DOMAIN_CONF[d]['stride_level']) for d in out_domains}


def step_losses(out, tasks_dict: Dict[str, torch.Tensor], masks: Dict[str, torch.Tensor], patch_size: int = 16,
                loss_fns=None, contra_weight: float = 0.3):
    """pretrain_mmae.py:479-500 with NoWeightingStrategy (utils/task_balancing.py:11-19)."""
    preds, _, pooled, _, _, *rets = out
    loss_fns = loss_fns or make_loss_fns(tuple(preds.keys()), patch_size)
    task_losses = {}
    for task in preds:
        pred = preds[task].float()
        fn = loss_fns[task]
        if isinstance(pred, mc.PredTokens):
            task_losses[task] = fn.forward_tokens(pred.tokens, tasks_dict[task], mask=masks.get(task, None))
        else:
            task_losses[task] = fn(pred, tasks_dict[task], mask=masks.get(task, None))
    feats = torch.chunk(pooled, pooled.shape[1], dim=1)                        # :489-490 (squeeze the token axis)
    loss_contra = sum(dino_loss_func(r.squeeze(1), f.squeeze(1)) for r, f in zip(rets, feats))   # :493
    loss = sum(task_losses.values()) + contra_weight * loss_contra             # :499-500
    return task_losses, loss_contra, loss


class PretrainStep:
    """One optimizer step on a batch of tiles already resident on the device.  No host synchronisation."""

    def __init__(self, model, optimizer, num_encoded_tokens: int, in_domains=('s1', 's2', 'dem'), alphas: float = 1.0,
                 sample_tasks_uniformly: bool = False, autocast: bool = True, patch_size: int = 16,
                 grad_reducer=None, side_stream_wgrad: bool = False):
        self.model, self.opt = model, optimizer
        self.N, self.in_domains, self.alphas, self.uniform = num_encoded_tokens, tuple(in_domains), alphas, sample_tasks_uniformly
        self.autocast, self.patch = autocast, patch_size
        self.loss_fns = make_loss_fns(tuple(model.output_adapters.keys()), patch_size)
        self.reducer = grad_reducer
        model.fuse_unpatchify_loss = True
        model.side_stream_wgrad = side_stream_wgrad

    def __call__(self, tasks_dict: Dict[str, torch.Tensor], task_masks: Optional[Dict[str, torch.Tensor]] = None):
        x = {t: v for t, v in tasks_dict.items() if t in self.in_domains}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
            out = self.model(x, task_masks=task_masks, num_encoded_tokens=self.N, alphas=self.alphas,
                             sample_tasks_uniformly=self.uniform)
            task_losses, loss_contra, loss = step_losses(out, tasks_dict, out[1], self.patch, self.loss_fns)
        self.opt.zero_grad(set_to_none=True)               # (FlatAdamW: also clears the flat gradient buffer)
        if self.reducer is not None:
            self.reducer.prepare()
        loss.backward()
        ops.join_wgrad_stream()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return {'loss': loss.detach(), 'loss_contra': loss_contra.detach(),
                **{k + '_loss': v.detach() for k, v in task_losses.items()}}
