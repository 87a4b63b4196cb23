"""The 4-modality sibling of the fusion-token model, behind the reference's module names
(reference: pretraining/multimae/multimae_quadruplet.py -- class MultiMAE :39-490, factories :493-560; driver
pretraining/pretrain_mmae_my.py:35-36, :248-255).

Same packed pipeline as multimae_crossattn.MultiMAE (same kernels, same row space) with what that file does not have switched
off: no per-layer Block_Fusion (the encoder is the Zorro-masked Block stack over [s1 | s2 | dem | dnw | fusion], :430-432), no
mask_embedding, no per-modality contrastive return tokens; forward returns the 5-tuple
(preds, task_masks, return_tokens, ori_tokens, encoder_fusion_tokens) (:490).  State-dict keys are the reference's
(tests/test_gpu_quad.py strict-loads a reference-generated state dict).
"""
from typing import Dict, Optional, Tuple

from torch import nn

from . import multimae_crossattn as _mc
from .zorro_utils import LayerNorm
from .zorro_utils_quadruplet import TokenTypes

__all__ = ['pretrain_multimae_tiny', 'pretrain_multimae_base', 'pretrain_multimae_large', 'MultiMAE']


class MultiMAE(_mc.MultiMAE):
    def __init__(self, input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]],
                 num_global_tokens: int = 1, dim_tokens: int = 768, depth: int = 12, dim_head: int = 64, heads: int = 8,
                 ff_mult: int = 4, num_fusion_tokens: int = 16,
                 return_token_types: Tuple[TokenTypes] = (TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.DNW,
                                                          TokenTypes.FUSION),
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = LayerNorm):
        super().__init__(input_adapters, output_adapters, num_global_tokens=num_global_tokens, dim_tokens=dim_tokens,
                         depth=depth, dim_head=dim_head, heads=heads, ff_mult=ff_mult, num_fusion_tokens=num_fusion_tokens,
                         return_token_types=return_token_types, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
                         fusion_blocks=False, contrastive_tokens=False)


def _factory(dim_tokens, depth, heads):
    def build(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
        return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=dim_tokens, depth=depth,
                        dim_head=64, heads=heads, ff_mult=4, norm_layer=LayerNorm, **kwargs)
    return build


pretrain_multimae_tiny = _factory(384, 12, 8)      # reference :493-514 (the driver's choice, pretrain_mmae_my.py:248)
pretrain_multimae_base = _factory(768, 12, 8)      # :517-538
pretrain_multimae_large = _factory(1024, 24, 8)    # :541-560
