"""Per-modality reconstruction decoder with the reference's constructor / state-dict surface
(reference: pretraining/multimae/output_adapters_simple.py:33-188).

forward(encoder_tokens, input_info, ids_keep, ids_restore) -> (B, C, H, W) like the reference (ids_* accepted and
unused there, too).  `forward_tokens` is the packed hot-path entry: it returns the out_proj tokens so that MultiMAE's
training step can feed them to the fused unpatchify+masked-loss kernel without materialising the image.
"""
from functools import partial
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from .multimae_utils import Block, build_2d_sincos_posemb, pair, trunc_normal_
from .zorro_utils import compute_dtype, linear, wcast


class SpatialOutputAdapter(nn.Module):
    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens_enc: Optional[int] = None, dim_tokens: int = 256, depth: int = 0,
                 learnable_pos_emb: int = False, image_size: Union[int, Tuple[int]] = 224, mlp_ratio: int = 4.0,
                 num_heads: int = 8, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6),
                 use_task_queries: bool = True, task: Optional[str] = None, context_tasks: Optional[list] = None,
                 use_xattn: bool = True):
        super().__init__()
        assert drop_rate == 0.0 and attn_drop_rate == 0.0
        self.num_channels, self.stride_level = num_channels, stride_level
        self.patch_size_full, self.image_size = pair(patch_size_full), pair(image_size)
        self.dim_tokens_enc, self.dim_tokens = dim_tokens_enc, dim_tokens
        self.learnable_pos_emb, self.use_task_queries, self.task, self.use_xattn = \
            learnable_pos_emb, use_task_queries, task, use_xattn
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        self.task_embeddings = None
        if context_tasks is not None:
            self.task_embeddings = nn.ParameterDict(
                {t: nn.Parameter(trunc_normal_(torch.zeros(1, 1, dim_tokens), std=0.02)) for t in context_tasks})
        # pos_emb is part of the checkpoint but never read by forward (output_adapters_simple.py:104-111)
        gh = self.image_size[0] // (stride_level * self.P_H)
        gw = self.image_size[1] // (stride_level * self.P_W)
        if not learnable_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=gh, w=gw, embed_dim=dim_tokens), requires_grad=False)
        else:
            self.pos_emb = nn.Parameter(trunc_normal_(torch.zeros(1, gh, gw, dim_tokens), std=0.02))
        if depth > 0:
            dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # stochastic depth decay rule (reference :115)
            self.decoder_transformer = nn.Sequential(*[
                Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop_path=dpr[i],
                      norm_layer=norm_layer)
                for i in range(depth)])
        else:
            self.decoder_transformer = nn.Identity()
        self.dim_patch = num_channels * self.P_H * self.P_W
        self.out_proj = nn.Linear(dim_tokens, self.dim_patch)
        if dim_tokens_enc is not None:
            self.init(dim_tokens_enc=dim_tokens_enc)

    def init(self, dim_tokens_enc: int = 768):
        self.dim_tokens_enc = dim_tokens_enc
        self.proj_context = nn.Linear(dim_tokens_enc, self.dim_tokens)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb', 'task_embeddings'}

    def forward_tokens(self, enc_rows: torch.Tensor, B: int, P: int, seg=None, once: bool = False) -> torch.Tensor:
        """enc_rows (B*P, D_enc) in the compute dtype -> out_proj tokens (B*P, C*P_H*P_W) in (c ph pw) order.
        once: this adapter runs exactly once in the current optimizer step (see ops.linear)."""
        assert self.dim_tokens_enc is not None, 'Need to call init(dim_tokens_enc) function first'
        ctx = linear(enc_rows, self.proj_context.weight, self.proj_context.bias, once=once)
        x = ctx.float()
        if self.task_embeddings is not None and self.task in self.task_embeddings:
            x = x + self.task_embeddings[self.task].reshape(1, -1)
        return self.decode_rows(x, B, P, seg, once, enc_rows.dtype)

    def own_task_embedding(self):
        """The task embedding forward adds to the projected context (None when the adapter has none for its task)."""
        if self.task_embeddings is not None and self.task in self.task_embeddings:
            return self.task_embeddings[self.task]
        return None

    def decode_rows(self, x: torch.Tensor, B: int, P: int, seg=None, once: bool = False, T=None) -> torch.Tensor:
        """x (B*P, dim_tokens) fp32 = proj_context(encoder rows) + task embedding -> out_proj tokens.  MultiMAE's packed forward
        computes every decoder's context projection in ONE GEMM (ops.kv_ctx_projections) and enters here."""
        T = compute_dtype(self.out_proj.weight) if T is None else T
        delta = None
        if isinstance(self.decoder_transformer, nn.Sequential):
            for blk in self.decoder_transformer:
                x, delta = blk.forward_rows(x, delta, B, P, seg, once)
        y = x if delta is None else x + delta.float()
        return linear(wcast(y, T), self.out_proj.weight, self.out_proj.bias, once=once)

    def forward(self, encoder_tokens: torch.Tensor, input_info: Dict, ids_keep: torch.Tensor = None,
                ids_restore: torch.Tensor = None):
        H, W = input_info['image_size']
        B, P, D = encoder_tokens.shape
        assert self.P_H == self.P_W
        T = compute_dtype(self.out_proj.weight)
        tok = self.forward_tokens(wcast(encoder_tokens.reshape(B * P, D), T).contiguous(), B, P)
        return ops.unpatchify(tok, B, self.num_channels, H, W, self.P_H)
