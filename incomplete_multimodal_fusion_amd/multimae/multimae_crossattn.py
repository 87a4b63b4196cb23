"""MultiMAE with learned fusion tokens -- MI355X-native hot path behind the reference's module API.

Reference: pretraining/multimae/multimae_crossattn.py (class MultiMAE :37-545, factories :548-599).  Same constructor,
same forward signature and 8-tuple result, same state-dict keys (435 at ViT-B / 3 modalities); the computation is a
packed, static-shape pipeline over hand-written gfx950 kernels:

  row space of one step:   [ B*N kept modality tokens | B*P fusion tokens | P mask-embedding rows ]
  residual stream in fp32, projections in the compute dtype (bf16 under autocast, else fp32) on hipBLASLt,
  everything between the GEMMs -- mask bookkeeping, patchify-gather, (residual add + double LayerNorm), Zorro-masked
  attention, modality attention, GEGLU/GELU, unpatchify, masked losses, DINO head -- in ../csrc kernels.

Exact algebraic shortcuts relative to the reference (rows are independent, results identical):
  * only kept patches are embedded (reference embeds all, then gathers, :369-373/:402-407);
  * Block_Fusion evaluates the query / output projection of the fusion slot only (reference computes all M+1 slots and
    discards M of them, zorro_utils.py:255-256) and K/V of masked slots once per patch (they come from mask_embedding);
  * the (B,P,M+1,D) `all_tokens` tensor (:454-462) and the (B,h,S,S) score tensor are never materialised.
"""
import itertools
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn
from torch.distributions.dirichlet import Dirichlet

from .. import ops
from .input_adapters import interp_posemb
from .multimae_utils import trunc_normal_
from .zorro_utils import (Attention, Block, Block_Fusion, LayerNorm, Mlp, TokenTypes, compute_dtype, exists, linear,
                          wcast)

__all__ = ['pretrain_multimae_tiny', 'pretrain_multimae_base', 'pretrain_multimae_large', 'MultiMAE']


# Implementation switches (module attributes, no environment reads here: tools/tuning_env.py maps MMAE_* variables onto them for A/B runs)
FUSED_FINAL_CAST = True     # bf16 copy of the final norm's output from the same pass
FUSED_DECODER_CTX = True    # one GEMM for every decoder's proj_context
DUAL_LAYERNORM = True       # one pass for the modality rows' two LayerNorm pairs
DECODER_STREAMS = False     # the per-modality decoders on separate HIP streams (measured: no gain)
ASYNC_DRAW_COPY = True          # the Dirichlet draw reaches the device through pinned memory, asynchronously (generate_random_masks)

class PredTokens:
    """Decoder output kept token-major, (B*P, C*p*p) in (c ph pw) order, so that the masked loss can be fused with the
    unpatchify.  Quacks like the prediction tensor for the reference driver (`preds[task].float()`,
    pretrain_mmae.py:484-486); `.image()` materialises (B,C,H,W)."""

    def __init__(self, tokens, B, C, H, W, patch):
        self.tokens, self.meta = tokens, (B, C, H, W, patch)
        self.shape = torch.Size((B, C, H, W))

    def float(self):
        return self

    def image(self):
        B, C, H, W, patch = self.meta
        return ops.unpatchify(self.tokens, B, C, H, W, patch)


class MultiMAE(nn.Module):
    def __init__(self,
                 input_adapters: Dict[str, nn.Module],
                 output_adapters: Optional[Dict[str, nn.Module]],
                 num_global_tokens: int = 1,
                 dim_tokens: int = 768,
                 depth: int = 12,
                 dim_head: int = 64,
                 heads: int = 8,
                 ff_mult: int = 4,
                 num_fusion_tokens: int = 16,
                 return_token_types: Tuple[TokenTypes] = (TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                 drop_path_rate: float = 0.0,
                 norm_layer: nn.Module = LayerNorm,
                 fusion_blocks: bool = True,
                 contrastive_tokens: bool = True):
        """fusion_blocks / contrastive_tokens (keyword extras, default = this file's reference class): False builds the
        sibling pretraining/multimae/multimae_quadruplet.py model -- same encoder without the per-layer Block_Fusion, without
        mask_embedding and without the per-modality contrastive return tokens (see multimae_quadruplet.py here)."""
        super().__init__()
        for adapter in input_adapters.values():
            adapter.init(dim_tokens=dim_tokens)
        self.input_adapters = nn.ModuleDict(input_adapters)
        if output_adapters is not None:
            for adapter in output_adapters.values():
                adapter.init(dim_tokens_enc=dim_tokens)
            self.output_adapters = nn.ModuleDict(output_adapters)
        else:
            self.output_adapters = None
        self.domains = [d for d in input_adapters if d != 'fusion']
        assert num_fusion_tokens == input_adapters['s1'].num_patches          # reference :87
        # Any number M <= 7 of modalities (the reference hard-codes s1/s2/dem, :402-407; its 4-modality sibling
        # multimae_quadruplet.py adds dnw): one return token per modality in adapter order, then the fusion type last --
        # the pool / Zorro rules are built from that order on the device (csrc/masks.hip).
        M = len(self.domains)
        assert 1 <= M <= 7, "1..7 modalities"
        assert len(return_token_types) == M + 1, "one return token per modality plus fusion"
        assert [t.value for t in return_token_types] == list(range(M + 1)), \
            "return_token_types must be (modality 0, ..., modality M-1, FUSION) with values 0..M"

        self.dim_tokens, self.depth, self.heads, self.dim_head = dim_tokens, depth, heads, dim_head
        self.max_return_tokens = len(return_token_types)
        self.return_token_types = return_token_types
        self.register_buffer('return_token_types_tensor',
                             torch.tensor([t.value for t in return_token_types]), persistent=False)

        self.return_tokens = nn.Parameter(trunc_normal_(torch.zeros(1, self.max_return_tokens, dim_tokens), std=0.02))
        self.attn_pool = Attention(dim=dim_tokens, dim_head=dim_head, heads=heads)
        self.fusion_tokens = nn.Parameter(trunc_normal_(torch.zeros(1, num_fusion_tokens, dim_tokens), std=0.02))
        self.has_fusion_blocks, self.has_contrastive_tokens = bool(fusion_blocks), bool(contrastive_tokens)
        self.dual_layernorm = DUAL_LAYERNORM
        if contrastive_tokens:
            for d in self.domains:                                            # return_token_s1 / _s2 / _dem (:105-109)
                setattr(self, 'return_token_' + d, nn.Parameter(torch.randn(1, 1, dim_tokens)))
        self.mlp = Mlp(in_features=dim_tokens, hidden_features=int(dim_tokens * 4.0))
        if fusion_blocks:
            self.fus_blocks = nn.ModuleList([
                Block_Fusion(dim=dim_tokens, dim_head=dim_head, heads=heads, ff_mult=ff_mult, norm_layer=norm_layer)
                for _ in range(depth)])
            self.mask_embedding = nn.Parameter(torch.zeros(1, num_fusion_tokens, dim_tokens))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # stochastic depth decay rule (reference :132)
        self.blocks = nn.ModuleList([
            Block(dim=dim_tokens, dim_head=dim_head, heads=heads, ff_mult=ff_mult, drop_path=dpr[i], norm_layer=norm_layer)
            for i in range(depth)])
        self.norm = LayerNorm(dim_tokens)

        # behaviour switches of the native path (not part of the reference API)
        self.per_sample_masks = False      # True: every sample draws / uses its own mask row (packed superset)
        self.fuse_unpatchify_loss = False  # True: preds are PredTokens (fused unpatchify + masked loss)
        self.check_masks = True            # explicit task_masks: verify kept count == num_encoded_tokens (host sync)
        self.decoder_streams = DECODER_STREAMS
        self._dec_streams = []
        self.side_stream_wgrad = False     # True: encoder weight-gradient GEMMs overlap the backward chain on a side
                                           # stream; the trainer must ops.join_wgrad_stream() before reading .grad
        self.layer_timer = None            # an ops.LayerTimer: HIP-event brackets around that encoder layer (bench.py roofline_block)

        self._reset_parameters()

    # ---- initialisation: same distributions as the reference (:140-165) ------------------------------------------------
    def _reset_parameters(self):
        for name, m in self.named_modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
                if 'qkv' in name or 'kv' in name:
                    parts = 3 if 'qkv' in name else 2      # Q/K/V treated as separate matrices
                    val = math.sqrt(6. / float(m.weight.shape[0] // parts + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -val, val)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, nn.Conv2d) and '.proj' in name:
                nn.init.xavier_uniform_(m.weight.data.view([m.weight.shape[0], -1]))

    def never_used_parameters(self):
        """Trainable parameters the training graph never reaches (the reference needs DDP's find_unused_parameters for
        them): the pooled `return_tokens` (their loss path is detached, criterion.py:333) and every
        decoder's task embeddings of OTHER tasks (output_adapters_simple.py:172-174 reads self.task only)."""
        out = [self.return_tokens]
        for task, ad in (self.output_adapters or {}).items():
            if getattr(ad, "task_embeddings", None) is not None:
                out += [p for k, p in ad.task_embeddings.items() if k != ad.task]
        return out

    @torch.jit.ignore
    def no_weight_decay(self):
        no_wd = {'global_tokens'}
        for group, adapters in (('input_adapters', self.input_adapters), ('output_adapters', self.output_adapters or {})):
            for task, adapter in adapters.items():
                if hasattr(adapter, 'no_weight_decay'):
                    no_wd |= {f'{group}.{task}.{n}' for n in adapter.no_weight_decay()}
        return no_wd

    # ---- mask generation (reference :188-278) --------------------------------------------------------------------------
    def sample_alphas(self, B: int, n_tasks: int, alphas: float = 1.0, eps: float = 1e-5):
        choices = torch.Tensor([list(i) for i in itertools.product([0, 1], repeat=n_tasks)][1:])
        pick = torch.randint(0, len(choices), (B,))
        return torch.index_select(choices, 0, pick) * torch.tensor(alphas) + eps

    def draw_mask_distribution(self, R: int, M: int, alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False):
        """The host-side draw of generate_random_masks (reference :237-241): (R, M) token shares, a CPU tensor."""
        alphas = [alphas] * M if isinstance(alphas, float) else alphas
        if sample_tasks_uniformly:
            return Dirichlet(self.sample_alphas(R, M, alphas=alphas)).sample()
        return Dirichlet(torch.Tensor(alphas)).sample((R,))

    def generate_random_masks(self, input_tokens: Dict[str, torch.Tensor], num_encoded_tokens: int,
                              alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False):
        """Same random draws, in the same order and on the same devices, as the reference (:223-264); the integer
        bookkeeping after the draws runs in csrc/masks.hip.  `input_tokens` only supplies B, P and the device (values
        are not read) -- the native forward passes cheap placeholders."""
        vals = list(input_tokens.values())
        B, device = vals[0].shape[0], vals[0].device
        P = vals[0].shape[1]
        assert all(v.shape[1] == P for v in vals), "all modalities share one patch grid on this path"
        M = len(vals)
        R = B if self.per_sample_masks else 1
        dist = getattr(self, "mask_draws", None)
        if dist is not None:
            # a captured step (pretrain.PretrainStep.capture): the host draws and copies into this static device tensor before each
            # replay -- the draw itself cannot sit inside a hipGraph (host RNG, host->device copy)
            assert dist.is_cuda and tuple(dist.shape) == (R, M)
        else:
            dist = self.draw_mask_distribution(R, M, alphas, sample_tasks_uniformly)
        if dist.is_cuda:
            pass
        elif device.type == 'cuda' and ASYNC_DRAW_COPY:
            # through pinned memory, asynchronously: a pageable host->device copy makes the host wait until the stream has drained,
            # i.e. one host/GPU synchronisation at the top of every step (the GPU then idles while the step's first launches arrive)
            dist = dist.pin_memory().to(device, non_blocking=True)
        else:
            dist = dist.to(device)
        noise = torch.stack([torch.rand(R, P, device=device) for _ in range(M)], dim=1)       # (R, M, P)
        noise_all = torch.rand(R, M * P, device=device)
        mask_all, ids_keep, ids_restore = ops.masks_from_draws(dist, noise, noise_all, num_encoded_tokens)
        rep = (lambda t: t) if R == B else (lambda t: t.repeat(B, 1))
        task_masks = {d: rep(mask_all[:, i * P:(i + 1) * P]) for i, d in enumerate(input_tokens.keys())}
        return task_masks, rep(ids_keep), rep(ids_restore)

    @staticmethod
    def make_mask(N_H, N_W, xy_idxs, full_tasks=[], indicate_visible=True, flatten=True, device='cuda'):
        """Masks from lists of un-masked (x, y) patch coordinates (reference :280-308)."""
        out = {}
        for k, v in xy_idxs.items():
            m = torch.ones(N_H, N_W, device=device)
            v = torch.as_tensor(v, dtype=torch.long)
            if len(v) > 0:
                m[v[:, 1], v[:, 0]] = 0
            if k in full_tasks:
                m[:] = 0
            if not indicate_visible:
                m = 1 - m
            out[k] = m.flatten().unsqueeze(0) if flatten else m
        return out

    def generate_input_info(self, input_task_tokens, image_size):
        info = OrderedDict(tasks={})
        i = 0
        for domain, n in input_task_tokens.items():
            n = n if isinstance(n, int) else n.shape[1]
            info['tasks'][domain] = {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': i, 'end_idx': i + n}
            i += n
        info['image_size'] = image_size
        info['num_task_tokens'] = i
        return info

    # ---- packed encoder shared by the pretraining forward and the downstream backbone ------------------------------------
    def _encode(self, x, doms, mask_all, N, explicit, taps=()):
        """Patch-embed the kept patches of modalities `doms`, then run the depth x (Block_Fusion + Block) stack on the
        packed row space.  Returns (descriptors, xm, xf, pending residual deltas, tap list): the residual stream is
        (xm, xf) PLUS the pending deltas, which the caller's next add+LayerNorm pass folds in.  `taps`: layer indices
        after which the fusion tokens (B, P, D) are materialised (downstream feature taps)."""
        B, H, W = x[doms[0]].shape[0], x[doms[0]].shape[-2], x[doms[0]].shape[-1]
        device = x[doms[0]].device
        M, D, Hh, dh = len(doms), self.dim_tokens, self.heads, self.dim_head
        ps = self.input_adapters[doms[0]].P_H
        nh, nw = H // ps, W // ps
        P = nh * nw
        T = compute_dtype(self.fusion_tokens)
        desc = ops.Descriptors(mask_all.contiguous(), B, M, P, N)
        if explicit and self.check_masks:
            desc.check()
        BN, BP = B * N, B * P

        # -- patch embedding of the kept patches: one gather kernel + one GEMM for all modalities --------------------------
        Ks = [self.input_adapters[d].packed_channels * ps * ps for d in doms]
        koff = [sum(Ks[:i]) for i in range(M)]
        onehot = sum(Ks)
        Kcat = onehot + ((M + 7) // 8) * 8
        pcat = ops.patchify_gather([self.input_adapters[d].packed_image(x[d]) for d in doms], koff, onehot, Kcat, ps,
                                   desc.tok_mod, desc.tok_patch, N, T)
        wcat = torch.cat([self.input_adapters[d].packed_weight() for d in doms] +
                         [torch.stack([self.input_adapters[d].packed_bias() for d in doms], dim=1),
                          pcat.new_zeros(D, Kcat - onehot - M, dtype=torch.float32)], dim=1)
        tok = linear(pcat, wcat, once=True)                                                   # (B*N, D), bias included
        pe_table = torch.cat([interp_posemb(self.input_adapters[d].pos_emb, nh, nw) for d in doms], dim=0)
        if pe_table.requires_grad:
            xm = pe_table.index_select(0, desc.tok_pe.long())
        else:
            xm = ops.gather_rows(pe_table.detach().contiguous(), desc.tok_pe)      # (B*N, D) fp32
        fus_pe = self.input_adapters['fusion'].posemb_rows()
        xf = (self.fusion_tokens[0] + fus_pe).unsqueeze(0).expand(B, P, D).reshape(BP, D).contiguous()
        me = self.mask_embedding[0].contiguous() if self.has_fusion_blocks else None   # (P, D) shared rows
        # pending residual deltas (compute dtype): modality part / fusion part, as (tensor, row offset)
        dm, dm_off, df, df_off = tok, 0, None, -1
        tap_out = []

        def one_delta(a, a_off, b, b_off):
            """parts_add_ln takes ONE delta tensor; the two pending deltas are either the same tensor or one is None."""
            if a is None:
                return b, -1, b_off
            if b is None:
                return a, a_off, -1
            assert a is b
            return a, a_off, b_off

        sw = self.side_stream_wgrad
        lt = self.layer_timer
        for l in range(self.depth):
            blk = self.blocks[l]
            timed = lt is not None and l == lt.layer and dm is df and dm is not None
            if timed:
                xm, xf, dm = ops.layer_mark(lt, 0, xm, xf, dm)
                df = dm
            if self.has_fusion_blocks:
                fus = self.fus_blocks[l]
                # ---- Block_Fusion (DSI-MM zorro_utils.py:252-258 on multimae_crossattn.py:454-468) -----------------------
                dl, o1, o2 = one_delta(dm, dm_off, df, df_off)
                # the modality rows do not change until the Block's attention: both of their double LayerNorms -- this
                # stage's and the Block's (zorro_utils.py:238) -- come out of ONE pass over the residual (ops dual mode);
                # zb is the Block's attention input, its fusion rows are filled after the fusion feed-forward below
                if self.dual_layernorm:
                    zb = torch.empty(BN + BP, D, dtype=T, device=xm.device)
                    (xm, xf, _), z, zb = ops.parts_add_ln([xm, xf, me], dl, [o1, o2, -1], fus.norm1.gamma, None,
                                                          fus.attn.norm.gamma, None, out_dtype=T,
                                                          dual=(0, blk.norm1.gamma, blk.attn.norm.gamma, zb, 0))
                else:                                                                      # two-pass form (A/B, tests)
                    (xm, xf, _), z = ops.parts_add_ln([xm, xf, me], dl, [o1, o2, -1], fus.norm1.gamma, None,
                                                      fus.attn.norm.gamma, None, out_dtype=T)       # (BN+BP+P, D)
                # K/V of every slot source + queries of the fusion slots only, one autograd node (ops._KvQ)
                kv, q = ops.kv_q_projections(z, BN, BP, fus.attn.to_q.weight, fus.attn.to_kv.weight)
                a = ops.modattn(q, kv, desc.slot_row, B, P, M + 1, Hh, dh, desc.shared_base, fus.attn.scale)
                o = linear(a, fus.attn.to_out.weight, side_wgrad=sw, once=True)
                (xf,), y = ops.parts_add_ln([xf], o, [0], fus.norm2.gamma, None, fus.mlp[0].gamma, None, out_dtype=T)
                f = ops.feedforward_geglu(y, fus.mlp[1].weight, fus.mlp[3].weight)                    # (BP, D)
                # ---- Block (zorro_utils.py:237-240), Zorro mask as segments ----------------------------------------------
                if self.dual_layernorm:
                    (xf,), z = ops.parts_add_ln([xf], f, [0], blk.norm1.gamma, None, blk.attn.norm.gamma, None, out_dtype=T,
                                                y_into=(zb, BN))                            # (BN+BP, D)
                else:
                    (xm, xf), z = ops.parts_add_ln([xm, xf], f, [-1, 0], blk.norm1.gamma, None, blk.attn.norm.gamma, None,
                                                   out_dtype=T)
            else:                                     # multimae_quadruplet.py:430-432: the Zorro-masked Block only
                dl, o1, o2 = one_delta(dm, dm_off, df, df_off)
                (xm, xf), z = ops.parts_add_ln([xm, xf], dl, [o1, o2], blk.norm1.gamma, None, blk.attn.norm.gamma, None,
                                               out_dtype=T)                                 # (BN+BP, D)
            qkv = linear(z, [blk.attn.to_q.weight, blk.attn.to_kv.weight], side_wgrad=sw, once=True)
            a = ops.mha_self(qkv, Hh, dh, desc.enc_seg, blk.attn.scale)
            o = blk.drop_rows(linear(a, blk.attn.to_out.weight, side_wgrad=sw, once=True), B, (N, P))   # DropPath :238 (rate 0: identity)
            (xm, xf), y = ops.parts_add_ln([xm, xf], o, [0, BN], blk.norm2.gamma, None, blk.mlp[0].gamma, None,
                                           out_dtype=T)
            f = blk.drop_rows(ops.feedforward_geglu(y, blk.mlp[1].weight, blk.mlp[3].weight), B, (N, P))   # (BN+BP, D); DropPath :239
            if timed:
                xm, xf, f = ops.layer_mark(lt, 1, xm, xf, f)
            dm, dm_off, df, df_off = f, 0, f, BN
            if l in taps:
                tap_out.append((xf + f[BN:BN + BP].float()).reshape(B, P, D))

        return desc, xm, xf, (dm, dm_off, df, df_off), tap_out

    # ---- the hot path ----------------------------------------------------------------------------------------------------
    def forward(self,
                x: Union[Dict[str, torch.Tensor], torch.Tensor],
                mask_inputs: bool = True,
                task_masks: Dict[str, torch.Tensor] = None,
                num_encoded_tokens: int = 128,
                alphas: Union[float, List[float]] = 1.0,
                sample_tasks_uniformly: bool = False,
                fp32_output_adapters: List[str] = [],
                return_token_indices: Optional[Tuple[int]] = None):
        x = {'s1': x} if isinstance(x, torch.Tensor) else x
        B, _, H, W = x['s1'].shape                      # KeyError without 's1', as in the reference (:360)
        device = x['s1'].device
        doms = self.domains
        for d in doms:
            _ = x[d]                                    # every configured modality must be present (:402-407)
        if exists(return_token_indices):                # reference :478-484
            assert len(set(return_token_indices)) == len(return_token_indices), 'all indices must be unique'
            assert all(i < self.max_return_tokens for i in return_token_indices), \
                'indices must range from 0 to max_num_return_tokens - 1'
        M, D, Hh, dh = len(doms), self.dim_tokens, self.heads, self.dim_head
        I = Hh * dh
        ad0 = self.input_adapters[doms[0]]
        ps = ad0.P_H
        nh, nw = H // ps, W // ps
        P = nh * nw
        assert P == self.fusion_tokens.shape[1], "fusion tokens are tied to the patch grid (reference :87)"
        T = compute_dtype(self.fusion_tokens)
        N = num_encoded_tokens if mask_inputs else M * P

        # -- masks + device-side descriptors (no host sync on the random path) ---------------------------------------------
        if task_masks is None:
            placeholders = {d: torch.empty(B, P, 0, device=device) for d in doms}
            task_masks, ids_keep, ids_restore = self.generate_random_masks(
                placeholders, N, alphas=alphas, sample_tasks_uniformly=sample_tasks_uniformly)
            mask_all = torch.cat([task_masks[d][:(B if self.per_sample_masks else 1)] for d in doms], dim=1)
            explicit = False
        else:
            mask_full = torch.cat([task_masks[d] for d in doms], dim=1).to(torch.int64)
            ids_shuffle = torch.argsort(mask_full, dim=1, stable=True)           # reference :397-399 (outputs unused)
            ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
            ids_keep = ids_shuffle[:, :N]
            mask_all = mask_full if self.per_sample_masks else mask_full[:1]      # row 0 drives the batch (:402-406)
            explicit = True
        desc, xm, xf, pend, _ = self._encode(x, doms, mask_all, N, explicit)
        dm, dm_off, df, df_off = pend
        BN, BP = B * N, B * P

        def one_delta(a, a_off, b, b_off):
            if a is None:
                return b, -1, b_off
            if b is None:
                return a, a_off, -1
            assert a is b
            return a, a_off, b_off

        # ---- final norm (:472) -------------------------------------------------------------------------------------------
        dl, o1, o2 = one_delta(dm, dm_off, df, df_off)
        # fp32 tokens (returned: ori_tokens / enc_fus) and, under autocast, their bf16 copy for the pool / decoder projections from
        # the same pass over the residual (the copy's gradient then reaches the LayerNorm backward in bf16: no cast either way)
        res = ops.parts_add_ln([xm, xf], dl, [o1, o2], self.norm.gamma, None, out_dtype=torch.float32,
                               cast_copy=(T == torch.bfloat16 and FUSED_FINAL_CAST))
        (xm, xf), tokens = res[0], res[1]
        ori_tokens = tokens[:BN].reshape(B, N, D)                                           # :495
        enc_fus = tokens[BN:].reshape(B, P, D)                                              # :504
        tokens_T = res[2] if len(res) == 3 else wcast(tokens, T)

        # ---- attention pooling into the return tokens (:475-497) ---------------------------------------------------------
        ap = self.attn_pool
        R = self.max_return_tokens
        # pool K/V over every row + (training forward) the context projection of EVERY decoder over the fusion rows: one node,
        # one GEMM for the three proj_context layers (ops._KvCtx).  fp32 output adapters keep their own fp32 projection.
        fused_ctx = FUSED_DECODER_CTX and self.output_adapters is not None and len(self.output_adapters) > 0 and not fp32_output_adapters and \
            all(hasattr(a, 'decode_rows') and a.dim_tokens_enc is not None for a in self.output_adapters.values())
        if fused_ctx:
            ads = list(self.output_adapters.values())
            kvp, ctx_all = ops.kv_ctx_projections(tokens_T, BN, BP, ap.to_kv.weight, [a.proj_context.weight for a in ads],
                                                  [a.proj_context.bias for a in ads])
            ctx_rows = dict(zip(self.output_adapters.keys(),
                                ops.split_cols_f32(ctx_all, [a.dim_tokens for a in ads], [a.own_task_embedding() for a in ads])))
        else:
            kvp = linear(tokens_T, ap.to_kv.weight, once=True)                                         # (BN+BP, 2I), context un-normalised
        gk = None
        if self.output_adapters is not None and self.has_contrastive_tokens:
            # the contrastive queries' keys (:530-543) are a row gather of kvp; taken here, in one node with the pass-through that
            # the pooling attention reads, so that kvp's two gradients are merged by a scatter-add instead of a full-size sum
            # (its one consumer below, ops.mha_cross, returns a gradient tensor it allocates itself: fresh_grad)
            kvp, gk = ops.fork_gather_rows(kvp, desc.tok_fus, filt=desc.tok_mod, nfilt=M, fresh_grad=True)   # (B*N, 2I)
        rq = linear(ops.layernorm(self.return_tokens[0].contiguous(), ap.norm.gamma, out_dtype=T), ap.to_q.weight)
        a = ops.mha_cross(rq.repeat(B, 1), kvp, Hh, dh, desc.pool_q, desc.enc_seg, ap.scale, empty_mode=0)
        pooled = linear(a, ap.to_out.weight).float()                                        # (B*R, D)
        pooled = pooled + self.mlp(ops.layernorm(pooled, self.norm.gamma, out_dtype=T)).float()
        return_tokens = pooled.reshape(B, R, D)
        if exists(return_token_indices):                # a row subset of the pooled queries (each query row is independent)
            return_tokens = return_tokens[:, torch.tensor(list(return_token_indices), dtype=torch.long, device=device)]

        if self.output_adapters is None:
            full = torch.cat([ori_tokens, enc_fus], dim=1)
            return full, return_tokens, task_masks                                          # :500-501

        # ---- per-modality reconstruction decoders on the encoded fusion tokens (:507-527) --------------------------------
        input_info = self.generate_input_info({d: P for d in doms}, image_size=(H, W))
        dec_seg = ops.Segments.dense(B, P, device)
        enc_rows = tokens_T[BN:]
        preds = {}
        # The decoders are independent chains of SMALL kernels (B*P rows x 256 columns: a GEMM of 256 tiles, one per CU, no second
        # wave to hide its tail).  decoder_streams runs them side by side on their own HIP streams (autograd replays each
        # chain's backward on the stream its forward ran on), joined before the losses.
        main = torch.cuda.current_stream()
        use_streams = self.decoder_streams and len(self.output_adapters) > 1 and enc_rows.is_cuda
        if use_streams:
            while len(self._dec_streams) < len(self.output_adapters) - 1:
                self._dec_streams.append(torch.cuda.Stream(device=enc_rows.device))
        for i, (d, adapter) in enumerate(self.output_adapters.items()):
            st_ = self._dec_streams[i - 1] if (use_streams and i > 0) else main
            if st_ is not main:
                st_.wait_stream(main)
            with torch.cuda.stream(st_):
                # fp32 adapters (:518-527) take the final norm's fp32 output itself, as the reference's do (layer_norm leaves an
                # autocast region in fp32) -- not the bf16 copy widened again
                rows = tokens[BN:] if d in fp32_output_adapters else enc_rows
                with torch.autocast("cuda", enabled=False) if d in fp32_output_adapters else _nullctx():
                    tk = adapter.decode_rows(ctx_rows[d], B, P, dec_seg, True, T) if fused_ctx else \
                        adapter.forward_tokens(rows, B, P, dec_seg, once=True)
                if st_ is not main:
                    (ctx_rows[d] if fused_ctx else rows).record_stream(st_)
                    tk.record_stream(main)
            C = adapter.num_channels
            preds[d] = PredTokens(tk, B, C, H, W, adapter.P_H) if self.fuse_unpatchify_loss else \
                ops.unpatchify(tk, B, C, H, W, adapter.P_H)
        if use_streams:
            for st_ in self._dec_streams[:len(self.output_adapters) - 1]:
                main.wait_stream(st_)

        if not self.has_contrastive_tokens:
            return (preds, task_masks, return_tokens, ori_tokens, enc_fus)                  # multimae_quadruplet.py:490
        # ---- contrastive return tokens: one query per modality over the fusion tokens at its kept patches (:530-543) -----
        rt = torch.cat([getattr(self, 'return_token_' + d)[0] for d in doms], dim=0).contiguous()   # (M, D)
        cq = linear(ops.layernorm(rt, ap.norm.gamma, out_dtype=T), ap.to_q.weight)                  # (M, I)
        a = ops.mha_cross(cq.repeat(B, 1), gk, Hh, dh, desc.ctr_q, desc.ctr_k, ap.scale, empty_mode=1)
        r = linear(a, ap.to_out.weight).float()                                             # (B*M, D)
        r = r + self.mlp(ops.layernorm(r, self.norm.gamma, out_dtype=T)).float()
        r = r.reshape(B, M, D)
        rets = [r[:, i:i + 1, :] for i in range(M)]
        return (preds, task_masks, return_tokens, ori_tokens, enc_fus, *rets)               # :545


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _factory(dim_tokens, depth, heads):
    def build(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
        return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=dim_tokens,
                        depth=depth, dim_head=64, heads=heads, ff_mult=4, norm_layer=LayerNorm, **kwargs)
    return build


pretrain_multimae_tiny = _factory(192, 12, 3)      # reference :548-563
pretrain_multimae_base = _factory(768, 12, 8)      # reference :566-581  (inner width 8*64 = 512, not 768)
pretrain_multimae_large = _factory(1024, 24, 8)    # reference :584-599
