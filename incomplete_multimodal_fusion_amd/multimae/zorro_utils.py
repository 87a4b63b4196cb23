"""Encoder building blocks with the reference's names, constructor signatures and state-dict keys
(reference: downstream/instance_segmentation/modeling/multimae/zorro_utils.py -- the working copy of
pretraining/multimae/zorro_utils.py), computed by the gfx950 kernels in ../csrc.

State-dict ABI kept:  LayerNorm.{gamma, beta(buffer)}  Attention.{norm, to_q, to_kv, to_out}
FeedForward = Sequential(LayerNorm, Linear, GEGLU, Linear) -> mlp.{0.gamma,0.beta,1.weight,3.weight}
Block.{norm1, attn, norm2, mlp}   Block_Fusion.{norm1, norm2, attn, mlp}   Mlp.{fc1, fc2}.

The standalone `forward`s below serve API parity and per-module tests.  The pretraining hot path
(MultiMAE.forward) drives the same kernels through a packed, static-shape pipeline instead of calling them.
Device tensors only: there is no CPU path in this package.
"""
from enum import Enum
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


class TokenTypes(Enum):          # zorro_utils.py:14-18
    S1 = 0
    S2 = 1
    DEM = 2
    FUSION = 3


class TokenTypesQuad(Enum):      # pretraining/multimae/zorro_utils_quadruplet.py:17-22 (4-modality driver)
    S1 = 0
    S2 = 1
    DEM = 2
    DNW = 3
    FUSION = 4


def exists(val):
    return val is not None


def compute_dtype(ref: torch.Tensor) -> torch.dtype:
    """bf16 inside a torch autocast region (the reference driver wraps the step in autocast, pretrain_mmae.py:466),
    otherwise the parameter dtype (fp32)."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return torch.float32 if ref.dtype not in (torch.float32, torch.bfloat16) else ref.dtype


def wcast(w: torch.Tensor, dtype) -> torch.Tensor:
    return w if w.dtype == dtype else w.to(dtype)


def linear(x, weight, bias=None, side_wgrad=False, once=False):
    """Dense projection: library GEMMs (hipBLASLt/rocBLAS through torch) for y and dx, split-K batched GEMM with an fp32
    sum for the weight gradient (ops._Linear).  `weight` may be a list of master weights to be row-concatenated."""
    return ops.linear(x, weight, bias, side_wgrad, once)


def segments_from_mask(mask: torch.Tensor, B: int):
    """Host helper (synchronises): turn a Zorro-family boolean mask (n_q, n_k) shared by the batch into segment
    descriptors.  Supported structure: consecutive query rows with the same allowed key set form a group; the allowed
    set of a group is one contiguous key range, empty, or all keys (only the last group).  Anything else raises."""
    m = mask.detach().to("cpu", torch.bool)
    nq, nk = m.shape
    groups = []          # (q0, q1, lo, hi, full)
    i = 0
    while i < nq:
        j = i
        while j + 1 < nq and torch.equal(m[j + 1], m[i]):
            j += 1
        idx = m[i].nonzero().flatten()
        if len(idx) == nk and nk > 0:
            lo, hi, full = 0, nk, True
        elif len(idx) == 0:
            lo, hi, full = 0, 0, False
        else:
            lo, hi = int(idx[0]), int(idx[-1]) + 1
            if hi - lo != len(idx):
                raise NotImplementedError("attention mask rows must allow one contiguous key range")
            full = False
        groups.append((i, j + 1, lo, hi, full))
        i = j + 1
    if any(g[4] for g in groups[:-1]):
        raise NotImplementedError("attend-all (fusion) query rows must come last")
    segs = [(g[0], g[1] - g[0], g[2], g[3] - g[2]) for g in groups if not g[4]]
    covered = torch.zeros(nk, dtype=torch.bool)
    for _, _, lo, ln in segs:
        if covered[lo:lo + ln].any():
            raise NotImplementedError("key ranges of different query groups overlap")
        covered[lo:lo + ln] = True
    # keys nobody attends specifically: contiguous runs become query-less key segments; the last run joins the fusion segment
    runs, k = [], 0
    while k < nk:
        if not covered[k]:
            e = k
            while e < nk and not covered[e]:
                e += 1
            runs.append((k, e - k))
            k = e
        else:
            k += 1
    last_q = (groups[-1][0], groups[-1][1] - groups[-1][0]) if groups and groups[-1][4] else (nq, 0)
    last_k = runs.pop() if runs else (nk, 0)
    segs += [(nq, 0, lo, ln) for lo, ln in runs]
    segs.append((last_q[0], last_q[1], last_k[0], last_k[1]))
    if len(segs) > 8:
        raise NotImplementedError("more than 8 segments")
    dev = mask.device
    ar = torch.arange(B, dtype=torch.int32, device=dev)[:, None]
    qs = ar * nq + torch.tensor([s[0] for s in segs], dtype=torch.int32, device=dev)[None]
    ql = torch.tensor([s[1] for s in segs], dtype=torch.int32, device=dev)[None].expand(B, -1)
    ks = ar * nk + torch.tensor([s[2] for s in segs], dtype=torch.int32, device=dev)[None]
    kl = torch.tensor([s[3] for s in segs], dtype=torch.int32, device=dev)[None].expand(B, -1)
    return ops.Segments(qs.contiguous(), ql.contiguous(), nq), ops.Segments(ks.contiguous(), kl.contiguous(), nk)


# bias-less layernorm (zorro_utils.py:103-110)
class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x):
        shp = x.shape
        y = ops.layernorm(x.reshape(-1, shp[-1]).float().contiguous(), self.gamma, self.beta, out_dtype=torch.float32)
        return y.reshape(shp)


class GEGLU(nn.Module):          # zorro_utils.py:115-118
    def forward(self, x):
        return ops.geglu(x)


def FeedForward(dim, mult=4):    # zorro_utils.py:121-128
    inner_dim = int(dim * mult * 2 / 3)
    return _FeedForward(
        LayerNorm(dim),
        nn.Linear(dim, inner_dim * 2, bias=False),
        GEGLU(),
        nn.Linear(inner_dim, dim, bias=False),
    )


class _FeedForward(nn.Sequential):
    def forward(self, x, pre_gamma: Optional[torch.Tensor] = None):
        """x (.., D) fp32.  With pre_gamma the preceding Block.norm2 is fused into the same pass (double LN)."""
        shp = x.shape
        T = compute_dtype(self[1].weight)
        x2 = x.reshape(-1, shp[-1]).float().contiguous()
        if pre_gamma is None:
            y = ops.layernorm(x2, self[0].gamma, self[0].beta, out_dtype=T)
        else:
            y = ops.layernorm(x2, pre_gamma, None, self[0].gamma, self[0].beta, out_dtype=T)
        h = linear(y, self[1].weight)
        f = linear(ops.geglu(h), self[3].weight)
        return f.reshape(*shp[:-1], f.shape[-1])


class Mlp(nn.Module):            # zorro_utils.py:131-148
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        assert drop == 0.0, "dropout is 0 on this path (reference default)"

    def forward(self, x):
        T = compute_dtype(self.fc1.weight)
        h = linear(wcast(x, T), self.fc1.weight, self.fc1.bias)
        return linear(ops.gelu(h), self.fc2.weight, self.fc2.bias)


class Attention(nn.Module):      # zorro_utils.py:152-194
    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        self.norm = LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x, context=None, attn_mask=None, pre_gamma: Optional[torch.Tensor] = None,
                segments: Optional[Tuple[ops.Segments, ops.Segments]] = None, empty_mode: int = 0):
        """x (B, n, D) fp32; context (B, m, D) (NOT normalised, :177); attn_mask bool (n, m) of the Zorro family or None.
        pre_gamma fuses the caller's norm1 into the same pass.  Returns (B, n, D) in the compute dtype."""
        B, n, D = x.shape
        T = compute_dtype(self.to_q.weight)
        H, dh = self.heads, self.dim_head
        x2 = x.reshape(B * n, D).float().contiguous()
        if pre_gamma is None:
            y = ops.layernorm(x2, self.norm.gamma, self.norm.beta, out_dtype=T)
        else:
            y = ops.layernorm(x2, pre_gamma, None, self.norm.gamma, self.norm.beta, out_dtype=T)
        m = n if context is None else context.shape[1]
        if segments is None:
            if attn_mask is None:
                segments = (ops.Segments.dense(B, n, x.device), ops.Segments.dense(B, m, x.device))
            else:
                segments = segments_from_mask(attn_mask, B)
        qseg, kseg = segments
        if context is None:
            qkv = linear(y, [self.to_q.weight, self.to_kv.weight])
            a = ops._MHA.apply(qkv, None, 0, H * dh, 2 * H * dh, H, dh, qseg, kseg, self.scale, empty_mode)
        else:
            q = linear(y, self.to_q.weight)
            if m == 0:
                a = torch.zeros_like(q)          # empty context: softmax over nothing -> zeros (:189-191)
            else:
                kv = linear(wcast(context.reshape(B * m, D), T).contiguous(), self.to_kv.weight)
                a = ops.mha_cross(q, kv, H, dh, qseg, kseg, self.scale, empty_mode)
        return linear(a, self.to_out.weight).reshape(B, n, D)


_DROP_PATH_DRAWS = None          # tests: a list of (B,) uniform draws consumed in call order instead of torch.rand


def set_drop_path_draws(draws):
    """Inject the uniform draws of the following training forward(s): one (B,) tensor per DropPath application, in the order the
    reference's modules call torch.rand (per Block with rate > 0: attention branch, then feed-forward branch; encoder layers first,
    then the decoders in domain order).  None restores torch.rand.  CPU and GPU generators differ, so parity tests inject."""
    global _DROP_PATH_DRAWS
    _DROP_PATH_DRAWS = None if draws is None else list(draws)


class DropPath(nn.Module):       # zorro_utils.py:69-96 (= multimae_utils.py:105-135): stochastic depth per sample
    """output = x / keep * floor(keep + u), u ~ U[0, 1) drawn per sample with shape (B, 1, ..., 1) in x's dtype (:79-83); identity
    when drop_prob == 0 or in eval mode (:77-78).  On the packed row space `rows()` scales a residual branch's output row by row
    (csrc/rowops.hip scale_rows_kernel)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def active(self) -> bool:
        return bool(self.drop_prob) and self.training

    def draw(self, B: int, dtype, device) -> torch.Tensor:
        if _DROP_PATH_DRAWS is not None:
            return _DROP_PATH_DRAWS.pop(0).to(device=device).reshape(B)
        return torch.rand((B, 1, 1), dtype=dtype, device=device).reshape(B)           # reference shape / dtype (:80-81)

    def rows(self, x2d: torch.Tensor, B: int, counts) -> torch.Tensor:
        """x2d: row blocks stacked as B * counts[0] rows, then B * counts[1] rows, ... (sample-major inside each block)."""
        if not self.active():
            return x2d
        u = self.draw(B, x2d.dtype, x2d.device)
        return ops.scale_rows(x2d, ops.drop_path_row_scale(u, self.drop_prob, counts))

    def forward(self, x):
        if not self.active():
            return x
        B, W = x.shape[0], x.shape[-1]
        n = x.numel() // (B * W)
        return self.rows(x.reshape(B * n, W).contiguous(), B, (n,)).reshape(x.shape)

    def extra_repr(self) -> str:
        return 'p={}'.format(self.drop_prob)


class Block(nn.Module):          # zorro_utils.py:227-240
    def __init__(self, dim=768, dim_head=64, heads=8, ff_mult=4, drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim=dim, dim_head=dim_head, heads=heads)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()          # :233
        self.norm2 = norm_layer(dim)
        self.mlp = FeedForward(dim=dim, mult=ff_mult)

    def drop_rows(self, x2d, B, counts):
        """DropPath of this block on a packed residual branch (identity at rate 0 / eval): :238-239."""
        return self.drop_path.rows(x2d, B, counts) if isinstance(self.drop_path, DropPath) else x2d

    def forward(self, x, attn_mask, segments=None):
        x = x.float() + self.drop_path(self.attn(x, attn_mask=attn_mask, pre_gamma=self.norm1.gamma, segments=segments)).float()
        x = x + self.drop_path(self.mlp(x, pre_gamma=self.norm2.gamma)).float()
        return x


class Block_Fusion(nn.Module):   # DSI-MM zorro_utils.py:243-258 (canonical)
    def __init__(self, dim=768, dim_head=64, heads=8, ff_mult=4, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)
        self.attn = Attention(dim=dim, dim_head=dim_head, heads=heads)
        self.mlp = FeedForward(dim=dim, mult=ff_mult)

    def forward(self, x, attn_mask=None):
        """x (B, n, m, D): m slots per position, the last one is the fusion token.  Only the fusion slot's query is
        evaluated (the reference discards the others at :256); K/V come from all m slots."""
        assert attn_mask is None
        B, n, m, D = x.shape
        T = compute_dtype(self.attn.to_q.weight)
        H, dh = self.attn.heads, self.attn.dim_head
        rows = x.reshape(B * n * m, D).float().contiguous()
        z = ops.layernorm(rows, self.norm1.gamma, None, self.attn.norm.gamma, None, out_dtype=T)
        kv = linear(z, self.attn.to_kv.weight)
        zf = z.reshape(B * n, m, D)[:, -1, :].contiguous()
        q = linear(zf, self.attn.to_q.weight)
        # every slot is a real row here (no shared mask-embedding rows): append n dummy shared rows for the kernel ABI
        slot = torch.arange(B * n * m, dtype=torch.int32, device=x.device).reshape(B * n, m)
        kv_ext = torch.cat([kv, kv.new_zeros(n, kv.shape[1])], dim=0)
        a = ops.modattn(q, kv_ext, slot.contiguous(), B, n, m, H, dh, B * n * m, self.attn.scale)
        o = linear(a, self.attn.to_out.weight)
        xf = x[:, :, -1, :].reshape(B * n, D).float() + o.float()
        xf = xf + self.mlp(xf, pre_gamma=self.norm2.gamma).float()
        return xf.reshape(B, n, D)
