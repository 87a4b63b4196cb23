"""MMAE_* tuning variables -> implementation switches of the product modules, for same-box A/B runs (tools/ab_bench.sh).

The product package reads NO tuning variable itself (only MMAE_HIP_LIB, the library path, and the launcher's RANK / WORLD_SIZE):
`python bench.py --tuning-env ...` calls apply() once before the model is built; nothing else does.

    MMAE_MHA_VARIANT     ops.MHA_SELF_VARIANT       kernel variant of the encoder's self-attention calls (csrc/mmae_internal.h)
    MMAE_FUSED_CAST      multimae_crossattn.FUSED_FINAL_CAST
    MMAE_FUSED_CTX       multimae_crossattn.FUSED_DECODER_CTX
    MMAE_DUAL_LN         multimae_crossattn.DUAL_LAYERNORM
    MMAE_DECODER_STREAMS multimae_crossattn.DECODER_STREAMS
    MMAE_SPLITK_CAP / MMAE_SPLITK_MAX / MMAE_WGRAD_PRIO / MMAE_FF_CHUNKS      the ops.py constants of the same names
    MMAE_OWN_GEMM        ops.OWN_GEMM               bit 0: own 8-phase GEMM for forward / input-gradient projections, bit 1: for weight gradients
    MMAE_ASYNC_DRAW      multimae_crossattn.ASYNC_DRAW_COPY   0: the mask draw's host->device copy from pageable memory (one host/GPU sync per step)
    MMAE_DEFER_SPLITK    ops.DEFER_SPLITK           0: one split-K sum launch per weight gradient instead of one per layer
    MMAE_MHA_FUSED_BWD   ops.MHA_FUSED_BWD          0: the dQ + dK/dV kernel pair instead of the fused attention backward
    MMAE_PAD_FF_MIN_TILES ops.PAD_FF_MIN_TILES      output tiles of FeedForward[3] from which the padded route is taken
    MMAE_OWN_GEMM_MIN_TILES ops._OWN_GEMM_MIN_TILES output tiles from which a projection runs on the own GEMM (512 = two tiles per CU)
    MMAE_PAD_FF          ops.PAD_FF                 0: FeedForwards whose GEGLU width fits none of the own GEMM's tiles (ViT-L) stay on the library GEMMs
"""
import os


def apply(verbose=True):
    from incomplete_multimodal_fusion_amd import ops
    from incomplete_multimodal_fusion_amd.multimae import multimae_crossattn as mc
    env = os.environ
    changed = {}

    def put(mod, attr, key, conv):
        if key in env and hasattr(mod, attr):
            setattr(mod, attr, conv(env[key]))
            changed[key] = getattr(mod, attr)
    flag = lambda v: v != "0"
    put(ops, "MHA_SELF_VARIANT", "MMAE_MHA_VARIANT", int)
    put(ops, "_SPLITK_CAP", "MMAE_SPLITK_CAP", int)
    put(ops, "_SPLITK_MAX", "MMAE_SPLITK_MAX", int)
    put(ops, "WGRAD_STREAM_PRIORITY", "MMAE_WGRAD_PRIO", int)
    put(ops, "FF_CHUNKS", "MMAE_FF_CHUNKS", int)
    put(ops, "OWN_GEMM", "MMAE_OWN_GEMM", int)
    put(ops, "DEFER_SPLITK", "MMAE_DEFER_SPLITK", flag)
    put(ops, "PAD_FF", "MMAE_PAD_FF", flag)
    put(ops, "MHA_FUSED_BWD", "MMAE_MHA_FUSED_BWD", flag)
    put(ops, "PAD_FF_MIN_TILES", "MMAE_PAD_FF_MIN_TILES", int)
    put(ops, "_OWN_GEMM_MIN_TILES", "MMAE_OWN_GEMM_MIN_TILES", int)
    put(mc, "FUSED_FINAL_CAST", "MMAE_FUSED_CAST", flag)
    put(mc, "FUSED_DECODER_CTX", "MMAE_FUSED_CTX", flag)
    put(mc, "DUAL_LAYERNORM", "MMAE_DUAL_LN", flag)
    put(mc, "DECODER_STREAMS", "MMAE_DECODER_STREAMS", flag)
    put(mc, "ASYNC_DRAW_COPY", "MMAE_ASYNC_DRAW", flag)
    if verbose and changed:
        import sys
        print("[tuning_env] " + ", ".join("%s=%s" % kv for kv in sorted(changed.items())), file=sys.stderr)
    return changed
