"""MI355X-native fusion-token pretraining path (MultiMAE-style) behind the reference's `multimae` module API.

    from incomplete_multimodal_fusion_amd.multimae.multimae_crossattn import pretrain_multimae_base
    from incomplete_multimodal_fusion_amd.multimae.zorro_utils import TokenTypes
    ...

The compute path is libmmae_hip.so (hand-written gfx950 kernels, C ABI in include/mmae_hip.h) + hipBLASLt GEMMs through
torch.  There is no CPU or eager fallback: modules raise on host tensors / a missing library.
"""
__version__ = "0.1.0"
