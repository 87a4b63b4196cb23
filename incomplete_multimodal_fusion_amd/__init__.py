"""MI355X-native fusion-token pretraining path (MultiMAE-style) behind the reference's `multimae` module API.

    from incomplete_multimodal_fusion_amd.multimae.multimae_crossattn import pretrain_multimae_base
    from incomplete_multimodal_fusion_amd.multimae.zorro_utils import TokenTypes
    ...

The compute path is libmmae_hip.so (hand-written gfx950 kernels, C ABI in include/mmae_hip.h) + hipBLASLt GEMMs through
torch.  There is no CPU or eager fallback: modules raise on host tensors / a missing library.
"""
__version__ = "0.1.0"


def install_as_multimae():
    """Register this package's `multimae` sub-package under the top-level name `multimae`, so that the reference driver's
    own import lines (pretraining/pretrain_mmae.py:35-39: `from multimae.multimae_crossattn import ...`; the 4-modality
    driver's pretrain_mmae_my.py:35-40: `from multimae.multimae_quadruplet import ...`) resolve to the
    native modules without editing the driver.  Call before the driver is imported."""
    import importlib
    import sys
    native = importlib.import_module(__name__ + ".multimae")
    sys.modules["multimae"] = native
    for sub in ("multimae_crossattn", "zorro_utils", "criterion", "input_adapters", "output_adapters_simple",
                "multimae_utils", "multimae_quadruplet", "zorro_utils_quadruplet"):
        sys.modules["multimae." + sub] = importlib.import_module(__name__ + ".multimae." + sub)
    return native
