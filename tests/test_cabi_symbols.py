"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mmae_hip.h declares,
the product path refuses host tensors (no fallback), and the module surface keeps the reference's state-dict ABI."""
import ctypes
import inspect
import json
import os

import pytest
import torch

from incomplete_multimodal_fusion_amd import _lib


def test_header_symbols_exported():
    protos = _lib.parse_header()
    assert len(protos) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), "missing export: " + name
    assert _lib.lib().mmae_abi_version() == 7
    # pure host helpers (no GPU needed)
    assert _lib.lib().mmae_add_ln_bwd_ws_floats(1000, 768) == 250 * 4 * 768
    assert _lib.lib().mmae_add_ln_bwd_ws_floats(40960, 768) == 1024 * 4 * 768
    off = (ctypes.c_long * 15)()
    total = _lib.lib().mmae_descriptor_layout(2, 3, 16, 24, ctypes.cast(off, ctypes.c_void_p))
    assert total == off[14] and off[13] + 4 == off[14] and off[6] - off[5] == 2 * 24


def test_invalid_arguments_are_rejected_without_launch():
    l = _lib.lib()
    # null pointers / bad head dim -> MMAE_ERR_ARG, nothing touches a device
    assert l.mmae_geglu_fwd(0, 4, 8, None, None, None) == -1
    assert l.mmae_mha_fwd(1, 48, 1, 1, 1, None, None, None, None, None, 8, 8, 8, 8, 8, None, None, None, None, 8, 8, 1.0, 0, None) == -1
    assert l.mmae_add_ln_fwd(0, 0, 4, 6, None, None, None, None, None, None, 1e-5, None, None, 1e-5, None, None) == -1


def test_product_path_has_no_cpu_fallback():
    from incomplete_multimodal_fusion_amd import ops
    x = torch.randn(4, 32)
    with pytest.raises(_lib.MmaeLibraryError):
        ops.geglu(x)
    with pytest.raises(_lib.MmaeLibraryError):
        ops.layernorm(x, torch.ones(32))


def test_product_never_imports_oracle():
    import incomplete_multimodal_fusion_amd as pkg
    root = os.path.dirname(pkg.__file__)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "product file mentions the oracle: " + f


def build_model(cfg, channels):
    from incomplete_multimodal_fusion_amd.multimae import (FusionInputAdapter, MultiMAE, PatchedInputAdapter,
                                                           SpatialOutputAdapter, TokenTypes)
    doms = [c[0] for c in channels]
    ia = {d: PatchedInputAdapter(num_channels=c, stride_level=1, patch_size_full=cfg["patch_size"],
                                 image_size=cfg["image_size"]) for d, c in channels}
    ia["fusion"] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=cfg["patch_size"],
                                      image_size=cfg["image_size"])
    oa = {d: SpatialOutputAdapter(num_channels=c, stride_level=1, patch_size_full=cfg["patch_size"],
                                  dim_tokens=cfg["decoder_dim"], depth=cfg["decoder_depth"],
                                  num_heads=cfg["decoder_heads"], use_task_queries=True, task=d,
                                  context_tasks=list(doms), use_xattn=True,
                                  drop_path_rate=cfg.get("decoder_drop_path_rate", 0.0)) for d, c in channels}
    P = (cfg["image_size"] // cfg["patch_size"]) ** 2
    return MultiMAE(ia, oa, num_global_tokens=1, dim_tokens=cfg["dim_tokens"], depth=cfg["depth"],
                    dim_head=cfg["dim_head"], heads=cfg["heads"], ff_mult=4, num_fusion_tokens=P,
                    return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                    drop_path_rate=cfg.get("drop_path_rate", 0.0))


def test_state_dict_abi_matches_reference(g_e2e):
    """Keys, shapes, parameter-vs-buffer split and parameter order of the reference checkpoint (strict load)."""
    cfg = g_e2e.json("config")
    model = build_model(cfg, cfg["channels"])
    state = g_e2e.sub("state")
    missing, unexpected = model.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    ref_param_names = g_e2e.json("param_names")
    assert [n for n, _ in model.named_parameters()] == ref_param_names
    assert not model.input_adapters["s1"].pos_emb.requires_grad
    assert "return_token_types_tensor" not in model.state_dict()
    assert sorted(model.state_dict().keys()) == sorted(state.keys())


def test_factory_presets_and_signatures():
    from incomplete_multimodal_fusion_amd.multimae import multimae_crossattn as mc
    sig = inspect.signature(mc.MultiMAE.forward)
    assert list(sig.parameters)[1:] == ["x", "mask_inputs", "task_masks", "num_encoded_tokens", "alphas",
                                        "sample_tasks_uniformly", "fp32_output_adapters", "return_token_indices"]
    assert sig.parameters["num_encoded_tokens"].default == 128
    cfg = dict(patch_size=16, image_size=64, decoder_dim=64, decoder_depth=1, decoder_heads=2)
    from incomplete_multimodal_fusion_amd.multimae import FusionInputAdapter, PatchedInputAdapter, TokenTypes
    ia = {d: PatchedInputAdapter(num_channels=c, stride_level=1, patch_size_full=16, image_size=64)
          for d, c in (("s1", 1), ("s2", 3), ("dem", 1))}
    ia["fusion"] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=16, image_size=64)
    m = mc.pretrain_multimae_tiny(ia, None, num_global_tokens=1, num_fusion_tokens=16,
                                  return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                                  drop_path_rate=0.0)
    assert m.dim_tokens == 192 and m.depth == 12 and m.heads == 3 and m.dim_head == 64
    assert m.blocks[0].mlp[1].weight.shape == (2 * int(192 * 8 / 3), 192)
    with pytest.raises(AssertionError):
        mc.pretrain_multimae_tiny(ia, None, num_fusion_tokens=8)     # must equal s1.num_patches (reference :87)


def test_install_as_multimae_resolves_driver_imports():
    """The reference driver's import lines (pretrain_mmae.py:35-39) must bind the native modules after aliasing."""
    import subprocess, sys
    code = (
        "import incomplete_multimodal_fusion_amd as n; n.install_as_multimae()\n"
        "from multimae.multimae_crossattn import pretrain_multimae_base, pretrain_multimae_tiny\n"
        "from multimae.zorro_utils import TokenTypes as T\n"
        "from multimae.criterion import MaskedL1Loss, MaskedMSELoss, vicreg, HardNegtive_loss, DINOLoss, byol_loss_func, dino_loss_func\n"
        "from multimae.input_adapters import PatchedInputAdapter, FusionInputAdapter\n"
        "from multimae.output_adapters_simple import SpatialOutputAdapter\n"
        "assert pretrain_multimae_base.__module__.startswith('incomplete_multimodal_fusion_amd')\n"
        "assert [t.value for t in T] == [0, 1, 2, 3]\n"
        # the 4-modality driver's lines (pretrain_mmae_my.py:35-40)
        "from multimae.multimae_quadruplet import pretrain_multimae_base as qb, pretrain_multimae_tiny as qt\n"
        "from multimae.zorro_utils_quadruplet import TokenTypes as TQ\n"
        "from multimae.criterion import MaskedCrossEntropyLoss\n"
        "from multimae.input_adapters import SemSegInputAdapter\n"
        "assert qt.__module__.startswith('incomplete_multimodal_fusion_amd') and [t.value for t in TQ] == [0, 1, 2, 3, 4]\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_downstream_backbone_state_dict_abi():
    """ViTBaseline keeps the downstream reference's checkpoint keys (tests/golden/downstream.npz state, strict load)."""
    from tests.conftest import Golden
    from incomplete_multimodal_fusion_amd.multimae import FusionInputAdapter, PatchedInputAdapter, TokenTypes
    from incomplete_multimodal_fusion_amd.multimae.multimae_big_imcomplete import ViTBaseline
    g = Golden("downstream.npz")
    cfg = g.json("config")
    ia = {d: PatchedInputAdapter(num_channels=c, stride_level=1, patch_size_full=cfg["patch_size"], image_size=cfg["image_size"])
          for d, c in cfg["channels"]}
    ia["fusion"] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=cfg["patch_size"], image_size=cfg["image_size"])
    m = ViTBaseline(input_adapters=ia, output_adapters=None, num_fusion_tokens=(cfg["image_size"] // cfg["patch_size"]) ** 2,
                    return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION),
                    dim_tokens=cfg["dim_tokens"], depth=cfg["depth"], dim_head=cfg["dim_head"], heads=cfg["heads"],
                    in_domains=[c[0] for c in cfg["channels"]], pretrained="/nonexistent")
    state = g.sub("state")
    missing, unexpected = torch.nn.Module.load_state_dict(m, state, strict=True)
    assert not missing and not unexpected
    assert m.flags == cfg["flags"]
