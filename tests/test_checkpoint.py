"""Checkpoint compatibility: file naming / auto-resume of pretraining/utils/checkpoint.py, the
reference's dict layout, strict model load; the flat-engine optimizer round trip runs on the GPU."""
import os

import pytest
import torch

from tests.test_cabi_symbols import build_model


def _tiny(g_e2e):
    cfg = g_e2e.json("config")
    m = build_model(cfg, cfg["channels"])
    m.load_state_dict(g_e2e.sub("state"), strict=True)
    return m


def test_save_and_auto_resume_layout(tmp_path, g_e2e):
    from incomplete_multimodal_fusion_amd import checkpoint as C
    model = _tiny(g_e2e)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)
    opt.step()
    for ep in (3, 19, 7):
        C.save_model(str(tmp_path), ep, model, opt, args={"lr": 1e-3})
    assert os.path.basename(C.latest_checkpoint(str(tmp_path))) == "checkpoint-19.pth"
    ck = torch.load(os.path.join(tmp_path, "checkpoint-19.pth"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 19
    assert sorted(ck["model"]) == sorted(g_e2e.sub("state"))                  # reference key names
    assert set(ck["optimizer"]) == {"state", "param_groups"}
    model2 = _tiny(g_e2e)
    with torch.no_grad():
        for p in model2.parameters():
            p.zero_()
    opt2 = torch.optim.AdamW(model2.parameters(), lr=5.0)
    start = C.auto_load_model(str(tmp_path), model2, opt2)
    assert start == 20
    for (n, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), n
    assert opt2.param_groups[0]["lr"] == 1e-3
    assert C.auto_load_model(str(tmp_path / "empty"), model2, opt2) == 0


@pytest.mark.gpu
def test_flat_engine_optimizer_round_trip(tmp_path, g_e2e):
    """FlatAdamW state is written in torch.optim.AdamW layout and both optimizers can resume from it."""
    from incomplete_multimodal_fusion_amd import checkpoint as C
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    model = _tiny(g_e2e).to("cuda")
    opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    for _ in range(2):
        opt.zero_grad()
        for p in opt.params:
            p.grad = torch.randn_like(p)
            opt._on_grad(p)
        opt.step()
    C.save_model(str(tmp_path), 4, model, opt)
    ref = torch.optim.AdamW(model.parameters(), lr=9.0)
    model_t = _tiny(g_e2e).to("cuda")
    ref_t = torch.optim.AdamW(model_t.parameters(), lr=9.0)
    assert C.auto_load_model(str(tmp_path), model_t, ref_t, map_location="cuda") == 5
    model_f = _tiny(g_e2e).to("cuda")
    opt_f = FlatAdamW(model_f.parameters(), lr=9.0, betas=(0.9, 0.95), weight_decay=0.0, exclude=model_f.never_used_parameters())
    assert C.auto_load_model(str(tmp_path), model_f, opt_f, map_location="cuda") == 5
    assert opt_f.steps == 2 and opt_f.param_groups[0]["lr"] == 1e-3 and opt_f.param_groups[0]["weight_decay"] == 0.05
    assert torch.equal(opt_f.exp_avg, opt.exp_avg) and torch.equal(opt_f.exp_avg_sq, opt.exp_avg_sq)
    assert torch.equal(opt_f.master, opt.master)
    assert torch.equal(opt_f.shadow, opt_f.master.to(torch.bfloat16))
    st = ref_t.state_dict()["state"]
    assert len(st) == len(opt.params) and all(float(v["step"]) == 2.0 for v in st.values())
