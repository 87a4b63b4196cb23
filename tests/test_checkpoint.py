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


def test_reference_way_optimizer_layout(g_e2e):
    """create_optimizer builds what the reference trainer builds (utils/optim_factory.py:136-150): two groups, trainable
    parameters only (the frozen sin-cos pos_emb are NOT optimizer parameters), weight decay everywhere."""
    from incomplete_multimodal_fusion_amd.pretrain import NoWeightingStrategy, UncertaintyWeightingStrategy, create_optimizer
    model = _tiny(g_e2e)
    n_train = sum(1 for p in model.parameters() if p.requires_grad)
    n_all = sum(1 for _ in model.parameters())
    assert n_all > n_train                                                   # pos_emb parameters are frozen
    sd = create_optimizer(model, NoWeightingStrategy()).state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [n_train, 0]
    sd = create_optimizer(model, UncertaintyWeightingStrategy(["s1", "s2", "dem"]), balancer_lr_scale=2.0).state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [n_train, 1]
    assert sd["param_groups"][1]["lr_scale"] == 2.0 and sd["param_groups"][0]["weight_decay"] == 0.05


@pytest.mark.gpu
def test_flat_engine_optimizer_round_trip(tmp_path, g_e2e):
    """FlatAdamW state is written in the REFERENCE trainer's optimizer layout (two groups over the trainable parameters,
    pos_emb excluded) and exchanges files with an optimizer built the reference way, in both directions."""
    from incomplete_multimodal_fusion_amd import checkpoint as C
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import NoWeightingStrategy, create_optimizer
    model = _tiny(g_e2e).to("cuda")
    opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    for _ in range(2):
        opt.zero_grad()
        for p in opt.params:
            p.grad = torch.randn_like(p)
            opt._on_grad(p)
        opt.step()
    C.save_model(str(tmp_path), 4, model, opt, loss_balancer=NoWeightingStrategy())
    ck = torch.load(os.path.join(tmp_path, "checkpoint-4.pth"), weights_only=False)
    n_train = sum(1 for p in model.parameters() if p.requires_grad)
    assert [len(g["params"]) for g in ck["optimizer"]["param_groups"]] == [n_train, 0]
    # (1) engine file -> reference-way torch optimizer (torch's own load_state_dict: group sizes / shapes must match)
    model_t = _tiny(g_e2e).to("cuda")
    ref_t = create_optimizer(model_t, NoWeightingStrategy(), lr=9.0)
    assert C.auto_load_model(str(tmp_path), model_t, ref_t, map_location="cuda") == 5
    st = ref_t.state_dict()["state"]
    assert len(st) == len(opt.params) and all(float(v["step"]) == 2.0 for v in st.values())
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    by_name = dict(model.named_parameters())
    for i, v in st.items():
        p = by_name[names[i]]
        o = opt.offsets[id(p)]
        assert torch.equal(v["exp_avg"], opt.exp_avg[o:o + p.numel()].view_as(p)), names[i]
    # (2) engine file -> engine
    model_f = _tiny(g_e2e).to("cuda")
    opt_f = FlatAdamW(model_f.parameters(), lr=9.0, betas=(0.9, 0.95), weight_decay=0.0, exclude=model_f.never_used_parameters())
    assert C.auto_load_model(str(tmp_path), model_f, opt_f, map_location="cuda") == 5
    assert opt_f.steps == 2 and opt_f.param_groups[0]["lr"] == 1e-3 and opt_f.param_groups[0]["weight_decay"] == 0.05
    assert torch.equal(opt_f.exp_avg, opt.exp_avg) and torch.equal(opt_f.exp_avg_sq, opt.exp_avg_sq)
    assert torch.equal(opt_f.master, opt.master)
    assert torch.equal(opt_f.shadow, opt_f.master.to(torch.bfloat16))
    # (3) reference-way torch optimizer file -> engine: step both once more on the same gradients, weights must agree
    for p in model_t.parameters():
        p.grad = None
    g = torch.Generator(device="cuda").manual_seed(5)
    grads = {n: torch.randn(p.shape, device="cuda", generator=g) for n, p in model_t.named_parameters() if p.requires_grad}
    unused = {id(p) for p in model_t.never_used_parameters()}
    for n, p in model_t.named_parameters():
        if p.requires_grad and id(p) not in unused:
            p.grad = grads[n].clone()
    for grp in ref_t.param_groups:
        grp["lr"] = 1e-3
    ref_t.step()
    sub = tmp_path / "ref"
    C.save_model(str(sub), 7, model_t, ref_t, loss_balancer=NoWeightingStrategy())
    model_g = _tiny(g_e2e).to("cuda")
    opt_g = FlatAdamW(model_g.parameters(), lr=1.0, betas=(0.9, 0.95), weight_decay=0.0, exclude=model_g.never_used_parameters())
    assert C.auto_load_model(str(sub), model_g, opt_g, map_location="cuda") == 8
    assert opt_g.steps == 3
    opt_g.zero_grad()
    for n, p in model_g.named_parameters():
        if id(p) in opt_g.offsets:
            p.grad = grads[n].clone()
            opt_g._on_grad(p)
    opt_g.step()
    ref_t.step()                                                             # same gradients again on the torch side
    for (n, a), (_, b) in zip(model_g.named_parameters(), model_t.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=3e-6), n


class _FakeFlat:
    """The attributes of engine.FlatAdamW that checkpoint.optimizer_state_dict / load_optimizer_state_dict touch, on the CPU
    (the real engine needs the GPU: tests/test_gpu_engine.py runs the same round trip with it)."""

    def __init__(self, params, lr=1e-3):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, o = {}, 0
        for p in self.params:
            self.offsets[id(p)] = o; o += p.numel()
        self.exp_avg = torch.randn(o); self.exp_avg_sq = torch.rand(o)
        self._pstep = {id(p): 3 for p in self.params}
        self.param_groups = [{"lr": lr, "weight_decay": 0.05, "lr_scale": 1.0}]
        self.betas, self.eps, self.steps = (0.9, 0.95), 1e-8, 3

    def skipped_steps(self):
        return 0

    def set_param_steps(self, steps):
        self._pstep = dict(steps); self.steps = max(steps.values()) if steps else 0

    def refresh_shadow(self):
        pass


def test_balancer_group_of_the_flat_engine_checkpoint_layout(tmp_path, g_e2e):
    """the loss balancer's optimizer group next to a flat engine.  The companion AdamW's state is indexed from 0 there
    and from len(model params) in the reference's joint two-group layout (len(params) + i on save, i - n0 on load), and group 1
    records lr * lr_scale as the reference trainer's optimizer does (utils/optim_factory.py:136-150, pretrain_mmae.py:439-445)."""
    from incomplete_multimodal_fusion_amd import checkpoint as C
    from incomplete_multimodal_fusion_amd.pretrain import UncertaintyWeightingStrategy, create_optimizer
    model = _tiny(g_e2e)
    bal = UncertaintyWeightingStrategy(["s1", "s2", "dem"])
    eng = _FakeFlat(model.parameters(), lr=1e-3)
    comp = torch.optim.AdamW(bal.parameters(), lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05)
    for _ in range(2):
        bal.log_vars.grad = torch.tensor([0.5, -1.0, 2.0])
        comp.step()
    sd = C.optimizer_state_dict(eng, model, bal, balancer_lr_scale=2.0, balancer_optimizer=comp)
    n = len(eng.params)
    assert [len(g["params"]) for g in sd["param_groups"]] == [n, 1] and sd["param_groups"][1]["params"] == [n]
    assert sd["param_groups"][1]["lr"] == pytest.approx(2e-3) and sd["param_groups"][1]["lr_scale"] == 2.0
    assert sd["param_groups"][0]["lr"] == pytest.approx(1e-3)
    assert float(sd["state"][n]["step"]) == 2.0 and torch.equal(sd["state"][n]["exp_avg"], comp.state[bal.log_vars]["exp_avg"])
    # (1) the file loads into ONE optimizer built the reference way (torch's own checks: group sizes, shapes)
    C.save_model(str(tmp_path), 2, model, eng, loss_balancer=bal, balancer_lr_scale=2.0, balancer_optimizer=comp)
    model_t, bal_t = _tiny(g_e2e), UncertaintyWeightingStrategy(["s1", "s2", "dem"])
    ref = create_optimizer(model_t, bal_t, lr=9.0, balancer_lr_scale=2.0)
    assert C.auto_load_model(str(tmp_path), model_t, ref, loss_balancer=bal_t) == 3
    assert torch.equal(bal_t.log_vars, bal.log_vars)
    st = ref.state_dict()["state"]
    assert torch.equal(st[n]["exp_avg_sq"], comp.state[bal.log_vars]["exp_avg_sq"]) and float(st[n]["step"]) == 2.0
    assert ref.param_groups[1]["lr"] == pytest.approx(2e-3)
    # (2) ... and back into an engine + companion pair: the companion receives group 1, re-indexed from 0
    model_f, bal_f = _tiny(g_e2e), UncertaintyWeightingStrategy(["s1", "s2", "dem"])
    eng_f = _FakeFlat(model_f.parameters(), lr=7.0)
    comp_f = torch.optim.AdamW(bal_f.parameters(), lr=7.0)
    assert C.auto_load_model(str(tmp_path), model_f, eng_f, loss_balancer=bal_f, balancer_optimizer=comp_f) == 3
    assert torch.equal(comp_f.state[bal_f.log_vars]["exp_avg"], comp.state[bal.log_vars]["exp_avg"])
    assert float(comp_f.state[bal_f.log_vars]["step"]) == 2.0
    assert torch.equal(eng_f.exp_avg, eng.exp_avg) and eng_f.param_groups[0]["lr"] == pytest.approx(1e-3)
    # one more identical step on both companions: identical log_vars (state AND hyper-parameters arrived)
    for b_, c_ in ((bal, comp), (bal_f, comp_f)):
        for g_ in c_.param_groups:
            g_["lr"] = 2e-3; g_["betas"] = (0.9, 0.95); g_["weight_decay"] = 0.05
        b_.log_vars.grad = torch.tensor([1.0, 1.0, -1.0]); c_.step()
    assert torch.allclose(bal_f.log_vars, bal.log_vars, rtol=0, atol=1e-7)
