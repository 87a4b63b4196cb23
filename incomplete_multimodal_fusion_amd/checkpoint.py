"""Checkpoint format compatible with the reference trainer.

Reference: pretraining/utils/checkpoint.py -- save_model (:75-100) writes
`output_dir/checkpoint-{epoch}.pth` = {'model', 'optimizer', 'epoch', 'scaler', 'args'[, 'loss_balancer']} from rank 0
only; auto_load_model (:103-152) resumes from the highest-numbered file, restores model / optimizer / scaler and sets
args.start_epoch = epoch + 1.  Consumers of the 'model' entry: pretraining/infer_mmae.py:146-147 (strict load) and the
downstream backbone (downstream/.../multimae_big_imcomplete.py:456-460).

The 'model' entry is the module's own state_dict (reference key names / shapes, tested by
tests/test_cabi_symbols.py).  The 'optimizer' entry is written in the layout the REFERENCE trainer's optimizer has
(torch.optim.AdamW over two groups: the model's trainable parameters, then the loss balancer's; create_optimizer's dict
path, utils/optim_factory.py:136-150) even when the fused flat engine (engine.FlatAdamW) produced it, so either optimizer
can resume from either file (tests/test_checkpoint.py builds the reference-way optimizer and exchanges files with it).
"""
import glob
import os
import re
from typing import Optional

import torch
import torch.distributed as dist


def _is_main() -> bool:
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _trainable(model):
    """The reference's model parameter group: [p for n, p in model.named_parameters() if p.requires_grad]
    (utils/optim_factory.py:136-141) -- the frozen sin-cos pos_emb Parameters are NOT in the optimizer."""
    return [p for _, p in model.named_parameters() if p.requires_grad]


def optimizer_state_dict(optimizer, model, balancer=None, balancer_lr_scale: float = 1.0, balancer_optimizer=None) -> dict:
    """torch.optim.AdamW state_dict in the reference trainer's layout for a torch optimizer or an engine.FlatAdamW:
    two parameter groups -- model (lr_scale 1) then loss balancer (lr_scale balancer_lr_scale; empty for
    NoWeightingStrategy) -- indexed over the trainable parameters in named_parameters() order; state only for parameters
    that have been stepped (utils/optim_factory.py:136-150, pretrain_mmae.py:351-352)."""
    if isinstance(optimizer, torch.optim.Optimizer):
        return optimizer.state_dict()
    params = _trainable(model)
    idx = {id(p): i for i, p in enumerate(params)}
    skipped = optimizer.skipped_steps()
    state = {}
    for p in optimizer.params:
        st = max(0, optimizer._pstep[id(p)] - skipped)
        if st == 0 or id(p) not in idx:
            continue                                       # never stepped: torch keeps no state for it
        o = optimizer.offsets[id(p)]
        sl = slice(o, o + p.numel())
        state[idx[id(p)]] = {"step": torch.tensor(float(st)),
                             "exp_avg": optimizer.exp_avg[sl].view_as(p).detach().clone(),
                             "exp_avg_sq": optimizer.exp_avg_sq[sl].view_as(p).detach().clone()}
    g = optimizer.param_groups[0]
    common = {"lr": g["lr"], "betas": tuple(optimizer.betas), "eps": optimizer.eps, "weight_decay": g["weight_decay"],
              "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
              "fused": None, "decoupled_weight_decay": True}
    nb = len([p for p in balancer.parameters() if p.requires_grad]) if balancer is not None else 0
    if balancer_optimizer is not None:                     # PretrainStep's companion AdamW of the balancer group: its state
        bsd = balancer_optimizer.state_dict()              # is indexed from 0 there, from len(params) in the joint layout
        for i, st in bsd["state"].items():
            st = dict(st)
            if torch.is_tensor(st.get("step")):            # the companion keeps its step counts on the device (capturable);
                st["step"] = st["step"].detach().float().cpu()   # the reference layout has CPU step tensors
            state[len(params) + int(i)] = st
    groups = [dict(common, lr_scale=g.get("lr_scale", 1.0), params=list(range(len(params)))),
              # the reference stores lr * lr_scale in a group (optim_factory.py:136-150 scales the group's lr when it is set)
              dict(common, lr=g["lr"] * balancer_lr_scale, lr_scale=balancer_lr_scale,
                   params=list(range(len(params), len(params) + nb)))]
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(optimizer, model, sd: dict, balancer_optimizer=None):
    """Accepts the reference layout (two groups over the trainable parameters) written by either optimizer.
    balancer_optimizer: PretrainStep.balancer_opt (flat engine + trainable loss balancer) -- receives the second group."""
    if isinstance(optimizer, torch.optim.Optimizer):
        optimizer.load_state_dict(sd)
        return
    if balancer_optimizer is not None and len(sd["param_groups"]) > 1:
        n0 = len(sd["param_groups"][0]["params"])
        bsd = balancer_optimizer.state_dict()
        bsd["state"] = {int(i) - n0: st for i, st in sd["state"].items() if int(i) >= n0}
        balancer_optimizer.load_state_dict(bsd)
    params = _trainable(model)
    n_model = len(sd["param_groups"][0]["params"])
    if n_model != len(params):
        raise ValueError("optimizer state has %d model parameters, the model has %d trainable ones" % (n_model, len(params)))
    steps = {}
    optimizer.exp_avg.zero_(); optimizer.exp_avg_sq.zero_()
    for i, st in sd["state"].items():
        i = int(i)
        if i >= n_model:
            continue                                       # balancer group: not held by the flat engine
        p = params[i]
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError("optimizer state %d has shape %s, parameter has %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
        if id(p) not in optimizer.offsets:
            continue                                       # excluded from the engine (never receives a gradient)
        o = optimizer.offsets[id(p)]
        optimizer.exp_avg[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
        optimizer.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
        steps[id(p)] = int(float(st["step"]))
    optimizer.set_param_steps(steps)
    g = sd["param_groups"][0]
    optimizer.param_groups[0]["lr"] = g["lr"]
    optimizer.param_groups[0]["weight_decay"] = g["weight_decay"]
    optimizer.refresh_shadow()


def save_model(output_dir: str, epoch: int, model, optimizer, args=None, loss_scaler=None, loss_balancer=None,
               balancer_lr_scale: float = 1.0, balancer_optimizer=None) -> Optional[str]:
    """checkpoint.py:75-93 -- rank 0 only.  balancer_lr_scale / balancer_optimizer: the loss balancer's group of the reference
    optimizer (PretrainStep.balancer_lr_scale / .balancer_opt when the model runs on the flat engine)."""
    if not _is_main():
        return None
    os.makedirs(output_dir, exist_ok=True)
    to_save = {"model": model.state_dict(),
               "optimizer": optimizer_state_dict(optimizer, model, loss_balancer, balancer_lr_scale, balancer_optimizer), "epoch": epoch,
               "scaler": loss_scaler.state_dict() if loss_scaler is not None else {}, "args": args}
    if loss_balancer is not None:
        to_save["loss_balancer"] = loss_balancer.state_dict()
    path = os.path.join(output_dir, "checkpoint-%s.pth" % str(epoch))
    torch.save(to_save, path)
    return path


def latest_checkpoint(output_dir: str) -> Optional[str]:
    """Highest-numbered checkpoint-*.pth (checkpoint.py:107-117)."""
    best, best_path = -1, None
    for f in glob.glob(os.path.join(output_dir, "checkpoint-*.pth")):
        m = re.search(r"checkpoint-(\d+)\.pth$", f)
        if m and int(m.group(1)) > best:
            best, best_path = int(m.group(1)), f
    return best_path


def auto_load_model(output_dir: str, model, optimizer=None, loss_scaler=None, resume: str = "", map_location="cpu",
                    loss_balancer=None, balancer_optimizer=None) -> int:
    """Returns the epoch to start from (0 when nothing was found).  checkpoint.py:103-132."""
    path = resume or (latest_checkpoint(output_dir) if output_dir else None)
    if not path:
        return 0
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["model"], strict=True)
    if loss_balancer is not None and "loss_balancer" in ckpt:
        loss_balancer.load_state_dict(ckpt["loss_balancer"])
    if optimizer is not None and "optimizer" in ckpt and "epoch" in ckpt:
        load_optimizer_state_dict(optimizer, model, ckpt["optimizer"], balancer_optimizer)
        if loss_scaler is not None and ckpt.get("scaler"):
            loss_scaler.load_state_dict(ckpt["scaler"])
        return int(ckpt["epoch"]) + 1
    if hasattr(optimizer, "refresh_shadow"):
        optimizer.refresh_shadow()
    return 0
