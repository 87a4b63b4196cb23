#!/usr/bin/env python
"""Full-size training sanity: ViT-B, bf16, fused engine, Dirichlet masks per step, ONE fixed synthetic batch repeated --
the loss must fall steadily and stay finite (run on the GPU box; ~30 s for 150 steps at batch 256)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--lr", type=float, default=3e-4)
    args = ap.parse_args()
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = get_model("base", input_size=256).to(dev).train()
    opt = FlatAdamW(model.parameters(), lr=args.lr, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 384, autocast=True)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = {"s1": torch.randn(args.batch, 1, 256, 256, generator=g).to(dev), "s2": torch.randn(args.batch, 3, 256, 256, generator=g).to(dev),
         "dem": torch.randn(args.batch, 1, 256, 256, generator=g).to(dev)}
    hist = []
    for i in range(args.steps):
        out = step(x)
        if i % 10 == 0 or i == args.steps - 1:
            l = {k: float(v) for k, v in out.items()}
            hist.append(l["loss"])
            print("step %4d  loss %.4f  (s1 %.3f s2 %.3f dem %.3f contra %.3f)  gnorm %.3f" %
                  (i, l["loss"], l["s1_loss"], l["s2_loss"], l["dem_loss"], l["loss_contra"], float(opt.grad_norm())), flush=True)
    assert all(h == h for h in hist), "NaN loss"
    assert hist[-1] < hist[0], "loss did not fall"
    print("ok: %.4f -> %.4f" % (hist[0], hist[-1]))


if __name__ == "__main__":
    main()
