#!/usr/bin/env python
"""BASELINE config 5's shape on ONE GPU (the config itself is an 8-GPU run): ViT-Large (D1024 / L24 / 8 heads), 4 modalities
(s1 2ch, s2 4ch, dem, dnw class map), 256x256 tiles, N = 512 of 1024 tokens kept, per-sample masks with modality dropout,
hard-negative contrastive head, bf16, flat AdamW engine.  `--quadruplet` runs the reference's own 4-modality model (no fusion
blocks, task losses only) instead of the fusion-token extension.  Prints one line: samples/s and ms/step.
    python tools/bench_config5.py [--batch 64] [--steps 8] [--warmup 4] [--quadruplet]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="large")
    ap.add_argument("--quadruplet", action="store_true")
    ap.add_argument("--tunable", type=int, default=1)
    a = ap.parse_args()
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    dev = torch.device("cuda", 0)
    if a.tunable:
        import torch.cuda.tunable as tun
        tun.enable(True); tun.tuning_enable(True)
        tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
        tun.set_filename(os.path.join("/tmp", "mmae_tunableop_c5.csv"))
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)
    torch.manual_seed(0)
    doms = ("s1", "s2", "dem", "dnw")
    model = get_model(a.model, in_domains=doms, input_size=256, fusion_blocks=not a.quadruplet).to(dev).train()
    model.per_sample_masks = True
    opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 512, autocast=True, sample_tasks_uniformly=True,
                        contra="none" if a.quadruplet else "hardneg", clip_grad=1.0)
    g = torch.Generator(device="cpu").manual_seed(1)
    B = a.batch
    x = {"s1": torch.randn(B, 2, 256, 256, generator=g).to(dev), "s2": torch.randn(B, 4, 256, 256, generator=g).to(dev),
         "dem": torch.randn(B, 1, 256, 256, generator=g).to(dev), "dnw": torch.randint(0, 9, (B, 256, 256), generator=g).to(dev)}
    for _ in range(a.warmup):
        out = step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("config-5 shape, %s, ViT-%s 4-mod 256x256 N=512 per-sample masks + dropout, B=%d, %.1f M params: %.1f samples/s, %.1f ms/step, "
          "loss %.4f, peak mem %.1f GB" % ("multimae_quadruplet (no fusion blocks)" if a.quadruplet else "fusion-token model + hard-negative head",
                                          a.model, B, n_params / 1e6, B / dt, dt * 1e3, float(out["loss"]),
                                          torch.cuda.max_memory_allocated() / 1e9))


if __name__ == "__main__":
    main()
