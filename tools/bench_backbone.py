#!/usr/bin/env python
"""Throughput of the downstream backbone forward on one GPU: ViTBaseline.forward (encoder taps + necks),
eval mode = all three modalities, every token kept (S = 3P + P = 1024 at 256x256), bf16 autocast, synthetic tiles."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incomplete_multimodal_fusion_amd.multimae import FusionInputAdapter, PatchedInputAdapter, TokenTypes
from incomplete_multimodal_fusion_amd.multimae.multimae_big_imcomplete import ViTBaseline

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--preset", default="tiny", choices=["tiny", "base"])
ap.add_argument("--train", action="store_true", help="training mode: random modality subset, 90 %% keep, fwd+bwd")
a = ap.parse_args()
D, L, h = {"tiny": (192, 12, 3), "base": (768, 12, 8)}[a.preset]
ch = (("s1", 1), ("s2", 3), ("dem", 1))
ia = {d: PatchedInputAdapter(num_channels=c, stride_level=1, patch_size_full=16, image_size=256) for d, c in ch}
ia["fusion"] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=16, image_size=256)
m = ViTBaseline(input_adapters=ia, output_adapters=None, num_fusion_tokens=256,
                return_token_types=(TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION), dim_tokens=D,
                depth=L, dim_head=64, heads=h, in_domains=[c[0] for c in ch], pretrained="/nonexistent").cuda()
m.train(a.train)
x = {d: torch.randn(a.batch, c, 256, 256, device="cuda") for d, c in ch}

def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if a.train:
            feats = m(x)
            sum(f.float().mean() for f in feats).backward()
            m.zero_grad(set_to_none=True)
        else:
            with torch.no_grad():
                m(x)

for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "backbone_%s_samples_per_sec" % ("train_fwd_bwd" if a.train else "eval_forward"),
                  "value": round(a.batch / dt, 1), "ms": round(dt * 1e3, 2), "preset": a.preset, "batch": a.batch,
                  "dtype": "bf16", "tokens_per_sample": 1024 if not a.train else "0.9*(M'*256)+256"}))
