#!/usr/bin/env python
"""Where does the host wait for the GPU inside one pretraining step?  Runs the bench step (ViT-B, B = 256, engine optimizer) with
torch.cuda.set_sync_debug_mode("warn") for one step and prints every synchronising call with its Python stack, then the host's
enqueue time per step against the whole region (equal = the host is held back once per step; host << region = it runs ahead)."""
import os
import sys
import time
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd.engine import FlatAdamW  # noqa: E402
from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = get_model("base", in_domains=("s1", "s2", "dem"), input_size=256, patch_size=16, decoder_dim=256, decoder_depth=2,
                  decoder_num_heads=8, fusion_blocks=True).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
step = PretrainStep(model, opt, 384, autocast=True)
x = {"s1": torch.randn(B, 1, 256, 256, device=dev), "s2": torch.randn(B, 3, 256, 256, device=dev), "dem": torch.randn(B, 1, 256, 256, device=dev)}
for _ in range(3):
    step(x)
torch.cuda.synchronize()


def show(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", message, flush=True)
    for fr in traceback.extract_stack()[:-2][-10:]:
        if "incomplete_multimodal_fusion_amd" in fr.filename or "sync_probe" in fr.filename:
            print("    %s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name), flush=True)


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(x)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B %d: host enqueue %.1f ms/step, whole region %.1f ms/step" % (B, (t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n))
