#!/usr/bin/env python
"""Host-side cost of one pretraining step (the Python that enqueues it): cProfile over a few steps of a SMALL configuration (ViT-Small,
128 x 128, where the step is host-bound), main thread = forward + optimizer, autograd thread = backward (profiled separately through
threading.setprofile is not possible for the engine's C++ threads, so the backward is run with the engine's worker on the main thread:
torch.autograd.set_multithreading_enabled(False))."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd.engine import FlatAdamW  # noqa: E402
from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = get_model("small", in_domains=("s1", "s2", "dem"), input_size=128, patch_size=16, decoder_dim=256, decoder_depth=2,
                  decoder_num_heads=8, fusion_blocks=True).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
step = PretrainStep(model, opt, 64, autocast=True)
x = {"s1": torch.randn(B, 1, 128, 128, device=dev), "s2": torch.randn(B, 3, 128, 128, device=dev), "dem": torch.randn(B, 1, 128, 128, device=dev)}
for _ in range(4):
    step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager: host enqueue %.1f ms/step, region %.1f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
torch.autograd.set_multithreading_enabled(False)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
