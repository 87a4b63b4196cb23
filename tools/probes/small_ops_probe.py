#!/usr/bin/env python
"""Which Python lines launch the step's small ATen kernels (adds, copies, fills)?  One profiled bench step (torch.profiler, with stacks);
prints, per ATen op name, the launch count and the innermost frames of this repository that issued it -- for backward ops the autograd
node that ran them (the `evaluate_function` parent)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd.engine import FlatAdamW  # noqa: E402
from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(x)
    torch.cuda.synchronize()
want = ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::contiguous", "aten::clone", "aten::index_select", "aten::sum", "aten::cat")
by = collections.defaultdict(collections.Counter)
events = prof.events()
for e in events:
    if e.name not in want:
        continue
    frames = [f for f in (e.stack or []) if "incomplete_multimodal_fusion_amd" in f or "bench" in f]
    where = " < ".join(f.split("incomplete_multimodal_fusion_amd/")[-1] for f in frames[:3]) if frames else "(autograd engine / no repo frame)"
    p = e.cpu_parent
    node = ""
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function") or "Backward" in p.name:
            node = p.name.replace("autograd::engine::evaluate_function: ", "")
            break
        p = p.cpu_parent
    shapes = str(e.input_shapes)[:60]
    by[e.name][(where[:150], node[:40], shapes)] += 1
for name in want:
    tot = sum(by[name].values())
    if not tot:
        continue
    print("\n%s: %d" % (name, tot))
    for (where, node, shapes), c in by[name].most_common(30):
        print("   %4d  %-150s %-40s %s" % (c, where, node, shapes))
