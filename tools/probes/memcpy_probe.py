"""Which torch ops launch the device-to-device memcpys of a bench step?  (rocprofv3 shows ~130 `__amd_rocclr_copyBuffer` per step; only ~11 of
them come from torch ops -- select_backward copies, loss clones: the rest belongs to the library GEMM calls, whose `UserArgs` kernels get their
arguments through a runtime blit.)   python tools/probes/memcpy_probe.py"""
import argparse, collections, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from incomplete_multimodal_fusion_amd.engine import FlatAdamW
from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
args = argparse.Namespace(model="base", domains="s1,s2,dem", input_size=256, fusion_blocks=1, batch=256)
dev = torch.device("cuda", 0)
model = bench.build(args, dev)
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
step = PretrainStep(model, opt, 384, autocast=True, contra="dino")
x = bench.synthetic_tiles(args, 256, 256, dev, 1)
for _ in range(3): step(x)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(x); torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU: continue
    ks = getattr(e, "kernels", [])
    for k in ks:
        if "emcpy" in k.name or "copyBuffer" in k.name:
            chain, q = [], e
            while q is not None:
                chain.append(q.name); q = q.cpu_parent
            chain = [c for c in reversed(chain) if not c.startswith("ProfilerStep")]
            key = " > ".join(chain[:4])[:200]
            cnt[key] += 1; dur[key] += k.duration
for k, n in cnt.most_common(25):
    print("%4d x %8.1f us  %s" % (n, dur[k], k))
