#!/usr/bin/env python
"""Which python lines launch the small ATen kernels of the bench step (fills, casts, adds, copies)?  torch.profiler with stacks
over two steps of the default bench configuration; per kernel-name pattern the launching op + innermost repo frame, by count.
    python tools/probes/glue_probe.py [--batch 256]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    import bench
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    args = argparse.Namespace(model="base", domains="s1,s2,dem", input_size=256, fusion_blocks=1, batch=a.batch)
    dev = torch.device("cuda", 0)
    model = bench.build(args, dev)
    opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 384, autocast=True, contra="dino")
    x = bench.synthetic_tiles(args, a.batch, 256, dev, 1)
    for _ in range(3):
        step(x)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            step(x)
        torch.cuda.synchronize()
    # kernel events carry no stack; link them to the CPU op that launched them through the correlation id
    evs = prof.events()
    small = collections.Counter()
    where = collections.defaultdict(collections.Counter)
    dur = collections.Counter()
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        ks = getattr(e, "kernels", [])
        if not ks:
            continue
        for k in ks:
            name = k.name
            if name.startswith("Cijk") or name.startswith("Custom_Cijk") or "mmae" in name or (name.startswith("_Z") and "at6native" not in name and "colsum" not in name):
                continue
            if not ("at::native" in name or "rocclr" in name or "at6native" in name or "colsum" in name or "splitk" in name):
                continue
            short = name.split("<")[0][-40:] + ("<" + name.split("<")[1][:60] if "<" in name else "")
            # attribution: the chain of enclosing CPU ops (autograd nodes carry the Function's name), outermost first
            chain, q = [], e
            while q is not None:
                chain.append(q.name)
                q = q.cpu_parent
            chain = [c for c in reversed(chain) if not c.startswith("ProfilerStep")]
            site = " > ".join(chain[:3])
            small[short] += 1
            dur[short] += k.duration
            where[short][(e.name, site)] += 1
    for short, n in small.most_common():
        print("%5d x  %7.1f us total  %s" % (n, dur[short], short))
        for (op, site), c in where[short].most_common(8):
            print("        %4d  %-28s %s" % (c, op, site.replace(ROOT + "/", "")))


if __name__ == "__main__":
    main()
