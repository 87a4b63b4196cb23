#!/usr/bin/env python
"""Input staging (row f3) on the GPU box: kernel time for one batch of raw tiles, and the PCIe-inclusive rate of the pinned
ring (pageable host batch -> pinned slot -> async H2D -> staging kernels), which is what a training loop overlaps with the
step.  Raw batch: s1 fp32 (B,1,n,n), s2 uint8 (B,3,n,n), dem fp32 (B,1,n,n), n = 256*factor."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--factor", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from incomplete_multimodal_fusion_amd import staging
    B, n = args.batch, 256 * args.factor
    g = np.random.default_rng(0)
    raw = {'s1': g.gamma(2.0, 0.1, size=(B, 1, n, n)).astype(np.float32),
           's2': g.integers(0, 256, size=(B, 3, n, n), dtype=np.uint8),
           'dem': g.normal(5.0, 7.0, size=(B, 1, n, n)).astype(np.float32)}
    nbytes = sum(v.nbytes for v in raw.values())
    dev = {d: torch.from_numpy(v).cuda() for d, v in raw.items()}
    outs = {d: torch.empty(B, v.shape[1], 256, 256, device="cuda") for d, v in raw.items()}

    def kernels():
        for d, r in staging.DFC2023.items():
            staging.stage_tiles(dev[d], r['kind'], 256, r['mean'], r['std'], out=outs[d])
    for _ in range(3):
        kernels()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        kernels()
    b.record()
    torch.cuda.synchronize()
    k_ms = a.elapsed_time(b) / args.iters
    out_bytes = sum(o.numel() * 4 for o in outs.values())
    print("staging kernels: %.3f ms per batch of %d (%.0f GB/s raw+out)" % (k_ms, B, (nbytes + out_bytes) / k_ms / 1e6))

    st = staging.TileStager("cuda:0", image_size=256, slots=2)
    st.submit(raw)
    for _ in range(2):
        st.get(); st.submit(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        x = st.get()
        st.submit(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print("pinned ring, PCIe inclusive: %.2f ms per batch = %.0f samples/s (%.1f GB/s host->device, %.2f MB raw per sample)"
          % (dt * 1e3, B / dt, nbytes / dt / 1e9, nbytes / B / 1e6))


if __name__ == "__main__":
    main()
