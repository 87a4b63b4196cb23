#!/usr/bin/env python
"""Build-container check behind bench.py's cpu_baseline ("the restatement must be within +-15 % of the reference on
this container before its GPU-box number is quoted"): the ACTUAL reference step and the oracle step, same weights, inputs,
masks and thread count, timed alternately so machine load hits both alike.  Needs /root/reference; prints one summary line.
    python tools/cpu_port_vs_reference.py [--batch 8] [--rounds 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mmae_oracle as O, ref_loader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--spelled-out", action="store_true", help="time the oracle's spelled-out LayerNorm / GELU formulas")
    a = ap.parse_args()
    O.set_fused_primitives(not a.spelled_out)          # what bench.py's cpu_baseline leg runs
    ref = ref_loader.load()
    torch.manual_seed(0)
    cfg = dict(dim_tokens=768, depth=12, dim_head=64, heads=8, image_size=256, patch_size=16, decoder_dim=256, decoder_depth=2,
               decoder_heads=8)
    model = ref_loader.build_reference_model(ref, **cfg).train()
    fns = {"s1": ref.cr.MaskedMSELoss(16, 1), "s2": ref.cr.MaskedMSELoss(16, 1), "dem": ref.cr.MaskedL1Loss(16, 1)}
    opt_r = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    p = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("pos_emb") and not k.endswith("beta"))
         for k, v in model.state_dict().items()}
    opt_o = torch.optim.AdamW([t for t in p.values() if t.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    B, P, N = a.batch, 256, 384
    g = torch.Generator().manual_seed(1)
    x = {"s1": torch.randn(B, 1, 256, 256, generator=g), "s2": torch.randn(B, 3, 256, 256, generator=g),
         "dem": torch.randn(B, 1, 256, 256, generator=g)}
    masks = {}
    for d, k in (("s1", 150), ("s2", 130), ("dem", 104)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P, generator=g)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)

    def ref_step():
        out = model(x, task_masks=masks, num_encoded_tokens=N)
        tl = {t: fns[t](out[0][t].float(), x[t], mask=masks[t]) for t in out[0]}
        feats = [c.squeeze() for c in torch.chunk(out[2], 4, dim=1)]
        lc = sum(ref.cr.dino_loss_func(r.squeeze(), f) for r, f in zip(out[5:], feats))
        loss = sum(tl.values()) + 0.3 * lc
        opt_r.zero_grad(set_to_none=True); loss.backward(); opt_r.step()
        return float(loss)

    def port_step():
        _, (_, _, loss) = O.train_step_loss(p, x, masks, N, 8, 8, 16)
        opt_o.zero_grad(set_to_none=True); loss.backward(); opt_o.step()
        return float(loss)

    tr, to = [], []
    for i in range(a.rounds + 1):
        t0 = time.perf_counter(); lr_ = ref_step(); t1 = time.perf_counter(); lo = port_step(); t2 = time.perf_counter()
        print("round %d: reference %.2f s (loss %.5f)   port %.2f s (loss %.5f)" % (i, t1 - t0, lr_, t2 - t1, lo), flush=True)
        if i:
            tr.append(t1 - t0); to.append(t2 - t1)
    r, o = min(tr), min(to)                            # a shared VM: the quietest round of each is the comparable figure
    ratios = sorted(b / a_ for a_, b in zip(tr, to))
    print("ViT-B 3-mod 256x256 B=%d N=%d fp32, %d threads: reference best %.2f s/step, oracle port best %.2f s/step, "
          "port/reference = %.3f (best/best; median of the per-round ratios %.3f)"
          % (B, N, torch.get_num_threads(), r, o, o / r, ratios[len(ratios) // 2]))


if __name__ == "__main__":
    main()
