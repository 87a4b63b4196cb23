"""Calibration aid for tests/test_gpu_trajectory.py: the native fp32 trajectory against the oracle's, per tensor -- relative L2 of the total
update, and the max-abs error over the elements whose gradient is a signal in every step (|g| >= FRAC x the tensor's max |g|)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_trajectory as T                                    # noqa: E402


def main():
    state, x, masks = T._setup()
    ref_losses, ref_w, grads = T._oracle_trajectory(state, x, masks, bf16=False, keep_grads=True)
    got_losses, got_w = T._native_trajectory(state, x, masks, autocast=False)
    print("loss err", T._loss_errs(got_losses, ref_losses))
    rows = []
    for k, r in ref_w.items():
        if not r.dtype.is_floating_point or k not in grads[0]:
            continue
        d0 = r - state[k].double()
        if float(d0.abs().max()) == 0:
            continue
        err = got_w[k] - r
        l2 = float(err.norm() / d0.norm())
        out = [k, l2]
        for frac in (1e-1, 1e-2, 1e-3):
            sig = torch.ones_like(r, dtype=torch.bool)
            for g in grads:
                gk = g[k]
                if gk is None or float(gk.abs().max()) == 0:
                    continue
                sig &= gk.abs().double() >= frac * float(gk.abs().max())
            scale = max(float(r.abs().max()), T.LR * T.STEPS)
            out.append(float(err[sig].abs().max()) / scale if bool(sig.any()) else 0.0)
            out.append(float(sig.double().mean()))
        rows.append(out)
    rows.sort(key=lambda r: -r[1])
    print("worst update rel L2:")
    for r in rows[:12]:
        print("  %-60s l2 %.2e | sig>=1e-1: %.2e (%.2f) | >=1e-2: %.2e (%.2f) | >=1e-3: %.2e (%.2f)" % tuple(r))
    for col, name in ((2, "1e-1"), (4, "1e-2"), (6, "1e-3")):
        w = max(rows, key=lambda r: r[col])
        print("worst signal-element err at frac %s: %.2e  %s (share %.2f)" % (name, w[col], w[0], w[col + 1]))
    import statistics
    print("median l2 %.2e  max l2 %.2e  tensors %d" % (statistics.median(r[1] for r in rows), rows[0][1], len(rows)))


if __name__ == "__main__":
    main()
