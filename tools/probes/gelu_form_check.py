"""CPU check (numpy, float32 emulation) of the GELU form of the FF1 + GEGLU GEMM epilogue (csrc/gemm.hip gm_gelu, GM_GELU_FAST 1):
x * sigma(x (c0 + c1 x^2 + c2 x^4)) with the argument clamped to |x| <= 9, against the exact x Phi(x) over EVERY bf16 input --
the epilogue's inputs are bf16 values of h -- and after rounding both to bf16.   python tools/probes/gelu_form_check.py [--fit]"""
import sys

import numpy as np
import torch
from scipy.special import erf

C = np.array([1.59501577e+00, 7.40112920e-02, -7.03033577e-04])


def gelu_fast32(u, c=C):
    L = np.float32(1.4426950408889634)
    k = (-L * c.astype(np.float32)).astype(np.float32)
    u = u.astype(np.float32)
    uc = np.clip(u, np.float32(-9), np.float32(9))
    u2 = uc * uc
    p = (k[2] * u2 + k[1]) * u2 + k[0]
    return u * (np.float32(1) / (np.float32(1) + np.exp2(uc * p)))


def main():
    if "--fit" in sys.argv:
        from scipy.optimize import minimize
        x = np.linspace(-9, 9, 600001)
        g = x * 0.5 * (1 + erf(x / np.sqrt(2)))

        def obj(c):
            t = x * (c[0] + c[1] * x * x + c[2] * x ** 4)
            return np.abs(x / (1 + np.exp(-t)) - g).max()
        best = minimize(obj, C, method="Nelder-Mead", options=dict(xatol=1e-11, fatol=1e-13, maxiter=40000))
        print("fitted coefficients", repr(best.x), "max |err| on [-9, 9]: %.3e" % obj(best.x))
    allb = torch.arange(-2 ** 15, 2 ** 15, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
    allb = allb[torch.isfinite(allb)].numpy().astype(np.float64)
    exact = allb * 0.5 * (1 + erf(allb / np.sqrt(2)))
    fast = gelu_fast32(allb).astype(np.float64)
    assert np.isfinite(fast).all()
    # beyond the clamp the factor is the constant sigma(t(+-9)): 1 for x > 9 (exact), 2.4e-12 for x < -9 (exact: < 1e-19) -- an error
    # of 2.4e-12 |x|, below any bf16 resolution of the activations; the statistics below are over |x| <= 64
    tail = allb < -9
    print("x < -9: fast / x = %.3e (exact Phi < 1.2e-19)" % np.abs(fast[tail] / allb[tail]).max())
    keep = np.abs(allb) <= 64
    allb, exact, fast = allb[keep], exact[keep], fast[keep]
    print("every bf16 input with |x| <= 64 (%d values): max |fast - exact| = %.3e" % (len(allb), np.abs(fast - exact).max()))
    eb = torch.tensor(exact).to(torch.bfloat16).double().numpy()
    fb = torch.tensor(fast).to(torch.bfloat16).double().numpy()
    diff = eb != fb
    sig = np.abs(exact) > 0.02
    rel = np.abs(eb - fb)[diff & sig] / np.abs(eb)[diff & sig]
    print("bf16-rounded results differ for %d inputs (%.2f %%); among the %d inputs with |gelu| > 0.02: %d, largest relative step %.2e "
          "(one bf16 ulp = 3.9e-3 .. 7.8e-3)" % (diff.sum(), 100 * diff.mean(), sig.sum(), (diff & sig).sum(), rel.max() if len(rel) else 0.0))


if __name__ == "__main__":
    main()
