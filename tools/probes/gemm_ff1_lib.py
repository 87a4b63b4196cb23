#!/usr/bin/env python
"""The library GEMM (hipBLASLt through torch, TunableOp on) at the FF1 shape in isolation: the number the own-GEMM probe
(gemm_ff1_probe.hip) is gated against."""
import torch, time
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30); tun.set_filename('/tmp/gemm_ff1_tun.csv')
M, N, K = 163840, 4096, 768
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
for _ in range(5):
    c = torch.nn.functional.linear(a, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    c = torch.nn.functional.linear(a, w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("library FF1 GEMM  M %d N %d K %d : %.1f us per launch = %.0f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
