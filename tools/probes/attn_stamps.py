#!/usr/bin/env python
"""Where does an iteration of the 32x32x16 forward spend its cycles?  Runs the diagnostic build (variant 9: s_memtime stamps
around the end-of-tile barrier and the staging block) on the bench shape and prints per-wave averages.  Never a timing.
Needs the diagnostic library: `make -C incomplete_multimodal_fusion_amd/csrc DIAG=1` (libmmae_hip_diag.so); this script selects it
through MMAE_HIP_LIB before the package is imported."""
import ctypes, os, sys
os.environ.setdefault("MMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "incomplete_multimodal_fusion_amd", "csrc", "libmmae_hip_diag.so"))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from incomplete_multimodal_fusion_amd import _lib, ops
B, H, dh, P = 256, 8, 64, 256
nm = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,128,128").split(",")]
N = sum(nm); S = N + P; I = H * dh
lens = torch.tensor([nm + [P]] * B, dtype=torch.int32); st = torch.zeros_like(lens)
for b in range(B):
    off = 0
    for s_ in range(len(nm)):
        st[b, s_] = b * N + off; off += nm[s_]
    st[b, len(nm)] = B * N + b * P
seg = ops.Segments(st.cuda(), lens.cuda(), S)
qkv = torch.randn(B * S, 3 * I, device="cuda").to(torch.bfloat16)
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
for v in (0, 9):
    for _ in range(3):
        ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=v)
torch.cuda.synchronize()
lib.mmae_debug_mha_stamps(buf)                      # clear
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=9); e1.record(); torch.cuda.synchronize()
lib.mmae_debug_mha_stamps(buf)
loop, bar, stage, iters, pro, waves = [int(buf[i]) for i in range(6)]
print("diagnostic launch %.1f us; waves with work %d, tile iterations %d (%.1f per wave)" % (e0.elapsed_time(e1) * 1e3, waves, iters, iters / waves))
print("per wave: prologue %.0f cycles, key loop %.0f cycles = %.0f per iteration" % (pro / waves, loop / waves, loop / iters))
print("prologue: entry -> segment table in registers %.0f, -> Q and first K/V tile arrived (vmcnt 0) %.0f, -> first iteration %.0f"
      % (int(buf[6]) / waves, int(buf[7]) / waves, (pro - int(buf[6]) - int(buf[7])) / waves))
print("per iteration: end-of-tile barrier %.0f cycles (%.0f %%), staging block (vmcnt wait + ds_write) %.0f (%.0f %%), rest %.0f"
      % (bar / iters, 100.0 * bar / loop, stage / iters, 100.0 * stage / loop, (loop - bar - stage) / iters))
