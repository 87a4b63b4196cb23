#!/usr/bin/env python
"""Where does a tile iteration of the sample-head forward (csrc/mha_sh.hip) spend its cycles?  Runs the stamped diagnostic
build (variant 8) on the bench shape and prints per-wave averages for the two wave roles.  Never a timing.
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
buf = (ctypes.c_ulonglong * 32)()
for v in (5, 8):
    for _ in range(3):
        ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=v)
torch.cuda.synchronize()
lib.mmae_debug_sh_stamps(buf)                      # clear
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=8); e1.record(); torch.cuda.synchronize()
lib.mmae_debug_sh_stamps(buf)
print("diagnostic launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
names = ["vmcnt wait", "barrier", "ring issue + tile record", "slot switch", "QK^T", "softmax", "PV", "finish"]
for role, o in (("global", 0), ("local", 16)):
    v = [int(buf[o + i]) for i in range(16)]
    waves, iters, units = max(v[11], 1), max(v[10], 1), max(v[9], 1)
    print("%s waves: %d, iterations per wave %.1f, units per wave %.1f, loop %.0f cycles per wave = %.0f per iteration"
          % (role, waves, iters / waves, units / waves, v[8] / waves, v[8] / iters))
    print("   per iteration: " + ", ".join("%s %.0f" % (names[i], v[i] / iters) for i in range(4)) + ", finish %.0f" % (v[7] / iters))
    print("   per unit     : " + ", ".join("%s %.0f" % (names[i], v[i] / units) for i in (4, 5, 6)))
    print("   slot switch  : wait for staged Q %.0f, LDS reads + conversion %.0f, next-target scan %.0f, rest (Q DMA issue) %.0f  [cycles per iteration]"
          % (v[12] / iters, v[13] / iters, v[14] / iters, (v[3] - v[12] - v[13] - v[14]) / iters))
