"""Probe: split-K weight gradient in both orientations -- dW = g^T x (n_out x n_in, what ops._wgrad runs) against dW^T = x^T g
(n_in x n_out) -- at the bench's shapes, TunableOp on.  If the transposed problem runs faster, the gradient can be produced
transposed and transposed back in the (tiny) fp32 sum."""
import torch, time
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
tun.set_filename("/tmp/probe_orient.csv")
dev = "cuda"
# (n_out, n_in, rows, S): the step's big weight gradients at B = 256 with the split ops._split_k picks
shapes = [(4096, 768, 163840, 4), (768, 2048, 163840, 8), (1536, 768, 163840, 8), (768, 512, 163840, 32),
          (4096, 768, 65536, 4), (768, 2048, 65536, 8), (1024, 768, 164096, 16)]
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for no, ni, rows, S in shapes:
    g = torch.randn(rows, no, device=dev, dtype=torch.bfloat16); x = torch.randn(rows, ni, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * no * ni * rows
    gs = g.view(S, rows // S, no); xs = x.view(S, rows // S, ni)
    ta = timeit(lambda: torch.bmm(gs.transpose(1, 2), xs))
    tb = timeit(lambda: torch.bmm(xs.transpose(1, 2), gs))
    print("n_out %5d n_in %5d rows %6d S %2d | g^T x %7.1f us %5.0f TF/s | x^T g %7.1f us %5.0f TF/s" %
          (no, ni, rows, S, ta * 1e6, fl / ta / 1e12, tb * 1e6, fl / tb / 1e12), flush=True)
