"""N > 1 path on CPU: two gloo ranks run the bucketed, hook-driven gradient all-reduce (incomplete_multimodal_fusion_amd/dp.py)
on a small torch module and must end with the single-process gradient of the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 32)
        self.unused = torch.nn.Parameter(torch.ones(7))      # never receives a gradient (like return_tokens in the model)
        self.c = torch.nn.Linear(32, 4)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from incomplete_multimodal_fusion_amd import dp
    assert dp.init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = Net()
    red = dp.GradAllReducer(net.parameters(), bucket_bytes=3000)      # several buckets
    assert len(red.buckets) >= 3
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    res = []
    for step in range(3):                                              # step 0 detects the unused parameter
        net.zero_grad(set_to_none=True)
        red.prepare()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        red.finish()
        res.append({n: p.grad.tolist() for n, p in net.named_parameters() if p.grad is not None})
    unused = [n for n, p in net.named_parameters() if any(p is u for u in red.unused_parameters())]
    launched_early = sum(1 for b in red.buckets if b.expected > 0)
    q.put((rank, res, unused, launched_early))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_bucketed_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = Net()
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    ((net(X) - Y) ** 2).mean().backward()                             # mean over the global batch == average of rank means
    for rank, res, unused, launched in out:
        assert unused == ["unused"]
        for step_grads in res:
            for n, p in net.named_parameters():
                if n == "unused":
                    continue
                assert torch.allclose(torch.tensor(step_grads[n]), p.grad, atol=1e-6), (rank, n)
