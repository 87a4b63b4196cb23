"""Child process of tests/test_gpu_dp_engine.py: the pretraining step with the RCCL gradient reducer captured into one hipGraph.

One rank on the one GPU of the box (MMAE_DIST_SINGLE_RANK=1: backend "nccl" = RCCL, every bucket all-reduce really issued), two copies of
the same model: one stepped eagerly, one captured after two warm-up steps and replayed.  Prints one JSON line with the comparison.
A separate process so that a capture that hangs inside the collective library costs the test its timeout, not the session."""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from incomplete_multimodal_fusion_amd import dp
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    from tests.test_gpu_graph import _setup
    assert dp.init_distributed() and torch.distributed.get_backend() == "nccl" and torch.distributed.get_world_size() == 1
    dev = torch.device("cuda", 0)
    base, x, masks = _setup()

    def build():
        model = copy.deepcopy(base).to(dev).train()
        opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
        red = dp.GradAllReducer(None, bucket_bytes=1 << 20, engine=opt)
        return opt, red, PretrainStep(model, opt, 96, grad_reducer=red, check_finite=True)
    opt_e, red_e, step_e = build()
    opt_g, red_g, step_g = build()
    assert red_e.collective and len(red_e.buckets) > 2
    ok0, why0 = red_g.capturable()                    # before any step: the unused-parameter set is not agreed yet
    losses_e = [float(step_e(x, task_masks=masks)["loss"]) for _ in range(6)]
    # many collectives still in flight when capture() is entered: nothing below waits for them -- capture()'s own hand-off
    # (GradAllReducer.quiesce: handles complete, barrier, watchdog retirement read from c10d's flight recorder) has to
    junk = torch.randn(1 << 22, device=dev)
    inflight = [torch.distributed.all_reduce(junk, async_op=True) for _ in range(64)]
    step_g.capture(x, masks, warmup=2)
    assert all(w.is_completed() for w in inflight)
    losses_g = [float(step_g.replay()["loss"]) for _ in range(4)]
    torch.cuda.synchronize()
    print(json.dumps({
        "quiesce": red_g.last_quiesce, "exposed_after_capture": red_g.exposed_ms(), "capturable_before_first_step": ok0, "why": why0, "buckets": len(red_g.buckets), "sent_buckets": red_e.stats["buckets"],
        "losses_eager": losses_e[2:], "losses_replay": losses_g,
        "master_equal": bool(torch.equal(opt_g.master, opt_e.master)), "exp_avg_equal": bool(torch.equal(opt_g.exp_avg, opt_e.exp_avg)),
        "steps": [opt_e.steps, opt_g.steps]}), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
