"""CPU checks of the parity harness itself (tests/parity.py): the chunked evaluation of the oracle equals the batched one."""
import torch

from tests import parity
from tests.test_cabi_symbols import build_model


def test_chunked_oracle_equals_batched_oracle():
    torch.manual_seed(3)
    cfg = dict(dim_tokens=64, depth=2, dim_head=32, heads=2, image_size=64, patch_size=16, decoder_dim=32, decoder_depth=1,
               decoder_heads=2)
    channels = (("s1", 1), ("s2", 3), ("dem", 1))
    model = build_model(cfg, channels)
    B, P, N = 6, 16, 20
    x = {d: torch.randn(B, c, 64, 64) for d, c in channels}
    masks = {}
    for d, k in (("s1", 9), ("s2", 4), ("dem", 7)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = parity.oracle_step(state, x, masks, N, cfg["heads"], cfg["decoder_heads"])
    got = parity.chunked_oracle(state, x, masks, N, cfg["heads"], cfg["decoder_heads"], chunk=4)      # chunks of 4 and 2
    assert set(ref) == set(got)
    for k in ref:
        err = float((ref[k] - got[k]).abs().max()) / max(float(ref[k].abs().max()), 1e-6)
        assert err < 2e-5, (k, err)
