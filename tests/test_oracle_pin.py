"""Live pin of the oracle against the reference itself (authoring container only).

Skipped wherever /root/reference is absent (e.g. the GPU box); the committed
golden vectors (tests/test_oracle_golden.py) carry the pin there.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import _refload
from oracle import restatement as R

pytestmark = pytest.mark.skipif(not _refload.available(), reason="reference not present")


@pytest.fixture(scope="module")
def ref():
    return _refload.load_reference()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gru_forward_and_greedy_random_shapes(ref, seed):
    rng = np.random.RandomState(seed)
    E, H, V, L, B = int(rng.choice([16, 32])), int(rng.choice([16, 48])), 30, int(rng.randint(1, 4)), int(rng.randint(1, 6))
    torch.manual_seed(seed)
    m = ref.rnn.RNN(E, H, V, L).eval()
    params = {k: v.detach() for k, v in m.state_dict().items()}
    cap, lens = R.synthetic_captions(B, V, seed=seed, mean=6, std=2, lo=3, hi=9)
    feat = torch.randn(B, E)
    with torch.no_grad():
        np.testing.assert_allclose(R.rnn_forward(params, feat, cap, lens).numpy(),
                                   m(feat, cap, lens).numpy(), atol=2e-5)
        assert torch.equal(R.rnn_greedy(params, feat), m.sentence_index(feat))


def test_attn_gru_forward_random(ref):
    torch.manual_seed(3)
    m = ref.rnn_attn.RNN_Attn(16, 24, 20, 32, 30, 2).eval()
    params = {k: v.detach() for k, v in m.state_dict().items()}
    cap, lens = R.synthetic_captions(3, 30, seed=3, mean=6, std=2, lo=3, hi=9)
    feat = torch.randn(3, 24, 49)
    with torch.no_grad(), _refload.cpu_cuda():
        lg, al = m(feat, cap, lens)
    lo, ao = R.attn_forward(params, feat, cap, lens)
    np.testing.assert_allclose(lo.detach().numpy(), lg.numpy(), atol=2e-5)
    np.testing.assert_allclose(ao.detach().numpy(), al.numpy(), atol=2e-5)
