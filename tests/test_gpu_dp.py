"""Data-parallel SEMANTICS on a real training run ("N replicas with averaged grads vs the N-rank run"):
two gloo ranks share the one GPU of the test box, start from DIFFERENT weights (Trainer broadcasts rank 0's), train three
Trainer steps on different minibatches; the final trainable parameters must equal a single-process emulation that computes
both ranks' gradients on the same weights, averages them and applies one SGD step.  fp32 kernels: the only differences are
fp32 atomics' summation order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_rank_training_equals_emulated_data_parallel(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "rank0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_rank.py"), port, str(r), out], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("DP_RANK_OK" in l for l in logs), "\n".join(l[-3000:] for l in logs)
    got = np.load(out)

    # single-process emulation: rank 0's initial weights, gradient = mean of the two ranks' gradients, one update per step
    sys.path.insert(0, HERE)
    import _dp_rank as D
    from showtell_amd.head import linear_bn1d
    cnn, rnn, opt = D.build(seed=50)
    data = [D.batches(0), D.batches(1)]
    losses0 = []
    for i in range(D.STEPS):
        opt.zero_grad()
        acc = torch.zeros_like(opt.flat_grad)
        for r in range(2):
            img, cap, lens = data[r][i]
            opt.zero_grad()
            # each rank's BatchNorm1d uses ITS batch statistics and updates ITS running buffers (DDP default, parallel.py);
            # rank 1's running buffers are its own, so restore rank 0's after emulating rank 1
            keep = {k: v.clone() for k, v in cnn.last_layer.state_dict().items()} if r == 1 else None
            feat = linear_bn1d(cnn.backbone_features(img), cnn.linear_secondlast_layer, cnn.last_layer, True, cnn.compute_dtype)
            loss = rnn.loss(feat, cap, lens)
            loss.backward()
            if keep is not None:
                cnn.last_layer.load_state_dict(keep)
            if r == 0:
                losses0.append(float(loss.detach()))
            acc += opt.flat_grad
        opt.flat_grad.copy_(acc)
        opt.grad_scale = 0.5
        opt.step()
    torch.cuda.synchronize()
    ref = opt.flat.detach().cpu().numpy()
    assert np.allclose(got["losses"], losses0, rtol=1e-4, atol=1e-5), (got["losses"], losses0)
    err = np.abs(got["flat"] - ref).max() / np.abs(ref).max()
    assert err < 1e-5, f"2-rank parameters differ from the emulated data-parallel run: {err:.3e}"
    assert np.allclose(got["rm"], cnn.last_layer.running_mean.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
    # and they did move: three updates of lr 0.05
    init = D.build(seed=50)[2].flat.detach().cpu().numpy()
    assert np.abs(ref - init).max() > 1e-3
