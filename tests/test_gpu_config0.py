"""BASELINE configs[0] (main.py GRU decoder, ResNet-101 encoder, emb = 512, bs = 8, 32 synthetic 224x224 images + random
captions): the reference runs it on the CPU as a plumbing check.  The MI355X build has no CPU path by design, so its
rendition is the SAME model in the fp32 kernels on the GPU: 32 images in 4 minibatches of 8 through the reference's loop
(main.py:136-152: zero_grad, encoder in train mode, decoder, CrossEntropyLoss, backward, SGD(lr, momentum) step), every
step's loss against the CPU oracle running the same loop (SURVEY 8(d) "Config 1 ... loss equality vs restatement"), and
the trained weights after the four updates."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config0_four_sgd_steps_match_oracle():
    from oracle import restatement as R
    from showtell_amd import optim
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    E = H = 512
    L, V, B, NB = 5, 10000, 8, 4
    lr, mom = 0.01, 0.9                                         # main.py:48-51 defaults
    enc = R.init_encoder_params(101, E, seed=1)
    dec = R.init_decoder_params(E, H, V, L, "gru", seed=1)
    data = []
    for i in range(NB):
        cap, lens = R.synthetic_captions(B, V, seed=1 + i)
        img = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1 + i))
        data.append((img, cap, lens))

    # ---- HIP path, written like the reference's loop (plain modules + optimizer, no Trainer) -----------------------
    cnn = ResNet(101, E); cnn.load_state_dict(enc)
    rnn = RNN(E, H, V, L); rnn.load_state_dict(dec)
    params = list(rnn.parameters()) + list(cnn.linear_secondlast_layer.parameters()) + list(cnn.last_layer.parameters())   # main.py:96
    opt = optim.SGD(params, lr=lr, momentum=mom, shadow_dtype=None)     # constructed BEFORE .cuda(), as main.py does
    cnn, rnn = cnn.cuda().train(), rnn.cuda().train()
    crit = torch.nn.CrossEntropyLoss()
    got = []
    for img, cap, lens in data:
        opt.zero_grad()
        logits = rnn(cnn(img.cuda()), cap.cuda(), lens)                 # main.py:147-148
        target = torch.nn.utils.rnn.pack_padded_sequence(cap.cuda(), lens, batch_first=True)[0]   # main.py:145
        loss = crit(logits, target)
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    torch.cuda.synchronize()

    # ---- oracle: the same loop on the CPU ---------------------------------------------------------------------------
    po = {k: v.clone() for k, v in enc.items()}
    head_keys = ("linear_secondlast_layer.weight", "linear_secondlast_layer.bias", "last_layer.weight", "last_layer.bias")
    for k in head_keys:
        po[k].requires_grad_(True)
    do = {k: v.clone().requires_grad_(True) for k, v in dec.items()}
    bufs, ref = {}, []
    for img, cap, lens in data:
        loss, _, _ = R.gru_train_loss(do, R.encoder_forward(po, img, 101, train=True), cap, lens)
        loss.backward()
        with torch.no_grad():
            for name, t in list(do.items()) + [(k, po[k]) for k in head_keys]:
                bufs[name] = R.sgd_momentum_step(t, t.grad, bufs.get(name), lr, mom)
                t.grad = None
        ref.append(float(loss.detach()))

    for a, b in zip(got, ref):
        assert abs(a - b) < 1e-3 * max(1.0, abs(b)), (got, ref)
    # trained weights after four updates (fp32 kernels vs fp32 oracle)
    sd = rnn.state_dict()
    for k in ("linear.weight", "unit.weight_hh_l4", "unit.weight_ih_l0", "embeddings.weight"):
        d = (sd[k].float().cpu() - do[k].detach()).abs().max().item()
        assert d < 2e-4 * max(1.0, do[k].detach().abs().max().item()), (k, d)
    d = (cnn.linear_secondlast_layer.weight.detach().cpu() - po["linear_secondlast_layer.weight"].detach()).abs().max().item()
    assert d < 2e-4, d
    # the running buffers of the frozen backbone moved like nn.BatchNorm2d's in train mode (main.py:125)
    esd = cnn.state_dict()
    assert int(esd["model.1.num_batches_tracked"]) == NB
    for k in ("model.1.running_var", "model.6.22.bn3.running_mean", "last_layer.running_var"):
        r = po[k].detach()
        assert ((esd[k].float().cpu() - r).abs().max() / (r.abs().max() + 1e-6)).item() < 2e-3, k
