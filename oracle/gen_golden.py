"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own classes (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Each fixture holds data only: random-init weights (seeded), inputs and the
reference's outputs (logits, loss, gradients, greedy / beam token ids).  No
reference source is stored.  The fixtures travel to the GPU box; the reference
does not.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refload  # noqa: E402
from oracle import restatement as R  # noqa: E402  (only for synthetic_captions / pack_rows)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def sd_np(module, prefix="p/"):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="g/"):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def gen_rnn(ref, cell, name, E=64, H=64, V=50, L=5, B=4, seed=7):
    torch.manual_seed(seed)
    cls = ref.rnn.RNN if cell == "gru" else ref.rnn_lstm.RNN
    m = cls(E, H, V, L)
    cap, lens = R.synthetic_captions(B, V, seed=seed, mean=7, std=2, lo=4, hi=9)
    feat = torch.randn(B, E)
    target = nn.utils.rnn.pack_padded_sequence(cap, lens, batch_first=True)[0]   # main.py:145
    logits = m(feat, cap, lens)                                                   # main.py:148
    loss = nn.CrossEntropyLoss()(logits, target)                                  # main.py:149
    loss.backward()
    d = sd_np(m)
    d.update(grads_np(m))
    d.update(feat=feat.numpy(), caption=cap.numpy(), lens=np.array(lens), target=target.numpy(),
             logits=logits.detach().numpy(), loss=np.array(loss.item(), dtype=np.float64))
    m.eval()
    with torch.no_grad():
        d["greedy"] = m.sentence_index(feat).numpy()                              # (B,25)
        d["greedy_b1"] = m.sentence_index(feat[:1]).numpy()                       # (25,) after squeeze
        if cell == "gru":
            for k in (1, 3, 5):
                d[f"qbeam{k}"] = m.sentence_index(feat[:1], beam_size=k).numpy()  # rnn.py:60-108
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, "loss", loss.item(), "logits", logits.shape)


def gen_attn(ref, cell, name, E=32, Fd=48, A=40, H=64, V=50, L=3, B=4, P=49, seed=11, alpha_c=1.0):
    torch.manual_seed(seed)
    mod = ref.rnn_attn if cell == "gru" else ref.rnn_attn_lstm
    m = mod.RNN_Attn(E, Fd, A, H, V, L)
    cap, lens = R.synthetic_captions(B, V, seed=seed, mean=7, std=2, lo=4, hi=9)
    feat = torch.randn(B, Fd, P)
    target = nn.utils.rnn.pack_padded_sequence(cap, lens, batch_first=True)[0]
    with _refload.cpu_cuda():
        logits, alphas = m(feat, cap, lens)                                       # main_attn.py:129
        loss = nn.CrossEntropyLoss()(logits, target)
        loss = loss + alpha_c * ((1. - alphas.sum(dim=1)) ** 2).mean()            # main_attn.py:131
        loss.backward()
        d = sd_np(m)
        d.update(grads_np(m))
        d.update(feat=feat.numpy(), caption=cap.numpy(), lens=np.array(lens), target=target.numpy(),
                 logits=logits.detach().numpy(), alphas=alphas.detach().numpy(),
                 loss=np.array(loss.item(), dtype=np.float64), alpha_c=np.array(alpha_c))
        m.eval()
        vocab = lambda w: {"<pad>": 0, "<start>": 1, "<end>": 2, "<unk>": 3}[w]
        with torch.no_grad():
            d["greedy"] = m.sentence_index(feat, vocab).numpy()
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, "loss", loss.item(), "alphas", alphas.shape)


def gen_beam(ref, name, E=32, H=48, V=40, L=2, B=6, seed=5, scale=10.0, boost=0.5):
    """beam_search.py driven by callbacks over the reference's own RNN sub-modules.

    With torch-default random weights the softmax is nearly flat and the shortest
    hypothesis always wins, so the vocab projection is sharpened (x``scale``) and
    <end> gets ``boost`` (SURVEY 8(c) recipe) to obtain varied lengths, including
    images for which no hypothesis completes (beam_search.py returns []).
    """
    torch.manual_seed(seed)
    m = ref.rnn.RNN(E, H, V, L).eval()
    with torch.no_grad():
        m.linear.weight *= scale
        m.linear.bias[2] += boost
    feats = torch.randn(B, E)
    d = sd_np(m)
    d["feat"] = feats.numpy()
    for bw, ml, nh in ((5, 25, 3), (4, 12, 1)):
        seq = np.zeros((B, nh, ml + 1), dtype=np.int64)
        ln = np.zeros((B, nh), dtype=np.int64)
        cost = np.full((B, nh), np.nan, dtype=np.float64)
        for b in range(B):
            f = feats[b:b + 1]

            def init(_X, f=f):
                with torch.no_grad():
                    _, h = m.unit(f.unsqueeze(1), None)
                return h.numpy().astype(np.float32)

            def gen(_X, Y, st):
                n = len(Y)
                with torch.no_grad():
                    h = torch.from_numpy(st).view(n, L, H).transpose(0, 1).contiguous()
                    x = m.embeddings(torch.from_numpy(Y.astype(np.int64))).unsqueeze(1)
                    o, h2 = m.unit(x, h)
                    p = torch.softmax(m.linear(o.squeeze(1)), 1).numpy().astype(np.float32)
                    s = h2.transpose(0, 1).contiguous().view(n, L * H).numpy().astype(np.float32)
                return s, p, [None] * n

            hyp = ref.beam_search.beam_search(init, gen, [0], 1, 2, beam_width=bw,
                                              num_hypotheses=nh, max_length=ml)
            for i, hp in enumerate(hyp):
                v = hp.to_sequence_of_values()
                seq[b, i, :len(v)] = v
                ln[b, i] = len(v)
                cost[b, i] = hp.cum_cost
        d[f"bw{bw}_seq"], d[f"bw{bw}_len"], d[f"bw{bw}_cost"] = seq, ln, cost
        d[f"bw{bw}_maxlen"] = np.array(ml)
        print(name, "bw", bw, "lens", ln.tolist())
    np.savez_compressed(os.path.join(OUT, name), **d)


def gen_bleu(ref, name):
    rng = np.random.RandomState(3)
    words = ["a", "man", "dog", "on", "the", "beach", "with", "red", "ball", "sits", "runs", "cat", "two"]
    gts, res = {}, {}
    for i in range(12):
        refs = [" ".join(rng.choice(words, size=rng.randint(5, 10))) for _ in range(rng.randint(1, 4))]
        hyp = refs[0].split()
        for j in range(len(hyp)):
            if rng.rand() < 0.3:
                hyp[j] = str(rng.choice(words))
        if rng.rand() < 0.5:
            hyp = hyp[:-1]
        gts[str(i)] = refs
        res[str(i)] = [" ".join(hyp)]
    score, _ = ref.metrics.Bleu(4).compute_score(gts, res)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump({"gts": gts, "res": res, "bleu": list(map(float, score))}, f, indent=1)
    print(name, score)


def gen_eval(ref, name):
    """Inputs + outputs of the reference's evaluate() (evaluation_metrics.py:662) and of its three scorers."""
    rng = np.random.RandomState(5)
    words = ["a", "man", "dog", "on", "the", "beach", "with", "red", "ball", "sits", "runs", "cat", "two", "<unk>"]
    cases = {}
    for case, nimg in (("batch24", 24), ("single", 1), ("pair", 2)):
        target, predicted = [], []
        for i in range(nimg):
            refs = [[str(w) for w in rng.choice(words, size=rng.randint(3, 12))] for _ in range(rng.randint(1, 5))]
            hyp = list(refs[rng.randint(len(refs))])
            for j in range(len(hyp)):
                if rng.rand() < 0.35:
                    hyp[j] = str(rng.choice(words))
            if rng.rand() < 0.4:
                hyp = hyp[:rng.randint(1, len(hyp) + 1)]
            if rng.rand() < 0.2:
                hyp = hyp + [str(w) for w in rng.choice(words, size=rng.randint(1, 6))]
            if case == "batch24" and i == 7:
                hyp = []                                   # a caption that starts with <end>
            if case == "batch24" and i == 11:
                hyp = list(refs[0])                        # exact copy of a reference
            target.append(refs)
            predicted.append(hyp)
        scores = ref.metrics.evaluate(target, predicted)
        gts = {i: [" ".join(s) for s in target[i]] for i in range(nimg)}
        res = {i: [" ".join(predicted[i])] for i in range(nimg)}
        _, bleu_img = ref.metrics.Bleu(4).compute_score(gts, res)
        _, cider_img = ref.metrics.Cider().compute_score(gts, res)
        _, rouge_img = ref.metrics.Rouge().compute_score(gts, res)
        cases[case] = {"target": target, "predicted": predicted, "scores": {k: float(v) for k, v in scores.items()},
                       "bleu_per_image": [list(map(float, b)) for b in bleu_img],
                       "cider_per_image": list(map(float, cider_img)), "rouge_per_image": list(map(float, rouge_img))}
        print(name, case, scores)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(cases, f)


def main():
    assert _refload.available(), "reference not present"
    if sys.argv[1:] == ["eval"]:
        return gen_eval(_refload.load_reference(), "eval_small.json")
    os.makedirs(OUT, exist_ok=True)
    ref = _refload.load_reference()
    gen_rnn(ref, "gru", "gru_small.npz")
    gen_rnn(ref, "lstm", "lstm_small.npz")
    gen_attn(ref, "gru", "attn_gru_small.npz")
    gen_attn(ref, "lstm", "attn_lstm_small.npz")
    gen_beam(ref, "beam_small.npz")
    gen_bleu(ref, "bleu_small.json")
    gen_eval(ref, "eval_small.json")


if __name__ == "__main__":
    main()
