"""Caption scorers and the test_model loop against the outputs of the reference's own scorers
(tests/golden/eval_small.json, written by oracle/gen_golden.py from evaluation/evaluation_metrics.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

with open(os.path.join(ROOT, "tests", "golden", "eval_small.json")) as f:
    GOLD = json.load(f)


@pytest.mark.parametrize("case", sorted(GOLD))
def test_evaluate_matches_reference_scorers(case):
    from showtell_amd import evaluation as E
    g = GOLD[case]
    got = E.evaluate(g["target"], g["predicted"])
    assert sorted(got) == sorted(g["scores"])
    for k, v in g["scores"].items():
        assert got[k] == pytest.approx(v, rel=1e-12, abs=1e-15), k
    gts = {i: [" ".join(s) for s in t] for i, t in enumerate(g["target"])}
    res = {i: [" ".join(p)] for i, p in enumerate(g["predicted"])}
    # the reference's evaluate() reports the LAST image's CIDEr / ROUGE-L (evaluation_metrics.py:706-712 rebinds `score`)
    assert g["scores"]["CIDEr"] == g["cider_per_image"][-1] and g["scores"]["ROUGE_L"] == g["rouge_per_image"][-1]
    mean = E.evaluate(g["target"], g["predicted"], corpus_mean=True)
    assert mean["CIDEr"] == pytest.approx(np.mean(g["cider_per_image"]), rel=1e-12)
    assert mean["ROUGE_L"] == pytest.approx(np.mean(g["rouge_per_image"]), rel=1e-12)
    np.testing.assert_allclose(E.bleu_score(gts, res)[1], g["bleu_per_image"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(E.cider_score(gts, res)[1], g["cider_per_image"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(E.rouge_score(gts, res)[1], g["rouge_per_image"], rtol=1e-12, atol=0)


def test_bleu_matches_the_earlier_fixture():
    from showtell_amd import evaluation as E
    with open(os.path.join(ROOT, "tests", "golden", "bleu_small.json")) as f:
        j = json.load(f)
    np.testing.assert_allclose(E.bleu_score(j["gts"], j["res"])[0], j["bleu"], rtol=1e-12)


def test_evaluate_live_against_reference_when_present():
    from oracle import _refload
    if not _refload.available():
        pytest.skip("reference tree not present (GPU box)")
    from showtell_amd import evaluation as E
    ref = _refload.load_reference()
    rng = np.random.RandomState(17)
    words = ["w%d" % i for i in range(9)]
    target = [[[str(w) for w in rng.choice(words, size=rng.randint(2, 9))] for _ in range(rng.randint(1, 4))] for _ in range(10)]
    predicted = [[str(w) for w in rng.choice(words, size=rng.randint(1, 9))] for _ in range(10)]
    want = ref.metrics.evaluate(target, predicted)
    got = E.evaluate(target, predicted)
    for k in want:
        assert got[k] == pytest.approx(float(want[k]), rel=1e-12, abs=1e-15)


class _Vocab:
    """The slice of vocab_builder.Vocabulary that utils.create_caption_word_format reads."""
    def __init__(self, n):
        self.index_to_word = {0: "<pad>", 1: "<start>", 2: "<end>", 3: "<unk>", **{i: "w%d" % i for i in range(4, n)}}
        self.word_to_index = {w: i for i, w in self.index_to_word.items()}

    def start_token(self):
        return "<start>"

    def end_token(self):
        return "<end>"


@pytest.mark.gpu
def test_test_model_loop(tmp_path):
    """utils.py:147-247 end to end on synthetic batches: checkpoint -> load -> loss + greedy captions -> scores."""
    from showtell_amd import evaluation as E, optim
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    from showtell_amd.train import Trainer, synthetic_batch
    from showtell_amd.utils import create_checkpoint
    V, Edim = 60, 64
    torch.manual_seed(2)
    cnn = ResNet(18, Edim).cuda().eval()
    rnn = RNN(Edim, Edim, V, 2).cuda().eval()
    opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
    params = {"output_dir": str(tmp_path)}
    create_checkpoint(cnn, rnn, opt, 3, 0, [1.0], params)
    batches = []
    for i in range(3):
        img, cap, lens = synthetic_batch(6, V, seed=70 + i, image_size=64, device="cpu")
        batches.append((tuple("im%d_%d.jpg" % (i, j) for j in range(6)), img, cap, lens))
    with pytest.raises(ValueError):
        E.test_model(cnn, rnn, opt, torch.nn.CrossEntropyLoss(), batches, _Vocab(V), params, "model_3", "cpu")
    r = E.test_model(cnn, rnn, opt, torch.nn.CrossEntropyLoss(), batches, _Vocab(V), params, "model_3", "gpu", sub_batch_size=2)
    assert np.isfinite(r["test_loss"]) and abs(r["test_loss"] - np.log(V)) < 1.0
    assert len(r["target"]) == 12 and all(len(v) == 1 for v in r["candidate"].values())
    for k in ("Bleu_1", "Bleu_4", "CIDEr", "ROUGE_L"):
        assert 0.0 <= r[k] <= 10.0
    # the scores are those of evaluate() on the same captions, minibatch by minibatch
    with torch.no_grad():
        _, img, cap, lens = batches[0]
        ids = rnn.sentence_index(cnn(img.cuda())).cpu().numpy()
    from showtell_amd.utils import create_caption_word_format as words
    one = E.evaluate(words(cap.numpy(), _Vocab(V), True), words(ids, _Vocab(V), False))
    assert 0.0 <= one["Bleu_1"] <= 1.0
    assert os.path.isfile(os.path.join(str(tmp_path), "Target_Words_Dict.pickle"))
