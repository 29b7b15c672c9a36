"""The training step of the reference (main.py:136-152) on MI355X, plus a synthetic-data driver.

    optimizer.zero_grad(); feat = cnn(image); loss = CE(rnn(feat, cap, lens), packed(cap));
    loss.backward(); optimizer.step()

`Trainer.step` performs exactly that arithmetic, ordered so that data-parallel communication
hides behind the frozen backbone:  the previous step's gradient all-reduce and optimizer
update are completed AFTER this step's backbone forward (which reads only frozen weights,
cnn.py:47) and BEFORE its trainable part.  With one GPU the order is irrelevant and the result
is identical to the reference loop; `flush()` applies the last pending update.
"""
import numpy as np
import torch

from .head import linear_bn1d
from .parallel import GradAllReducer


def synthetic_batch(B, V, seed=1, device="cuda", image_size=224, mean=12.5, std=2.5, lo=6, hi=25):
    """COCO-shaped synthetic minibatch in the layout utils.create_batch produces (utils.py:61-77):
    images (B,3,224,224) fp32, captions LongTensor (B,Tmax) zero padded, lengths sorted descending,
    tokens [<start>=1] + randint(4,V) + [<end>=2] (vocab_builder.py:66-69).  SURVEY 8(d)."""
    rng = np.random.RandomState(seed)
    lens = np.clip(np.rint(rng.normal(mean, std, size=B)), lo, hi).astype(np.int64)
    lens = np.sort(lens)[::-1].copy()
    cap = np.zeros((B, int(lens[0])), dtype=np.int64)
    for b, l in enumerate(lens):
        cap[b, 0] = 1
        cap[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
        cap[b, l - 1] = 2
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = torch.randn(B, 3, image_size, image_size, generator=g)
    return images.to(device), torch.from_numpy(cap).to(device), [int(l) for l in lens]


class Trainer:
    def __init__(self, cnn, rnn, optimizer, world_size=1):
        self.cnn, self.rnn, self.opt = cnn, rnn, optimizer
        self.reducer = GradAllReducer(world_size)
        self.pending = False

    def trainable_params(cnn, rnn):
        """main.py:96: rnn.parameters() + cnn.linear_secondlast_layer + cnn.last_layer."""
        return list(rnn.parameters()) + list(cnn.linear_secondlast_layer.parameters()) + list(cnn.last_layer.parameters())
    trainable_params = staticmethod(trainable_params)

    def _apply_pending(self):
        if self.pending:
            self.opt.grad_scale = self.reducer.finish()
            self.opt.step()
            self.pending = False

    def step(self, image, caption, caption_len):
        cnn, rnn = self.cnn, self.rnn
        pooled = cnn.backbone_features(image)          # frozen, detached (cnn.py:46-47): overlaps the all-reduce
        self._apply_pending()                          # previous step's optimizer.step() (main.py:152)
        self.opt.zero_grad()                           # main.py:146
        feat = linear_bn1d(pooled, cnn.linear_secondlast_layer, cnn.last_layer, cnn.training, cnn.compute_dtype)
        loss = rnn.loss(feat, caption, caption_len)    # main.py:148-149
        loss.backward()                                # main.py:151
        self.reducer.start(self.opt.flat_grad)
        self.pending = True
        return loss

    def flush(self):
        self._apply_pending()
