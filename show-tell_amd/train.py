"""The training step of the reference (main.py:136-152) on MI355X, plus a synthetic-data driver.

    optimizer.zero_grad(); feat = cnn(image); loss = CE(rnn(feat, cap, lens), packed(cap));
    loss.backward(); optimizer.step()

`Trainer.step` performs exactly that arithmetic, ordered so that data-parallel communication
hides behind the frozen backbone:  the previous step's gradient all-reduce and optimizer
update are completed AFTER this step's backbone forward (which reads only frozen weights,
cnn.py:47) and BEFORE its trainable part.  With one GPU the order is irrelevant and the result
is identical to the reference loop; `flush()` applies the last pending update.

Software pipelining across steps (`step(..., upcoming=[next images])`): because the backbone is frozen, the
forward of a LATER minibatch depends on nothing this step computes.  When the caller hands the next images over (a data
loader always has them), their backbone forwards are issued on side HIP streams (up to three in flight) and run beside this step's head /
decoder / backward (launch-bound wavefront, few-tile GEMMs) AND beside each other: two backbone forwards in flight fill
each other's launch tails and pair the HBM-bound normalise passes with the MFMA-bound convolutions (6.8 -> 5.5 ms per
forward measured, tools/two_stream_encoder.py).  The arithmetic and its order per sample are unchanged -- the running
BatchNorm buffers are updated in minibatch order (cnn.py) -- only the schedule is.
"""
import os

import numpy as np
import torch

from .head import linear_bn1d
from .parallel import GradAllReducer, broadcast_state


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
        if world_size > 1:
            broadcast_state([cnn, rnn], optimizer)         # replicas start from rank 0's weights and BN buffers
        self.pending = False
        optimizer._pending_owner = self                    # optimizer.state_dict() / utils.create_checkpoint flush the deferred step
        self._pre = []        # FIFO of (images, pooled features, event): backbone forwards issued ahead on the side streams
        self._side = []
        self._rr = 0
        self.depth = int(os.environ.get("ST_PIPE_DEPTH", "3"))        # backbone forwards kept in flight ahead of the trainable part (B=128 step: depth 1: 6.8 ms per forward, 2: 5.6, 3: 5.3 = 5.96 ms/step; 4: 6.35 ms/step, 5: 7.46 -- more forwards in flight evict each other's activations from the 256 MB Infinity Cache)

    def trainable_params(cnn, rnn):
        """main.py:96: rnn.parameters() + cnn.linear_secondlast_layer + cnn.last_layer."""
        return list(rnn.parameters()) + list(cnn.linear_secondlast_layer.parameters()) + list(cnn.last_layer.parameters())
    trainable_params = staticmethod(trainable_params)

    def _apply_pending(self):
        if self.pending:
            self.opt.grad_scale = self.reducer.finish()
            self.opt.step()
            self.pending = False

    def _backbone(self, image):
        """Backbone features of `image` on the current stream; takes the result issued ahead by `_prefetch` if the oldest one in
        flight is this batch."""
        main = torch.cuda.current_stream()
        if self._pre and self._pre[0][0] is image:
            _, pooled, ev = self._pre.pop(0)
            main.wait_event(ev)
            pooled.record_stream(main)
            return pooled
        for s_ in self._side:                          # unexpected batch: drop what was prefetched, stay ordered with the side streams
            main.wait_stream(s_)
        self._pre = []
        return self.cnn.backbone_features(image)

    def _prefetch(self, image):
        if not self._side:
            self._side = [torch.cuda.Stream() for _ in range(self.depth)]
        side = self._side[self._rr % len(self._side)]
        self._rr += 1
        side.wait_stream(torch.cuda.current_stream())  # after everything already queued on the main stream
        with torch.cuda.stream(side):
            pooled = self.cnn.backbone_features(image)
            ev = torch.cuda.Event()
            ev.record(side)
        self._pre.append((image, pooled, ev))

    def step(self, image, caption, caption_len, upcoming=()):
        """One training step.  `upcoming`: the images of the next minibatches, in order (at most `depth` are used)."""
        cnn, rnn = self.cnn, self.rnn
        pooled = self._backbone(image)                 # frozen, detached (cnn.py:46-47): overlaps the all-reduce
        # keep up to `depth` later minibatches' frozen backbones in flight, beside this step's trainable part and each other
        upcoming = [im for im in list(upcoming)[:self.depth] if im is not None]
        for j, im in enumerate(upcoming):
            if j < len(self._pre):
                if self._pre[j][0] is not im:          # the caller changed its mind: start over from here
                    for s_ in self._side:
                        torch.cuda.current_stream().wait_stream(s_)
                    self._pre = self._pre[:j]
                    self._prefetch(im)
            else:
                self._prefetch(im)
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
