"""A few software-pipelined train steps of bench.py's configuration (the program tools/prof_pipeline.sh traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda", 0)
E, H, L, V, B = 512, 512, 5, 10000, 128
torch.manual_seed(1)
cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
trainer = Trainer(cnn, rnn, opt, 1)
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for k in range(n):
    trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, n - 1 - k))
trainer.flush(); torch.cuda.synchronize()
