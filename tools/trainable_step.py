"""A few plain (unpipelined) train steps of bench.py's configuration -- the program tools/prof_trainable.sh traces."""
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
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    trainer.step(image, caption, lens)
trainer.flush(); torch.cuda.synchronize()
print("T =", int(lens.max()), "sum(lens) =", int(lens.sum()))
