import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from showtell_amd.rnn import RNN
from showtell_amd.beam import beam_search
torch.manual_seed(0)
rnn = RNN(512, 512, 10000, 5, dtype=torch.bfloat16).cuda().eval()
feat = torch.randn(256, 512, device="cuda")
for _ in range(2): beam_search(rnn, feat, beam_width=5, num_hypotheses=1, max_length=25)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): beam_search(rnn, feat, beam_width=5, num_hypotheses=1, max_length=25)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
