"""Launches per training step by kernel name: run under `rocprofv3 --kernel-trace --stats` with two step counts and
difference the `Calls` columns (tools/step_kernel_counts.py diff a.csv b.csv n_extra_steps).
  rocprofv3 --kernel-trace --stats --output-format csv -d out5 -- python3 tools/step_kernel_counts.py run 5
  rocprofv3 --kernel-trace --stats --output-format csv -d out15 -- python3 tools/step_kernel_counts.py run 15"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n):
    import torch
    from showtell_amd import optim
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    from showtell_amd.train import Trainer, synthetic_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    cnn = ResNet(101, 512, dtype=torch.bfloat16).to(dev).train()
    rnn = RNN(512, 512, 10000, 5, dtype=torch.bfloat16).to(dev).train()
    tr = Trainer(cnn, rnn, optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9), 1)
    image, caption, lens = synthetic_batch(128, 10000, seed=1, device=dev)
    for k in range(n):
        tr.step(image, caption, lens, upcoming=[image] * min(tr.depth, n - 1 - k))
    tr.flush()
    torch.cuda.synchronize()


def diff(a, b, extra):
    def load(f):
        return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
    A, B = load(a), load(b)
    rows = []
    for k, (c, t) in B.items():
        c0, t0 = A.get(k, (0, 0.0))
        if c != c0:
            rows.append(((t - t0) / extra / 1e3, (c - c0) / extra, k))
    print("us_per_step,launches_per_step,kernel")
    for us, c, k in sorted(rows, reverse=True):
        print("%.1f,%.1f,%s" % (us, c, k[:110]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        diff(sys.argv[2], sys.argv[3], int(sys.argv[4]))
