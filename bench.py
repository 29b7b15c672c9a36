#!/usr/bin/env python3
"""bench.py -- training images/sec of the ResNet-101 + GRU captioner (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the reference training step (main.py:136-152) over one synthetic
minibatch already resident in HBM: zero_grad, ResNet-101 forward in train mode (frozen, detached),
Linear+BatchNorm1d head, 5-layer GRU over the packed captions, vocabulary projection + cross
entropy, full backward, optimizer step.  Workload = BASELINE.json configs[1]: bf16, B=128/GPU,
E=H=512, L=5, V=10000, SGD(lr=0.01, momentum=0.9) (main.py:48-51 defaults).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the MFMA implicit-GEMM
convolution): algorithmic FLOPs of its launches / their summed duration, measured with HIP events
on the launch stream in extra, separately instrumented steps right after the timed region (the
timed steps themselves run uninstrumented).  `cpu_baseline` times the oracle's CPU restatement of
the same step on the host cores (rank 0, N=1 only) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
ENC_GFLOP_PER_IMG = 15.60           # 7.7994 GMAC of convolution per 224x224 image


def _progress(msg):
    """One line on stderr per phase: a long run must never look hung to whoever watches it."""
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def host_cores():
    """Cores this job may use: the affinity mask, capped at the 16-core share a one-GPU box grants (os.cpu_count() is the
    whole host: oversubscribing it turned the CPU baseline into minutes)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("SHOWTELL_BENCH_CORES", "16"))))


def cpu_baseline(threads, B=8, budget_s=8.0, min_steps=3, warm=True):
    """Oracle (CPU restatement, fp32) of the same train step; returns (images/sec, seconds, steps) on `threads` cores."""
    from oracle import restatement as R
    torch.set_num_threads(threads)
    enc = R.init_encoder_params(101, 512, seed=1)
    dec = {k: v.clone().requires_grad_(True) for k, v in R.init_decoder_params(512, 512, 10000, 5, "gru", seed=1).items()}
    head = {k: enc[k].clone().requires_grad_(True) for k in ("linear_secondlast_layer.weight", "linear_secondlast_layer.bias",
                                                             "last_layer.weight", "last_layer.bias")}
    cap, lens = R.synthetic_captions(B, 10000, seed=1)
    img = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    bufs = {}

    def step():
        p = dict(enc); p.update(head)
        feat = R.encoder_forward(p, img, 101, train=True)
        loss, _, _ = R.gru_train_loss(dec, feat, cap, lens, "gru")
        loss.backward()
        with torch.no_grad():
            for k, t in list(dec.items()) + list(head.items()):
                bufs[k] = R.sgd_momentum_step(t, t.grad, bufs.get(k), 0.01, 0.9)
                t.grad = None
    if warm:
        step()  # warm-up (allocations, thread pool)
    t0 = time.time()
    steps = 0
    while steps < min_steps or time.time() - t0 < budget_s:
        step()
        steps += 1
    dt = time.time() - t0
    return B * steps / dt, dt, steps


def cpu_greedy_baseline(threads, B=128):
    """Oracle greedy decode (rnn.py:37-58), 25 steps at the BASELINE decoder shape; returns (us per step, seconds)."""
    from oracle import restatement as R
    torch.set_num_threads(threads)
    dec = R.init_decoder_params(512, 512, 10000, 5, "gru", seed=1)
    feat = torch.randn(B, 512, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        R.rnn_greedy(dec, feat[:8], steps=2)
        t0 = time.time()
        R.rnn_greedy(dec, feat)
        dt = time.time() - t0
    return dt / 25 * 1e6, dt


def encoder_algorithmic_bytes(B, es=2, size=224):
    """HBM bytes one train-mode ResNet-101 forward must move when every fusion short of cross-layer on-chip residency is
    made: each convolution reads its (already normalised or raw) input and its weights once and writes its raw output once
    (batch statistics force that write: they must be complete before the output can be normalised); each block end reads the
    raw conv3 output and the identity and writes the block output; the stem's BN + ReLU ride in the max pool."""
    tot = 0
    h = size // 2
    tot += B * 3 * size * size * 4 + B * h * h * 64 * es            # stem conv: fp32 image in, raw out
    tot += B * h * h * 64 * es + B * (h // 2) ** 2 * 64 * es        # max pool (with bn1 + relu)
    h //= 2
    inpl = 64
    for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)):
        for bi in range(blocks):
            s_ = stride if bi == 0 else 1
            ho = h // s_
            tot += (B * h * h * inpl + inpl * planes + B * h * h * planes) * es                      # conv1 1x1
            tot += (B * h * h * planes + 9 * planes * planes + B * ho * ho * planes) * es            # conv2 3x3 (stride here, v1.5)
            tot += (B * ho * ho * planes + 4 * planes * planes + B * ho * ho * 4 * planes) * es      # conv3 1x1
            if bi == 0:
                tot += (B * h * h * inpl + inpl * 4 * planes + B * ho * ho * 4 * planes) * es        # downsample 1x1
            tot += 3 * B * ho * ho * 4 * planes * es                                                 # bn3 + identity + relu
            inpl, h = 4 * planes, ho
    return tot


def _corpus_bleu4(refs, hyps):
    """Corpus BLEU-4 (uniform weights, brevity penalty) over token-id sequences, one reference per hypothesis."""
    import math
    from collections import Counter
    num, den = [0] * 4, [0] * 4
    rl = hl = 0
    for r, h in zip(refs, hyps):
        rl += len(r); hl += len(h)
        for n in range(1, 5):
            rc = Counter(tuple(r[i:i + n]) for i in range(len(r) - n + 1))
            hc = Counter(tuple(h[i:i + n]) for i in range(len(h) - n + 1))
            num[n - 1] += sum(min(c, rc[g]) for g, c in hc.items())
            den[n - 1] += max(0, len(h) - n + 1)
    if min(num) == 0 or min(den) == 0:
        return 0.0
    bp = 1.0 if hl >= rl else math.exp(1.0 - rl / max(1, hl))
    return bp * math.exp(sum(math.log(a / b) for a, b in zip(num, den)) / 4.0)


def self_launch(n):
    """Run this script under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous, a free port)."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc or 1)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 timed steps after 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (BASELINE: 128)")
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--no-pipeline", action="store_true", help="do not issue the next minibatch's frozen backbone ahead on a second stream")
    ap.add_argument("--optimizer", default="SGD", choices=["SGD", "Adam"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU) BEFORE this
        # process touches the GPU, relay rank 0's JSON line and the launcher's return code.  Never exec.
        return self_launch(a.gpus)

    from showtell_amd import optim, parallel
    from showtell_amd._lib import lib
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    from showtell_amd.train import Trainer, synthetic_batch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    # RCCL (backend "nccl") is the product path; SHOWTELL_DIST_BACKEND=gloo only exists to rehearse N>1 on a 1-GPU box
    backend = os.environ.get("SHOWTELL_DIST_BACKEND", "nccl")
    _progress("start")
    rank, world, local = parallel.init_from_env(backend, device_index=(int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16
    E, H, L, V, B = 512, 512, 5, a.vocab, a.batch

    torch.manual_seed(1)                                     # main.py:26-27
    cnn = ResNet(101, E, dtype=dtype).to(dev).train()        # main.py:92,125
    rnn = RNN(E, H, V, L, dtype=dtype).to(dev).train()       # main.py:93,126
    params = Trainer.trainable_params(cnn, rnn)              # main.py:96
    opt = optim.SGD(params, lr=0.01, momentum=0.9) if a.optimizer == "SGD" else optim.Adam(params, lr=1e-4)
    trainer = Trainer(cnn, rnn, opt, world)
    image, caption, lens = synthetic_batch(B, V, seed=1 + rank, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Steps are software-pipelined: the frozen backbone forwards of minibatches k+1 .. k+3 are issued on side streams
    # while step k runs its trainable part (train.py).  Nothing is carried across the timing boundaries: the last warm-up
    # steps and the last timed steps prefetch nothing beyond their loop, so the timed region holds exactly `steps` complete
    # steps (the first one unpipelined, the second half-pipelined).
    pipe = not a.no_pipeline
    def ahead(k, n):   # the minibatches after step k that exist inside this loop (never across a timing boundary)
        return dict(upcoming=[image] * min(trainer.depth, n - 1 - k) if pipe else ())
    _progress("warm-up")
    for k in range(a.warmup):
        trainer.step(image, caption, lens, **ahead(k, a.warmup))
    trainer.flush()
    barrier()
    t0 = time.perf_counter()
    loss = None
    for k in range(a.steps):
        loss = trainer.step(image, caption, lens, **ahead(k, a.steps))
    trainer.flush()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item()) if loss is not None else float("nan")

    _progress("timed region done: %.3f ms/step" % (dt / a.steps * 1e3))
    # ---- roofline of the dominant kernel: instrumented steps, HIP events on the launch stream --------------
    roof = None
    if rank == 0:
        lib().st_prof_enable(1)
    for _ in range(max(1, a.profile_steps)):           # every rank steps: the gradient all-reduce is collective
        trainer.step(image, caption, lens)
    trainer.flush()
    torch.cuda.synchronize()
    if rank == 0:
        ms, fl, n = (C.c_double * 32)(), (C.c_double * 32)(), (C.c_long * 32)()
        lib().st_prof_collect(ms, fl, n)
        lib().st_prof_enable(0)
        names = {0: "igemm 128x128-tile family: igemm_kernel<bf16,128,128,2,4,{8,4},{1,2}> + igemm_s3b_kernel", 1: "igemm_kernel<bf16,128,64,4,1,8>",
                 2: "igemm_kernel<bf16,64,128,1,4,8>", 3: "igemm_kernel<bf16,256,128,4,2,8>"}
        prefixes = {0: ("igemm_kernel<bf16,128,128,2,4,", "igemm_s3b_kernel")}
        for i_, c_ in enumerate((64, 128, 256, 512)):          # csrc/conv_img.hip: one slot per kernel symbol (csrc/prof.h)
            names[8 + i_] = "conv3x3_img_kernel<%d,...> (image-resident 3x3, csrc/conv_img.hip)" % c_
            prefixes[8 + i_] = ("conv3x3_img_kernel<%d," % c_, "conv3x3_img_kernel_occ2<%d," % c_)
            names[12 + i_] = "conv1x1_wreg_kernel<%d,...> (pointwise, filter slice in registers, csrc/conv_img.hip)" % c_
            prefixes[12 + i_] = ("conv1x1_wreg_kernel<%d," % c_,)
        for i_, c_ in enumerate((1024, 2048)):
            names[16 + i_] = "conv1x1_kstream_kernel<%d,...> (pointwise, long K, csrc/conv_img.hip)" % c_
            prefixes[16 + i_] = ("conv1x1_kstream_kernel<%d," % c_,)
        for i_, c_ in enumerate((256, 512)):
            names[18 + i_] = "conv1x1_astat_kernel<%d,...> + conv1x1_cstat_kernel (pointwise, activation-stationary; statistics-only form, csrc/conv_img.hip)" % c_
            prefixes[18 + i_] = ("conv1x1_astat_kernel<%d," % c_, "conv1x1_cstat_kernel<%d," % c_)
        names[20] = "stem_pool_kernel (conv 7x7/2 + statistics + 3x3/2 pool in one kernel, csrc/conv_stem.hip)"
        prefixes[20] = ("stem_pool_kernel<",)
        names[21] = "conv_b2b_kernel (conv3 recomputed + bn3 + identity + ReLU (+ next conv1) in one pass, csrc/conv_b2b.hip)"
        prefixes[21] = ("conv_b2b_kernel<",)
        names[23] = "conv3x3s2_kstream_kernel (the three stride-2 3x3 convs, K-streaming implicit GEMM, csrc/conv_s2.hip)"
        prefixes[23] = ("conv3x3s2_kstream_kernel<",)
        names[22] = "conv_c3c1_kernel (14x14 Bottlenecks: conv3 256->1024 + block end + next conv1 1024->256 in one kernel, csrc/conv_c3c1.hip)"
        prefixes[22] = ("conv_c3c1_kernel<256,1024,256,7,1,true>", "conv_c3c1_kernel<true>")      # the train-mode form (the eval form appears in `secondary`'s config-5 encoder)
        names[24] = "conv_c3c1_kernel<128,512,128> (28x28 Bottlenecks: conv3 128->512 + block end + next conv1 512->128, HBM-bound, csrc/conv_c3c1.hip)"
        prefixes[24] = ("conv_c3c1_kernel<128,512,128,5,2,true>",)
        v = max(range(32), key=lambda i: ms[i])
        ach = fl[v] / (ms[v] * 1e-3) / 1e12 if ms[v] > 0 else 0.0
        tot_ms, tot_fl = sum(ms), sum(fl)
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside the process; the figure comes from the
        # committed summary of the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command
        # (profiles/README.md; 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction), launch-weighted over the 128x128-tile forms
        traffic = traffic_src = None
        try:
            import csv, glob
            f = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_hbm_traffic.csv")))[-1]
            traffic_src = "profiles/" + os.path.basename(f) + " (committed rocprofv3 --pmc passes of an earlier run of this command, NOT this run)"
            rows = [r for r in csv.reader(open(f)) if r and any(pf in r[0] for pf in prefixes.get(v, ("\0",)))]
            if rows:
                traffic = round(sum(float(r[1]) * float(r[4]) for r in rows) / sum(float(r[1]) for r in rows) * 1e6)
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                "kernel": names.get(v, f"variant{v}"), "launches": int(n[v]),
                "avg_launch_us": round(ms[v] * 1e3 / max(1, n[v]), 2),
                "all_conv_TFLOPs": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else 0.0,
                "all_conv_ms_per_step": round(tot_ms / max(1, a.profile_steps), 3),
                "by_kernel": {names.get(i, f"variant{i}").split(" (")[0].split(":")[0]: {"ms_per_step": round(ms[i] / max(1, a.profile_steps), 3), "launches": int(n[i]),
                                                                       "TFLOPs": round(fl[i] / (ms[i] * 1e-3) / 1e12, 1)} for i in range(32) if n[i] > 0}}
    _progress("phases")
    # ---- per-phase split of one plain (unpipelined) step, SURVEY 8(d) config 2: HIP events on the launch stream ----------
    phases = None
    if rank == 0 and world == 1:
        try:
            from showtell_amd.head import linear_bn1d
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(3)]
            for e in ev:
                e[0].record()
                pooled = cnn.backbone_features(image)                                    # frozen encoder (cnn.py:46-47)
                e[1].record()
                opt.zero_grad()
                feat_ = linear_bn1d(pooled, cnn.linear_secondlast_layer, cnn.last_layer, cnn.training, cnn.compute_dtype)
                ls = rnn.loss(feat_, caption, lens)                                      # head + decoder forward + loss
                e[2].record()
                ls.backward()
                e[3].record()
                opt.step()
                e[4].record()
            torch.cuda.synchronize()
            names_ = ["encoder_forward", "head_decoder_forward_loss", "backward", "optimizer"]
            phases = {nm: round(sum(e[i].elapsed_time(e[i + 1]) for e in ev[1:]) / 2, 3) for i, nm in enumerate(names_)}
            enc_s = phases["encoder_forward"] * 1e-3
            enc_bytes = encoder_algorithmic_bytes(B)
            phases["encoder_mfma_frac"] = round(ENC_GFLOP_PER_IMG * 1e9 * B / enc_s / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)   # whole forward, one in flight
            phases["encoder_hbm_frac"] = round(enc_bytes / enc_s / 8e12, 4)
            phases["encoder_algorithmic_GB"] = round(enc_bytes / 1e9, 2)
            # the same encoder under cnn.eval() (utils.py:163-164 / main.py:173-174: folded BatchNorm, no statistics, no block-end pass)
            cnn.eval()
            ee = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(2):
                cnn.backbone_features(image)
            ee[0].record()
            for _ in range(3):
                cnn.backbone_features(image)
            ee[1].record()
            torch.cuda.synchronize()
            cnn.train()
            phases["encoder_forward_eval"] = round(ee[0].elapsed_time(ee[1]) / 3, 3)
            phases["encoder_eval_mfma_frac"] = round(ENC_GFLOP_PER_IMG * 1e9 * B / (phases["encoder_forward_eval"] * 1e-3) / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)
            phases["unit"] = "ms per plain step (no forwards in flight); the pipelined step overlaps the encoder of later minibatches with the rest"
        except Exception as e:
            phases = {"error": repr(e)}
    _progress("secondary: decode / beam / attention / input transform")
    # ---- secondary (rank 0, N=1): greedy decode step against the HBM roofline, beam=5 captions/sec -------------
    secondary = None
    if rank == 0 and world == 1:
        try:
            rnn.eval()
            feat = torch.randn(B, E, device=dev)
            for _ in range(2):
                rnn.sentence_index(feat)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                rnn.sentence_index(feat)
            e1.record(); torch.cuda.synchronize()
            us_step = e0.elapsed_time(e1) / reps / 25 * 1e3
            nw = sum(p.numel() for n_, p in rnn.named_parameters() if not n_.startswith("embeddings"))
            byts = (nw + 2 * L * B * H + B * E) * 2 + 16 * B            # 27.46 MB/step in bf16
            f256 = torch.randn(256, E, device=dev)
            rnn.beam_search(f256[:8], 5, 1, 25)
            torch.cuda.synchronize(); tb = time.perf_counter()
            rnn.beam_search(f256, 5, 1, 25)
            torch.cuda.synchronize(); tb = time.perf_counter() - tb
            # BASELINE configs[4] END TO END (utils.py:177-194): 256 synthetic images -> eval-mode ResNet-101 + head -> beam search
            # (beam 5, max_length 25) -> token lists on the host; the decoder-only figure above starts from ready-made features
            cnn.eval()
            img256 = torch.randn(256, 3, 224, 224, device=dev)
            with torch.no_grad():
                rnn.beam_search(cnn(img256[:8]), 5, 1, 25)            # warm-up (workspace of another batch size)
                cnn(img256)
                torch.cuda.synchronize(); te = time.perf_counter()
                feats256 = cnn(img256)
                torch.cuda.synchronize(); t_enc = time.perf_counter() - te
                hyp256 = rnn.beam_search(feats256, 5, 1, 25)
                torch.cuda.synchronize(); te = time.perf_counter() - te
            cnn.train()
            del img256
            # Config 5 quality check against the REFERENCE's own outputs: tests/golden/beam_small.npz holds weights, image
            # features and the beam-5 hypotheses the reference produced for them (oracle/gen_golden.py); the fp32 kernels must
            # reproduce those token ids, the bf16 id-match rate is reported beside it.  (BLEU against the CPU oracle on longer
            # captions lives in tests/test_gpu_beam.py: the oracle is test infrastructure.)
            import numpy as np
            gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "beam_small.npz"))
            gp = {k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("p/")}
            gV, gE = gp["embeddings.weight"].shape
            gH = gp["unit.weight_hh_l0"].shape[1]
            gL = sum(1 for k in gp if k.startswith("unit.weight_hh_l"))
            r32 = RNN(gE, gH, gV, gL, dtype=torch.float32); r32.load_state_dict(gp); r32 = r32.to(dev).eval()
            r16 = RNN(gE, gH, gV, gL, dtype=torch.bfloat16); r16.load_state_dict(gp); r16 = r16.to(dev).eval()
            gfeat = torch.from_numpy(gold["feat"]).to(dev)
            ml = int(gold["bw5_maxlen"])
            h32, h16 = r32.beam_search(gfeat, 5, 1, ml), r16.beam_search(gfeat, 5, 1, ml)
            nimg, same32, same16, nonempty = gfeat.shape[0], 0, 0, 0
            for b_ in range(nimg):
                ln = int(gold["bw5_len"][b_][0])
                ref_seq = gold["bw5_seq"][b_, 0, :ln].tolist() if ln > 0 else []
                nonempty += int(ln > 0)
                same32 += int((h32[b_][0][0] if h32[b_] else []) == ref_seq)
                same16 += int((h16[b_][0][0] if h16[b_] else []) == ref_seq)
            del r32, r16
            # BLEU-4 of 25-token greedy captions against the reference's own greedy captions (tests/golden/gru_small.npz, rnn.py:37-58)
            gg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "gru_small.npz"))
            gq = {k[2:]: torch.from_numpy(gg[k]) for k in gg.files if k.startswith("p/")}
            qV, qE = gq["embeddings.weight"].shape
            qH = gq["unit.weight_hh_l0"].shape[1]
            qL = sum(1 for k in gq if k.startswith("unit.weight_hh_l"))
            bleu = {}
            for nm_, dt_ in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
                rq = RNN(qE, qH, qV, qL, dtype=dt_); rq.load_state_dict(gq); rq = rq.to(dev).eval()
                hyp = rq.sentence_index(torch.from_numpy(gg["feat"]).to(dev)).cpu().numpy()
                bleu[nm_] = _corpus_bleu4([list(map(int, r_)) for r_ in gg["greedy"]], [list(map(int, h_)) for h_ in hyp])
                del rq
            # HBM bytes per greedy step from the committed PMC passes of tools/time_decode.py (profiles/README.md, r01i):
            # L fused-x cell launches + one vocabulary arg-max launch, 2*FETCH_SIZE + WRITE_SIZE each
            dec_traffic = dec_traffic_src = None
            try:
                import csv, glob
                f_ = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_decode_hbm_traffic.csv")))[-1]
                rows_ = [r_ for r_ in csv.reader(open(f_)) if r_]
                pipe_ = [float(r_[4]) for r_ in rows_ if r_[0].startswith("decode_pipe_kernel")]
                cell_ = [float(r_[4]) for r_ in rows_ if r_[0].startswith("rnn_gemm_kernel<bool") and ", true, 1>" in r_[0]]
                voc_ = [float(r_[4]) for r_ in rows_ if r_[0] == "vocab_argmax_lds_kernel"]
                if pipe_:                                             # one launch = the whole 25-step decode
                    dec_traffic = round(pipe_[0] * 1e6 / 25)
                elif cell_ and voc_:
                    dec_traffic = round((L * cell_[0] + voc_[0]) * 1e6)
                if dec_traffic is not None:
                    dec_traffic_src = "profiles/" + os.path.basename(f_) + " (committed PMC passes of an earlier run, NOT this run)"
            except Exception:
                dec_traffic = None
            secondary = {"greedy_decode_us_per_step": round(us_step, 1),
                         "greedy_fp32_bleu4_vs_reference_vectors": round(bleu["fp32"], 4),
                         "greedy_bf16_bleu4_vs_reference_vectors": round(bleu["bf16"], 4),
                         "beam5_fp32_id_match_vs_reference_vectors": round(same32 / max(1, nimg), 3),
                         "beam5_bf16_id_match_vs_reference_vectors": round(same16 / max(1, nimg), 3),
                         "beam5_reference_vectors_images": nimg, "beam5_reference_nonempty": nonempty,
                         "greedy_algorithmic_MB_per_step": round(byts / 1e6, 2),
                         "greedy_hbm_roofline": {"bound": "hbm", "achieved": round(byts / us_step / 1e3, 1), "peak": 8000.0,
                                                 "unit": "GB/s", "frac": round(byts / us_step / 1e3 / 8000.0, 4), "traffic": dec_traffic, "traffic_source": dec_traffic_src},
                         "greedy_captions_per_sec": round(B / (us_step * 25e-6), 0),
                         "beam5_bs256_captions_per_sec": round(256 / tb, 0),
                         "beam5_bs256_images_to_captions_per_sec": round(256 / te, 0),
                         "beam5_bs256_images_to_captions_ms": {"encoder_eval_forward": round(t_enc * 1e3, 2), "total": round(te * 1e3, 2)},
                         "beam5_bs256_images_to_captions_hypotheses": len(hyp256)}
            _progress("secondary: beam-5 bf16 vs fp32 at the full decoder shape")
            # Config 5 quality at the FULL decoder shape (E = H = 512, L = 5, V = 10000): beam-5 captions of the bf16 kernels
            # scored with the reference's BLEU (evaluation.py:bleu_score = evaluation_metrics.py:117-317) against the fp32
            # kernels' captions on the same weights, 32 images.  A random-init decoder never emits <end> (and a constant <end>
            # bias ends every caption at the first word or never: the hidden state settles within a few steps), so ONE hidden
            # unit of the top layer is hand-set to a leaky counter, h_c(t) = 1 - 0.9^t, and <end> reads it with a large weight:
            # its logit climbs past the word logits after a number of steps that depends on the image -- captions of 5 - 20
            # words, as a trained model's.  Every other weight stays random.
            try:
                from showtell_amd.evaluation import bleu_score
                torch.manual_seed(5)
                r32 = RNN(E, H, V, L, dtype=torch.float32)
                sd_ = {k: v.clone() for k, v in r32.state_dict().items()}
                r16 = RNN(E, H, V, L, dtype=torch.bfloat16)
                r32, r16 = r32.to(dev).eval(), r16.to(dev).eval()
                f32_ = torch.randn(32, E, device=dev)
                boost_used, h32 = None, None
                sd_["linear.weight"] = sd_["linear.weight"] * 12.0    # a random-init decoder's logits are nearly flat: give them spread
                top = "_l%d" % (L - 1)
                for nm in ("unit.weight_ih" + top, "unit.weight_hh" + top):
                    for gate in range(3):
                        sd_[nm][gate * H] = 0.0                         # unit 0 of the top layer listens to nothing ...
                sd_["unit.bias_ih" + top][0] = 0.0; sd_["unit.bias_hh" + top][0] = 0.0
                sd_["unit.bias_ih" + top][H] = 2.1972; sd_["unit.bias_hh" + top][H] = 0.0        # ... keeps 0.9 of itself (z = sigmoid(2.197))
                sd_["unit.bias_ih" + top][2 * H] = 3.0; sd_["unit.bias_hh" + top][2 * H] = 0.0    # ... and moves towards tanh(3) = 0.995
                sd_["linear.weight"][:, 0] = 0.0
                sd_["linear.weight"][2, 0] = 21.0                       # <end> (id 2) reads the counter
                for boost in [-14.0 + 0.5 * i for i in range(0, 40)]:  # the smallest <end> bias that lets most captions complete within 25 steps
                    sd_b = dict(sd_); sd_b["linear.bias"] = sd_["linear.bias"].clone(); sd_b["linear.bias"][2] += boost
                    r32.load_state_dict(sd_b)
                    h = r32.beam_search(f32_, 5, 1, 25)
                    done = [len(x[0][0]) for x in h if x]
                    if len(done) >= 28:                             # first (= longest-caption) setting that completes 7/8 of the images
                        boost_used, h32 = boost, h
                        break
                if h32 is None:
                    boost_used, h32 = boost, h
                sd_b = dict(sd_); sd_b["linear.bias"] = sd_["linear.bias"].clone(); sd_b["linear.bias"][2] += boost_used
                r32.load_state_dict(sd_b); r16.load_state_dict(sd_b)
                h32, h16 = r32.beam_search(f32_, 5, 1, 25), r16.beam_search(f32_, 5, 1, 25)
                gts = {i: [" ".join(map(str, h32[i][0][0])) if h32[i] else ""] for i in range(32)}
                res = {i: [" ".join(map(str, h16[i][0][0])) if h16[i] else ""] for i in range(32)}
                keep = [i for i in range(32) if gts[i][0]]
                b4 = bleu_score({i: gts[i] for i in keep}, {i: res[i] for i in keep}, 4)[0][3] if keep else None
                secondary["beam5_bf16_bleu4_vs_fp32_kernels_fullshape"] = None if b4 is None else round(b4, 4)
                secondary["beam5_fullshape_end_bias_boost"] = boost_used
                secondary["beam5_fullshape_images"] = 32
                secondary["beam5_fullshape_fp32_completed"] = len(keep)
                secondary["beam5_fullshape_exact_match"] = sum(int(gts[i] == res[i]) for i in range(32))
                lens_ = [len(gts[i][0].split()) for i in keep]
                secondary["beam5_fullshape_mean_len"] = round(sum(lens_) / max(1, len(keep)), 1)
                secondary["beam5_fullshape_len_min_max"] = [min(lens_), max(lens_)] if lens_ else None
                secondary["beam5_fullshape_distinct_tokens"] = len({t for i in keep for t in gts[i][0].split()})
                del r32, r16
            except Exception as e:
                secondary["beam5_fullshape_error"] = repr(e)
            rnn.train()
            _progress("secondary: attention config")
            # BASELINE configs[2]: soft-attention GRU decoder (Attention/main_attn.py), bs=64, alpha_c=1.0: train steps/sec
            try:
                from showtell_amd.cnn_attn import ResNet as ResNetAttn
                from showtell_amd.rnn_attn import RNN_Attn
                cnn_a = ResNetAttn(101, E, dtype=dtype).to(dev).train()
                rnn_a = RNN_Attn(E, 2048, 512, H, V, L, dtype=dtype).to(dev).train()
                opt_a = optim.SGD(list(rnn_a.parameters()), lr=0.01, momentum=0.9)
                img_a, cap_a, lens_a = synthetic_batch(64, V, seed=5, device=dev)

                def astep():
                    opt_a.zero_grad()
                    la = rnn_a.loss(cnn_a(img_a), cap_a, lens_a, 1.0)      # main_attn.py:126-131
                    la.backward()
                    opt_a.step()
                    return la
                for _ in range(3):
                    astep()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(8):
                    la = astep()
                torch.cuda.synchronize()
                ta = (time.perf_counter() - ta) / 8
                secondary["attention_gru_bs64_train_images_per_sec"] = round(64 / ta, 1)
                secondary["attention_gru_bs64_ms_per_step"] = round(ta * 1e3, 3)
                secondary["attention_gru_bs64_loss_finite"] = bool(torch.isfinite(la.detach()).item())
                del cnn_a, rnn_a, opt_a
            except Exception as e:
                secondary["attention_error"] = repr(e)
            _progress("secondary: input transform")
            # input pipeline (utils.py:84-88 on the GPU, data.py): 128 COCO-shaped uint8 images -> (128, 3, 224, 224) fp32
            try:
                import numpy as np
                from showtell_amd.data import DeviceTransform
                rng = np.random.default_rng(0)
                shapes = [(480, 640), (640, 480), (427, 640), (375, 500)]
                imgs = [rng.integers(0, 256, size=shapes[i % 4] + (3,), dtype=np.uint8) for i in range(B)]
                tfm = DeviceTransform()
                for _ in range(2):
                    tfm(imgs)
                torch.cuda.synchronize()
                tt = time.perf_counter()
                for _ in range(5):
                    tfm(imgs)
                torch.cuda.synchronize()
                tt = (time.perf_counter() - tt) / 5
                secondary["input_transform_host_arrays_images_per_sec"] = round(B / tt, 0)      # staging pass + PCIe + 2 kernels
                flat = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
                hs, ws = [im.shape[0] for im in imgs], [im.shape[1] for im in imgs]
                for _ in range(2):
                    tfm.packed(flat, hs, ws)
                torch.cuda.synchronize()
                tt = time.perf_counter()
                for _ in range(5):
                    tfm.packed(flat, hs, ws)
                torch.cuda.synchronize()
                secondary["input_transform_pinned_buffer_images_per_sec"] = round(B * 5 / (time.perf_counter() - tt), 0)   # PCIe + 2 kernels
            except Exception as e:
                secondary["input_transform_error"] = repr(e)
        except Exception as e:
            secondary = {"error": repr(e)}
    if world > 1:
        torch.distributed.barrier()

    out = None
    if rank == 0:
        value = world * B * a.steps / dt
        out = {"metric": "training images/sec (ResNet101+GRU, emb=512, bs=128) at 1/2/4/8 MI355X",
               "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: ResNet-101 (train-mode BN, frozen) + 5-layer GRU decoder, "
                                      "E=H=512, V=%d, fwd+bwd+%s step, synthetic COCO-shaped batch" % (V, a.optimizer),
                          "batch_per_gpu": B, "global_batch": B * world, "image": "3x224x224",
                          "tokens_per_batch": int(sum(lens)), "parallelism": "dp%d" % world,
                          "schedule": ("frozen-backbone forwards of the next 3 minibatches in flight on side streams (train.py); "
                                       "every step's full work is inside the timed region") if pipe else "plain loop (--no-pipeline)",
                          "final_loss": round(final_loss, 4)},
               "roofline": roof, "phases": phases, "secondary": secondary}
        if world == 1 and not a.no_cpu_baseline:
            ncpu = host_cores()
            _progress("cpu baseline on %d cores" % ncpu)
            try:
                v, secs, nst = cpu_baseline(ncpu)
                out["cpu_baseline"] = {"value": round(v, 2), "unit": "images/sec", "cores": ncpu, "kind": "port",
                                       "sample": "oracle/restatement.py (torch CPU fp32) on the same train step at B=8 on the %d host "
                                                 "cores of this job's share: %d timed steps after 1 warm-up, %.1f s" % (ncpu, nst, secs)}
                extra = {}
                _progress("cpu baseline on 8 cores, B=128 step, greedy decode")
                v8, s8, n8 = cpu_baseline(min(8, ncpu))
                extra["train_B8_8cores_images_per_sec"] = round(v8, 2)
                extra["train_B8_8cores_sample"] = "%d steps, %.1f s on %d cores" % (n8, s8, min(8, ncpu))
                vb, sb, nb = cpu_baseline(ncpu, B=128, budget_s=0.0, min_steps=2, warm=True)
                extra["train_B128_allcores_images_per_sec"] = round(vb, 2)
                extra["train_B128_allcores_sample"] = "%d timed steps after 1 warm-up at B=128 (the metric's batch), %.1f s on %d cores" % (nb, sb, ncpu)
                us, sg = cpu_greedy_baseline(ncpu)
                extra["greedy_decode_B128_us_per_step"] = round(us, 1)
                extra["greedy_decode_sample"] = "25 steps, B=128, L=5, V=10000 fp32, %.1f s on %d cores" % (sg, ncpu)
                out["cpu_baseline"]["more"] = extra
            except Exception as e:  # the baseline must never take the bench line down
                out.setdefault("cpu_baseline", {"value": None, "unit": "images/sec", "cores": ncpu, "kind": "port"})["error"] = "failed: %r" % (e,)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
