"""bench.py's pipelined loop with the trainable chain on a CU-masked stream (hipExtStreamCreateWithCUMask): ms per step.
usage: python tools/time_step_cumask.py [cus_per_group_of_32 ...]   (0 = no mask)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
E, H, L, V, B = 512, 512, 5, 10000, 128
torch.manual_seed(1)
cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
trainer = Trainer(cnn, rnn, opt, 1)
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev, **({"mean": 4.0, "std": 0.5, "lo": 3, "hi": 5} if os.environ.get("SHORT") else {}))
ahead = lambda k, n: dict(upcoming=[image] * min(trainer.depth, n - 1 - k))
for per32 in [int(a) for a in sys.argv[1:]] or [0, 8, 12, 16, 0]:
    if per32 == 0:
        st = torch.cuda.current_stream()
    else:
        # bit i set when (i / 8) % 4 < per32 / 8 ... : per32 CUs of every 32, the same under a linear and an XCC-round-robin reading of the mask
        words = (C.c_uint32 * 8)()
        for i in range(256):
            if ((i // 8) % 4) * 8 < per32:
                words[i // 32] |= 1 << (i % 32)
        h = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
        assert rc == 0, rc
        st = torch.cuda.ExternalStream(h.value)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for k in range(10):
            trainer.step(image, caption, lens, **ahead(k, 10))
        trainer.flush(); torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for k in range(n):
            trainer.step(image, caption, lens, **ahead(k, n))
        trainer.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    torch.cuda.current_stream().wait_stream(st)
    print(f"trainable chain on {per32 * 8 if per32 else 256} CUs: {1e3 * dt / n:.3f} ms/step  {B * n / dt:.0f} img/s", flush=True)
