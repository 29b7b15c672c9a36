"""The command the driver runs for N > 1 -- `python bench.py --gpus N ...` -- under test on the one GPU of the test box:
two ranks (gloo instead of RCCL: both ranks share the card, SHOWTELL_DIST_BACKEND is bench.py's rehearsal switch), the
BASELINE configs[3] shape per rank (ResNet-101 + GRU, B = 128 per rank).  bench.py must start its ranks itself, finish with
rc 0 and print exactly one JSON line with n_gpus = 2, global_batch = 256 and a finite loss."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, SHOWTELL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 256 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["unit"] == "images/sec" and d["value"] > 0
    assert math.isfinite(d["config"]["final_loss"]) and 5.0 < d["config"]["final_loss"] < 12.0      # ~ log(10000) = 9.2 at random init
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None          # N > 1: no CPU leg, no secondary block
