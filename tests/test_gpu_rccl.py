"""The RCCL ("nccl" backend) branch of the data-parallel trainer on the one GPU a test box has; see _rccl_single.py.
The N>1 semantics (mean over ranks, bucketing) are covered on CPU by test_parallel_gloo.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_trainer_through_rccl_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_single.py"), str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
