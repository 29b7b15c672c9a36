"""bench.py --gpus N without a launcher must start its ranks itself (fresh child processes, 127.0.0.1 rendezvous) before
touching a GPU, relay rank 0's JSON line and the return code.  CPU test: the child launcher is replaced by a stub."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(lines))
        self._rc = rc

    def wait(self):
        return self._rc


def test_self_launch_builds_the_torchrun_command_and_relays(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_popen(cmd, stdout=None, text=None, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return _FakeProc(["rank noise\n", json.dumps({"metric": "m", "n_gpus": 2}) + "\n"], 0)

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main()                                      # --gpus 2 and no WORLD_SIZE: must self-launch, never reach the GPU code
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2


def test_self_launch_propagates_failure(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: _FakeProc(["boom\n"], 3))
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2)
    assert e.value.code == 3
