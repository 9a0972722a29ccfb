"""``python bench.py --gpus N`` without a launcher: the parent must start its N ranks as CHILD processes through
``torch.distributed.run`` on 127.0.0.1, never touch the GPU itself, give the job one plan cache, relay exactly rank 0's JSON line
and fail when a rank fails.  The ranks are replaced by a stub here (CPU box); the real thing is rehearsed on the GPU box
(profiles/r04/bench_*rank_gloo_one_gpu.json)."""
import importlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout, self._rc = io.StringIO("".join(l + "\n" for l in lines)), rc

    def wait(self):
        return self._rc


def _launch(monkeypatch, capsys, lines, rc, argv):
    bench = importlib.import_module("bench")
    assert bench.torch is None, "importing bench.py must not import torch (the self-launching parent never touches the GPU)"
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        seen["cache_exists"] = os.path.isdir(env["HFEM_PLAN_CACHE"])
        return _FakeProc(lines, rc)

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HFEM_PLAN_CACHE", raising=False)
    a = bench.parse()
    got = bench.self_launch(a)
    out = capsys.readouterr()
    return got, seen, out


def test_self_launch_starts_children_relays_one_line_and_cleans_up(monkeypatch, capsys):
    line = json.dumps({"metric": "element-evals/sec", "value": 1.0, "n_gpus": 2})
    rc, seen, out = _launch(monkeypatch, capsys, ["[Gloo] Rank 0 is connected", line, "trailing noise"], 0,
                            ["--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5"])
    assert rc == 0
    assert out.out.strip() == line, "stdout carries exactly rank 0's JSON line"
    assert "Rank 0 is connected" in out.err and "trailing noise" in out.err        # everything else goes to stderr
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=2" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5"][-6:] and os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["cache_exists"] and not os.path.exists(seen["env"]["HFEM_PLAN_CACHE"]), "one plan cache per job, removed afterwards"


def test_self_launch_fails_when_a_rank_fails_or_nothing_was_printed(monkeypatch, capsys):
    line = json.dumps({"metric": "m", "value": 1.0})
    rc, _, out = _launch(monkeypatch, capsys, [line], 3, ["--gpus", "4"])
    assert rc == 3 and out.out.strip() == "", "a failed rank: non-zero exit, no result line"
    rc, _, out = _launch(monkeypatch, capsys, ["no json here"], 0, ["--gpus", "4"])
    assert rc != 0 and out.out.strip() == ""
