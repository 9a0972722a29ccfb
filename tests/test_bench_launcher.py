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
    pid = 2 ** 22 + 12345                 # never a real process group (and os.killpg is replaced in these tests anyway)

    def __init__(self, lines, rc, hang=False, on_start=None):
        self.stdout, self._rc, self._hang, self._on_start = io.StringIO("".join(l + "\n" for l in lines)), rc, hang, on_start

    def wait(self, timeout=None):
        import subprocess
        if self._on_start:
            self._on_start()
            self._on_start = None
        if self._hang:
            self._hang = False
            raise subprocess.TimeoutExpired("ranks", timeout)
        return self._rc


def _launch(monkeypatch, capsys, lines, rc, argv, hang=False, provisional=None):
    bench = importlib.import_module("bench")
    assert bench.torch is None, "importing bench.py must not import torch (the self-launching parent never touches the GPU)"
    seen = {"killed": []}

    def fake_popen(cmd, env=None, stdout=None, text=None, preexec_fn=None):
        seen["cmd"], seen["env"], seen["own_group"] = cmd, env, callable(preexec_fn)      # own session + parent-death signal
        seen["cache_exists"] = os.path.isdir(env["HFEM_PLAN_CACHE"])

        def rank0_writes():                # what rank 0 does between sections: rewrite its provisional line
            if provisional is not None:
                with open(env["HFEM_BENCH_PROVISIONAL"], "w") as f:
                    f.write(json.dumps(provisional))
        return _FakeProc(lines, rc, hang=hang, on_start=rank0_writes)

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(os, "killpg", lambda pid, sig: seen["killed"].append((pid, sig)), raising=False)
    import signal
    monkeypatch.setattr(signal, "signal", lambda *a_: None)          # the parent's SIGTERM / SIGINT forwarders: not in the test process
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


def test_a_failed_or_hung_optional_leg_does_not_cost_the_headline(monkeypatch, capsys):
    """Rank 0 rewrites a provisional line after every section; when a rank fails, or the ranks outlive --deadline, the parent
    (which stops ITS OWN children's process group) prints that line marked `partial` instead of nothing."""
    prov = {"metric": "element-evals/sec", "value": 7.0, "n_gpus": 8, "partial": "collective-path legs", "config": {"notes": []}}
    # (i) a rank dies in a later leg
    rc, seen, out = _launch(monkeypatch, capsys, ["some log"], 1, ["--gpus", "8"], provisional=prov)
    got = json.loads(out.out.strip())
    assert rc == 0 and got["value"] == 7.0 and "a rank failed" in got["partial"] and "collective-path legs" in got["partial"]
    assert any("PARTIAL RESULT" in n for n in got["config"]["notes"]) and seen["own_group"] is True
    assert not os.path.exists(seen["env"]["HFEM_BENCH_PROVISIONAL"]), "the provisional file is removed"
    # (ii) the ranks hang: stopped at the deadline, provisional line printed
    rc, seen, out = _launch(monkeypatch, capsys, [], 0, ["--gpus", "8", "--deadline", "1"], hang=True, provisional=prov)
    got = json.loads(out.out.strip())
    assert rc == 0 and "--deadline" in got["partial"] and seen["killed"] and seen["killed"][0][0] == _FakeProc.pid
    # (iii) no provisional line yet (failure before the headline): still a failure
    rc, _, out = _launch(monkeypatch, capsys, [], 2, ["--gpus", "8"])
    assert rc == 2 and out.out.strip() == ""
    # (iv) the complete line wins over the provisional one
    line = json.dumps({"metric": "m", "value": 9.0})
    rc, _, out = _launch(monkeypatch, capsys, [line], 0, ["--gpus", "8"], provisional=prov)
    assert rc == 0 and json.loads(out.out.strip())["value"] == 9.0
