#!/usr/bin/env python3
"""Headline benchmark: fused TRI3+EDGE2 elastic-energy forward+backward ("element-evals/s").

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU over RCCL -- either launched as ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
...`` or, given plain ``python bench.py --gpus N``, bench.py starts its N ranks itself as child processes (self_launch: the
parent never touches the GPU and never execs).  Every host plan is built ONCE per job (rank 0 writes the plan blob to the
job's plan cache, the other ranks deserialise it).
A *step* is one pass of the hot path over the synthetic mesh: loss + d/d node_coords_free + d/d u_free, inputs resident
in HBM.  The timed region is ONE hipGraph of exactly K steps, bracketed by barrier + synchronize on both sides; it is
replayed ``--repeats`` (9) times and the MEDIAN replay is reported (max over ranks per replay).

Workload at N = 1: BASELINE.json configs[3] "Example 4", reading (i) of SURVEY F11 = "T1M": plate [0,2]x[0,1], 1001x501
nodes -> 1,000,000 TRI3 (each structured quad split in two), interior nodes jittered 0.2 h (seed 0), outer boundary fixed,
left edge Dirichlet, right edge Neumann (500 edges), u_free ~ 1e-5 N(0,1), E=10e9, nu=0.3, gauss_order=4, fp64, r-adaptivity on.

N > 1: `value` is WEAK scaling (the plate grows to N x 1,000,000 elements; every rank evaluates its contiguous tile range,
then ONE small all_gather of interface parameter rows + partial energies -- owner-sharded mode, in-library RCCL on the
kernel's stream, the K steps in one hipGraph).  Beside it, in `config`:
  eval_exchange_overlap             the headline step with the exchange of step k under the interior tiles of step k + 1
  train_step / train_step_overlap   whole Adam iterations (exchange on the critical path / hidden under the interior tiles)
  train_step_fused[_overlap]        the same with Adam applied by the energy kernel's own write-out (three launches per step)
  peer_exchange                     the evaluation / training steps with the interface rows STORED by the pack launch into every
                                    rank's receive window (csrc/peer.hip: IPC-mapped memory, xGMI stores + flags; no collective,
                                    no second stream), verified in the run against the collective path; headline if faster
  alt_exchange                      the north-star's literal wording: a dense sum all-reduce of [gX|gU|loss]
  strong_scaling                    BASELINE configs[3] and [4] AS STATED: 10^6 TRI3 FIXED sharded over the N ranks (and, at
                                    N = 8, the 4.1 M-element Delaunay mesh), kernel-only per rank and end to end
At N = 1 `config.strong_scaling_emulated` rehearses those shards on the one GPU: the kernel over the tile range rank r of
N would evaluate (`--emulate-shard r/N` runs just that).

One JSON line on stdout (rank 0).  ``roofline`` is for the dominant kernel (tri3_energy_pair_kernel): algorithmic bytes
(12 Ne + 64 Nn + 8) over its average launch time, HIP events on the launch stream.  ONE regime per line:
the top level is the regime the timed step itself runs in (`replayed`: the same buffers every launch, a 44 MB working set the
256 MB Infinity Cache holds -- what a training loop on this mesh sees), so `roofline.kernel_us <= ms_per_step`; the regime whose
reads really come from HBM (`rotating_sets`: 10 parameter / gradient sets, 313 MB) is `roofline.hbm_regime`, with the whole
K-step region timed there as well (`hbm_regime.step`); `regimes` keeps every leg (replayed, rewritten_inputs, rotating_sets),
each with the kernel's in-run span from s_memrealtime stamps (hfem_plan_set_span_stamps) and, labelled `rocprof_*`, the
profiler's average from the committed run.  `roofline.traffic` is measured IN THE RUN (two `rocprofv3 --pmc` child passes of the
replayed leg: FETCH_SIZE x 2 + WRITE_SIZE) when rocprofv3 is present, else the committed figure, labelled "committed".
``cpu_baseline`` times the oracle's op-for-op PyTorch restatement of the reference chain on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

np = torch = None            # imported by main(): the self-launching parent of an N > 1 run needs neither (and must not touch the GPU)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r04")      # committed rocprofv3 summaries of this round (r03 as fallback)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=9, help="replays of the K-step graph, each timed on its own; the median is reported")
    ap.add_argument("--nx", type=int, default=1001, help="nodes along x PER GPU (+1 shared column)")
    ap.add_argument("--ny", type=int, default=501)
    ap.add_argument("--tile-elems", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph of K steps")
    ap.add_argument("--cpu-evals", type=int, default=2)
    ap.add_argument("--inline-loss-sum", action="store_true", help="N = 1: reduce the tile energies with a separate "
                    "1-block launch after every energy kernel instead of inside the next launch")
    ap.add_argument("--prewarm", type=float, default=0.75, help="seconds of untimed replays before each timed leg")
    ap.add_argument("--option", action="append", default=[], help="name=value for hfem_set_option (lab A/B runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip config.extra / train_step_1gpu / strong_scaling_emulated")
    ap.add_argument("--no-regimes", action="store_true", help="skip the cache-regime legs (roofline = the replayed leg)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip config.strong_scaling")
    ap.add_argument("--strong-cfg5u", action="store_true", help="N > 1: run the 4.1 M-element Delaunay strong-scaling leg at any N "
                    "(default: from N = 8; ~2 min of host-side meshing and planning per rank)")
    ap.add_argument("--time-budget", type=float, default=480.0, help="seconds: optional N > 1 legs are skipped, with a note, when the run "
                    "is already past a share of it -- the peer-window legs and the sharded L-BFGS leg past 60 %, the strong-scaling legs "
                    "past 70 %, the 4.1 M-element strong-scaling mesh (~2 min of host-side meshing and planning per rank) past half")
    ap.add_argument("--deadline", type=float, default=540.0, help="self-launched N > 1 runs: seconds after which the parent stops the "
                    "ranks and prints rank 0's last PROVISIONAL line (headline + the legs finished so far, marked `partial`) instead of nothing")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not measure roofline.traffic in the run (two rocprofv3 --pmc child "
                    "passes of the replayed leg, ~15 s each); the committed figure is reported instead, labelled so")
    ap.add_argument("--no-lbfgs", action="store_true", help="skip the L-BFGS legs (config.lbfgs_step, lbfgs_step_1gpu, lbfgs_sharded_emulated)")
    ap.add_argument("--no-peer", action="store_true", help="skip the peer-window exchange legs (config.peer_exchange / sharded_step_1gpu)")
    ap.add_argument("--only-regime", default="", help="profiler helper: run ONLY this roofline leg (replayed | "
                    "rewritten_inputs | rotating_sets) and exit")
    ap.add_argument("--only-extra", default="", help="profiler helper: run ONLY this config.extra workload (Q1M | T2M | cfg5 | "
                    "cfg5auto | cfg5r | cfg5u) and exit")
    ap.add_argument("--emulate-shard", default="", help="N = 1: 'r/N' -- time the kernel over the tile range rank r of N "
                    "would evaluate on the FIXED 10^6-element mesh (strong-scaling rehearsal) and exit")
    ap.add_argument("--reorder", default="auto", help="row storage order of the triangular models (auto | hilbert | off); A/B runs")
    ap.add_argument("--rotating-sets", type=int, default=10, help="parameter/gradient sets of the rotating leg (32 MB each)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "for rehearsing the multi-rank path on fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(mesh6, u_free, n_evals):
    """oracle/ref_chain.py (the reference's ATen op chain + autograd) on the host cores."""
    from oracle import ref_chain as R
    coords, conn, geom, bc, mn, edges = mesh6
    # the GPU box gives one GPU's share of the host (16 cores); more threads than that only thrash
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    mesh = dict(n_nodes=coords.shape[0], conn=conn, free_mask=~geom, boundary_mask=geom,
                coords_fixed=coords[geom], u_free_mask=~bc, dirichlet_mask=bc,
                u_fixed=torch.tensor(0.0, dtype=torch.float64), edges=edges)
    xf, uf = coords[~geom].clone(), u_free.clone()
    R.energy_and_grads(xf, uf, mesh)                       # warm-up
    best = float("inf")
    for _ in range(n_evals):
        t0 = time.perf_counter()
        loss, gx, gu = R.energy_and_grads(xf, uf, mesh)
        best = min(best, time.perf_counter() - t0)
    return dict(value=conn.shape[0] / best, unit="element-evals/s", cores=torch.get_num_threads(), kind="port",
                sample=f"same workload ({conn.shape[0]} TRI3, fp64), best of {n_evals} fwd+bwd evaluations "
                       f"after 1 warm-up, {best:.3f} s/eval"), loss.item()


def pmc_traffic_in_run(kernel_substr, timeout_s=150.0):
    """FETCH_SIZE / WRITE_SIZE per launch of the dominant kernel, measured now: this script is run twice as a CHILD under
    ``rocprofv3 --kernel-trace --pmc <counter>`` (one counter per pass, nothing but --kernel-trace beside it, the program itself
    after ``--``), on the replayed leg only.  Returns (dict, None) or (None, reason).  Never raises; a pass that hangs is
    killed with its process group."""
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hfem_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--no-cpu-baseline", "--steps", "50", "--only-regime", "replayed", "--prewarm", "0.02"]
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return None, f"the {ctr} pass exceeded {timeout_s:.0f} s"
            if rc != 0:
                return None, f"the {ctr} pass exited with {rc}"
            vals = []
            for root, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        with open(os.path.join(root, fn)) as f:
                            for r in csv.DictReader(f):
                                if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                                    vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, f"the {ctr} pass recorded no launch of {kernel_substr}"
            out[ctr + "_KB"] = sum(vals) / len(vals)
            out["launches"] = len(vals)
        except Exception as e:  # noqa: BLE001
            return None, f"{type(e).__name__}: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["read_bytes_corrected"] = 2.0 * out["FETCH_SIZE_KB"] * 1024.0
    out["write_bytes"] = out["WRITE_SIZE_KB"] * 1024.0
    out["traffic_bytes_per_launch"] = out["read_bytes_corrected"] + out["write_bytes"]
    return out, None


def self_launch(a):
    """``python bench.py --gpus N`` (N > 1) without a launcher: THIS process -- which has made no GPU call and makes none --
    starts the N ranks as CHILDREN through ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1), relays
    rank 0's JSON line and returns the launcher's exit code (non-zero if any rank failed).  Never an exec.  The plan cache
    directory is created here, so that the N ranks build every host plan ONCE (TilePlan(cache_dir=...))."""
    import shutil
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL / peer windows across processes
    made = None
    if not env.get("HFEM_PLAN_CACHE"):
        base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
        made = env["HFEM_PLAN_CACHE"] = tempfile.mkdtemp(prefix="hfem_plans_", dir=base)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {a.gpus} without WORLD_SIZE: starting {a.gpus} rank processes as children ({' '.join(cmd[1:8])} ...)",
          file=sys.stderr, flush=True)
    # A leg that hangs or kills a rank on hardware nobody could rehearse on must not cost the whole measurement: rank 0 rewrites
    # a PROVISIONAL line (headline step, roofline, the legs finished so far) after every section; if the ranks fail or outlive
    # --deadline, the parent stops them (its own children: their process group) and prints that line, marked `partial`.
    import signal
    import threading
    prov = env["HFEM_BENCH_PROVISIONAL"] = os.path.join(tempfile.gettempdir(), f"hfem_bench_provisional_{os.getpid()}.json")
    line, why = [None], None
    try:
        def own_group_and_die_with_parent():                  # in the child, before exec: a session of its own (so that the parent
            os.setsid()                                         # can stop launcher + ranks together) that still ends if the parent
            try:                                                # is killed outright (PR_SET_PDEATHSIG; the launcher sets the same
                import ctypes                                   # for its ranks)
                ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM))
            except OSError:
                pass
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, preexec_fn=own_group_and_die_with_parent)   # stderr passes through

        def forward(signum, frame):                             # the driver stops the parent: the ranks go with it
            try:
                os.killpg(pr.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
            raise SystemExit(128 + signum)
        for sg in (signal.SIGTERM, signal.SIGINT):
            try:
                signal.signal(sg, forward)
            except ValueError:                                  # not the main thread (never the case for `python bench.py`)
                pass

        def pump():
            for ln in pr.stdout:
                if ln.startswith("{") and '"metric"' in ln:
                    line[0] = ln.strip()                                           # rank 0's one JSON line
                else:
                    sys.stderr.write(ln)
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        try:
            rc = pr.wait(timeout=a.deadline)
        except subprocess.TimeoutExpired:
            why = f"the ranks were still running after --deadline {a.deadline:.0f} s and were stopped"
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(pr.pid, sig)
                except (ProcessLookupError, PermissionError, AttributeError):
                    pass
                try:
                    rc = pr.wait(timeout=20)
                    break
                except subprocess.TimeoutExpired:
                    rc = -9
        th.join(timeout=10)
        if line[0] is None and why is None and rc != 0:
            why = f"a rank failed (launcher exit code {rc})"
        if line[0] is None and why is not None and os.path.exists(prov):
            try:
                with open(prov) as f:
                    d = json.loads(f.read())
                d["partial"] = f"{why}; this is rank 0's provisional line, written after '{d.get('partial')}'"
                d.setdefault("config", {}).setdefault("notes", []).append("PARTIAL RESULT: " + d["partial"])
                print(f"[bench] {why}: printing the provisional line", file=sys.stderr, flush=True)
                line[0], rc = json.dumps(d), 0
            except (OSError, ValueError):
                pass
    finally:
        if made:
            shutil.rmtree(made, ignore_errors=True)
        try:
            os.unlink(prov)
        except OSError:
            pass
    if rc == 0 and line[0] is None:
        print("[bench] the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if line[0] is not None and rc == 0:
        print(line[0], flush=True)
    return rc


def main():
    t_start = time.perf_counter()
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    global np, torch
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm device; there is no CPU fallback"
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and world > ndev:
        sys.exit(f"bench.py: {world} ranks but {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy

    f64 = torch.float64
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    notes = []                                   # everything that did not run the way the docstring says (-> config.notes)

    def note(msg):
        notes.append(msg)
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr)

    wall = {}                                    # section -> seconds since start when it finished (rank 0's clock) -> config.wall_s

    prov_state = {}                              # what rank 0's provisional line holds so far (self-launched N > 1 runs)

    def provisional(section):
        """Rewrite the provisional result line (self_launch: $HFEM_BENCH_PROVISIONAL): headline, roofline and notes as of now."""
        path = os.environ.get("HFEM_BENCH_PROVISIONAL")
        if not path or rank != 0 or world == 1 or "value" not in prov_state:
            return
        d = dict(metric="element-evals/sec (fwd+bwd energy) + achieved HBM GB/s, 2D quad mesh", value=prov_state["value"],
                 unit="element-evals/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=prov_state["ms_per_step"],
                 higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                 config=dict(prov_state.get("config", {}), notes=list(notes), wall_s=dict(wall)), roofline=prov_state.get("roofline"),
                 partial=section)
        try:
            with open(path + ".tmp", "w") as f:
                f.write(json.dumps(d))
            os.replace(path + ".tmp", path)
        except OSError:
            pass

    def progress(section):
        """One stderr line per finished section with the elapsed time: a driver log shows where an N-rank run spends its limit."""
        wall[section] = round(time.perf_counter() - t_start, 1)
        if rank == 0:
            print(f"[bench] {wall[section]:7.1f} s  {section}", file=sys.stderr, flush=True)
        provisional(section)

    for kv in a.option:
        name, val = kv.split("=")
        _lib.check(L.hfem_set_option(name.encode(), int(val)), "hfem_set_option")

    # ---- N > 1: the job's plan cache.  A host plan costs ~1 s per 10^6 elements; every rank needs the same ones (the model's
    #      default tiling for its row order, the plan sharded N ways), so rank 0 builds them and writes the blobs, the other
    #      ranks deserialise (TilePlan(cache_dir=$HFEM_PLAN_CACHE), hfem_plan_deserialize).  The self-launching parent made the
    #      directory; under an external launcher rank 0 makes it here.
    cache_made = None
    if world > 1:
        box = [os.environ.get("HFEM_PLAN_CACHE") or None]
        if rank == 0 and box[0] is None:
            import tempfile
            box[0] = cache_made = tempfile.mkdtemp(prefix="hfem_plans_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
        dist.broadcast_object_list(box, src=0)
        os.environ["HFEM_PLAN_CACHE"] = box[0]

    def staged(fn):
        """fn() on rank 0 first (it builds and caches the host plans), then on the other ranks (cache hits)."""
        if world == 1:
            return fn()
        out = fn() if rank == 0 else None
        dist.barrier()
        return out if rank == 0 else fn()

    # ------------------------------------------------------------------------------------------------ timing helpers
    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    peer_on = [False]      # the legs being timed exchange through peer windows (no collective: capturable on any backend)

    def capture(body, n, begin=None, end=None, warm=2):
        """One hipGraph of `n` calls of body() (bracketed by begin() / end()); (graph, None) or (None, reason)."""
        if a.no_graph:
            return None, "--no-graph"
        if world > 1 and comm is None and not peer_on[0]:
            return None, "torch.distributed collectives are not capturable; needs the in-library RCCL communicator"
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                if begin:
                    begin()
                for _ in range(warm):
                    body()
                if end:
                    end()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if begin:
                    begin()
                for _ in range(n):
                    body()
                if end:
                    end()
            return g, None
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize()
            return None, f"{type(e).__name__}: {str(e)[:160]}"

    def timed_steps(body, n, begin=None, end=None, fixed_prewarm=None):
        """W warm-up steps, then `--repeats` timed regions of EXACTLY n steps (one graph replay each), each bracketed by
        barrier + synchronize on both sides; per region the MAX over ranks; returns (median seconds, all regions, launch)."""
        g, why = capture(body, n, begin, end)

        def run():
            if g is not None:
                g.replay()
            else:
                if begin:
                    begin()
                for _ in range(n):
                    body()
                if end:
                    end()
        # untimed pre-warm: the whole default run is a few ms of GPU time, shorter than the clock ramp of an idle chip.
        # Fixed count when N > 1 (every rank must issue the same number of collectives).
        if world == 1 and g is not None:
            t_pw = time.perf_counter()
            while time.perf_counter() - t_pw < a.prewarm:
                g.replay()
                torch.cuda.synchronize()
        else:
            for _ in range(fixed_prewarm if fixed_prewarm is not None else max(1, int(a.prewarm * 400 / max(n, 1)))):
                run()
        if begin:
            begin()
        for _ in range(a.warmup + (a.warmup & 1)):      # W warm-up steps (rounded up to even: the fused steps alternate between
            body()                                      # two parameter buffers and a captured graph assumes the parity it saw)
        if end:
            end()
        run()
        regions = []
        for _ in range(max(1, a.repeats)):
            sync_all()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=f64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            regions.append(el)
        sync_all()
        launch = "hipgraph" if g is not None else f"eager ({why})"
        if g is None and why != "--no-graph" and f"eager: {why}" not in notes:
            note(f"eager: {why}")
        return sorted(regions)[len(regions) // 2], regions, launch

    stream_box = [torch.cuda.current_stream()]

    def time_launches(body, kreps):
        """Average time of one body(i) over `kreps` back-to-back calls (one hipGraph unless --no-graph), HIP events on the
        launch stream, median of 5 regions after the clock pre-warm.  Returns (us, regions)."""
        g = None
        if not a.no_graph:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    stream_box[0] = s
                    body(0)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    stream_box[0] = torch.cuda.current_stream()
                    for i in range(kreps):
                        body(i)
            except Exception as e:  # noqa: BLE001
                note(f"kernel-only leg: hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); eager launches")
                g = None
        stream_box[0] = torch.cuda.current_stream()
        for i in range(5):
            body(i)
        t_pw = time.perf_counter()
        while g is not None and time.perf_counter() - t_pw < a.prewarm:
            g.replay()
            torch.cuda.synchronize()
        regions = []
        for _ in range(5):                      # 5 timed regions of `kreps` back-to-back bodies each
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream_box[0])
            if g is not None:
                g.replay()
            else:
                for i in range(kreps):
                    body(i)
            ev1.record(stream_box[0])
            torch.cuda.synchronize()
            regions.append(ev0.elapsed_time(ev1) * 1e3 / kreps)
        return sorted(regions)[len(regions) // 2], regions       # median region; each value is a kreps-launch average

    def span_launches(plan, body, kreps):
        """In-run duration of the energy kernel inside ANY launch sequence: s_memrealtime stamps written by the kernel itself
        (hfem_plan_set_span_stamps), first-workgroup-start to last-workgroup-end per launch, averaged over `kreps` launches
        of a replayed graph.  Returns (span_us, gap_us) -- gap = from one launch's last end to the next launch's first
        start -- or (None, None) if the plan's kernel does not stamp."""
        nt = plan.n_tiles
        buf = torch.zeros(kreps * nt * 2, dtype=torch.int64, device=dev)
        _lib.check(L.hfem_plan_set_span_stamps(plan.handle, buf.data_ptr(), kreps), "hfem_plan_set_span_stamps")
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                stream_box[0] = s
                body(0)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            _lib.check(L.hfem_plan_set_span_stamps(plan.handle, buf.data_ptr(), kreps), "hfem_plan_set_span_stamps")   # cursor -> 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                stream_box[0] = torch.cuda.current_stream()
                for i in range(kreps):
                    body(i)
            stream_box[0] = torch.cuda.current_stream()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            buf.zero_()
            g.replay()
            torch.cuda.synchronize()
            sp = buf.view(kreps, nt, 2).cpu().numpy()
            ok = sp[:, :, 0] > 0
            if not ok.any():
                return None, None
            start = np.where(ok, sp[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
            end = np.where(ok, sp[:, :, 1], 0).max(axis=1)
            span = float((end - start).mean()) * 0.01                                  # 100 MHz ticks -> us
            gap = float((start[1:] - end[:-1]).mean()) * 0.01 if kreps > 1 else None
            del g
            return span, gap
        except Exception as e:  # noqa: BLE001
            note(f"span stamps failed: {type(e).__name__}: {str(e)[:120]}")
            return None, None
        finally:
            stream_box[0] = torch.cuda.current_stream()
            L.hfem_plan_set_span_stamps(plan.handle, None, 0)

    # ------------------------------------------------------------------------------------------------ workloads
    def build_model(mesh6):
        coords, conn, geom, bc, mn, edges = mesh6
        torch.manual_seed(0)
        return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                        neumann_edges=edges, reorder=a.reorder).to(dev)

    def t1m_mesh(nranks=1):
        nx = (a.nx - 1) * nranks + 1
        return structured_tri_mesh(nx, a.ny, length=2.0 * nranks, height=1.0, jitter=0.2, seed=0, dtype=f64)

    class KernelOnly:
        """hfem_tri3_energy_plan on a tile range with every host-side lookup hoisted (HFEM_FLAG_NO_LOSS_SUM)."""

        def __init__(self, model, loss_fn, plan, lo=0, hi=-1):
            self.plan, self.lo, self.hi = plan, int(lo), int(hi)
            self.xf, self.uf = model.node_coords_free.detach(), model.u_free.detach()
            self.xfix, self.ufix = model.node_coords_fixed, model.u_fixed_rows()
            _, Tconst = loss_fn._traction(model, None)
            self.mat, self.W, self.Bk, self.Tc = dv(loss_fn._mat), loss_fn._W, dv([0.0] * 6), dv(Tconst)
            self.gx, self.gu = torch.empty_like(self.xf), torch.empty_like(self.uf)
            self.loss = torch.zeros((), dtype=f64, device=dev)

        def __call__(self, bufs=None, flags=8, stream=None):
            x_, u_, gx_, gu_ = bufs if bufs is not None else (self.xf, self.uf, self.gx, self.gu)
            _lib.check(L.hfem_tri3_energy_plan(self.plan.handle, x_.data_ptr(), self.xfix.data_ptr() if self.xfix.numel() else None,
                                               u_.data_ptr(), self.ufix.data_ptr() if self.ufix.numel() else None, self.mat,
                                               self.W, self.Bk, None, self.Tc, self.lo, self.hi, self.loss.data_ptr(),
                                               gx_.data_ptr(), gu_.data_ptr(), flags,
                                               stream if stream is not None else stream_box[0].cuda_stream), "hfem_tri3_energy_plan")

    class RotatingStep:
        """The headline step (lagged loss sum and all) on R rotating parameter / gradient sets: step i reads set i mod R."""

        def __init__(self, ko, sets):
            self.ko, self.sets, self.i, self.lag = ko, sets, 0, False

        def begin(self):
            self.i, self.lag = 0, False

        def __call__(self):
            self.ko(self.sets[self.i % len(self.sets)], flags=8 | (32 if self.lag else 0), stream=torch.cuda.current_stream().cuda_stream)
            self.i, self.lag = self.i + 1, True

        def end(self):
            _lib.check(L.hfem_plan_loss_sum(self.ko.plan.handle, self.ko.lo, self.ko.hi if self.ko.hi >= 0 else self.ko.plan.n_tiles,
                                            self.ko.loss.data_ptr(), torch.cuda.current_stream().cuda_stream), "hfem_plan_loss_sum")
            self.lag = False

    def range_work(plan, lo, hi):
        """(home elements, owned nodes, algorithmic bytes) of a launch over tiles [lo, hi)."""
        td = plan.export("tile_desc")
        hi = plan.n_tiles if hi < 0 else hi
        if (lo, hi) == (0, plan.n_tiles):
            ne_ = plan.n_elems
        else:
            ne_ = int(sum(int(plan.tile_elements(t)[2].sum()) for t in range(lo, hi)))
        nn_ = int(td[lo:hi, 4].sum())
        return ne_, nn_, 12 * ne_ + 64 * nn_ + 8

    class AdamRange:
        """hfem_tri3_energy_adam_step_ex over a tile range (energy + Adam at write-out, no exchange): launch i reads parameter
        buffer i mod 2 and writes the rows the range owns into the other one (both start complete; lr tiny)."""

        def __init__(self, ko):
            self.ko = ko
            self.x, self.u = [ko.xf.clone(), ko.xf.clone()], [ko.uf.clone(), ko.uf.clone()]
            self.st = [torch.zeros_like(ko.xf), torch.zeros_like(ko.xf), torch.zeros_like(ko.uf), torch.zeros_like(ko.uf)]
            self.bc = torch.tensor([0.1, 0.0316], dtype=f64, device=dev)        # bias corrections of step 1 (held fixed: timing only)

        def __call__(self, i):
            k, i_, o_ = self.ko, i & 1, (i + 1) & 1
            _lib.check(L.hfem_tri3_energy_adam_step_ex(
                k.plan.handle, 0, self.x[i_].data_ptr(), k.xfix.data_ptr() if k.xfix.numel() else None, self.u[i_].data_ptr(),
                k.ufix.data_ptr() if k.ufix.numel() else None, k.mat, float(k.W), None, None, k.Tc, self.x[o_].data_ptr(),
                self.u[o_].data_ptr(), self.st[0].data_ptr(), self.st[1].data_ptr(), self.st[2].data_ptr(), self.st[3].data_ptr(),
                1e-9, 1e-12, 0.9, 0.999, 1e-8, self.bc.data_ptr(), k.lo, k.hi, k.loss.data_ptr(), 8, stream_box[0].cuda_stream),
                "hfem_tri3_energy_adam_step_ex")

    def shard_kernel_leg(model, loss_fn, nshards, ranks, kreps):
        """Kernel-only time of the tile ranges `ranks` of an `nshards`-way sharded plan of `model` (one GPU): the energy launch,
        and the fused energy + Adam launch (the whole compute of the rank's one-launch training step; its in-launch put / get
        of ~50 boundary tiles' interface rows is not emulated)."""
        plan = model.tile_plan(loss_fn.tile_elems, shards=nshards)
        out = []
        for r in ranks:
            lo, mid, hi = plan.shard_parts(r, nshards)
            ko = KernelOnly(model, loss_fn, plan, lo, hi)
            us, _ = time_launches(lambda i: ko(), kreps)
            ar = AdamRange(ko)
            us_adam, _ = time_launches(ar, kreps + (kreps & 1))
            del ar
            ne_, nn_, ab = range_work(plan, lo, hi)
            out.append(dict(rank=r, tiles=hi - lo, boundary_tiles=mid - lo, elements=ne_, owned_nodes=nn_, kernel_us=us,
                            alg_bytes=ab, frac=ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, energy_adam_launch_us=us_adam))
        st = plan.stats
        return dict(shards=nshards, tiles=st["n_tiles"], threads_per_tile=st["threads_per_tile"], slot_rows=st["slot_rows"],
                    max_tile_owned=st["max_tile_owned"], ranks=out, kernel_us_max=max(o["kernel_us"] for o in out))

    # ------------------------------------------------------------------------------------------------ helper modes (N = 1)
    if a.emulate_shard:
        r_, n_ = (int(v) for v in a.emulate_shard.split("/"))
        model = build_model(t1m_mesh())
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
        print(json.dumps(dict(emulate_shard=a.emulate_shard, **shard_kernel_leg(model, lf, n_, [r_], max(a.steps, 200)))), flush=True)
        return None

    # ------------------------------------------------------------------------------------------------ main workload
    mesh6 = t1m_mesh(world)
    coords, conn, geom, bc, mn, edges = mesh6
    ne, nn = conn.shape[0], coords.shape[0]
    t_plan0 = time.perf_counter()
    model = staged(lambda: build_model(mesh6))
    loss_fn = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
    comm, comm_state = None, "none (N = 1)"
    if world > 1:
        comm_state = f"torch.distributed {a.backend}"
        if a.backend == "nccl":      # in-library RCCL communicator: collectives on the kernel's stream, capturable
            try:
                comm = LibraryComm(dev)
                # self-check against torch.distributed on this very communicator size before anything is timed
                probe = torch.arange(6, dtype=f64, device=dev) + 100.0 * rank
                got_lib, got_pg = torch.zeros(6 * world, dtype=f64, device=dev), torch.zeros(6 * world, dtype=f64, device=dev)
                comm.all_gather(probe, got_lib)
                dist.all_gather_into_tensor(got_pg, probe)
                red_lib = torch.zeros_like(probe)
                comm.all_reduce_sum(probe, red_lib)
                red_pg = probe.clone()
                dist.all_reduce(red_pg)
                torch.cuda.synchronize()
                if not (torch.equal(got_lib, got_pg) and torch.allclose(red_lib, red_pg, rtol=1e-15, atol=0)):
                    raise RuntimeError("in-library collectives disagree with torch.distributed")
                comm_state = f"in-library RCCL (hfem_mg_*), verified in this run against torch.distributed on {world} ranks"
            except Exception as e:  # noqa: BLE001
                comm = None
                comm_state = f"torch.distributed nccl (in-library RCCL communicator unavailable: {type(e).__name__}: {str(e)[:120]})"
                note(comm_state)
            agreed = torch.tensor([1.0 if comm is not None else 0.0], dtype=f64, device=dev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)      # one verdict for all ranks: a mixed set of transports would hang
            if agreed.item() != 1.0 and comm is not None:
                comm.close()
                comm = None
                comm_state = "torch.distributed nccl (in-library RCCL communicator failed its check on another rank)"
                note(comm_state)
    sh = staged(lambda: ShardedTri3Energy(model, loss_fn, comm=comm))
    plan = sh.plan
    lo, hi = sh.lo, sh.hi
    plan_cache_state = None
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, (getattr(model, "plan_cache", None), plan.cache))
        plan_cache_state = dict(how="rank 0 builds each host plan and writes its blob, the other ranks deserialise it "
                                    "(hfem_plan_serialize / hfem_plan_deserialize)",
                                per_rank_model_plan_and_sharded_plan=got, blob_bytes=int(plan.to_bytes().size) if rank == 0 else None,
                                model_and_plans_s=round(time.perf_counter() - t_plan0, 2))

    progress("mesh, model, host plans")
    if world > 1:
        sh.setup_interfaces()
        sh.init_owner_adam(lr_x=1e-9, lr_u=1e-12, fused=True)      # tiny steps: the mesh stays valid over any number of iterations

    def step_dense():           # north-star literal: one all-reduce of [gX|gU|loss] (every rank gets everything)
        sh.evaluate_local()
        sh.exchange()

    # N = 1: the energy of step k is reduced by an extra workgroup of launch k+1 (HFEM_FLAG_SUM_PREVIOUS) and the last one
    # by a trailing 1-block launch, inside the timed region: every step's loss is produced, the reduction and its kernel
    # boundary just leave the critical path.  --inline-loss-sum restores the separate reduction after every launch.
    lagged = world == 1 and not a.inline_loss_sum
    step = (sh.evaluate_local_lagged if lagged else sh.evaluate_local) if world == 1 else sh.owner_step
    begin = sh.begin_lagged if lagged else None
    end = sh.flush_loss if lagged else None

    # ---- correctness guard: the benchmarked path must produce the oracle's numbers
    if begin:
        begin()
    step()
    if end:
        end()
    torch.cuda.synchronize()
    if world == 1:
        loss_gpu = sh._views(sh.send)[0].item()
    else:                       # the two independent exchange paths must agree on the global energy
        loss_gpu = sh.loss_global.item()
        step_dense()
        torch.cuda.synchronize()
        loss_dense = sh._views(sh.recv)[0].item()
        assert abs(loss_gpu - loss_dense) <= 1e-12 * abs(loss_dense), (loss_gpu, loss_dense)

    only = a.only_regime
    if not only and not a.only_extra:
        elapsed, regions, launch = timed_steps(step, a.steps, begin, end)
        if lagged:                  # the trailing flush delivered the last step's energy: same bits as the guard evaluation
            assert sh._views(sh.send)[0].item() == loss_gpu, (sh._views(sh.send)[0].item(), loss_gpu)
    else:
        elapsed, regions, launch = float("nan"), [], "skipped"
    ms_per_step = elapsed / a.steps * 1e3
    value = ne / (elapsed / a.steps)          # whole-job element-evals/s (all ranks' elements)

    # kernel-only time of this rank's launch (the roofline's numerator), taken HERE so that a multi-rank run has it on record
    # before its first exchange leg
    kreps = max(a.steps, 200)      # kernel-only legs: >= 200 launches per graph whatever --steps is (amortises the ~20 us graph launch)
    ko_main = KernelOnly(model, loss_fn, plan, lo, hi)
    k_us, samples = time_launches(lambda i: ko_main(), kreps) if only in ("", "replayed") and not a.only_extra else (float("nan"), [])
    if world > 1:
        _, _, ab_ = range_work(plan, lo, hi)
        prov_state.update(value=value, ms_per_step=ms_per_step,
                          config=dict(workload=f"Example 4 (T1M x {world}): {ne} TRI3, {nn} nodes, fwd+bwd (loss + dX + dU), owner-sharded",
                                      elements=ne, nodes=nn, elements_per_gpu=ne // world, launch=launch, collectives=comm_state,
                                      exchange_mode="serial: evaluation -> pack -> all_gather -> unpack on one stream"),
                          roofline=dict(bound="hbm", achieved=ab_ / (k_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=ab_ / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, traffic=None, kernel_us=k_us,
                                        alg_bytes_per_launch=ab_, regime="replayed; this rank's tile range of the weak-scaling mesh"))
    progress("headline step timed")
    # ---- N = 1, reported beside the headline: the step of a caller that needs ITS OWN loss before it goes on (a line
    #      search, an L-BFGS check): the 1-block reduction launched right after every energy launch, on the critical path
    inline_step = None
    if world == 1 and lagged and not a.no_extra and not only and not a.only_extra:
        el_, _, ln_ = timed_steps(sh.evaluate_local, a.steps)
        assert sh._views(sh.send)[0].item() == loss_gpu
        inline_step = dict(mode="loss of step k reduced by its own 1-block launch before step k+1 (--inline-loss-sum)",
                           ms_per_step=el_ / a.steps * 1e3, value=ne / (el_ / a.steps), launch=ln_)

    # ---- N > 1: the other readings of a multi-GPU step
    def leg(body, mode, n_elems, begin_=None, end_=None):
        el, regs, ln = timed_steps(body, a.steps, begin_, end_)
        return dict(mode=mode, value=n_elems / (el / a.steps), ms_per_step=el / a.steps * 1e3, launch=ln,
                    ms_per_step_replays=[round(r / a.steps * 1e3, 5) for r in regs])

    def lbfgs_leg(sh_, what, emulate=False, outer=8):
        """Example 4's optimiser (examples/example4.py:68-78: torch.optim.LBFGS defaults, 20 inner iterations per step, history
        100) NODE-SHARDED over the ranks of sh_ (optim.ShardedLBFGS): ms per inner iteration with the history full (outer steps
        6-8), max over ranks.  The parameters are restored afterwards."""
        from hidenn_fem_amd.optim import ShardedLBFGS
        m_ = sh_.model
        keep = (m_.node_coords_free.detach().clone(), m_.u_free.detach().clone())
        opt = ShardedLBFGS(sh_, emulate=emulate)
        ts, its = [], []
        for _ in range(outer):
            if not emulate:
                sync_all()
            else:
                torch.cuda.synchronize()
            n0, t0 = opt.state["n_iter"], time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1 and not emulate:
                t = torch.tensor([el], dtype=f64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            ts.append(el)
            its.append(opt.state["n_iter"] - n0)
        opt.finish()
        full = [(t_, i_) for t_, i_ in list(zip(ts, its))[5:] if i_ > 0] or [(t_, i_) for t_, i_ in zip(ts, its) if i_ > 0]
        per = sorted(t_ / i_ for t_, i_ in full)[len(full) // 2]
        n_loc, n_par = opt._n, m_.node_coords_free.numel() + m_.u_free.numel()
        byts = 4.0 * 100 * n_loc * m_.node_coords_free.element_size()
        out_ = dict(workload=what, parameters=n_par, parameters_this_rank=n_loc, ms_per_inner_iteration=per * 1e3,
                    inner_iterations_per_step=its, history_pairs=int(opt.status()[5]),
                    local_history_bytes_per_iteration=byts, achieved_this_rank=byts / per / 1e9, unit="GB/s",
                    frac_this_rank=byts / per / 1e9 / HBM_PEAK_GBS,
                    exchanges_per_iteration="2 small ones: the interface parameter rows before the energy launch, one L-BFGS payload "
                                            f"({opt._payload.numel() * 8} B per rank: per-slot dots, y.s, y.y, gradient statistics, max|d|, "
                                            "partial energy) gathered to every rank and summed there in rank order")
        with torch.no_grad():
            m_.node_coords_free.copy_(keep[0])
            m_.u_free.copy_(keep[1])
        del opt
        torch.cuda.empty_cache()
        return out_

    alt = train = train_ov = strong = eval_ov = train_fused = train_fused_ov = lbfgs_n = None
    exchange_mode = "serial: evaluation -> pack -> all_gather -> unpack on one stream"
    if world > 1:
        how = comm_state
        eval_ov = leg(sh.owner_step_overlapped, "evaluation + interface exchange, the exchange of step k on a side stream under the "
                      "interior tiles of step k+1 (fork / join inside the hipGraph: ~15 us per step on one rank, so it pays "
                      "when the exposed exchange latency is larger)", ne, end_=sh.finish_overlapped)
        # the headline is the faster of the two complete evaluation + exchange steps on THIS topology; both are reported
        serial = dict(value=value, ms_per_step=ms_per_step, launch=launch, ms_per_step_replays=[round(r / a.steps * 1e3, 6) for r in regions])
        if eval_ov["value"] > value:
            value, ms_per_step, launch = eval_ov["value"], eval_ov["ms_per_step"], eval_ov["launch"]
            regions = [r * a.steps * 1e-3 for r in eval_ov["ms_per_step_replays"]]
            exchange_mode = "overlapped: the exchange of step k under the interior tiles of step k+1 (faster than the serial step on this topology)"
        alt = leg(step_dense, f"dense: sum all-reduce of [gX|gU|loss] fp64, {sh.send.numel() * 8} B per rank, {how}", ne)
        train = leg(sh.owner_train_step, "owner-sharded training iteration: energy -> Adam on the rows the rank owns -> pack "
                    "(+ energy sum + step count) -> all_gather -> unpack; four launches + the collective, exchange on the "
                    "critical path", ne)
        train_ov = leg(sh.owner_train_step_overlapped, "the same iteration, exchange of step k on a side stream under the "
                       f"interior tiles of step k+1 (boundary tiles {sh.mid - sh.lo} of {sh.hi - sh.lo} on this rank)", ne,
                       end_=sh.finish_overlapped)
        if a.steps % 2 == 0:          # the fused steps alternate between two parameter buffers: an even number per hipGraph
            train_fused = leg(sh.owner_train_step_fused, "owner-sharded training iteration with Adam applied by the energy kernel's "
                              "write-out (hfem_tri3_energy_adam_step_ex on the rank's tile range): energy+Adam -> pack -> all_gather "
                              "-> unpack, three launches + the collective, no gradient traffic", ne)
            train_fused_ov = leg(sh.owner_train_step_fused_overlapped, "the fused iteration with the exchange under the next step's "
                                 "interior tiles", ne, end_=sh.finish_overlapped)
        else:
            note("fused training legs skipped: --steps must be even (ping-pong parameter buffers inside one hipGraph)")

    if world > 1:
        prov_state.update(value=value, ms_per_step=ms_per_step)
        prov_state["config"].update(exchange_mode=exchange_mode, launch=launch)
        progress("collective-path legs")
    def over_budget(share, what):
        """True (on every rank: the legs contain collectives) when more than `share` of --time-budget is already spent; notes it."""
        spent_ = torch.tensor([time.perf_counter() - t_start], dtype=f64, device=dev)
        dist.all_reduce(spent_, op=dist.ReduceOp.MAX)
        if spent_.item() > share * a.time_budget:
            note(f"{what} skipped ({spent_.item():.0f} s of the {a.time_budget:.0f} s --time-budget already spent)")
            return True
        return False

    # ---- N > 1: the same steps with the interface rows written straight into the peers' receive windows (csrc/peer.hip): no
    #      collective, no second stream.  Verified in this run against the collective path before anything is timed; a failure
    #      (IPC mapping, a flag that never arrives) is reported in notes and the legs are dropped -- never a silent fallback.
    peer_legs, peer_state = None, None

    def enable_peer(sh_, tag):
        """Switch sh_ to peer-window exchange after checking it against the collective path on this very topology."""
        l_ref = sh_.owner_step()[0].item()
        sh_.enable_peer_exchange(timeout_s=5.0)
        l_got = sh_.owner_step()[0].item()
        torch.cuda.synchronize()
        st = sh_.peer.status()
        ok = torch.tensor([1.0 if (st[0] == 0 and l_got == l_ref) else 0.0], dtype=f64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() != 1.0:
            sh_.close_peer_exchange(check=False)
            raise RuntimeError(f"{tag}: peer-window exchange disagrees with the collective path on some rank "
                               f"(this rank: status {st}, energy {l_got!r} vs {l_ref!r})")

    if world > 1 and not a.no_peer and over_budget(0.6, "peer-window legs"):
        a.no_peer = True
    if world > 1 and not a.no_peer:
        try:
            enable_peer(sh, "T1M x N")
            peer_on[0] = True
            peer_state = f"verified in this run against the collective path on {world} ranks (same global energy, to the bit)"
            peer_legs = dict(
                eval_exchange=leg(sh.owner_step, "evaluation + interface exchange: energy -> put (pack + energy sum + stores into every "
                                  "rank's window + flags) -> get (wait for the flags, unpack); three launches, one stream", ne),
                eval_exchange_overlap=leg(sh.owner_step_overlapped, "the same with the get of step k after the interior tiles of step k+1",
                                          ne, end_=sh.finish_overlapped),
                train_step=leg(sh.owner_train_step, "Adam iteration: energy -> Adam on owned rows -> put -> get", ne),
                train_step_overlap=leg(sh.owner_train_step_overlapped, "Adam iteration, the get of step k after the interior tiles of "
                                       "step k+1", ne, end_=sh.finish_overlapped))
            if a.steps % 2 == 0:
                peer_legs["train_step_fused"] = leg(sh.owner_train_step_fused, "Adam inside the energy launch -> put -> get", ne)
                peer_legs["train_step_fused_overlap"] = leg(sh.owner_train_step_fused_overlapped, "Adam inside the two energy launches, "
                                                            "the get after the next step's interior tiles", ne, end_=sh.finish_overlapped)
            st = torch.tensor([float(sh.peer.status()[0]), sh.verify_interfaces()], dtype=f64, device=dev)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            if st[0].item() != 0.0:
                raise RuntimeError("a rank timed out waiting for a peer's flags during the timed legs")
            if st[1].item() != 0.0:
                raise RuntimeError(f"after the peer-window legs a rank's copy of an interface row differs from its owner's by {st[1].item():.3e}")
            peer_legs["one_launch_step"] = bool(sh.inkernel_put)
            peer_legs["interface_rows_verified"] = ("after the timed legs every rank compared the interface rows it reads with the owners' "
                                                    "current rows over torch.distributed: identical")
            best = max(("eval_exchange", "eval_exchange_overlap"), key=lambda k: peer_legs[k]["value"])
            if peer_legs[best]["value"] > value:
                value, ms_per_step, launch = peer_legs[best]["value"], peer_legs[best]["ms_per_step"], peer_legs[best]["launch"]
                regions = [r * a.steps * 1e-3 for r in peer_legs[best]["ms_per_step_replays"]]
                exchange_mode = ("peer windows" + (", the get under the next step's interior tiles" if best.endswith("overlap") else "")
                                 + ": interface rows stored by the pack launch into every rank's window over xGMI (faster than the "
                                 "all_gather steps on this topology; config.peer_exchange)")
        except Exception as e:  # noqa: BLE001
            peer_legs = None
            peer_state = f"unavailable: {type(e).__name__}: {str(e)[:200]}"
            note("peer-window exchange " + peer_state)
            sh.close_peer_exchange(check=False)
        peer_on[0] = False

    if world > 1 and not a.no_peer:
        prov_state.update(value=value, ms_per_step=ms_per_step)
        prov_state["config"].update(exchange_mode=exchange_mode, launch=launch, peer_exchange=peer_state)
        progress("peer-window legs")
    # ---- N > 1: Example 4's own optimiser, node-sharded (weak: N x 10^6 elements, every rank keeps 2 x 10^6 parameters' history)
    if world > 1 and not a.no_lbfgs and over_budget(0.6, "sharded L-BFGS legs"):
        a.no_lbfgs = True
    if world > 1 and not a.no_lbfgs:
        try:
            was_peer = sh.peer is not None
            if was_peer:
                sh.close_peer_exchange(check=False)      # the L-BFGS legs exchange over the collective path
            lbfgs_n = lbfgs_leg(sh, f"T1M x {world} (weak), ShardedLBFGS as examples/example4.py drives torch.optim.LBFGS, history full")
        except Exception as e:  # noqa: BLE001
            note(f"lbfgs_step leg failed: {type(e).__name__}: {str(e)[:160]}")

    if world > 1 and not a.no_lbfgs:
        progress("sharded L-BFGS leg (weak)")
    # ---- N > 1: BASELINE configs[3] / [4] as stated -- a FIXED mesh sharded over the N ranks (strong scaling)
    def strong_leg(name, mesh6_s):
        m_s = staged(lambda: build_model(mesh6_s))
        lf_s = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
        sh_s = staged(lambda: ShardedTri3Energy(m_s, lf_s, comm=comm))
        sh_s.setup_interfaces()
        sh_s.init_owner_adam(lr_x=1e-9, lr_u=1e-12, fused=True)
        ne_s, nn_s = mesh6_s[1].shape[0], mesh6_s[0].shape[0]
        ko = KernelOnly(m_s, lf_s, sh_s.plan, sh_s.lo, sh_s.hi)
        us, _ = time_launches(lambda i: ko(), max(a.steps, 200))
        ne_r, nn_r, ab = range_work(sh_s.plan, sh_s.lo, sh_s.hi)
        t = torch.tensor([us, float(ab) / (us * 1e-6) / 1e9], dtype=f64, device=dev)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        st = sh_s.plan.stats
        res = dict(workload=name, elements=ne_s, nodes=nn_s, elements_per_gpu=ne_s // world, tiles=st["n_tiles"],
                   tiles_per_rank=sh_s.hi - sh_s.lo, boundary_tiles_this_rank=sh_s.mid - sh_s.lo,
                   threads_per_tile=st["threads_per_tile"], slot_rows=st["slot_rows"],
                   payload_bytes=sh_s.interface_stats["payload_bytes"],
                   kernel_only=dict(kernel_us_max_over_ranks=tmax[0].item(), kernel_us_rank0=us,
                                    value=ne_s / (tmax[0].item() * 1e-6), achieved_GBs_all_ranks=tsum[1].item(),
                                    frac_of_N_x_8TBs=tsum[1].item() / (HBM_PEAK_GBS * world)),
                   eval_exchange=leg(sh_s.owner_step, "evaluation + interface exchange (the N = 1 definition of a step + the exchange)", ne_s),
                   eval_exchange_overlap=leg(sh_s.owner_step_overlapped, "the same, exchange under the next step's interior tiles", ne_s,
                                             end_=sh_s.finish_overlapped),
                   train_step=leg(sh_s.owner_train_step, "Adam iteration, exchange on the critical path", ne_s),
                   train_step_overlap=leg(sh_s.owner_train_step_overlapped, "Adam iteration, exchange under the next step's interior tiles",
                                          ne_s, end_=sh_s.finish_overlapped))
        if a.steps % 2 == 0:
            res["train_step_fused"] = leg(sh_s.owner_train_step_fused, "Adam inside the energy launch, exchange on the critical path", ne_s)
            res["train_step_fused_overlap"] = leg(sh_s.owner_train_step_fused_overlapped, "Adam inside the energy launches, exchange "
                                                  "under the next step's interior tiles", ne_s, end_=sh_s.finish_overlapped)
        if peer_legs is not None:          # the peer-window steps on the fixed mesh (verified again: another plan, other tables)
            try:
                enable_peer(sh_s, name[:12])
                peer_on[0] = True
                res["peer_exchange"] = dict(
                    eval_exchange=leg(sh_s.owner_step, "energy -> put -> get", ne_s),
                    eval_exchange_overlap=leg(sh_s.owner_step_overlapped, "the get after the next step's interior tiles", ne_s,
                                              end_=sh_s.finish_overlapped),
                    train_step=leg(sh_s.owner_train_step, "energy -> Adam -> put -> get", ne_s),
                    train_step_overlap=leg(sh_s.owner_train_step_overlapped, "the get after the next step's interior tiles", ne_s,
                                           end_=sh_s.finish_overlapped))
                if a.steps % 2 == 0:
                    res["peer_exchange"]["train_step_fused"] = leg(sh_s.owner_train_step_fused, "energy+Adam -> put -> get", ne_s)
                    res["peer_exchange"]["train_step_fused_overlap"] = leg(sh_s.owner_train_step_fused_overlapped,
                                                                           "the get after the next step's interior tiles", ne_s,
                                                                           end_=sh_s.finish_overlapped)
                res["peer_exchange"]["status"] = sh_s.peer.status()[0]
            except Exception as e:  # noqa: BLE001
                note(f"peer-window exchange on {name[:12]}: {type(e).__name__}: {str(e)[:160]}")
            sh_s.close_peer_exchange(check=False)
            peer_on[0] = False
        if not a.no_lbfgs:
            try:
                res["lbfgs_step"] = lbfgs_leg(sh_s, name[:12] + ": ShardedLBFGS, history full -- the optimiser's passes shrink by the number of ranks")
            except Exception as e:  # noqa: BLE001
                note(f"lbfgs_step on {name[:12]}: {type(e).__name__}: {str(e)[:160]}")
        del sh_s, m_s, ko
        return res

    if world > 1 and not a.no_strong and over_budget(0.7, "strong_scaling legs"):
        a.no_strong = True
    if world > 1 and not a.no_strong:
        strong = [strong_leg("T1M FIXED (BASELINE configs[3] as stated): 10^6 TRI3 sharded over the ranks", t1m_mesh(1))]
        spent = torch.tensor([time.perf_counter() - t_start], dtype=f64, device=dev)
        dist.all_reduce(spent, op=dist.ReduceOp.MAX)          # one decision for all ranks (the legs contain collectives)
        if (world >= 8 or a.strong_cfg5u) and spent.item() > 0.5 * a.time_budget:
            note(f"strong_scaling: cfg5u leg skipped ({spent.item():.0f} s of the {a.time_budget:.0f} s --time-budget already spent)")
        elif world >= 8 or a.strong_cfg5u:
            from hidenn_fem_amd.mesh import unstructured_tri_mesh
            strong.append(strong_leg("cfg5u FIXED (BASELINE configs[4]): ~4.1 x 10^6 TRI3 Delaunay mesh sharded over the ranks",
                                     unstructured_tri_mesh(2_050_000, seed=2, dtype=f64)))

    if world > 1 and not a.no_strong:
        progress("strong-scaling legs")
    # ------------------------------------------------------------------------------------------------ roofline legs
    # (ko_main, k_us, samples: timed right after the headline step)
    # The same kernel in the cache regimes a training loop sees:
    #  replayed:         the same buffers every launch (44 MB working set: Infinity-Cache resident) -- what the headline step is
    #  rewritten_inputs: x and u are rewritten by another kernel before every launch (what an optimiser step does); HIP events
    #                    only see the pair, so events give (rewrite + energy) - (rewrite alone), an UPPER bound (the rewrite
    #                    kernels themselves slow down inside the pair); the kernel's own in-run time comes from its stamps
    #  rotating_sets:    R parameter / gradient sets (R x 32 MB > the 256 MB Infinity Cache) visited round-robin: every read of
    #                    x, u and every gradient line misses the Infinity Cache (plan arrays stay shared) -- the HBM regime
    regimes = {}
    if world == 1 and not a.no_regimes and not a.only_extra:
        xf, uf = ko_main.xf, ko_main.uf

        def rewrite(i):
            with torch.cuda.stream(stream_box[0]):
                xf.mul_(1.0)
                uf.mul_(1.0)
        if only in ("", "replayed"):
            sp, gap = span_launches(plan, lambda i: ko_main(), kreps)
            regimes["replayed"] = dict(kernel_us=k_us, wg_span_us=sp, launch_gap_us=gap)
        if only in ("", "rewritten_inputs"):
            t_rw, _ = time_launches(rewrite, kreps)
            t_pair, _ = time_launches(lambda i: (rewrite(i), ko_main()), kreps)
            sp, gap = span_launches(plan, lambda i: (rewrite(i), ko_main()), kreps)
            regimes["rewritten_inputs"] = dict(pair_us=t_pair, rewrite_alone_us=t_rw, events_upper_bound_us=t_pair - t_rw,
                                               wg_span_us=sp)
        if only in ("", "rotating_sets"):
            R = max(2, a.rotating_sets)
            sets = [(xf.clone(), uf.clone(), torch.empty_like(ko_main.gx), torch.empty_like(ko_main.gu)) for _ in range(R)]
            t_rot, rot_regions = time_launches(lambda i: ko_main(sets[i % R]), kreps)
            sp, gap = span_launches(plan, lambda i: ko_main(sets[i % R]), kreps)
            regimes["rotating_sets"] = dict(kernel_us=t_rot, kernel_us_regions=[round(v, 3) for v in rot_regions], wg_span_us=sp,
                                            launch_gap_us=gap, sets=R,
                                            working_set_mb=round(R * 4 * xf.numel() * 8 / 2 ** 20 + plan.stats["device_bytes"] / 2 ** 20, 1))
            if not only and lagged:      # the WHOLE timed step (same K, same lagged loss sum, same bracketing) in this regime
                rs = RotatingStep(ko_main, sets)
                el_, regs_, ln_ = timed_steps(rs, a.steps, rs.begin, rs.end)
                regimes["rotating_sets"]["step"] = dict(ms_per_step=el_ / a.steps * 1e3, value=ne / (el_ / a.steps), launch=ln_,
                                                        ms_per_step_replays=[round(r_ / a.steps * 1e3, 6) for r_ in regs_])
            # for information: the same leg on a plan created with "store_policy" = 2 (nt gradient stores) -- what plans of
            # >= 750 k nodes take by default and what a caller whose T1M-sized buffers are NOT cache-resident should set
            try:
                if only:
                    raise StopIteration          # profiler helper runs: the regime's own launches only
                from hidenn_fem_amd.plan import TilePlan
                prev_sp = L.hfem_get_option(b"store_policy")
                _lib.check(L.hfem_set_option(b"store_policy", 2), "hfem_set_option")
                try:
                    plan_nt = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                                       u_src=model._u_src, edges=model.neumann_edges, tile_elems=a.tile_elems, device=dev)
                finally:
                    L.hfem_set_option(b"store_policy", prev_sp)
                ko_nt = KernelOnly(model, loss_fn, plan_nt, 0, -1)
                t_nt, _ = time_launches(lambda i: ko_nt(sets[i % R]), kreps)
                t_nt_rep, _ = time_launches(lambda i: ko_nt(), kreps)
                regimes["rotating_sets"]["nt_stores"] = dict(rotating_kernel_us=t_nt, replayed_kernel_us=t_nt_rep)
                del ko_nt, plan_nt
            except StopIteration:
                pass
            except Exception as e:  # noqa: BLE001
                note(f"nt-store leg failed: {type(e).__name__}: {str(e)[:120]}")
            del sets
    if only:
        if rank == 0:
            print(json.dumps(dict(only_regime=only, kernel_us=k_us, regimes=regimes)), flush=True)
        return None

    ne_launch, nn_launch, alg_bytes = range_work(plan, lo, hi)
    # Counter traffic and the profiler's own per-kernel averages cannot be collected from inside this process: they come from
    # the committed rocprofv3 passes of the SAME kernel / workload (scripts/prof_regimes.sh, scripts/prof_extras.sh), keyed by
    # the workload's shape, else null.  FETCH_SIZE / WRITE_SIZE sit on the L2's fabric side: requests served by the Infinity
    # Cache are counted, so `traffic` is an UPPER bound on HBM bytes in the cache-resident regimes.
    prof = {}
    for fn in ("regimes_rocprof.json", "traffic.json"):
        prof[fn] = {}
        for d_ in (PROFILE_DIR, os.path.join(ROOT, "profiles", "r03")):
            try:
                with open(os.path.join(d_, fn)) as f:
                    prof[fn] = json.load(f)
                break
            except (OSError, ValueError):
                pass
    shape_key = f"{ne}/{nn}/{plan.stats['n_tiles']}"
    traffic_tab = prof["traffic.json"].get("workloads", {})
    traffic = traffic_tab.get("T1M", {}).get("traffic_bytes_per_launch") if world == 1 and traffic_tab.get("T1M", {}).get("shape") == shape_key else None
    rp = prof["regimes_rocprof.json"]
    rocprof_us = {k: v["avg_us"] for k, v in rp.get("regimes", {}).items()} if world == 1 and rp.get("shape") == shape_key else {}
    # in-run estimate of a regime's launch time from the kernel's own stamps: its span + the launch ramp that the replayed
    # leg shows between span and events time (dispatch of 1024 workgroups + the tail the stamps do not see)
    ramp = None
    if regimes.get("replayed", {}).get("wg_span_us"):
        ramp = regimes["replayed"]["kernel_us"] - regimes["replayed"]["wg_span_us"]
    for name, r in regimes.items():
        if name == "rewritten_inputs":
            if r.get("wg_span_us") and ramp is not None:
                r["kernel_us"] = r["wg_span_us"] + ramp
                r["kernel_us_source"] = ("in-run: the kernel's own span stamps inside the (rewrite, energy) sequence + the launch ramp "
                                         f"of the replayed leg ({ramp:.2f} us = events - span there)")
            else:
                r["kernel_us"] = r["events_upper_bound_us"]
                r["kernel_us_source"] = "HIP events, (rewrite + energy) - (rewrite alone): an upper bound"
        r["achieved"] = alg_bytes / (r["kernel_us"] * 1e-6) / 1e9
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
        if "nt_stores" in r:
            for k_ in ("rotating", "replayed"):
                r["nt_stores"][k_ + "_frac"] = alg_bytes / (r["nt_stores"][k_ + "_kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        if name in rocprof_us:                      # committed profiler run of the same leg, never the in-run figure
            r["rocprof_kernel_us"] = rocprof_us[name]
            r["rocprof_frac"] = alg_bytes / (rocprof_us[name] * 1e-6) / 1e9 / HBM_PEAK_GBS
    kname = "tri3_energy_pair_kernel" if plan.is_paired() else "tri3_energy_fast_kernel"
    # counter traffic of the dominant kernel, measured IN THIS RUN when rocprofv3 is at hand: two child passes of this very
    # script (--only-regime replayed) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes,
    # MI355X_MICROARCH.md; gfx950: FETCH_SIZE x 2); else the committed passes of the same shape, labelled as such
    traffic_kind = ("committed: FETCH_SIZE x2 + WRITE_SIZE of the rocprofv3 --pmc passes under profiles/ (same workload shape), "
                    "not measured in this run") if traffic is not None else None
    if world == 1 and not a.no_pmc and not a.no_regimes:
        got, why = pmc_traffic_in_run(kname)
        if got is not None:
            traffic, traffic_kind = got["traffic_bytes_per_launch"], (
                "measured in this run: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, two separate rocprofv3 --pmc child passes of "
                f"`bench.py --only-regime replayed` ({got['launches']} launches of {kname}); requests on the L2's fabric side "
                "(Infinity-Cache hits included): an upper bound on HBM bytes")
            regimes.setdefault("replayed", {})["pmc"] = got
        else:
            note(f"in-run PMC traffic unavailable ({why}); roofline.traffic is the committed figure")
    # ONE regime per line: the top level is the regime the timed headline step runs in (replayed: the same buffers every
    # launch -- a 44 MB working set, Infinity-Cache resident, as in a training loop on this mesh), so kernel_us <= ms_per_step
    # holds inside this object; the regime whose reads really come from HBM (rotating sets) sits beside it as `hbm_regime`,
    # with the whole step timed there too (hbm_regime.step)
    top = regimes.get("replayed")
    if top is not None and "kernel_us" in top:
        rot = regimes.get("rotating_sets")
        roofline = dict(bound="hbm", achieved=top["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=top["frac"], traffic=traffic,
                        traffic_kind=traffic_kind, kernel=kname, kernel_us=top["kernel_us"],
                        kernel_us_regions=[round(v, 3) for v in samples],
                        regime="replayed: the same buffers every launch -- the regime the timed step (ms_per_step) runs in; the 44 MB "
                               "working set is Infinity-Cache resident, so `achieved` is cache-fed algorithmic bandwidth held "
                               "against the HBM peak; hbm_regime is the leg whose reads come from HBM",
                        alg_bytes_per_launch=alg_bytes, elems_per_launch=ne_launch, nodes_per_launch=nn_launch,
                        hbm_regime=None if rot is None else dict(
                            regime=f"rotating_sets: {rot['sets']} parameter / gradient sets, {rot['working_set_mb']} MB > the 256 MB "
                                   "Infinity Cache -- reads of x, u and the gradient lines go to HBM",
                            kernel_us=rot["kernel_us"], achieved=rot["achieved"], frac=rot["frac"], step=rot.get("step")),
                        regimes=regimes)
    else:
        achieved = alg_bytes / (k_us * 1e-6) / 1e9
        roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                        traffic_kind=traffic_kind, kernel=kname, kernel_us=k_us, kernel_us_regions=[round(v, 3) for v in samples],
                        regime="replayed: the same buffers every launch (cache-resident working set)" + ("" if world == 1 else
                               "; this rank's tile range of the weak-scaling mesh"),
                        alg_bytes_per_launch=alg_bytes, elems_per_launch=ne_launch, nodes_per_launch=nn_launch)

    if world == 1:
        progress("roofline legs (regimes, in-run PMC traffic)")
    # ---- config.extra (N = 1): the other readings of "1 M quad elements" and BASELINE config 5, kernel only, each with
    #      its own roofline figures (algorithmic bytes of ITS element type over ITS kernel's average launch time) in the HBM
    #      regime (rotating sets; `replayed_*` = the same buffers every launch):
    #        Q1M      10^6 QUAD4-iso elements (the extension element; parity unpinned by the reference)
    #        T2M      the same 10^6 quads split in two: 2 x 10^6 TRI3 (reference-pinned element)
    #        cfg5     4 x 10^6 TRI3, random diagonals + random element / node permutation, rows stored as given (reorder="off")
    #        cfg5auto the same mesh through the plain model API (reorder="auto": rows stored along the locality curve)
    #        cfg5r    the same after mesh.reorder_for_locality (explicit renumbering by the caller)
    #        cfg5u    4.1 x 10^6 TRI3, genuinely unstructured: Delaunay of graded random points, plate with three holes
    extras = []
    if world == 1 and (not a.no_extra or a.only_extra):
        from hidenn_fem_amd.mesh import reorder_for_locality, structured_quad_mesh, unstructured_tri_mesh

        def extra(key, name, mesh, quad=False, model_kw=None):
            if a.only_extra and a.only_extra != key:
                return
            c_, cn_, g_, b_, _, e_ = mesh
            torch.manual_seed(0)
            m_ = PiecewiseLinearShapeNN2D(c_, cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_,
                                          **(model_kw or dict(reorder=a.reorder))).to(dev)
            lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
            pl = m_.tile_plan(0)
            x_, u_ = m_.node_coords_free.detach(), m_.u_free.detach()
            xfx, ufx = m_.node_coords_fixed, m_.u_fixed_rows()
            ls_ = torch.zeros((), dtype=f64, device=dev)
            _, Tc_ = lf_._traction(m_, None)
            Tcv, mat_ = dv(Tc_), dv(lf_._mat)
            # two regimes, as for the headline kernel: the same buffers every launch, and R rotating parameter / gradient sets
            # that together exceed the 256 MB Infinity Cache (the regime a training loop on a mesh of this size runs in)
            set_mb = 4 * x_.numel() * 8 / 2 ** 20
            R_ = max(2, int(320.0 / set_mb) + 1)
            sets_ = [(x_.clone(), u_.clone(), torch.empty_like(x_), torch.empty_like(u_)) for _ in range(R_)]

            def launch_on(xs, us_, gxs, gus):
                if quad:
                    _lib.check(L.hfem_quad4_energy_plan(pl.handle, xs.data_ptr(), xfx.data_ptr(), us_.data_ptr(), ufx.data_ptr(),
                                                        mat_, None, Tcv, 0, -1, ls_.data_ptr(), gxs.data_ptr(), gus.data_ptr(),
                                                        8, stream_box[0].cuda_stream))
                else:
                    _lib.check(L.hfem_tri3_energy_plan(pl.handle, xs.data_ptr(), xfx.data_ptr(), us_.data_ptr(), ufx.data_ptr(),
                                                       mat_, lf_._W, dv([0.0] * 6), None, Tcv, 0, -1, ls_.data_ptr(), gxs.data_ptr(),
                                                       gus.data_ptr(), 8, stream_box[0].cuda_stream))
            n_l = min(kreps, 60) // R_ * R_ or R_
            us_rep, _ = time_launches(lambda i: launch_on(*sets_[0]), n_l)
            us, _ = time_launches(lambda i: launch_on(*sets_[i % R_]), n_l)
            ne_, nn_ = cn_.shape[0], c_.shape[0]
            ab = (16 if quad else 12) * ne_ + 64 * nn_ + 8
            st_ = pl.stats
            stores_ = {16: "sc1 write-through", 2: "nt", 0: "plain"}.get(st_["store_policy"], str(st_["store_policy"]))
            del sets_
            tr_ = traffic_tab.get(key, {})
            tr_ok = tr_.get("shape") == f"{ne_}/{nn_}/{st_['n_tiles']}"
            extras.append(dict(key=key, name=name, element="QUAD4" if quad else "TRI3", elements=ne_, nodes=nn_, tiles=st_["n_tiles"],
                               halo_elem_factor=st_["tile_elem_total"] / ne_, halo_node_factor=st_["tile_node_total"] / nn_,
                               row_order=getattr(m_, "row_order", "as given"), row_line_factor=getattr(m_, "row_line_factor", None),
                               plan_row_line_factor=st_["row_line_factor"],
                               regime=f"rotating: {R_} parameter / gradient sets, {round(R_ * set_mb + st_['device_bytes'] / 2 ** 20)} MB "
                                      "(> the 256 MB Infinity Cache)", gradient_stores=stores_,
                               kernel_us=us, element_evals_per_s=ne_ / (us * 1e-6), alg_bytes_per_launch=ab,
                               achieved=ab / (us * 1e-6) / 1e9, frac=ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               replayed_kernel_us=us_rep, replayed_frac=ab / (us_rep * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               traffic=tr_.get("traffic_bytes_per_launch") if tr_ok else None,
                               traffic_over_alg=(tr_["traffic_bytes_per_launch"] / ab) if tr_ok else None))
            del m_, pl

        extra("Q1M", "Q1M: 10^6 QUAD4-iso (1001 x 1001 nodes), parity unpinned by the reference",
              structured_quad_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64), quad=True)
        extra("T2M", "T2M: the same 10^6 quads split in two (2 x 10^6 TRI3)",
              structured_tri_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64))
        cfg5 = structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True, dtype=f64)
        extra("cfg5", "cfg5: 4 x 10^6 TRI3, random diagonals, random element + node permutation, rows stored as the mesher "
              "numbered them (reorder='off')", cfg5, model_kw=dict(reorder="off"))
        extra("cfg5auto", "cfg5auto: the same mesh through the plain model API (reorder='auto')", cfg5)
        extra("cfg5r", "cfg5r: cfg5 after mesh.reorder_for_locality (Hilbert node renumbering by the caller)",
              reorder_for_locality(cfg5)[0])
        del cfg5
        extra("cfg5u", "cfg5u: genuinely unstructured (Delaunay, plate with three holes, graded), ~4.1 x 10^6 TRI3",
              unstructured_tri_mesh(2_050_000, seed=2, dtype=f64))
    if a.only_extra:
        if rank == 0:
            print(json.dumps(dict(only_extra=a.only_extra, extras=extras)), flush=True)
        return None

    if world == 1:
        progress("config.extra workloads")
    # ---- config.fp32_rows (N = 1): T1M as the reference would run it by default -- an fp32 model (src/loss.py:16,
    #      src/models.py:274): float rows widened on load, gradients rounded once on store, fp64 arithmetic and loss; kernel
    #      only, its own algorithmic bytes (12 Ne + 32 Nn + 8)
    fp32_leg = None
    if world == 1 and not a.no_extra:
        try:
            m32 = PiecewiseLinearShapeNN2D(coords.float(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                           neumann_edges=edges, reorder=a.reorder).to(dev)
            pl32 = m32.tile_plan(a.tile_elems)
            x32, u32 = m32.node_coords_free.detach(), m32.u_free.detach()
            xfx32, ufx32 = m32.node_coords_fixed, m32.u_fixed_rows()
            gx32, gu32 = torch.empty_like(x32), torch.empty_like(u32)
            ls32 = torch.zeros((), dtype=f64, device=dev)
            lf32 = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=torch.float32)
            _, Tc32 = lf32._traction(m32, None)
            mat32, Tcv32, Bk32 = dv(lf32._mat), dv(Tc32), dv([0.0] * 6)

            def launch32(flags32):
                def go(i):
                    _lib.check(L.hfem_tri3_energy_plan_f32(pl32.handle, x32.data_ptr(), xfx32.data_ptr(), u32.data_ptr(), ufx32.data_ptr(),
                                                           mat32, lf32._W, Bk32, None, Tcv32, 0, -1, ls32.data_ptr(), gx32.data_ptr(),
                                                           gu32.data_ptr(), flags32, stream_box[0].cuda_stream), "hfem_tri3_energy_plan_f32")
                return go
            ab32 = 12 * ne + 32 * nn + 8

            def fig(us_):
                return dict(kernel_us=us_, element_evals_per_s=ne / (us_ * 1e-6), achieved=ab32 / (us_ * 1e-6) / 1e9,
                            frac=ab32 / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS)
            us32, _ = time_launches(launch32(8 | 1024), kreps)           # HFEM_FLAG_FP32_MATH: what EnergyLoss2D(arithmetic="auto") runs
            us32_64, _ = time_launches(launch32(8), kreps)
            # accuracy of both against the fp64 kernel on the same float values (max-abs over max|g|)
            m64 = PiecewiseLinearShapeNN2D(coords.float().double(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                           neumann_edges=edges, reorder=a.reorder).to(dev)
            with torch.no_grad():
                m64.u_free.copy_(u32.double())
            lf64 = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
            lf64._mat, lf64._W, lf64._ci, lf64._cj = lf32._mat, lf32._W, lf32._ci, lf32._cj
            l64 = lf64.value_and_grad_(m64).item()
            acc = {}
            for tag, fl_ in (("fp32_arithmetic", 1024), ("fp64_arithmetic", 0)):
                _lib.check(L.hfem_tri3_energy_plan_f32(pl32.handle, x32.data_ptr(), xfx32.data_ptr(), u32.data_ptr(), ufx32.data_ptr(),
                                                       mat32, lf32._W, Bk32, None, Tcv32, 0, -1, ls32.data_ptr(), gx32.data_ptr(),
                                                       gu32.data_ptr(), fl_, torch.cuda.current_stream().cuda_stream), "hfem_tri3_energy_plan_f32")
                acc[tag] = dict(loss_rel_err=abs(ls32.item() - l64) / abs(l64),
                                gx_err_over_max=((gx32.double() - m64.node_coords_free.grad).abs().max() / m64.node_coords_free.grad.abs().max()).item(),
                                gu_err_over_max=((gu32.double() - m64.u_free.grad).abs().max() / m64.u_free.grad.abs().max()).item())
            fp32_leg = dict(workload="T1M as the reference runs it by default: an fp32 model (float parameter / gradient rows)",
                            alg_bytes_per_launch=ab32, regime="replayed (same buffers every launch)",
                            kernel="tri3_energy_pair_f32_kernel (csrc/tri3_pair_f32.hip): fp32 arithmetic, the two elements of a slot in "
                                   "the two halves of packed-fp32 registers; EnergyLoss2D(arithmetic='auto')",
                            **fig(us32), fp64_arithmetic=dict(kernel="tri3_energy_pair_kernel<float2>: rows widened on load, one rounding "
                                                                    "on store; EnergyLoss2D(arithmetic='fp64')", **fig(us32_64)),
                            error_vs_fp64_kernel_on_the_same_floats=acc)
            del m32, pl32, m64
        except Exception as e:  # noqa: BLE001
            note(f"fp32_rows leg failed: {type(e).__name__}: {str(e)[:160]}")

    # ---- config.strong_scaling_emulated (N = 1): BASELINE configs[3] AS STATED on one GPU -- the FIXED 10^6-element mesh
    #      sharded N ways by a plan prepared for N ranks (shard-aware tile policy), kernel only over the tile range of the
    #      first, a middle and the last rank; the max is what a strong-scaling step waits for
    strong_emu = None
    if world == 1 and not a.no_extra:
        strong_emu = []
        for n_ in (2, 4, 8):
            r_ = shard_kernel_leg(model, loss_fn, n_, sorted({0, n_ // 2, n_ - 1}), kreps)
            r_["workload"] = "T1M fixed"
            r_["speedup_vs_1gpu_kernel"] = regimes.get("replayed", {}).get("kernel_us", k_us) / r_["kernel_us_max"]
            strong_emu.append(r_)

    # ---- config.train_step_1gpu (N = 1): the hot path INSIDE an optimiser loop on T1M -- what a training iteration costs
    #      when the kernel's inputs are what the optimiser just wrote.  (i) energy launch + ONE multi-tensor FusedAdam launch
    #      (the reference's `loss.backward(); optimizer.step()`), (ii) one launch: the tiles apply Adam to the rows they own
    #      (hfem_tri3_energy_adam_step).  K iterations in one hipGraph each; lr tiny so the mesh stays valid.
    train1 = None
    if world == 1 and not a.no_extra:
        try:
            from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
            from hidenn_fem_amd.graphed import GraphedTraining
            res = {}
            K = max(2, (a.steps // 2) * 2)
            for mode in ("two_launch", "one_launch"):
                m_ = build_model(mesh6)
                lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
                if mode == "two_launch":
                    opt = FusedAdam([dict(params=[m_.node_coords_free], lr=1e-9), dict(params=[m_.u_free], lr=1e-12)],
                                    capturable=True)
                    gt = GraphedTraining(lambda: lf_.value_and_grad_(m_), opt, steps_per_replay=K, direct=True, warmup=2)
                else:
                    tr = EnergyAdamStep(m_, lf_, lr_x=1e-9, lr_u=1e-12)
                    gt = GraphedTraining(tr.step_lagged, None, steps_per_replay=K, direct=True, begin=tr.begin_lagged,
                                         end=tr.flush_loss)
                for _ in range(3):
                    gt.replay()
                torch.cuda.synchronize()
                t_pw = time.perf_counter()
                while time.perf_counter() - t_pw < a.prewarm:
                    gt.replay()
                    torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    gt.replay()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0_) / K)
                it = sorted(ts)[2]
                res[mode] = dict(us_per_iteration=it * 1e6, element_evals_per_s=ne / it)
                del gt, m_
            train1 = dict(workload="T1M, Adam on node_coords_free and u_free, K iterations per hipGraph", **res)
        except Exception as e:  # noqa: BLE001
            note(f"train_step_1gpu leg failed: {type(e).__name__}: {str(e)[:160]}")

    if world == 1:
        progress("fp32 rows, training-step legs")
    # ---- N = 1: Example 4's optimiser at full size -- LBFGS as the reference drives it (examples/example4.py:68-78: lr 1, max_iter 20,
    #      history 100, no line search) on T1M.  With the history full an inner iteration streams the 2 x 100 history vectors twice
    #      (multidot + direction passes, csrc/lbfgs.hip): 4 h n 8 B = 6.4 GB -- the optimiser, not the 9 us energy launch, is the iteration.
    lbfgs1 = None
    if world == 1 and not a.no_extra and not a.no_lbfgs and not only and not a.only_extra:
        try:
            from hidenn_fem_amd.optim import FusedLBFGS

            def lbfgs_run(cls, outer):
                m_ = build_model(mesh6)
                lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
                opt = cls(m_.parameters())
                n_par = sum(p.numel() for p in m_.parameters())
                ts = []
                for _ in range(outer):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    opt.step(lambda: lf_.value_and_grad_(m_))
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0_)
                del opt, m_
                torch.cuda.empty_cache()
                return n_par, ts
            n_par, ts = lbfgs_run(FusedLBFGS, 8)              # history (100 pairs) is full from outer step 5 on
            it = sorted(ts[5:])[1] / 20.0
            byts = 4.0 * 100 * n_par * 8
            lbfgs1 = dict(workload="T1M, FusedLBFGS as examples/example4.py drives torch.optim.LBFGS (20 inner iterations per step, history 100), "
                                   "history full", parameters=n_par, ms_per_inner_iteration=it * 1e3,
                          alg_bytes_per_iteration=byts, achieved=byts / it / 1e9, unit="GB/s", frac=byts / it / 1e9 / HBM_PEAK_GBS,
                          bytes_rule="4 h n 8 B: the 2 h history vectors read once by the multidot pass and once by the direction pass")
            _, tt = lbfgs_run(torch.optim.LBFGS, 7)
            lbfgs1["torch_optim_LBFGS_ms_per_inner_iteration"] = sorted(tt[5:])[len(tt[5:]) // 2] / 20.0 * 1e3
        except Exception as e:  # noqa: BLE001
            note(f"lbfgs_step_1gpu leg failed: {type(e).__name__}: {str(e)[:160]}")

    # ---- N = 1: the node-sharded L-BFGS of BASELINE configs[3] as stated (10^6 elements FIXED over 8 ranks) rehearsed on this GPU:
    #      rank r of 8 -- its tile range's energy launch, its 1/8 of the history (emulate=True: no peers; the payload gather is a
    #      local copy) -- ms per inner iteration against lbfgs_step_1gpu (the whole optimiser on one GPU)
    lbfgs_emu = None
    if world == 1 and not a.no_extra and not a.no_lbfgs and not only and not a.only_extra:
        try:
            lbfgs_emu = []
            for r_ in (0, 4, 7):
                m_ = build_model(mesh6)
                lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
                sh_ = ShardedTri3Energy(m_, lf_, rank=r_, world=8)
                leg_ = lbfgs_leg(sh_, f"T1M fixed, rank {r_} of 8 emulated on one GPU", emulate=True)
                leg_["rank"] = r_
                if lbfgs1 is not None:
                    leg_["speedup_vs_lbfgs_step_1gpu"] = lbfgs1["ms_per_inner_iteration"] / leg_["ms_per_inner_iteration"]
                lbfgs_emu.append(leg_)
                del sh_, m_
        except Exception as e:  # noqa: BLE001
            note(f"lbfgs_sharded_emulated leg failed: {type(e).__name__}: {str(e)[:160]}")

    if world == 1:
        progress("L-BFGS legs")
    # ---- N = 1: what the owner-sharded step machinery costs on ONE rank (no peer to talk to: every microsecond above the
    #      plain iteration is overhead of the exchange path) -- all_gather stand-in on one stream, the side-stream overlap,
    #      and the peer-window put / get; plain and fused (Adam inside the energy launch); K iterations per hipGraph.
    shard1 = None
    if world == 1 and not a.no_extra and not a.no_peer and not only and not a.only_extra:
        try:
            m_ = build_model(mesh6)
            lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
            sh_ = ShardedTri3Energy(m_, lf_)
            sh_.setup_interfaces()
            sh_.init_owner_adam(lr_x=1e-9, lr_u=1e-12, fused=True)
            n_t = sh_.hi - sh_.lo
            sh_.mid = sh_.lo + max(1, n_t // 20)            # 5 % of the tiles play the boundary part, as on a real shard
            K1 = max(2, (max(a.steps, 100) // 2) * 2)

            def us(body, end_=None):
                el, _, ln = timed_steps(body, K1, None, end_)
                return round(el / K1 * 1e6, 3)
            shard1 = dict(workload=f"T1M on one rank, {K1} iterations per hipGraph, boundary part = {sh_.mid - sh_.lo} of {n_t} tiles",
                          unit="us per iteration")
            shard1["collective_path"] = dict(train_step=us(sh_.owner_train_step),
                                             train_step_overlap_side_stream=us(sh_.owner_train_step_overlapped, sh_.finish_overlapped),
                                             train_step_fused=us(sh_.owner_train_step_fused),
                                             train_step_fused_overlap_side_stream=us(sh_.owner_train_step_fused_overlapped,
                                                                                     sh_.finish_overlapped))
            sh_.enable_peer_exchange()
            shard1["peer_windows"] = dict(train_step=us(sh_.owner_train_step),
                                          train_step_overlap=us(sh_.owner_train_step_overlapped, sh_.finish_overlapped),
                                          train_step_fused=us(sh_.owner_train_step_fused),
                                          train_step_fused_overlap=us(sh_.owner_train_step_fused_overlapped, sh_.finish_overlapped))
            shard1["peer_windows"]["status"] = sh_.peer.status()[0]
            sh_.close_peer_exchange()
            del sh_, m_
        except Exception as e:  # noqa: BLE001
            note(f"sharded_step_1gpu leg failed: {type(e).__name__}: {str(e)[:160]}")

    if world == 1:
        progress("owner-sharded steps on one rank")
    out = None
    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            cpu, loss_cpu = cpu_baseline(mesh6, model.to_caller_order(model.u_free.detach(), "u").cpu(), a.cpu_evals)
            rel = abs(loss_cpu - loss_gpu) / abs(loss_cpu)
            assert rel <= 1e-12, f"GPU loss {loss_gpu!r} != oracle loss {loss_cpu!r} (rel {rel:.2e})"
            cpu["loss_rel_err_vs_gpu"] = rel
        progress("cpu_baseline" if cpu is not None else "done")
        st = plan.stats
        out = dict(
            metric="element-evals/sec (fwd+bwd energy) + achieved HBM GB/s, 2D quad mesh",
            value=value, unit="element-evals/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
            ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
            data="synthetic",
            config=dict(workload=f"Example 4 (T1M x {world}): 2D plate linear elasticity, {ne} TRI3 "
                                 f"(structured quads split in 2), {nn} nodes, gauss_order=4, r-adaptivity on, "
                                 f"Neumann edges {edges.shape[0]}, fwd+bwd (loss + dX + dU)",
                        elements=ne, nodes=nn, elements_per_gpu=ne // world, tiles=st["n_tiles"],
                        tile_elems=st["tile_elems"], element_order="paired slots" if plan.is_paired() else "one element per slot",
                        threads_per_tile=st["threads_per_tile"],
                        halo_elem_factor=sum(len(plan.tile_elements(t)[0]) for t in range(st["n_tiles"])) / max(ne, 1),
                        lds_bytes=st["lds_bytes"], launch=launch,
                        timed_region=f"one hipGraph of {a.steps} steps; median of {len(regions)} replays",
                        ms_per_step_replays=[round(r / a.steps * 1e3, 6) for r in regions],
                        loss_sum=("by an extra workgroup of the next launch (HFEM_FLAG_SUM_PREVIOUS) + one trailing "
                                  "1-block launch" if lagged else "1-block launch after every energy kernel"),
                        exchange="none" if world == 1 else
                        f"owner-sharded: gradient rows stay with the rank whose tiles own the node; one all_gather per "
                        f"step of interface parameter rows + partial energy ({sh.interface_stats['payload_bytes']} B "
                        f"per rank), {comm_state}",
                        loss=loss_gpu),
            roofline=roofline,
        )
        if world > 1:
            out["config"]["collectives"] = comm_state
            out["config"]["plan_cache"] = plan_cache_state
        if extras:
            out["config"]["extra"] = extras
        if inline_step is not None:
            out["config"]["inline_loss_step"] = inline_step
        if train1 is not None:
            out["config"]["train_step_1gpu"] = train1
        if fp32_leg is not None:
            out["config"]["fp32_rows"] = fp32_leg
        if strong_emu is not None:
            out["config"]["strong_scaling_emulated"] = strong_emu
        if eval_ov is not None:
            out["config"]["eval_exchange_overlap"] = eval_ov
            out["config"]["eval_exchange_serial"] = serial
            out["config"]["exchange_mode"] = exchange_mode
        if alt is not None:
            out["config"]["alt_exchange"] = alt
        if train is not None:
            out["config"]["train_step"] = train
        if train_ov is not None:
            out["config"]["train_step_overlap"] = train_ov
        if train_fused is not None:
            out["config"]["train_step_fused"] = train_fused
            out["config"]["train_step_fused_overlap"] = train_fused_ov
        if strong is not None:
            out["config"]["strong_scaling"] = strong
        if lbfgs_n is not None:
            out["config"]["lbfgs_step"] = lbfgs_n
        if peer_state is not None:
            out["config"]["peer_exchange"] = dict(state=peer_state, **(peer_legs or {}))
        if shard1 is not None:
            out["config"]["sharded_step_1gpu"] = shard1
        if lbfgs1 is not None:
            out["config"]["lbfgs_step_1gpu"] = lbfgs1
        if lbfgs_emu is not None:
            out["config"]["lbfgs_sharded_emulated"] = lbfgs_emu
        if notes:
            out["config"]["notes"] = notes
        out["config"]["wall_s"] = dict(wall, what="seconds since the start of main() when each section finished (rank 0); the stderr "
                                                  "lines `[bench] <s> s  <section>` carry the same")
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if cache_made:
        import shutil
        shutil.rmtree(cache_made, ignore_errors=True)
    return out


if __name__ == "__main__":
    main()
