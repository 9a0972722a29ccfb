"""Build libhidenn_hip.so in-tree with hipcc for gfx950 (MI355X / CDNA4 only).

    python hidenn_fem_amd/csrc/build.py [--force] [--keep-temps] [--lab] [--tag NAME --define MACRO=VALUE ...]

One object per source (compiled in parallel), one link.  ``-munsafe-fp-atomics``
selects the hardware fp64 atomics (``global_atomic_add_f64`` / ``ds_add_f64``)
instead of compare-and-swap loops.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["plan.cpp", "mg.cpp", "tri3_energy.hip", "tri3_stream.hip", "tri3_pair.hip", "tri3_pair_f32.hip", "tri3_pair_lab.hip", "tri3_pair_pipe.hip", "tri3_det.hip", "tri3_eval.hip", "line_rect.hip", "quad4.hip", "optim.hip", "exchange.hip", "peer.hip", "post.hip", "lbfgs.hip"]
HEADERS = ["hfem_common.h", "hfem_device.h", "hfem_plan_dev.h", "tri3_energy_lab.inc", os.path.join(ROOT, "include", "hidenn_fem.h")]
OUT = os.path.join(HERE, "libhidenn_hip.so")
OUT_LAB = os.path.join(HERE, "libhidenn_hip_lab.so")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
            "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, keep_temps: bool = False, lab: bool = False, tag: str = "", defines=()) -> str:
    """``lab=True`` builds the second target, ``libhidenn_hip_lab.so`` (``-DHFEM_LAB``): the same library plus the
    kernel-lab instrumentation -- ablation instances, s_memrealtime stamps, start staggers, the pipelined and the
    streamed kernel variants -- that ``scripts/`` drives (``HFEM_LAB=1`` selects it in ``hidenn_fem_amd._lib``).
    None of that is compiled into the product library.  ``tag`` + ``defines`` (dev tool, A/B timing of a compile-time choice):
    a third target ``libhidenn_hip_<tag>.so`` of the product sources with ``-D<define>`` added, built in ``build/<tag>``; a
    script selects it by setting ``hidenn_fem_amd._lib.LIB_PATH`` before the first call."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build", "lab") if lab else os.path.join(HERE, "build")
    out = OUT_LAB if lab else OUT
    flags = CXXFLAGS + (["-DHFEM_LAB"] if lab else [])
    if tag:
        bdir, out = os.path.join(HERE, "build", tag), os.path.join(HERE, f"libhidenn_hip_{tag}.so")
        flags = flags + ["-D" + d for d in defines]
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    objs = [os.path.join(bdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    os.makedirs(bdir, exist_ok=True)

    def compile_one(pair):
        src, obj = pair
        res = obj + ".resource.txt"
        if not force and not _stale(obj, [src] + hdrs + [__file__]) and os.path.exists(res):
            return
        # the compiler's per-kernel resource report (VGPRs, scratch, occupancy) is always kept beside the object:
        # tests/test_build_resources.py holds the product to "no scratch, <= 100 pair-kernel instances"
        cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
        if keep_temps:
            cmd += ["-save-temps=obj"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=bdir)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        with open(res, "w") as f:
            f.write(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    if force or _stale(out, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out


def resource_usage(lab: bool = False):
    """``{kernel mangled name: dict(vgprs, scratch, occupancy, source)}`` of the last build (``-Rpass-analysis=kernel-resource-usage``)."""
    import re
    bdir = os.path.join(HERE, "build", "lab") if lab else os.path.join(HERE, "build")
    out = {}
    for s in SOURCES:
        path = os.path.join(bdir, os.path.splitext(s)[0] + ".o.resource.txt")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run build() first")
        text = open(path).read()
        for blk in text.split("Function Name: ")[1:]:
            name = blk.split()[0]
            get = lambda key: int(re.search(key + r": (\d+)", blk).group(1))
            out[name] = dict(vgprs=get(r" VGPRs"), scratch=get(r"ScratchSize \[bytes/lane\]"), occupancy=get(r"Occupancy \[waves/SIMD\]"),
                             source=s)
    return out


if __name__ == "__main__":
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else ""
    defs = [sys.argv[i + 1] for i, v in enumerate(sys.argv) if v == "--define"]
    print(build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv, lab="--lab" in sys.argv, tag=tag, defines=defs))
