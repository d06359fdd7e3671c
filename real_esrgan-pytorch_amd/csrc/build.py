"""Build libresr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python real_esrgan-pytorch_amd/csrc/build.py [--force] [--verbose]

One translation unit per .hip file, linked into csrc/libresr_hip.so.  Objects are cached by mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["api.hip", "conv3x3.hip", "conv3x3_ws.hip", "conv3x3_ws_mt1.hip", "conv3x3_ws_mt2.hip", "conv3x3_ws_x2_mt1.hip", "conv3x3_ws_x2_mt2.hip", "conv3x3_ws_sp.hip", "conv3x3_ws_chain.hip", "conv3x3_ws_chain_x2.hip", "wgrad.hip", "pack.hip", "layout.hip", "generator.hip", "degrade.hip", "degrade_int.hip", "disc.hip", "disc_native.hip", "loss.hip", "sustained.hip"]
LIB = os.path.join(HERE, "libresr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", HERE,
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "conv3x3.h"), os.path.join(HERE, "conv3x3_ws.h"), os.path.join(HERE, "conv3x3_ws_chain.h"), os.path.join(HERE, "wgrad.h"), os.path.join(ROOT, "include", "resr.h"), os.path.join(ROOT, "include", "resr_debug.h"), os.path.abspath(__file__)]
    extra = [s for s in os.listdir(HERE) if s.endswith(".hip") and s not in SOURCES]
    srcs = SOURCES + sorted(extra)
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    with ThreadPoolExecutor(max_workers=6) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose or r.returncode:
                sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
            if r.returncode:
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
