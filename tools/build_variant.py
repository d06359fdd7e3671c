"""Build a variant of libresr_hip.so with extra compiler flags into tools/ab/<name>.so (same-box A/B timing:
RESR_LIB_PATH=tools/ab/<name>.so python bench.py ...).

    git stash; python tools/build_variant.py prev; git stash pop        # the committed build next to the working tree's
    python tools/build_variant.py exp -DSOME_EXPERIMENT=1               # or the tree with an experiment's define
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real_esrgan-pytorch_amd", "csrc")


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "tools", "ab", name)
    os.makedirs(out_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-Wno-unused-result"] + extra
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

    def cc(f):
        obj = os.path.join(out_dir, f.replace(".hip", ".o"))
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, f), "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, srcs))
    lib = os.path.join(ROOT, "tools", "ab", name + ".so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
