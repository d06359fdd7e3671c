"""Registers / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py conv3x3_ws_x2_mt1.hip [more.hip ...]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "real_esrgan-pytorch_amd", "csrc")


def report(src):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(CSRC, "..", "..", "include"),
           "-I", CSRC, "-Wno-unused-result", "-Wno-inline-asm", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    def short(n):   # _ZN4resr17conv3x3_ws_kernelIDF16_Li1ELi2ELi8ELi33ELb1ELi0ELi2EEEv... -> conv3x3_ws_kernel<f16,1,2,8,33,1,0,2>
        m = re.match(r"_ZN4resr\d+([a-z0-9_]+?)I(.*?)EEv", n)
        if not m:
            return n[:70]
        args = re.findall(r"DF16_|f|L[ib](\d+)E", m.group(2))
        raw = re.findall(r"DF16_|L[ib]\d+E|f", m.group(2))
        out = ["f16" if t == "DF16_" else "f32" if t == "f" else re.sub(r"L[ib](\d+)E", r"\1", t) for t in raw]
        return m.group(1) + "<" + ",".join(out) + ">"
    for r in rows:
        n = short(r["name"])
        print(f"{n:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
              f"occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")


if __name__ == "__main__":
    for s in sys.argv[1:]:
        print("==", s)
        report(s)
