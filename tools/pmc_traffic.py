"""Reduce rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) of `bench.py` to per-launch HBM traffic
per kernel.  gfx950 note (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half of the bytes of wide coalesced
reads, so reads are doubled; both counters are in KiB.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys


def short(name):
    # conv3x3_ws_kernel<T, MT, NT, NWC, EPI, X2, SP, CH>: epilogue kinds are summed; X2 = 1 (three f16 stages per pair chunk) and X2 = 2 (one f16 +
    # one MX stage) share bench.py's in-situ name "f16x2"; CH != 0 = the passes of a dense block as one chained launch
    m = re.search(r"conv3x3_ws_kernelI(DF16_|f)Li(\d)ELi(\d)ELi(\d)ELi\d+E(?:Li(\d)E)?(?:Li(\d)E)?(?:Li(\d)E)?", name)
    if m:
        t = "f32" if m.group(1) != "DF16_" else ("f16x2" if m.group(5) not in (None, "0") else "f16")
        chain = ",chain" if m.group(7) not in (None, "0") else ""
        return f"conv3x3_ws_kernel<{t},{m.group(2)},{m.group(3)},{m.group(4)}{chain}>"
    m = re.search(r"conv3x3_kernelI(DF16_|f)Li(\d)ELi(\d)ELi(\d)E", name)
    if m:
        return f"conv3x3_kernel<{'f16' if m.group(1) == 'DF16_' else 'f32'},{m.group(2)},{m.group(3)},{m.group(4)}>"
    if "wgrad_quad_kernel" in name:
        return "wgrad_quad_kernel<mx>" if "ILb1E" in name or "<true>" in name else "wgrad_quad_kernel<f16>"
    m = re.search(r"wgrad_kernelI(DF16_|f)Li(\d)E", name)
    if m:
        return f"wgrad_kernel<{'f16' if m.group(1) == 'DF16_' else 'f32'},{m.group(2)}>"
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:60]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        if k not in write:
            continue
        f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2      # KiB -> B, x2 gfx950 correction for wide reads
        w = sum(write[k]) / len(write[k]) * 1024
        out[k] = {"launches": len(fetch[k]), "read_bytes_per_launch": round(f), "write_bytes_per_launch": round(w),
                  "hbm_bytes_per_launch": round(f + w)}
    # the degradation stage as a whole (SURVEY 8d asks for its HBM rate): every resr:: kernel of csrc/degrade.hip, per batch --
    # one crop_kernel launch closes each batch (train_realesrnet.py:374-377).  Plans differ from batch to batch (random resize
    # factors, noise kinds): this is the mean over the profiled run's batches.
    DEG = ("usm", "filter1d", "filter2d", "resize_kernel", "randn_kernel", "gauss_noise", "poisson_noise", "unique_", "jpeg_kernel", "crop_kernel")
    batches = out.get("resr::crop_kernel", {}).get("launches", 0)
    if batches:
        per = {k: v for k, v in out.items() if isinstance(v, dict) and "resr::" in k and any(d in k for d in DEG)}
        tot_r = sum(v["read_bytes_per_launch"] * v["launches"] for v in per.values())
        tot_w = sum(v["write_bytes_per_launch"] * v["launches"] for v in per.values())
        out["degradation_stage"] = {"batches": batches, "read_bytes_per_batch": round(tot_r / batches), "write_bytes_per_batch": round(tot_w / batches),
                                    "hbm_bytes_per_batch": round((tot_r + tot_w) / batches),
                                    "kernels": {k: {"launches_per_batch": round(v["launches"] / batches, 2), "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for k, v in per.items()}}
    # the build the counters were taken on: bench.py only quotes a traffic figure whose hash matches the current kernel sources
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        out["csrc_sha16"] = bench.kernel_sources_sha16()
        out["degrade_sha16"] = bench.degrade_sources_sha16()
    except Exception as e:   # pragma: no cover
        out["csrc_sha16"] = None
        out["csrc_sha16_error"] = repr(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
