"""Where a one-tile-per-CU chained dense-block launch spends its time: the consumer / producer stamps of the
LAST chained launch of a forward (six jobs: conv1..conv4 + the two halves of conv5, EPI 16) and of a backward pass (the four mirrored
passes + the two halves of g_x, EPI 33), reduced to microseconds per phase and job.

Needs the BUDGET trace build (192 stamps per wave):
    python tools/build_variant.py trace2 -DRESR_TRACE=2 && RESR_LIB_PATH=$PWD/tools/ab/trace2.so python tools/chain_budget.py [--batch 32 --res 64] [--json out.json]

Consumer wave 0 stamps: kernel entry; per stage (before the stage barrier, behind it, behind the stage's MFMAs); per tile one more behind the
epilogue + publication.  Producer wave 0 stamps, per stage: behind the barrier | weights requested | next stage worked out | polled | halo
requested | older requests landed.  s_memrealtime ticks are 10 ns."""
import argparse, json, os, sys
os.environ["RESR_TRACE_CHAIN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib
NS = 192
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--json", default="")
a = ap.parse_args()
lib = L.lib()
STAGES = [2, 3, 4, 5, 6, 6]      # fast mode: stages (= 32-channel chunks) of the six jobs


def capture(fn):
    tr = torch.zeros(32 * 2 * NS, dtype=torch.int64, device="cuda")
    lib.resr_debug_conv_trace(L.ptr(tr))
    fn()
    torch.cuda.synchronize()
    lib.resr_debug_conv_trace(None)
    return tr.cpu().view(32, 2, NS)


def reduce_consumer(v):
    """-> per job: barrier wait, multiply, epilogue + publish; plus fill (entry -> first barrier passed)."""
    v = [q for q in v if q > 0]
    need = 1 + sum(3 * s + 1 for s in STAGES)
    if len(v) < need:
        return None
    i, jobs = 1, []
    t_entry = v[0]
    for js, ns in enumerate(STAGES):
        wait = mult = 0.0
        first_b0 = v[i]
        for s in range(ns):
            b0, b1, m = v[i], v[i + 1], v[i + 2]
            wait += b1 - b0
            mult += m - b1
            i += 3
        td = v[i]
        i += 1
        jobs.append({"stages": ns, "barrier_wait_us": wait / 100, "multiply_us": mult / 100, "epilogue_publish_us": (td - v[i - 2]) / 100,
                     "job_us": (td - first_b0) / 100})
    return {"entry_to_first_stage_barrier_us": (v[1] - t_entry) / 100, "total_us": (v[i - 1] - t_entry) / 100, "jobs": jobs}


def reduce_producer(v):
    v = [q for q in v if q > 0]
    ph = {"w_issue": 0.0, "advance": 0.0, "poll": 0.0, "halo_issue": 0.0, "landing_wait": 0.0, "to_next_barrier": 0.0}
    n = 0
    for i in range(1, len(v) - 6, 6):
        d = [(v[i + k + 1] - v[i + k]) / 100 for k in range(6)]
        for k, key in enumerate(ph):
            ph[key] += d[k]
        n += 1
    return {"stages_seen": n, "span_us": (v[-1] - v[0]) / 100 if v else 0.0, "sum_us": ph}


def summarise(t, what):
    wgs = []
    for wg in range(32):
        c = reduce_consumer([int(q) for q in t[wg, 1]])
        if c is None:
            continue
        c["producer"] = reduce_producer([int(q) for q in t[wg, 0]])
        wgs.append(c)
    if not wgs:
        return {"what": what, "error": "no complete consumer timeline (is this the RESR_TRACE=2 build?)"}
    k = len(wgs)
    mean = lambda f: sum(f(w) for w in wgs) / k   # noqa: E731
    rec = {"what": what, "workgroups_traced": k,
           "launch_span_us": mean(lambda w: w["total_us"]),
           "fill_us (entry -> first stage's data landed)": mean(lambda w: w["entry_to_first_stage_barrier_us"] + w["jobs"][0]["barrier_wait_us"] / max(1, w["jobs"][0]["stages"])),
           "multiply_us": mean(lambda w: sum(j["multiply_us"] for j in w["jobs"])),
           "barrier_wait_us (consumers idle: data / flags not there yet)": mean(lambda w: sum(j["barrier_wait_us"] for j in w["jobs"])),
           "epilogue_publish_us": mean(lambda w: sum(j["epilogue_publish_us"] for j in w["jobs"])),
           "per_job": [{"job": j, "stages": STAGES[j], "multiply_us": round(mean(lambda w: w["jobs"][j]["multiply_us"]), 2),
                        "barrier_wait_us": round(mean(lambda w: w["jobs"][j]["barrier_wait_us"]), 2),
                        "epilogue_publish_us": round(mean(lambda w: w["jobs"][j]["epilogue_publish_us"]), 2)} for j in range(6)],
           "producer_sum_us": {key: round(mean(lambda w: w["producer"]["sum_us"][key]), 2) for key in wgs[0]["producer"]["sum_us"]},
           "producer_stages_seen": wgs[0]["producer"]["stages_seen"]}
    rec["unaccounted_us"] = rec["launch_span_us"] - rec["multiply_us"] - rec["barrier_wait_us (consumers idle: data / flags not there yet)"] - rec["epilogue_publish_us"] - wgs[0]["entry_to_first_stage_barrier_us"]
    return {k2: (round(v, 2) if isinstance(v, float) else v) for k2, v in rec.items()}


g = R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().train()
with torch.no_grad():
    g.conv4.bias.add_(0.5)
x = torch.rand(a.batch, 3, a.res, a.res, device="cuda")
gw = torch.randn(a.batch, 3, 4 * a.res, 4 * a.res, device="cuda")
for _ in range(2):
    (g(x) * gw).sum().backward()
torch.cuda.synchronize()
out = {"geometry": f"{a.batch} x {a.res}^2, fast mode, one dense block of 3 (the LAST chained launch of the pass)"}
y = None


def fwd():
    global y
    y = g(x)


out["forward"] = summarise(capture(fwd), "forward chain: conv1..conv4 (LeakyReLU + sign words) + the two halves of conv5")
loss = (y * gw).sum()
out["backward"] = summarise(capture(lambda: loss.backward()), "backward-data chain: the four mirrored passes (sign-word mask) + the two halves of g_x")
out["chain_errors"] = int(lib.resr_debug_chain_errors())
print(json.dumps(out, indent=1))
if a.json:
    with open(a.json, "w") as f:
        json.dump(out, f, indent=1)
