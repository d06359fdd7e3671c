"""Steady-state view of a rocprofv3 rocpd database (kernel trace) of a training loop: only the LAST `steps` whole optimisation
steps are summarised -- a step ends with `ema_kernel` (one launch per step in both train steps), so the window runs from the end
of the (steps+1)-th last `ema_kernel` to the end of the last one.  One-time launches (optimizer-state initialisation, warm-up
allocations, the first steps' compilation of kernels) stay outside.  Per-kernel table with PER-STEP figures, launch count per
step, busy / gap split of the window.

    python tools/rocpd_steady.py <results.db> [steps=4] [top=45]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("resr::", "")
    name = re.sub(r"\(.*\)$", "", name)
    m = re.match(r"_ZN4resr\d+([a-z0-9_]+?)I(.*?)EEv", name)
    if m:   # _ZN4resr17conv3x3_ws_kernelIDF16_Li1ELi2ELi8ELi33ELb1ELi0ELi2EEEv... -> conv3x3_ws_kernel<f16,1,2,8,33,1,0,2>
        raw = re.findall(r"DF16_|L[ib]\d+E|f", m.group(2))
        args = ["f16" if t == "DF16_" else "f32" if t == "f" else re.sub(r"L[ib](\d+)E", r"\1", t) for t in raw]
        return m.group(1) + "<" + ",".join(args) + ">"
    return name[:100]


def main(path, steps=4, top=45, marker="ema_kernel"):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    tab = "kernels" if "kernels" in tables else [t for t in tables if "kernel_dispatch" in t][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({tab})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {tab} order by start").fetchall()
    marks = [e for n, s, e in rows if marker in n]
    if len(marks) < steps + 1:
        print(f"only {len(marks)} `{marker}` launches: need {steps + 1}")
        return
    t0, t1 = marks[-steps - 1], marks[-1]
    win = [(n, s, e) for n, s, e in rows if s >= t0 and e <= t1]
    agg = {}
    for n, s, e in win:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    print(f"steady state: the last {steps} steps ({len(win)} dispatches, {len(win) / steps:.0f} per step, window {(t1 - t0) / 1e6 / steps:.3f} ms per step)")
    print(f"{'kernel':100s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
    aten = [0, 0.0]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if name.startswith("at::native") or "rocclr" in name or name.startswith("at::"):
            aten[0] += a[0]
            aten[1] += a[1]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name:100s} {a[0] / steps:10.1f} {a[1] / 1e6 / steps:9.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:8.2f} {a[3] / 1e3:8.2f} {100 * a[1] / total:6.2f}")
    print(f"TOTAL kernel time {total / 1e6 / steps:.3f} ms per step over {len(win) / steps:.0f} dispatches per step; "
          f"stock ATen / runtime-copy launches: {aten[0] / steps:.1f} per step, {aten[1] / 1e6 / steps:.3f} ms per step")
    # busy / gaps over the window (all streams merged)
    busy, gaps, cur = 0, [], t0
    for n, s, e in win:
        if s > cur:
            gaps.append((s - cur, n))
        busy += max(0, e - max(s, cur))
        cur = max(cur, e)
    span = t1 - t0
    gs = sorted(g for g, _ in gaps)
    print(f"window {span / 1e6 / steps:.3f} ms/step  busy {busy / 1e6 / steps:.3f} ms/step ({100 * busy / span:.1f} %)  gaps {sum(gs) / 1e6 / steps:.3f} ms/step in "
          f"{len(gaps) / steps:.0f} ({100 * sum(gs) / span:.1f} %), median gap {gs[len(gs) // 2] / 1e3:.2f} us, gaps > 50 us: {sum(1 for g in gs if g > 50e3) / steps:.1f} per step "
          f"= {sum(g for g in gs if g > 50e3) / 1e6 / steps:.3f} ms/step")
    small = sum(g for g in gs if g <= 50e3)
    big = sum(g for g in gs if g > 50e3)
    print(f"  of which launch latency between dependent kernels (gaps <= 50 us): {small / 1e6 / steps:.3f} ms/step = {100 * small / max(1, span - big):.1f} % of the span "
          f"without the > 50 us gaps -- those are the HOST falling behind under the tracer (the untraced step is printed by the tool's plain run: "
          f"its length is kernel time + launch latency, the host enqueues ahead)")
    by = {}
    for g, n in gaps:
        a = by.setdefault(short(n)[:70], [0, 0])
        a[0] += 1
        a[1] += g
    print("gap before kernel (top 10 by total, per step):")
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {k:70s} n={n / steps:7.1f} total {t / 1e6 / steps:8.3f} ms  avg {t / n / 1e3:7.2f} us")


def list_last_step(path, substr, marker="ema_kernel"):
    """Every launch of the LAST step whose name contains `substr`, in launch order: start offset in the step, duration."""
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    tab = "kernels" if "kernels" in tables else [t for t in tables if "kernel_dispatch" in t][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({tab})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {tab} order by start").fetchall()
    marks = [e for n, s, e in rows if marker in n]
    t0, t1 = marks[-2], marks[-1]
    print(f"launches of the last step matching `{substr}` (offset ms, duration us, name):")
    for n, s, e in rows:
        if s >= t0 and e <= t1 and (substr in n or substr == "*"):
            print(f"  {(s - t0) / 1e6:8.3f}  {(e - s) / 1e3:8.2f}  {short(n)[:90]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--list":
        list_last_step(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 45)
