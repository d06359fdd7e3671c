"""Idle time between consecutive kernel dispatches of a rocprofv3 rocpd database: how much of the busy span is launch
gaps / dependency bubbles rather than kernels.  Usage: python tools/rocpd_gaps.py <results.db> [min_span_fraction]"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    tab = "kernels" if "kernels" in tables else [t for t in tables if "kernel_dispatch" in t][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({tab})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {tab} order by start").fetchall()
    if not rows:
        print("no kernels")
        return
    # steady-state window: the last 60 % of the dispatches
    rows = rows[int(len(rows) * 0.4):]
    span = rows[-1][2] - rows[0][1]
    busy = 0
    gaps = []
    cur_end = rows[0][1]
    for name, s, e in rows:
        if s > cur_end:
            gaps.append((s - cur_end, name))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    gsum = sum(g for g, _ in gaps)
    print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)  "
          f"gaps {gsum / 1e6:.3f} ms in {len(gaps)} ({100 * gsum / span:.1f} %), median gap {sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3:.2f} us")
    by = {}
    for g, name in gaps:
        k = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("resr::", "").replace("void ", ""))[:70]
        a = by.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += g
    print("gap before kernel (top 12 by total):")
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {k:70s} n={n:6d} total {t / 1e6:8.3f} ms  avg {t / n / 1e3:7.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
