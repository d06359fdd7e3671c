"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table (like --stats)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("resr::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main(path, top=40):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, (end - start) from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name:110s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / total:6.2f}")
    print(f"TOTAL kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
