"""tools/timeline_gaps.py TRACE.csv -- from a rocprofv3 --kernel-trace CSV of bench.py: per train step (delimited by adam_kernel) the
wall time, the union of kernel-busy time, the idle time, the idle time by the kernel that FOLLOWS the gap, and the time with >= 2
kernels in flight.  Says how much of a step is launch gaps / host stalls rather than kernels."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
print(f"{len(rows)} kernels, {len(adam)} steps")
for s in range(max(1, len(adam) - 3), len(adam)):
    lo, hi = adam[s - 1] + 1, adam[s] + 1
    seg = rows[lo:hi]
    t0, t1 = rows[adam[s - 1]][1], rows[adam[s]][1]
    busy = 0
    multi = 0
    gaps = defaultdict(float)
    ngaps = defaultdict(int)
    cur_end = t0
    events = []
    for a, b, n in seg:
        events.append((a, 1))
        events.append((b, -1))
        if a > cur_end:
            key = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
            gaps[key] += a - cur_end
            ngaps[key] += 1
        cur_end = max(cur_end, b)
    events.sort()
    depth, last = 0, t0
    for t, d in events:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    wall = t1 - t0
    print(f"step {s}: wall {wall / 1e6:.2f} ms, {len(seg)} kernels, busy {busy / 1e6:.2f}, idle {(wall - busy) / 1e6:.2f}, >=2 in flight {multi / 1e6:.2f}, "
          f"sum of kernel durations {sum(b - a for a, b, _ in seg) / 1e6:.2f}")
    top = sorted(gaps.items(), key=lambda kv: -kv[1])[:12]
    for k, v in top:
        print(f"    idle before {k:60s} {v / 1e3:8.1f} us in {ngaps[k]:4d} gaps")
    big = sorted(((a - e, n) for (a, _, n), e in zip(seg[1:], [max(x[1] for x in seg[:i + 1]) for i in range(len(seg) - 1)])), reverse=True)[:8] if len(seg) < 4000 else []
    for g, n in big:
        print(f"    gap {g / 1e3:7.1f} us before {n[:80]}")
