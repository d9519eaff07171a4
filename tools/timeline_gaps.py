"""How busy is the GPU inside a train step?  Reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per dispatch) and prints,
for the steady-state part (after the first `skip` fraction of the dispatches): span, union of the kernels' intervals (time with at least one
kernel running), idle time (gaps), sum of the kernel durations, time with >= 2 / >= 3 kernels in flight, and the longest gaps with the kernels
either side of them.
    python tools/timeline_gaps.py <kernel_trace.csv> [skip_fraction=0.4]"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _n, _q in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = {1: 0, 2: 0, 3: 0, 4: 0}
    depth, last = 0, t0
    for t, d in ev:
        for k in busy:
            if depth >= k:
                busy[k] += t - last
        depth += d
        last = t
    span = t1 - t0
    ksum = sum(e - s for s, e, _n, _q in rows)
    print(f"dispatches {len(rows)}, queues {len(set(r[3] for r in rows))}, span {span / 1e6:.2f} ms, kernel-duration sum {ksum / 1e6:.2f} ms "
          f"({ksum / span:.2f} x span)")
    print(f"at least 1 kernel in flight {busy[1] / 1e6:.2f} ms ({100 * busy[1] / span:.1f} %), idle {(span - busy[1]) / 1e6:.2f} ms; "
          f">= 2 in flight {100 * busy[2] / span:.1f} %, >= 3 {100 * busy[3] / span:.1f} %, >= 4 {100 * busy[4] / span:.1f} %")
    # gaps: between the end of everything so far and the next start
    gaps = []
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, n, _q in rows[1:]:
        if s > end:
            gaps.append((s - end, prev, n))
        if e > end:
            end, prev = e, n
    gaps.sort(reverse=True)
    print(f"gaps: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.2f} ms; > 20 us: {sum(1 for g in gaps if g[0] > 20000)} "
          f"({sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms)")
    for g, a, b in gaps[:12]:
        print(f"  {g / 1e3:8.1f} us  after {a[:70]}  before {b[:70]}")
    # time during which exactly ONE kernel is in flight, by kernel: what a chain of dependent launches waits for; and the idle time in front
    # of a kernel (launch latency of a dependent node), by the kernel that ends the gap
    alone, before = {}, {}
    ev2 = []
    for i, (s_, e_, n_, _q) in enumerate(rows):
        ev2.append((s_, 1, i))
        ev2.append((e_, -1, i))
    ev2.sort()
    live, last = set(), t0
    for t, d, i in ev2:
        if len(live) == 1:
            n_ = rows[next(iter(live))][2]
            alone[n_] = alone.get(n_, 0) + (t - last)
        elif len(live) == 0 and d == 1 and t > last:
            n_ = rows[i][2]
            before[n_] = before.get(n_, 0) + (t - last)
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
        last = t
    tot_alone = sum(alone.values())
    print(f"exactly one kernel in flight: {tot_alone / 1e6:.2f} ms ({100 * tot_alone / span:.1f} % of the span), by kernel:")
    for n_, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {v / 1e6:8.2f} ms {100 * v / span:5.1f} %  {n_[:110]}")
    print("idle time in front of a kernel, by kernel:")
    for n_, v in sorted(before.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v / 1e6:8.2f} ms  {n_[:110]}")


if __name__ == "__main__":
    main()
