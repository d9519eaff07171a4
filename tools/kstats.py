"""summarise a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [steps] [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.2f} ms/step over {steps:g} steps")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.2f} ms/step {float(r['Calls']) / steps:6.1f} calls/step avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
