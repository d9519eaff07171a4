"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs:  python tools/pmc_summary.py <dir> [kernel-name substring]
Prints, for every kernel whose name contains the substring (default "gemm"), launches and the mean value of each counter."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "gemm"
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if want not in name:
                    continue
                m = re.search(r"(\w+_kernel)<([^>]*)>", name)
                short = f"{m.group(1)}<{m.group(2)}>" if m else name[:80]
                acc[short + f" grid={row['Grid_Size']}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main()
