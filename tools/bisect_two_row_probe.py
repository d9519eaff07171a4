"""Which change to the GEMM kernels removed round 1's multi-queue deviation?  For a commit: check its tree out under build/old/, put the
two-rows-per-trip LayerNorm forward (the kernel that deviated) into its rowwise.hip, build its library in place; the commit's own
tools/stream_order_probe.py then runs on the GPU box from that tree (GRAPPA_GEMM_PRECISION=bf16x3).

    python tools/bisect_two_row_probe.py <sha> [...]      # build only; run on the box:  cd build/old/w_<sha> && python tools/stream_order_probe.py
"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ln_two_rows_variant import MARK, TWO_ROWS  # noqa: E402


def build(sha):
    wt = os.path.join(ROOT, "build", "old", f"w_{sha}")
    if not os.path.isdir(wt):
        subprocess.run(["git", "-C", ROOT, "worktree", "add", "-f", wt, sha], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    csrc = os.path.join(wt, "grappa_amd", "csrc")
    p = os.path.join(csrc, "rowwise.hip")
    src = open(p).read()
    if "two rows per trip: the second row" not in src:
        if "ld4(xr" not in src:
            # before the kernels were templated on the element type: the file of the commit that still had the two-row kernel
            src = subprocess.run(["git", "-C", ROOT, "show", "1fd0283:grappa_amd/csrc/rowwise.hip"], check=True, capture_output=True, text=True).stdout
        else:
            assert src.count(MARK) == 1, sha
            two = TWO_ROWS
            if "y_amax" not in src:       # before the row maxima: the same kernel without them
                two = re.sub(r"\n\s*am[01] = max\(am[01], mag4\(o\)\);", "", two)
                two = two.replace("            unsigned am0 = 0u, am1 = 0u;\n", "")
                i, j = two.index("            if (y_amax) {"), two.index("        }\n        return;")
                two = two[:i] + two[j:]
            src = src.replace(MARK, two + MARK)
        open(p, "w").write(src)
    subprocess.run(["make", "-C", csrc, "-j8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print(sha, "built:", os.path.join(wt, "grappa_amd", "libgrappa_hip.so"))


if __name__ == "__main__":
    for s in sys.argv[1:]:
        build(s)
