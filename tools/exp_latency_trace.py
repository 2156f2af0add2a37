"""Timeline of ONE small call (a single read: 118 windows, fp32): per-kernel durations and the gaps between them.

  workload:  rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python3 tools/exp_latency_trace.py run
  analysis:  python3 tools/exp_latency_trace.py show /tmp/lt
The workload makes 200 synchronised calls; the analysis takes the last 100, aligns each call's kernels by order and prints the
median start offset, duration and gap to the previous kernel's end.
"""
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import bench
    from catfish_amd.engine import HipEngine
    eng = HipEngine(bench.load_weights(), device=0, max_windows_per_pass=4096)
    x = torch.randn(int(os.environ.get("LT_WINDOWS", 118)), 35, device="cuda")
    y = torch.empty(x.shape[0] * 35, device="cuda")
    for _ in range(200):
        eng.infer_device(x, out=y)
        torch.cuda.synchronize()
    eng.close()


def show(directory):
    f = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    rows = [r for r in rows if "catfish" in r[2] or "gru_" in r[2] or "res_" in r[2] or "head_kernel" in r[2]]
    # a call = the kernels from one residual launch to the next
    calls, cur = [], []
    for r in rows:
        if "res_" in r[2] and cur:
            calls.append(cur)
            cur = []
        cur.append(r)
    calls.append(cur)
    calls = [c for c in calls[-100:] if len(c) == len(calls[-1])]
    n = len(calls[0])
    print("%d calls of %d kernels" % (len(calls), n))
    total = np.median([c[-1][1] - c[0][0] for c in calls]) / 1e3
    for k in range(n):
        off = np.median([c[k][0] - c[0][0] for c in calls]) / 1e3
        dur = np.median([c[k][1] - c[k][0] for c in calls]) / 1e3
        gap = np.median([c[k][0] - c[k - 1][1] for c in calls]) / 1e3 if k else 0.0
        print("  +%7.1f us  %6.1f us  gap %5.1f us  %s" % (off, dur, gap, calls[0][k][2][:70]))
    print("first start -> last end: %.1f us; sum of durations %.1f us, of gaps %.1f us" % (
        total, sum(np.median([c[k][1] - c[k][0] for c in calls]) for k in range(n)) / 1e3,
        sum(np.median([c[k][0] - c[k - 1][1] for c in calls]) for k in range(1, n)) / 1e3))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
