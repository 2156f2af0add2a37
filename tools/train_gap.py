"""Kernel time against wall time of the config-5 training step (BASELINE configs[4]; reference rnn_class.py:201-210).

usage: train_gap.py <rocprofv3 *_kernel_stats.csv> <bench_train --profile-only JSON under the profiler> <the same without profiler>

The traced command (tools/bench_train.py --profile-only) runs nothing but whole training steps of one trainer, so every traced
launch belongs to a step.  Steps are counted by the optimizer kernel (one launch per updating step); a kernel's launches per step
are its calls over that count, rounded (the two eager warm-up runs before the graph capture skip the optimizer and add < 1 %).
Prints one JSON object: per kernel launches per step / average us / us per step, the kernel-time sum per step, the wall time per
step with and without the profiler attached, and the share of a step in which no kernel runs (launch gaps + host copies +
the loss read-back)."""
import csv
import json
import sys


def main():
    stats, prof, plain = sys.argv[1:4]
    with open(stats) as fh:
        rows = list(csv.DictReader(fh))
    with open(prof) as fh:
        under = json.loads(fh.read().strip().splitlines()[-1])
    with open(plain) as fh:
        free = json.loads(fh.read().strip().splitlines()[-1])
    opt = [r for r in rows if "opt_step" in r["Name"]]
    n_steps = int(opt[0]["Calls"]) if opt else 2 * under["steps"]
    kernels, total_ns = [], 0.0
    for r in rows:
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        per_step = calls / n_steps
        kernels.append({"name": r["Name"][:110], "calls": calls, "launches_per_step": round(per_step, 2), "avg_us": tot / calls / 1e3,
                        "us_per_step": tot / n_steps / 1e3})
        total_ns += tot
    kernels.sort(key=lambda k: -k["us_per_step"])
    ksum_ms = total_ns / n_steps / 1e6
    wall = free["ms_per_step"]
    print(json.dumps({
        "batch": free["batch"], "overlap_wgrad": free.get("overlap_wgrad"), "steps_traced": n_steps, "launches_per_step": round(sum(k["launches_per_step"] for k in kernels), 1),
        "kernel_sum_ms_per_step": ksum_ms, "wall_ms_per_step": wall, "wall_ms_per_step_under_profiler": under["ms_per_step"],
        "graph_device_ms": free["parts_ms"]["graph_device_ms"], "parts_ms": free["parts_ms"],
        "no_kernel_share_of_wall": 1.0 - ksum_ms / wall, "gaps_inside_graph_ms": free["parts_ms"]["graph_device_ms"] - ksum_ms,
        "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
