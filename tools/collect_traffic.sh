#!/bin/bash
# HBM traffic of the dominant kernel from rocprofv3 PMC passes of the CURRENT code -> profiles/traffic.json
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, they cost 3 + 2; MI355X_MICROARCH.md).
# usage (GPU box, repo root):  bash tools/collect_traffic.sh <commit> [precision]
# writes gpurun_out/traffic_<precision>.json; copy it to profiles/traffic.json (fp32) after the run.
set -e
COMMIT=${1:-unknown}
PREC=${2:-fp32}
ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python3 bench.py --precision $PREC --no-cpu-baseline --no-extra-precisions --no-sharded-leg --no-kernel-events --steps 3 --warmup 1 --prewarm-ms 50 > /tmp/pmc_$C.log 2>&1
done
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace -o kt -- python3 bench.py --precision $PREC --no-cpu-baseline --no-extra-precisions --no-sharded-leg --no-kernel-events --steps 20 --warmup 5 > /tmp/ktrace.log 2>&1
python3 - "$COMMIT" "$PREC" <<'PY'
import csv, glob, json, sys, time, collections
commit, prec = sys.argv[1], sys.argv[2]
pat = "gru_layer_kernel<128, false>" if prec == "fp32" else ("gru_bf16_pipe_kernel<128, false>" if prec == "bf16" else "gru_bf16x3_pipe_kernel<128, false>")
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"] and r["Counter_Name"] == c]
    vals[c] = sum(v) / len(v)
ks = glob.glob("/tmp/ktrace/**/*kernel_stats.csv", recursive=True)[0]
avg_ms = [float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(ks)) if pat in r["Name"]][0]
out = {"gru_layer_mid_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
       "precision": prec, "kernel": pat, "commit": commit, "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
       "kernel_avg_ms": avg_ms,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the gfx950 correction for 16 B/lane streaming reads "
                 "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16 B/lane stores; kernel_avg_ms from a separate "
                 "--kernel-trace --stats pass (20 steps)",
       "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"]}
json.dump(out, open("gpurun_out/traffic_%s.json" % prec, "w"), indent=1)
print(json.dumps(out))
PY
cp /tmp/ktrace/*/*kernel_stats.csv gpurun_out/kernel_stats_$PREC.csv 2>/dev/null || cp $(find /tmp/ktrace -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_$PREC.csv
