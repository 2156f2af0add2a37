#!/bin/bash
# BASELINE configs[3] (bench.py leg config4: 10 000 reads in 131 072-window bf16 launches): counter traffic and matrix-pipe
# occupancy of the mid biGRU kernel FOR THAT LAUNCH SHAPE, from rocprofv3 PMC passes over `bench.py --only-leg config4`
# (one warm pass + three timed passes over the reads, nothing else) -> gpurun_out/traffic_bf16_config4.json; copy it to
# profiles/traffic_bf16_config4.json, where bench.py's config4 leg reads it.
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots, they cost 3 + 2; MI355X_MICROARCH.md); the program comes
# directly after `--`.
# usage (GPU box, repo root):  bash tools/collect_traffic_config4.sh <commit>
set -e
COMMIT=${1:-unknown}
ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc4_$(echo $C | tr ' ' '_')
  rm -rf $D
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python3 bench.py --only-leg config4 > $D.log 2>&1
done
rm -rf /tmp/ktrace4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace4 -o kt -- python3 bench.py --only-leg config4 > /tmp/ktrace4.log 2>&1
python3 - "$COMMIT" <<'PY'
import csv, glob, json, sys, time
commit = sys.argv[1]
pat = "gru_bf16_pipe_kernel<128, false>"
leg = json.loads([l for l in open("/tmp/ktrace4.log") if l.startswith("{")][-1])
passes = leg["passes"]
def total(dirname, counter):
    f = glob.glob("/tmp/%s/**/*counter_collection.csv" % dirname, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v), len(v)
fetch, n_f = total("pmc4_FETCH_SIZE", "FETCH_SIZE")
write, n_w = total("pmc4_WRITE_SIZE", "WRITE_SIZE")
busy, n_b = total("pmc4_SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES")
active, _ = total("pmc4_SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE")
ks = glob.glob("/tmp/ktrace4/**/*kernel_stats.csv", recursive=True)[0]
row = [r for r in csv.DictReader(open(ks)) if pat in r["Name"]][0]
out = {"gru_layer_mid_bytes_per_pass": (2 * fetch + write) * 1024 / passes, "launches_per_pass": n_f // passes,
       "precision": "bf16", "kernel": pat, "commit": commit, "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
       "mid_layer_ms_per_pass": float(row["TotalDurationNs"]) / 1e6 / passes, "mid_layer_launches": int(row["Calls"]),
       "mfma_busy_frac": busy / (active / 8 * 1024),
       "method": "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in separate passes over "
                 "`python3 bench.py --only-leg config4` (%d passes over the 10 000 reads, %d launches of the kernel each); bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the launches / passes: FETCH_SIZE doubled per the gfx950 correction for "
                 "16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                 "(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), both summed over the launches; ms from a separate --kernel-trace --stats pass" % (
                     passes, n_f // passes),
       "FETCH_SIZE_KB_raw_per_pass": fetch / passes, "WRITE_SIZE_KB_per_pass": write / passes, "leg": leg}
json.dump(out, open("gpurun_out/traffic_bf16_config4.json", "w"), indent=1)
print(json.dumps(out))
PY
cp $(find /tmp/ktrace4 -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_config4.csv
