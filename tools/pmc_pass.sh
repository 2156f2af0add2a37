#!/bin/bash
# One rocprofv3 counter pass over the fp32 bench; prints per-kernel means of the requested counters.
# usage (GPU box, repo root):  bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_pass
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_pass -o pmc -- python3 bench.py --no-cpu-baseline --no-extra-precisions --no-kernel-events --steps 5 --warmup 2 > /tmp/pmc_pass.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_pass/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if 'gru_layer_kernel' in k or 'res_block' in k:
        print(k[:60], {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
