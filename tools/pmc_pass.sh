#!/bin/bash
# One rocprofv3 counter pass over the bench; prints per-kernel means of the requested counters as JSON lines.
# usage (GPU box, repo root):  PREC=bf16 bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
set -e
PREC=${PREC:-fp32}
ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rm -rf /tmp/pmc_pass
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_pass -o pmc -- python3 bench.py --precision $PREC --no-cpu-baseline --no-extra-precisions --no-sharded-leg --no-kernel-events --steps 5 --warmup 2 > /tmp/pmc_pass.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob('/tmp/pmc_pass/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if 'gru_' in k or 'res_block' in k or 'res_stack' in k:
        print(json.dumps({"kernel": k[:70], **{c: sum(v) / len(v) for c, v in cs.items()}, "launches": len(next(iter(cs.values())))}))
PY
