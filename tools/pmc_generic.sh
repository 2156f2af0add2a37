#!/bin/bash
# One rocprofv3 counter pass over the any-size path at one geometry (GEN="h,c,layers,blocks"); per-kernel counter means as JSON lines.
# usage (GPU box, repo root):  GEN=128,64,3,2 bash tools/pmc_generic.sh SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
set -e
export GEN=${GEN:-128,64,3,2}
ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rm -rf /tmp/pmc_gen
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_gen -o pmc -- python3 tools/exp_generic_parts.py > /tmp/pmc_gen.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob('/tmp/pmc_gen/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if 'gen_' in k:
        print(json.dumps({"kernel": k[:40], **{c: sum(v) / len(v) for c, v in cs.items()}, "launches": len(next(iter(cs.values())))}))
PY
