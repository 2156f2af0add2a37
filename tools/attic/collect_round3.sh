#!/bin/bash
# GPU box: the round-3 measurements the docs cite, written under gpurun_out/round3/ (copy what is judged into profiles/).
# usage:  bash tools/collect_round3.sh <commit>
COMMIT=${1:-unknown}
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/round3"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
echo "commit $COMMIT" > $OUT/commit.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done" >> $OUT/commit.txt
for P in fp32 bf16; do
  rm -rf /tmp/kt_$P
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$P -o kt -- python3 bench.py --precision $P --no-cpu-baseline --no-extra-precisions --no-sharded-leg > $OUT/bench_prof_$P.json 2>/dev/null
  cp $(find /tmp/kt_$P -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_$P.csv
done
echo "kernel stats done" >> $OUT/commit.txt
bash tools/collect_traffic.sh $COMMIT fp32 > $OUT/traffic_fp32.log 2>&1
cp gpurun_out/traffic_fp32.json $OUT/ 2>/dev/null
bash tools/collect_traffic.sh $COMMIT bf16 > $OUT/traffic_bf16.log 2>&1
cp gpurun_out/traffic_bf16.json $OUT/ 2>/dev/null
echo "traffic done" >> $OUT/commit.txt
PREC=bf16 bash tools/pmc_pass.sh SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU > $OUT/pmc_bf16_a.jsonl 2>&1
PREC=bf16 bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE > $OUT/pmc_bf16_b.jsonl 2>&1
PREC=fp32 bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES > $OUT/pmc_fp32.jsonl 2>&1
echo "pmc done" >> $OUT/commit.txt
python tools/bench_e2e.py > $OUT/e2e.log 2>/dev/null
echo "all done" >> $OUT/commit.txt
