#!/bin/bash
# GPU box: the round-4 measurements the docs cite, written under gpurun_out/round4/ (copy what is judged into profiles/).
# usage:  bash tools/collect_round4.sh <commit>
COMMIT=${1:-unknown}
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/round4"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
echo "commit $COMMIT" > $OUT/commit.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done" >> $OUT/commit.txt
for P in fp32 bf16 bf16x3; do
  rm -rf /tmp/kt_$P
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$P -o kt -- python3 bench.py --precision $P --no-cpu-baseline --no-extra-precisions --no-sharded-leg > $OUT/bench_prof_$P.json 2>/dev/null
  cp $(find /tmp/kt_$P -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_$P.csv
done
echo "kernel stats done" >> $OUT/commit.txt
bash tools/collect_traffic.sh $COMMIT bf16x3 > $OUT/traffic_bf16x3.log 2>&1
cp gpurun_out/traffic_bf16x3.json $OUT/ 2>/dev/null
bash tools/collect_traffic_config4.sh $COMMIT > $OUT/traffic_bf16_config4.log 2>&1
cp gpurun_out/traffic_bf16_config4.json gpurun_out/kernel_stats_config4.csv $OUT/ 2>/dev/null
echo "traffic done" >> $OUT/commit.txt
PREC=bf16x3 bash tools/pmc_pass.sh SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU > $OUT/pmc_bf16x3_a.jsonl 2>&1
PREC=bf16x3 bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE > $OUT/pmc_bf16x3_b.jsonl 2>&1
echo "pmc done" >> $OUT/commit.txt
python tools/merge_pmc.py $COMMIT bf16x3=$OUT/pmc_bf16x3_a.jsonl,$OUT/pmc_bf16x3_b.jsonl > $OUT/pmc.json
# diagnostic builds of the bf16x3 kernel (tools/build_x3_variants.sh stamp abl:1 abl:2, built before the run: tools/x3var/ travels)
if [ -f tools/x3var/libcatfish_x3_stamp_8_8_0.so ]; then
  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/x3var/libcatfish_x3_stamp_8_8_0.so python tools/exp_x3_stamps.py 2>&1 | grep -v "amdgpu.ids" > $OUT/x3_stamps.log
  X3_LAYERS=2 CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/x3var/libcatfish_x3_stamp_8_8_0.so python tools/exp_x3_stamps.py 2>&1 | grep -v "amdgpu.ids" > $OUT/x3_stamps_layer0.log
fi
if [ -f tools/x3var/libcatfish_x3_abl1.so ]; then
  bash tools/ab_x3_variants.sh 2 default tools/x3var/libcatfish_x3_abl1.so tools/x3var/libcatfish_x3_abl2.so > $OUT/x3_ablation.log 2>&1
fi
python tools/exp_latency_parts.py > $OUT/latency_parts.log 2>/dev/null
python tools/bench_latency.py > $OUT/latency.json 2>/dev/null
echo "all done" >> $OUT/commit.txt
