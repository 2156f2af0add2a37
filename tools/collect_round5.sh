#!/bin/bash
# GPU box: the round-5 measurements the docs cite, written under gpurun_out/round5/ (copy what is judged into profiles/).
# usage:  bash tools/collect_round5.sh <commit> [stage ...]     stages: probe tests bench rehearse prof clip ingest x3abl (default: probe tests bench rehearse prof)
COMMIT=${1:-unknown}; shift
STAGES=${@:-probe tests bench rehearse prof}
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/round5"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
echo "commit $COMMIT stages $STAGES" > $OUT/commit.txt
for S in $STAGES; do
  case $S in
    probe) bash tools/probe_sysfs.sh > $OUT/sysfs_probe.log 2>&1 ;;
    tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { echo "gpu tests FAILED" >> $OUT/commit.txt; tail -30 $OUT/gpu_tests.log; exit 1; } ;;
    bench) timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench FAILED" >> $OUT/commit.txt; tail -20 $OUT/bench_default.err; exit 1; } ;;
    rehearse)
      for N in 2 4; do
        CATFISH_BENCH_DEVICE=0 CATFISH_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
          --master-port $((29500 + N)) bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_${N}ranks_one_gpu.json 2> $OUT/bench_${N}ranks.err \
          || { echo "rehearsal $N FAILED" >> $OUT/commit.txt; tail -20 $OUT/bench_${N}ranks.err; exit 1; }
      done ;;
    x3abl)   # upper bounds for the bf16x3 layer-0 experiment: timing-only ablations, interleaved rounds on one box
      bash tools/ab_x3_variants.sh 2 default tools/x3var/libcatfish_x3_abl4.so tools/x3var/libcatfish_x3_abl1.so tools/x3var/libcatfish_x3_abl2.so \
        > $OUT/x3_layer0_ablation.log 2>&1 ;;
    clip)    # the CLI by precision + the file pool by thread count
      TMPDIR=/tmp python tools/exp_loader_threads.py 2>&1 | grep -v amdgpu.ids > $OUT/cli_by_precision.log ;;
    ingest)  # standalone times of the ingest / post-processing kernels, round 5 against round 1
      python tools/exp_ingest_post.py > $OUT/ingest_post_kernels.log 2>&1 ;;
    prof)
      for P in fp32 bf16 bf16x3; do
        rm -rf /tmp/kt_$P
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$P -o kt -- python3 bench.py --precision $P --no-cpu-baseline --no-extra-precisions --no-sharded-leg > $OUT/bench_prof_$P.json 2>/dev/null
        cp $(find /tmp/kt_$P -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_$P.csv
      done ;;
  esac
  echo "$S done" >> $OUT/commit.txt
done
echo "all done" >> $OUT/commit.txt
