#!/bin/bash
# GPU box: the round-6 measurements the docs cite, written under gpurun_out/round6/ (copy what is judged into profiles/).
# usage:  bash tools/collect_round6.sh <commit> [stage ...]
#   stages: tests bench train rehearse5 prof streams traffic   (default: tests bench train)
COMMIT=${1:-unknown}; shift
STAGES=${@:-tests bench train}
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/round6"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
echo "commit $COMMIT stages $STAGES" > $OUT/commit.txt
for S in $STAGES; do
  case $S in
    tests) timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { echo "gpu tests FAILED" >> $OUT/commit.txt; tail -40 $OUT/gpu_tests.log; exit 1; } ;;
    bench) timeout -k 10 700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench FAILED" >> $OUT/commit.txt; tail -20 $OUT/bench_default.err; exit 1; } ;;
    train)   # config 5 (the training step): host/device breakdown of a step + rocprofv3 kernel trace of the same command, at 256 and 4096 windows,
             # weight gradients beside the next layer's backward (default) and in round 5's order (--no-overlap), interleaved twice
      for B in 256 4096; do
        for R in 1 2; do for V in overlap serial; do
          F=""; [ $V = serial ] && F="--no-overlap"
          timeout -k 10 300 python3 tools/bench_train.py --batch $B --profile-only --steps 300 $F > $OUT/train_parts_${B}_${V}_$R.json 2> $OUT/train_parts_$B.err \
            || { echo "train parts $B $V FAILED" >> $OUT/commit.txt; tail -20 $OUT/train_parts_$B.err; exit 1; }
        done; done
        for V in overlap serial; do
          F=""; [ $V = serial ] && F="--no-overlap"
          rm -rf /tmp/kt_train_$B
          timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_train_$B -o kt -- python3 tools/bench_train.py --batch $B --profile-only --steps 300 $F \
            > $OUT/train_prof_${B}_$V.json 2> $OUT/train_prof_$B.err || { echo "train prof $B FAILED" >> $OUT/commit.txt; tail -20 $OUT/train_prof_$B.err; exit 1; }
          cp $(find /tmp/kt_train_$B -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats_${B}_$V.csv
          python3 tools/train_gap.py $OUT/train_kernel_stats_${B}_$V.csv $OUT/train_prof_${B}_$V.json $OUT/train_parts_${B}_${V}_2.json > $OUT/train_gap_${B}_$V.json
        done
      done ;;
    dxchunks)   # launch shape of the deferred input-gradient kernel in the training step (results do not depend on it), interleaved
      rm -f $OUT/train_dx_chunks.log
      for B in 256 1024; do for R in 1 2 3; do for V in 7 12 18 35; do
        CATFISH_DEBUG_KNOBS=1 CATFISH_DX_CHUNKS=$V timeout -k 10 300 python3 tools/bench_train.py --batch $B --profile-only --steps 300 2>> $OUT/train_dx_chunks.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, 'chunks', $V, $R, round(d['ms_per_step'],4), round(d['parts_ms']['graph_device_ms'],4))" >> $OUT/train_dx_chunks.log \
          || { echo "dxchunks $B $V FAILED" >> $OUT/commit.txt; tail -20 $OUT/train_dx_chunks.err; exit 1; }
      done; done; done ;;
    rehearse5)   # the N > 1 launch path over a directory of configs[2]'s full size (100 000 files), FIVE ranks on ONE card (the box allows at most 6 processes on its GPU and the launcher counts: a 6-rank try was stopped by its process guard): a rehearsal (value null), never a measurement
      CATFISH_BENCH_DEVICE=0 CATFISH_DEVICE=0 CATFISH_RCCL_PROBE_TIMEOUT_S=60 timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 \
        --master-addr 127.0.0.1 --master-port 29506 bench.py --gpus 5 --steps 20 --warmup 5 --reads-per-rank 20000 \
        > $OUT/bench_5ranks_100k_files_one_gpu.json 2> $OUT/bench_5ranks.err || { echo "rehearsal FAILED" >> $OUT/commit.txt; tail -30 $OUT/bench_5ranks.err; exit 1; } ;;
    traffic)   # counter traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-busy of the fp32 headline kernels at this commit
      bash tools/collect_traffic.sh $COMMIT fp32 > $OUT/traffic_fp32.log 2>&1 || { echo "traffic FAILED" >> $OUT/commit.txt; tail -20 $OUT/traffic_fp32.log; exit 1; }
      cp gpurun_out/traffic_fp32.json $OUT/ 2>/dev/null
      PREC=fp32 bash tools/pmc_pass.sh SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES > $OUT/pmc_fp32_a.jsonl 2>&1 \
        || { echo "pmc FAILED" >> $OUT/commit.txt; tail -20 $OUT/pmc_fp32_a.jsonl; exit 1; }
      python3 tools/merge_pmc.py $COMMIT fp32=$OUT/pmc_fp32_a.jsonl > $OUT/pmc.json 2>> $OUT/traffic_fp32.log || true ;;
    prof)
      for P in ${PRECS:-fp32}; do
        rm -rf /tmp/kt_$P
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$P -o kt -- python3 bench.py --precision $P --no-cpu-baseline --no-extra-precisions --no-sharded-leg > $OUT/bench_prof_$P.json 2>/dev/null
        cp $(find /tmp/kt_$P -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_$P.csv
      done ;;
    streams)   # two engines on two streams, alternate steps, configs[1] -- the gated look at the fp32 tail
      timeout -k 10 300 python3 tools/exp_two_engines.py > $OUT/two_engines_overlap.json 2> $OUT/two_engines_overlap.err \
        || { echo "streams FAILED" >> $OUT/commit.txt; tail -20 $OUT/two_engines_overlap.err; exit 1; } ;;
  esac
  echo "$S done" >> $OUT/commit.txt
done
echo "all done" >> $OUT/commit.txt
