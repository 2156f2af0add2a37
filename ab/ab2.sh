timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
for rep in 1 2 3; do
for lib in old new; do
    if [ $lib = old ]; then export CATFISH_HIP_LIB=$PWD/ab/libcatfish_hip_old.so; else unset CATFISH_HIP_LIB; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra-precisions --steps 60 --warmup 20 > gpurun_out/ab.json 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1]); print('$lib', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['kernels_ms']['gru_layer_first'], d['kernels_ms']['gru_layer_last'], d['parity']['max_abs_dp_vs_fp64_oracle'])"
done
done
