cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for f in tools/abl/libcatfish_old.so catfish_amd/csrc/libcatfish_hip.so; do
  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extra-precisions --no-sharded-leg --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$f'.split('/')[-1], 'value %.1f M  step %.3f ms' % (d['value']/1e6, d['ms_per_step']), {a: round(b*1e3,1) for a,b in k.items()})"
done; done
