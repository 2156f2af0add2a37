cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "4 128" "4 256" "8 64" "2 256" "2 128" "4 64"; do
  set -- $cfg
  CATFISH_DEBUG_KNOBS=1 CATFISH_BF16_WAVES=$1 CATFISH_BF16_WGS=$2 python bench.py --precision bf16 --no-cpu-baseline --no-extra-precisions --no-sharded-leg --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('waves $1 wgs/dir $2: step %.3f ms  gru0 %.1f mid %.1f last %.1f us' % (d['ms_per_step'], k['gru_layer_first']*1e3, k['gru_layer_mid']*1e3, k['gru_layer_last']*1e3))"
done
