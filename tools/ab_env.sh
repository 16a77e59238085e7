# same-box A/B of an engine switch read from the environment (GPU box): bash tools/ab_env.sh VAR [configs...]
# e.g. bash tools/ab_env.sh FGNN_PACK_IN_STRUCT cfg2 cfg5   -> ms/step with VAR=0 and VAR=1, twice each, alternating
VAR=$1; shift
for rep in 1 2; do
for v in 0 1; do
  for cfg in ${@:-cfg2}; do
    env $VAR=$v python bench.py --config $cfg --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v $cfg ms/step', round(d['ms_per_step'],4), 'module', (d.get('module_surface') or {}).get('ms_per_step'))
"
  done
done
done
