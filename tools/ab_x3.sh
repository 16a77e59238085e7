for rep in 1 2 3; do
for v in "f32|" "x3|--mfma x3" "x3pair|--mfma x3" "x3fwd|--mfma x3"; do
  name=${v%%|*}; a=${v#*|}
  parts="fwd,pair"; [ $name = x3pair ] && parts=pair; [ $name = x3fwd ] && parts=fwd
  FGNN_X3_PARTS=$parts python bench.py --no-cpu-baseline --no-extra-configs --profile-steps 0 --steps 20 --warmup 5 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-7s struct %.4f [%.4f %.4f]' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max']))" $name
done; done
