#!/bin/bash
# same-box A/B of one environment switch: tools/r05_ab.sh VAR "v1 v2 v1 v2" [forward|train] -> gpurun_out/r05/ab_<VAR>_<mode>.txt
set -u
out=$PWD/gpurun_out/r05; mkdir -p "$out"
var=$1; vals=$2; mode=${3:-forward}
for v in $vals; do
  if [ "$mode" = train ]; then
    env $var=$v python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $var=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'loss', d['final_loss'])"
  else
    env $var=$v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms; conv launches summed', d['roofline_conv']['sum_launch_ms'], 'ms; lookup frac', d['roofline']['frac'])"
  fi
done | tee "$out/ab_${var}_${mode}.txt"
