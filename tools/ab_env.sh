# A/B of one environment switch on one box: usage ab_env.sh VAR "<bench args>" [reps]   (VAR=1 against VAR unset, interleaved)
mkdir -p gpurun_out/ab
VAR="$1"; ARGS="$2"; REPS="${3:-2}"
for i in $(seq 1 $REPS); do
  for v in unset 1; do
    if [ $v = unset ]; then unset $VAR; else export $VAR=1; fi
    timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --roofline-steps 1 > gpurun_out/ab/ab_${v}_$i.json 2>/dev/null || exit 1
    python -c "
import json,sys;d=json.loads(open('gpurun_out/ab/ab_${v}_$i.json').read().strip().splitlines()[-1]);print('$ARGS $VAR=$v run $i', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['roofline'].get('executed_frac') or 0,4))"
  done
done
unset $VAR
