# A/B of the fused stage-2 bottleneck tail (RADNET_NO_BNECK_FUSE=1 = separate launches) on one box: usage ab_bneck.sh "<bench args>" [reps]
mkdir -p gpurun_out/ab
ARGS="$1"; REPS="${2:-2}"
for i in $(seq 1 $REPS); do
  for v in 0 1; do
    RADNET_NO_BNECK_FUSE=$v timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --roofline-steps 1 > gpurun_out/ab/ab_${v}_$i.json 2>/dev/null || exit 1
    python -c "
import json,sys;d=json.loads(open('gpurun_out/ab/ab_${v}_$i.json').read().strip().splitlines()[-1]);print('$ARGS nofuse=$v run $i', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['frac'], d['roofline'].get('executed_frac'))"
  done
done
