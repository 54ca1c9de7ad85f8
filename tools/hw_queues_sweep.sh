mkdir -p gpurun_out/q
for q in 4 6 8 10 12 16 8 4; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --roofline-steps 0 > gpurun_out/q/q$q.json 2>/dev/null || exit 1
  python -c "
import json;d=json.loads(open('gpurun_out/q/q$q.json').read().strip().splitlines()[-1]);print('GPU_MAX_HW_QUEUES=$q', round(d['value'],1))"
done
