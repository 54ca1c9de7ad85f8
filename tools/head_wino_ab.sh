mkdir -p gpurun_out/hw
X=res5a_branch2b,res5b_branch2b,res5c_branch2b
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --roofline-steps 0 > gpurun_out/hw/$name.json 2>gpurun_out/hw/$name.err || { tail -3 gpurun_out/hw/$name.err; return 1; }
  python -c "
import json;d=json.loads(open('gpurun_out/hw/$name.json').read().strip().splitlines()[-1]);print('$name', round(d['value'],1), round(d['ms_per_step'],4), d['losses']['det_cls'])"
}
for i in 1 2; do
run base_$i A=1 || exit 1
run stale_$i RADNET_WINOGRAD_EXTRA=$X RADNET_HEAD_WINO_STALE=1 || exit 1
run refresh_$i RADNET_WINOGRAD_EXTRA=$X || exit 1
done
