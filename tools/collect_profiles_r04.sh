# Round-4 profile collection (GPU box): bash tools/collect_profiles_r04.sh   -> gpurun_out/r04/*, summaries to copy into profiles/r04_*
set -e
TAG=${TAG:-r04}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/$TAG/tune.txt
rm -f $T
# the bench line with the shipped tables (what the driver runs), 300-step and driver form
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_20_steps_driver_form.json 2> gpurun_out/$TAG/bench20.err
python bench.py --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 40 --warmup 10 > /dev/null 2>&1      # table of this box for the traced runs
# pipelined run under the tracer (lane timeline)
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/pipe -o pipe --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 40 --warmup 20 > gpurun_out/$TAG/pipe.log 2>&1
python tools/lane_timeline.py gpurun_out/$TAG/pipe/pipe_kernel_trace.csv 0 > gpurun_out/$TAG/lanes.txt
cp gpurun_out/$TAG/pipe/pipe_kernel_stats.csv gpurun_out/$TAG/kernel_stats_pipelined.csv 2>/dev/null || true
# one-lane run: per-kernel durations without co-running kernels (what the roofline leg also measures)
export RADNET_SIDE_PREFETCH=0
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/serial -o serial --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 40 --warmup 20 > gpurun_out/$TAG/serial.log 2>&1
python tools/trace_summary.py gpurun_out/$TAG/serial/serial_kernel_trace.csv 44 > gpurun_out/$TAG/trace_summary.txt
python tools/trace_sequence.py gpurun_out/$TAG/serial/serial_kernel_trace.csv > gpurun_out/$TAG/kernel_sequence_one_lane.txt 2>/dev/null || true
cp gpurun_out/$TAG/serial/serial_kernel_stats.csv gpurun_out/$TAG/kernel_stats_one_lane.csv 2>/dev/null || true
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/$TAG/pmc1 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$TAG/pmc2 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$TAG/pmc3 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc3.log 2>&1
python tools/pmc_summary.py gpurun_out/$TAG/pmc1 gpurun_out/$TAG/pmc2 gpurun_out/$TAG/pmc3 > gpurun_out/$TAG/pmc_summary.txt
unset RADNET_SIDE_PREFETCH
cp $T gpurun_out/$TAG/tuned_launch_shapes.txt
rm -rf gpurun_out/$TAG/pmc1 gpurun_out/$TAG/pmc2 gpurun_out/$TAG/pmc3 gpurun_out/$TAG/pipe gpurun_out/$TAG/serial
bash tools/rpn_gemm_traffic.sh gpurun_out/$TAG/rpn_traffic > gpurun_out/$TAG/rpn_gemm_traffic.txt 2>&1 || true
rm -rf gpurun_out/$TAG/rpn_traffic
head -14 gpurun_out/$TAG/trace_summary.txt; head -16 gpurun_out/$TAG/pmc_summary.txt | cut -c1-220; cat gpurun_out/$TAG/rpn_gemm_traffic.txt
# the other bench lines (BASELINE cfg 3 / cfg 5 / cont_train.py mode / cfg 4 on one GPU / the data-parallel schedule rehearsed with 1-rank communicators)
for w in predict vgg16 cont; do python bench.py --workload $w > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err; done
python bench.py --per-gpu-batch 2 --no-cpu-baseline > gpurun_out/$TAG/bench_per_gpu_batch2_ordered.json 2>/dev/null
RADNET_DETERMINISTIC=0 python bench.py --per-gpu-batch 2 --no-cpu-baseline > gpurun_out/$TAG/bench_per_gpu_batch2_atomics.json 2>/dev/null
RADNET_BENCH_REHEARSAL=nccl1 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_dp_rehearsal_native_exchange.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_same_box_as_dp_rehearsal.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), round(r["frac"], 3), round(r.get("executed_frac") or 0, 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
