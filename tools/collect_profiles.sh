set -e
TAG=${TAG:-r02}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/$TAG/tune.txt
rm -f $T
python bench.py --tune-cache $T > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 1500 gpurun_out/$TAG/bench.json | head -c 400; echo
# pipelined run under the tracer (lane timeline)
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/pipe -o pipe --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 40 --warmup 20 > gpurun_out/$TAG/pipe.log 2>&1
python tools/lane_timeline.py gpurun_out/$TAG/pipe/pipe_kernel_trace.csv 0 > gpurun_out/$TAG/lanes.txt
# one-lane run: per-kernel durations without co-running kernels (what the roofline leg also measures)
export RADNET_SIDE_PREFETCH=0
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/serial -o serial --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 40 --warmup 20 > gpurun_out/$TAG/serial.log 2>&1
python tools/trace_summary.py gpurun_out/$TAG/serial/serial_kernel_trace.csv 44 > gpurun_out/$TAG/trace_summary.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/$TAG/pmc1 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$TAG/pmc2 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$TAG/pmc3 -o p --output-format csv -- python3 bench.py --tune-cache $T --no-cpu-baseline --steps 20 --warmup 8 --roofline-steps 0 > gpurun_out/$TAG/pmc3.log 2>&1
python tools/pmc_summary.py gpurun_out/$TAG/pmc1 gpurun_out/$TAG/pmc2 gpurun_out/$TAG/pmc3 > gpurun_out/$TAG/pmc_summary.txt
# keep the merge small
rm -f gpurun_out/$TAG/pmc*/p_counter_collection.csv gpurun_out/$TAG/pmc*/p_kernel_trace.csv gpurun_out/$TAG/pipe/pipe_kernel_trace.csv gpurun_out/$TAG/serial/serial_kernel_trace.csv
# secondary measurements: per-GPU batch 2 as one program (BASELINE cfg 4 on one GPU), proposal paths, Winograd vs direct
unset RADNET_SIDE_PREFETCH
python bench.py --tune-cache gpurun_out/$TAG/tune_b2.txt --per-gpu-batch 2 --steps 150 --warmup 20 --no-cpu-baseline > gpurun_out/$TAG/bench_batch2.json 2> gpurun_out/$TAG/bench_batch2.err
python tools/proposals_timing.py > gpurun_out/$TAG/proposals_timing.txt 2>&1
python tools/winograd_timing.py > gpurun_out/$TAG/winograd_timing.txt 2>&1
python tools/host_timeline.py 200 > gpurun_out/$TAG/host_timeline.txt 2>&1
ls -la gpurun_out/$TAG gpurun_out/$TAG/serial | head -40
head -5 gpurun_out/$TAG/lanes.txt; head -12 gpurun_out/$TAG/trace_summary.txt; head -12 gpurun_out/$TAG/pmc_summary.txt
