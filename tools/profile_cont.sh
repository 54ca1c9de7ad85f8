# One-lane kernel trace of the cont_train.py-mode step: bash tools/profile_cont.sh -> gpurun_out/cont/{trace_summary,kernel_sequence}.txt
set -e
mkdir -p gpurun_out/cont
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/cont/tune.txt
rm -f $T
python bench.py --workload cont --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 10 --warmup 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/cont/tr -o c --output-format csv -- python3 bench.py --workload cont --tune-cache $T --no-cpu-baseline --roofline-steps 0 --steps 12 --warmup 8 > gpurun_out/cont/trace.log 2>&1
ADAMS_PER_STEP=4 python tools/trace_summary.py gpurun_out/cont/tr/c_kernel_trace.csv 60 > gpurun_out/cont/trace_summary.txt
python tools/trace_sequence.py gpurun_out/cont/tr/c_kernel_trace.csv 4 > gpurun_out/cont/kernel_sequence.txt 2>/dev/null || true
rm -rf gpurun_out/cont/tr
head -50 gpurun_out/cont/trace_summary.txt
