# usage (GPU box): bash tools/rpn_gemm_traffic.sh <outdir>   -- FETCH_SIZE / WRITE_SIZE per launch of rpn_conv1's batched GEMM for several launch shapes
set -e
OUT=${1:-gpurun_out/rpn_traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "64 128 -1" "64 128 1" "32 64 -1" "32 64 1" "64 64 -1" "32 64 -3" "64 64 -3"; do
  tag=$(echo $cfg | tr ' -' '_m')
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $pmc -d $OUT/$tag.$pmc -o p --output-format csv -- python3 tools/rpn_gemm_traffic.py $cfg > $OUT/$tag.$pmc.log 2>&1
  done
  python3 - "$OUT" "$tag" "$cfg" <<'PY'
import csv, glob, sys
out, tag, cfg = sys.argv[1:4]
res = {}
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s.%s/**/*counter_collection.csv" % (out, tag, pmc), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "conv_igemm" in r["Kernel_Name"] and r["Counter_Name"] == pmc]
    res[pmc] = sum(vals) / max(len(vals), 1) / 1024.0
line = open("%s/%s.FETCH_SIZE.log" % (out, tag)).read().strip().splitlines()[-1]
print("%-12s FETCH_SIZE %7.1f MB (x2 = %7.1f MB)  WRITE_SIZE %6.1f MB  -> corrected total %7.1f MB | %s" % (cfg, res["FETCH_SIZE"], 2 * res["FETCH_SIZE"], res["WRITE_SIZE"], 2 * res["FETCH_SIZE"] + res["WRITE_SIZE"], line))
PY
  rm -rf $OUT/$tag.FETCH_SIZE $OUT/$tag.WRITE_SIZE
done
