#!/bin/bash
# A/B of one environment switch through bench.py, alternating, same box.  usage: tools/ab_env.sh VAR "a b a b" [bench.py args...]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-steps 0 "$@" 2>/dev/null | tail -1) || exit 1
  echo "$VAR=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "images/s", d["ms_per_step"])')"
done
