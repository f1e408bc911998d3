#!/bin/bash
# round 5, same-box A/B: the kept round-4 validation build against the current one (config-2 step, bf16), then a kernel-stats pass
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05a}
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  TIME_LIB=libtcvn_hip_dbg_r04.so timeout -k 10 200 python3 tools/time_dbg.py 10 > $OUT/ab_old_$rep.json 2> $OUT/ab_old_$rep.err || exit 1
  timeout -k 10 200 python3 tools/time_dbg.py 10 > $OUT/ab_new_$rep.json 2> $OUT/ab_new_$rep.err || exit 1
done
python3 - <<PY
import json
for n in ("ab_old_1","ab_new_1","ab_old_2","ab_new_2"):
    d=json.loads(open("$OUT/"+n+".json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], {k:v for k,v in list(d["survey"].items())[:7]})
PY
