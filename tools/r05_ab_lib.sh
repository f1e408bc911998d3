#!/bin/bash
# round 5, same-box A/B between two validation builds: bash tools/r05_ab_lib.sh <tag> libA.so libB.so [libA.so ...]   (TIME_BATCH / TIME_PRECISION honoured)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
i=0
for lib in "$@"; do
  i=$((i+1))
  TIME_LIB=$lib timeout -k 10 200 python3 tools/time_dbg.py 10 > $OUT/ab_$i.json 2> $OUT/ab_$i.err || { tail -5 $OUT/ab_$i.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/ab_$i.json").read().strip().splitlines()[-1])
print("$lib", d["ms_per_step"], "loss", round(d["loss"],5))
PY
done
