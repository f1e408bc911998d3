#!/bin/bash
# round 5, same-box A/B on the validation build: knobs given as "NAME=1,NAME2=1" lists, one run each (config-2 step, bf16; TIME_BATCH for batch 8)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=$(echo "$spec" | tr ',' ' ')
  [ "$spec" = "base" ] && envs=""
  env $envs timeout -k 10 200 python3 tools/time_dbg.py 10 > $OUT/ab_$i.json 2> $OUT/ab_$i.err || { tail -5 $OUT/ab_$i.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/ab_$i.json").read().strip().splitlines()[-1])
print("$spec", d["ms_per_step"], "loss", round(d["loss"],5), {k:v[1] for k,v in list(d["survey"].items())[:6]})
PY
done
