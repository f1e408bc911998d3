#!/bin/bash
# HBM traffic of the --sdxl bench step from the PMC counters (two separate passes, as MI355X_MICROARCH.md prescribes).
# usage (GPU box): bash tools/sdxl_pmc.sh r03
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/spmc_f -o f --output-format csv -- python3 $ROOT/bench.py --sdxl --steps 1 --warmup 1 --no-profile > /dev/null 2> $OUT/spmc_f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/spmc_w -o w --output-format csv -- python3 $ROOT/bench.py --sdxl --steps 1 --warmup 1 --no-profile > /dev/null 2> $OUT/spmc_w.err
python3 $ROOT/tools/pmc_traffic.py $OUT/spmc_f/f_counter_collection.csv $OUT/spmc_w/w_counter_collection.csv $OUT/sdxl_pmc_traffic.json 2 > $OUT/sdxl_pmc_traffic.log 2>&1
rm -rf $OUT/spmc_f $OUT/spmc_w
tail -30 $OUT/sdxl_pmc_traffic.log
