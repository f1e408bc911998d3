#!/bin/bash
# Collects the round's judged artifacts on a 1-GPU MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Everything lands in gpurun_out/<tag>final/; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[collect] default bench"; timeout -k 10 600 python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
tail -c 600 $OUT/bench_default.json; echo
echo "[collect] bf16 kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bf16 -o bf16 --output-format csv -- python3 $ROOT/bench.py --precision bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 > $OUT/bf16_bench_under_rocprof.json 2> $OUT/bf16.err || exit 2
echo "[collect] bf16 kernel trace, weight-gradient side stream off (per-kernel durations without sharing)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bf16_serial -o bf16_serial --output-format csv -- python3 $ROOT/bench.py --precision bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --no-bwd-overlap > $OUT/bf16_serial_bench_under_rocprof.json 2> $OUT/bf16_serial.err || exit 2
echo "[collect] fp32 kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/fp32 -o fp32 --output-format csv -- python3 $ROOT/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/fp32_bench_under_rocprof.json 2> $OUT/fp32.err || exit 3
echo "[collect] sdxl kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/sdxl -o sdxl --output-format csv -- python3 $ROOT/bench.py --sdxl --steps 2 --warmup 1 > $OUT/sdxl_bench_under_rocprof.json 2> $OUT/sdxl.err || exit 4
echo "[collect] PMC passes (FETCH_SIZE, WRITE_SIZE separately)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o f --output-format csv -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp32 --no-sdxl --no-batch8 --no-optimizer-leg > /dev/null 2> $OUT/pmc_f.err || exit 5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_w -o w --output-format csv -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp32 --no-sdxl --no-batch8 --no-optimizer-leg > /dev/null 2> $OUT/pmc_w.err || exit 6
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_f/f_counter_collection.csv $OUT/pmc_w/w_counter_collection.csv $OUT/pmc_traffic.json 2 > $OUT/pmc_traffic.log 2>&1
echo "[collect] other configurations (un-profiled bench lines)"
timeout -k 10 200 python3 $ROOT/bench.py --sdxl --steps 3 --warmup 1 > $OUT/sdxl_bench.json 2> /dev/null
timeout -k 10 200 python3 $ROOT/bench.py --ragged-inference --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 > $OUT/ragged_inference_bench.json 2> /dev/null
timeout -k 10 200 python3 $ROOT/bench.py --batch 8 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 > $OUT/batch8_bench.json 2> /dev/null
timeout -k 10 200 python3 $ROOT/bench.py --batch 64 --no-cpu-baseline --no-fp32 --steps 3 > $OUT/batch64_bench.json 2> /dev/null
echo "[collect] sdxl PMC passes"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/spmc_f -o f --output-format csv -- python3 $ROOT/bench.py --sdxl --steps 1 --warmup 1 --no-profile --no-optimizer-leg > /dev/null 2> $OUT/spmc_f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/spmc_w -o w --output-format csv -- python3 $ROOT/bench.py --sdxl --steps 1 --warmup 1 --no-profile --no-optimizer-leg > /dev/null 2> $OUT/spmc_w.err
python3 $ROOT/tools/pmc_traffic.py $OUT/spmc_f/f_counter_collection.csv $OUT/spmc_w/w_counter_collection.csv $OUT/sdxl_pmc_traffic.json 2 > $OUT/sdxl_pmc_traffic.log 2>&1
rm -rf $OUT/spmc_f $OUT/spmc_w
rm -rf $OUT/*/*_kernel_trace.csv $OUT/pmc_f $OUT/pmc_w       # traces are large; the stats CSVs and the reduced JSON stay
ls -la $OUT
