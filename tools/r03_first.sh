#!/bin/bash
# round-3 first GPU pass: test suite, default bench, batch-8 profile, 2-rank rehearsal of bench.py's own launcher
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
echo "[r03a] pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
echo "pytest rc $rc"
cd /tmp && export TMPDIR=/tmp
echo "[r03a] default bench"
timeout -k 10 400 python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 2
tail -c 400 $OUT/bench_default.json; echo
echo "[r03a] records dump b32 and b8"
timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --dump-records $OUT/records_b32.json > $OUT/bench_b32.json 2> /dev/null || exit 3
timeout -k 10 200 python3 $ROOT/bench.py --batch 8 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --dump-records $OUT/records_b8.json > $OUT/bench_b8.json 2> /dev/null || exit 3
echo "[r03a] batch-8 kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/b8 -o b8 --output-format csv -- python3 $ROOT/bench.py --batch 8 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --no-bwd-overlap > $OUT/b8_under_rocprof.json 2> $OUT/b8.err || exit 4
rm -f $OUT/b8/*/*_kernel_trace.csv $OUT/b8/*_kernel_trace.csv
echo "[r03a] 2-rank rehearsal (gloo, both ranks on cuda:0) through bench.py's own launcher"
TCVN_BENCH_REHEARSAL=1 timeout -k 10 300 python3 $ROOT/bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/rehearsal2.json 2> $OUT/rehearsal2.err; echo "rehearsal rc $?"
tail -c 300 $OUT/rehearsal2.json; echo
ls $OUT
