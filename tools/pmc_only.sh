set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_only
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp32 --no-sdxl --no-batch8"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o f --output-format csv -- python3 $ROOT/bench.py $FLAGS > /dev/null 2> $OUT/pmc_f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_w -o w --output-format csv -- python3 $ROOT/bench.py $FLAGS > /dev/null 2> $OUT/pmc_w.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_f/f_counter_collection.csv $OUT/pmc_w/w_counter_collection.csv $OUT/pmc_traffic.json 2 > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_f $OUT/pmc_w
cat $OUT/pmc_traffic.log
for i in 1 2 3; do python3 $ROOT/bench.py --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
