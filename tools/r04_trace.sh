#!/bin/bash
# kernel trace of the DEFAULT bf16 configuration (streams overlapped) for tools/gap_analysis.py
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r04trace}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/t -o t --output-format csv -- python3 $ROOT/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --no-profile > $OUT/bench.json 2> $OUT/err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/t/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
keep=["Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size_X","Grid_Size_Y","Grid_Size_Z","Workgroup_Size_X","Stream_Id","Queue_Id"]
keep=[k for k in keep if k in rows[0]]
w=csv.writer(open("$OUT/trace_small.csv","w"))
w.writerow(keep)
for r in rows: w.writerow([r[k].replace("tcvn::(anonymous namespace)::","") for k in keep])
PY
rm -rf $OUT/t
python3 $ROOT/tools/gap_analysis.py $OUT/trace_small.csv
python3 $ROOT/tools/per_block.py $OUT/trace_small.csv 3 > $OUT/per_block.txt 2>&1
gzip -f $OUT/trace_small.csv
