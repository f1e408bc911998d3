#!/bin/bash
# Round 4: matrix-pipe and SQ wait/active counters of the round-3 kernels (VERDICT r03 "missing #5"), one rocprofv3 --pmc pass each
# (program directly after `--`), bf16 config 2, weight-gradient side stream off so that every kernel runs alone.
#   bash tools/r04_pmc.sh [tag]     -> gpurun_out/<tag>/pmc_mfma.txt, pmc_sq.txt
set -u
TAG=${1:-r04pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp32 --no-sdxl --no-batch8 --no-bwd-overlap --no-optimizer-leg"
CMD="python3 bench.py $FLAGS"
echo "[pmc] mfma pass"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -d $OUT/m -o m --output-format csv -- python3 $ROOT/bench.py $FLAGS > /dev/null 2> $OUT/m.err || { tail -5 $OUT/m.err; exit 1; }
python3 $ROOT/tools/pmc_mfma.py $OUT/m/m_counter_collection.csv $OUT/m/m_kernel_trace.csv "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -- $CMD" 2500 > $OUT/pmc_mfma.txt
echo "[pmc] sq pass"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/s -o s --output-format csv -- python3 $ROOT/bench.py $FLAGS > /dev/null 2> $OUT/s.err || { tail -5 $OUT/s.err; exit 2; }
python3 $ROOT/tools/pmc_sq.py $OUT/s/s_counter_collection.csv $OUT/s/s_kernel_trace.csv "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace -- $CMD" > $OUT/pmc_sq.txt
rm -rf $OUT/m $OUT/s
cat $OUT/pmc_mfma.txt $OUT/pmc_sq.txt
