#!/bin/bash
# What issues the ~300 small rocclr copyBuffer launches per step?  One short bench run under rocprofv3 with the kernel, HIP-runtime and
# memory-copy traces; prints the copies' direction/size histogram and the kernels that surround the copy kernels.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $OUT/tr -o tr --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32 --no-sdxl --no-batch8 --no-profile > $OUT/bench.json 2> $OUT/bench.err
ls $OUT/tr
python3 - <<PY
import csv, glob, collections
base = "$OUT/tr/"
f = glob.glob(base + "**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("memory copies:", len(rows), rows[0].keys() if rows else "")
    h = collections.Counter((r.get("Direction"), r.get("Bytes") if "Bytes" in r else r.get("Size")) for r in rows)
    for k, v in h.most_common(25): print("  ", k, v)
f = glob.glob(base + "**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("kernels:", len(names), "copyBuffer launches:", len(idx))
# neighbours of the copy kernels in the second half of the run (the timed step)
ctx = collections.Counter()
for i in idx[len(idx)//2:]:
    prev = names[i-1][:60] if i else ""
    nxt = names[i+1][:60] if i + 1 < len(names) else ""
    ctx[(prev, nxt)] += 1
for k, v in ctx.most_common(20): print(v, k)
f = glob.glob(base + "**/*hip_api_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    h = collections.Counter(r["Function"] for r in rows)
    for k, v in h.most_common(25): print("  api", k, v)
PY
rm -rf $OUT/tr
