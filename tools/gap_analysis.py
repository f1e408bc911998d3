"""How much of a step is launch gaps?  From a rocprofv3 kernel trace (tools/per_block_trace.sh format: trace_small.csv) of the
DEFAULT configuration (two embedder streams + weight-gradient side stream), for the last full step in the trace:
span, union of busy intervals, time with >= 2 kernels resident, per-queue busy time and the gap histogram between consecutive
kernels of the same queue.   python tools/gap_analysis.py gpurun_out/perblock/trace_small.csv"""
import csv, sys, collections

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/perblock/trace_small.csv"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")) for r in csv.DictReader(open(path))]
rows.sort()
# a step starts with k_pack (weights -> kernel layout, first launch of DenseNetPlan::forward); two embedders -> two per step
starts = [s for s, e, n, q, st in rows if n.startswith("k_pack") or "k_pack<" in n]
if len(starts) < 4:
    print("not enough steps in the trace"); sys.exit(0)
# steps are separated by long idle gaps (host sync): split at gaps > 300 us
steps, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-50:]) > 300_000:
        steps.append(cur); cur = []
    cur.append(r)
steps.append(cur)
steps = [s for s in steps if len(s) > 500]
print("steps found:", [len(s) for s in steps])
s = steps[-1]
t0, t1 = s[0][0], max(x[1] for x in s)
ev = []
for a, b, n, q, st in s:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
busy = multi = 0; depth = 0; last = t0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
print(f"launches {len(s)}  span {(t1 - t0) / 1e6:.3f} ms  busy (>=1 kernel) {busy / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms  >=2 kernels resident {multi / 1e6:.3f} ms")
print(f"sum of kernel durations {sum(b - a for a, b, *_ in s) / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for a, b, n, q, st in s:
    byq[(q, st)].append((a, b, n))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    lst.sort()
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 50_000]
    dur = sum(b - a for a, b, n in lst)
    print(f"queue/stream {q}: {len(lst)} kernels, busy {dur / 1e6:.3f} ms, span {(lst[-1][1] - lst[0][0]) / 1e6:.3f} ms, "
          f"gaps<50us: n={len(small)} sum {sum(small) / 1e6:.3f} ms median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us; gaps>=50us sum {sum(g for g in pos if g >= 50_000) / 1e6:.3f} ms")
short = [(b - a) for a, b, *_ in s if b - a < 10_000]
print(f"kernels shorter than 10 us: {len(short)} ({sum(short) / 1e6:.3f} ms)")
