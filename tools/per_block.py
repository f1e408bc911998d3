"""Per (kernel, grid) time table from gpurun_out/perblock/trace_small.csv (tools/per_block_trace.sh): which dense block's launches
cost what.  Grid sizes identify the block (every block has its own pixel count) and the embedder (prong: 256 maps, event: 32)."""
import csv, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/perblock/trace_small.csv"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
acc = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:48], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in acc.values()) / steps / 1e3
print(f"kernel time per step {tot:.2f} ms, launches per step {sum(a[0] for a in acc.values()) / steps:.0f}")
byk = collections.defaultdict(list)
for (n, gx, gy, gz), (c, t) in acc.items():
    byk[n].append((gx, gy, gz, c / steps, t / c, t / steps / 1e3))
for n, lst in sorted(byk.items(), key=lambda kv: -sum(x[5] for x in kv[1])):
    s = sum(x[5] for x in lst)
    if s < 0.03: continue
    print(f"{n:50s} {s:7.3f} ms/step")
    for gx, gy, gz, c, avg, ms in sorted(lst, key=lambda x: -x[5]):
        if ms >= 0.02: print(f"      grid {gx:6d}x{gy}x{gz}  {c:6.1f} launches/step  {avg:8.1f} us  {ms:7.3f} ms")
