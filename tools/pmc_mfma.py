"""MFMA work per kernel from one rocprofv3 PMC pass with --kernel-trace:

    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace --output-format csv \\
              -d <dir> -o m -- python3 bench.py ...
    python tools/pmc_mfma.py <dir>/m_counter_collection.csv <dir>/m_kernel_trace.csv "<command line>" <peak TFLOP/s> > out.txt

MOPS counters are in units of 512 FLOP (MI355X_MICROARCH.md); durations come from the same trace (profiler attached)."""
import collections, csv, re, sys


def key(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    cc, kt, cmd, peak = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    ops = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        ops[key(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(kt)):
        k = key(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[k] += 1
    print(cmd)
    print(f"MFMA work per kernel from the SQ counters (MOPS x 512 FLOP), whole run; durations from the same trace; peak {peak:.0f} TFLOP/s dense\n")
    rows = []
    for k, v in ops.items():
        tf = (v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) + v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)) * 512 / 1e12
        if tf > 0:
            rows.append((tf, k))
    tot = 0.0
    for tf, k in sorted(rows, reverse=True)[:16]:
        tot += tf
        print(f"{k[:42]:42s} launches {cnt[k]:5d}  MFMA {tf:7.3f} TFLOP  time {dur[k]:8.2f} ms  -> {tf / dur[k] * 1e3:7.1f} TFLOP/s ({100 * tf / dur[k] * 1e3 / peak:4.1f}% of peak)")
    print(f"total MFMA work of the listed kernels: {tot:.2f} TFLOP")


if __name__ == "__main__":
    main()
