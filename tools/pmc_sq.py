"""Where the waves' cycles go, per kernel, from one rocprofv3 PMC pass with --kernel-trace:

    python tools/pmc_sq.py <dir>/s_counter_collection.csv <dir>/s_kernel_trace.csv "<command line>" > out.txt

All SQ_* counters are summed over the launches of a kernel; fractions are of SQ_WAVE_CYCLES."""
import collections, csv, re, sys


def key(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    cc, kt, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    v = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        v[key(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(kt)):
        k = key(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[k] += 1
    print(cmd)
    print("fractions of SQ_WAVE_CYCLES per kernel (whole run); time from the same trace (profiler attached)\n")
    cols = ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"]
    print(f"{'kernel':44s} {'launches':>8s} {'ms':>8s} " + " ".join(f"{c[3:]:>16s}" for c in cols) + f" {'LDS_BANK_CONFLICT/ACTIVE_LDS':>28s}")
    for k in sorted(dur, key=lambda k: -dur[k])[:24]:
        wc = v[k].get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        lds = v[k].get("SQ_ACTIVE_INST_LDS", 0.0)
        bc = v[k].get("SQ_LDS_BANK_CONFLICT", 0.0)
        print(f"{k[:44]:44s} {cnt[k]:8d} {dur[k]:8.2f} " + " ".join(f"{v[k].get(c, 0.0) / wc:16.3f}" for c in cols)
              + f" {bc / lds if lds else 0.0:28.3f}")


if __name__ == "__main__":
    main()
