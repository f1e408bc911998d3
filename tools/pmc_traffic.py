"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each in its own run with --kernel-trace):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv out.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads (LDS-DMA alike), so it is doubled; WRITE_SIZE is exact
for 16-B-per-lane stores.  Keys of the output are bench.py's kernel labels."""
import collections
import csv
import json
import sys

LABELS = {            # bench.py label -> substring of the demangled kernel name
    "k_bwd1x1_fused_bf16": "k_bwd1x1_fused_bf16",
    "k_fwd1x1_fused_bf16": "k_fwd1x1_",            # both kernels behind the label: k_fwd1x1_fused_bf16<NKC> and k_fwd1x1_wide_bf16 (K > 256)
    "k_gemm_nt_bf16<dgrad1x1>": "k_gemm_nt_bf16<1, ",
    "k_gemm_nt_bf16<dgradtrans>": "k_gemm_nt_bf16<2, ",
    "k_conv3x3_dgrad_bf16": "k_conv3x3_dgrad",
    "k_conv3x3_fwd_bf16": "k_conv3x3_fwd_",
    "k_conv3x3_wgrad_bf16": "k_conv3x3_wgrad_bf16",
    "k_stem_fwd_bf16": "k_stem_fwd2_bf16",
    "k_act_bf16": "k_act_bf16",
    "k_eff_mat": "k_eff_mat",
    "k_gemm_tn_bf16<conv1>": "k_gemm_tn_bf16<0>",
    "k_gemm_nt_bf16<fwd1x1>": "k_gemm_nt_bf16<0, ",
    "k_encoder_fwd": "k_encoder_fwd",
    # --sdxl embedder: bench labels name the operator, rocprof the kernel that served it (k_sconv3_c64 runs the forward AND the data
    # gradient of the 64->64 layers: the average over both is attributed to either label)
    "k_sconv_fwd<bf16,3x3/1,64->64>": "k_sconv3_c64(",
    "k_sconv_dgrad<bf16,3x3/1,64->64>": "k_sconv3_c64(",
    "k_sconv_wgrad<bf16,3x3/1,64->64>": "k_sconv3_c64_wgrad<1, false>",
    "k_sconv3_g": "k_sconv3_g(",
    "k_encoder_bwd": "k_encoder_bwd",
}


def per_kernel(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {"_note": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024, averaged over every launch of the run"}
    for label, sub in LABELS.items():
        f = [v for k, vs in fetch.items() if sub in k for v in vs]
        w = [v for k, vs in write.items() if sub in k for v in vs]
        if not f or not w:
            continue
        out[label] = {"launches": len(f), "fetch_bytes_per_launch": 2 * 1024 * sum(f) / len(f),
                      "write_bytes_per_launch": 1024 * sum(w) / len(w),
                      "traffic_bytes_per_launch": 2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w)}
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    if steps:          # whole run / steps: every kernel of the two passes (weights packing, token path, torch helpers included)
        tf = 2 * 1024 * sum(v for vs in fetch.values() for v in vs) / steps
        tw = 1024 * sum(v for vs in write.values() for v in vs) / steps
        out["_whole_step"] = {"steps_in_run": steps, "fetch_bytes": tf, "write_bytes": tw, "traffic_bytes": tf + tw}
        print(f"whole step: {tf / 1e9:.1f} GB read + {tw / 1e9:.1f} GB written = {(tf + tw) / 1e9:.1f} GB")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if not k.startswith("_"):
            print(f"{k:32s} {v['launches']:5d} launches  {v['traffic_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
