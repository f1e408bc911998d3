"""Where the fp32 3x3 tile kernels spend their cycles (staging / MFMA / epilogue), summed over workgroups.

Needs a library whose conv3x3_f32.hip was compiled with -DTCVN_PHASE_PROF (adds clock64() markers and the
`tcvn_debug_phases` export; never part of the product build):
    cd dune-transformercvn_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTCVN_PHASE_PROF -c conv3x3_f32.hip \
        -o build/conv3x3_f32.o && make product
    python tools/f32_phase_cycles.py
Runs bench.py --precision fp32 for 3 steps, then prints the counters."""
import ctypes, os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--precision", "fp32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-profile"]
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
import glob
lib = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dune-transformercvn_amd/lib/libtcvn_hip.so"))[0])
buf = (ctypes.c_ulonglong * 32)()
lib.tcvn_debug_phases(buf, 0)
v = list(buf)
names = {0: "wgrad mfma", 1: "wgrad tbl", 2: "wgrad stage", 8: "fwd epilogue", 9: "fwd tbl", 10: "fwd stage", 11: "fwd mfma",
         16: "dgrad epilogue", 17: "dgrad tbl", 18: "dgrad stage", 19: "dgrad mfma"}
for k in sorted(names):
    print(f"{names[k]:16s} {v[k]/1e6:12.1f} Mcycles summed over WGs")
