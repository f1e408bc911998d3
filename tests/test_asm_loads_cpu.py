"""Build-time guard of the one kernel that issues its loads by inline assembly (csrc/bwd1x1_wide.hip, a validated variant that the
product library never selects): tools/check_asm_loads.sh recompiles it to ISA and replays the main loop against the vmcnt queue -- no
instruction may touch the destination of a load that is still in flight (the compiler believes an asm-loaded register is ready and has
been seen spilling it right behind the load), no compiler-made vmcnt(0) may drain the prefetch pipeline, no scratch access in the loop.
The Makefile runs the same check before it links either library (build/bwd1x1_wide.chk)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_hand_issued_loads_are_not_touched_before_their_wait():
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_asm_loads.sh")], capture_output=True, text=True, timeout=600)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("NS=")]
    assert len(lines) == 3 and all("hazards 0, vmcnt(0) drains in the loop 0, scratch operations in the loop 0" in ln for ln in lines), lines
