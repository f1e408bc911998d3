"""Child processes on the -DTCVN_DEBUG_KNOBS build (libtcvn_hip_dbg.so): the variant-vs-variant tests compare the product kernels
with the generic / fallback kernels that build selects through its validation switches.  The debug library is chosen explicitly
in the child (transformercvn.hip._libselect.use) -- the product loader reads no environment variable -- and the result travels
through a file private to this call."""
import os
import subprocess
import sys
import tempfile

import torch


def run_on_debug_build(body: str, knobs: dict):
    """Run `body` (python source; must leave its result in a variable named `result`) in a child process bound to the debug
    build with the validation switches `knobs` set; returns the unpickled result."""
    fd, path = tempfile.mkstemp(prefix="tcvn_variant_", suffix=".pt")
    os.close(fd)
    os.remove(path)                       # the child creates it: a stale file can never be mistaken for this call's result
    code = (f"import sys, torch\nsys.path[:0] = {sys.path!r}\n"
            "from transformercvn.hip import _libselect\n_libselect.use('libtcvn_hip_dbg.so')\n"
            + body + f"\ntorch.save(result, {path!r})\n")
    try:
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, **knobs))
        return torch.load(path)
    finally:
        if os.path.exists(path):
            os.remove(path)
