"""The oracle against the LIVE reference (only where /root/reference exists, i.e. in the build container -- never on the GPU box):
a fresh seeded case that is not among the committed goldens, eval logits and one training step (loss + a few gradients).
The reference is imported read-only with the stub modules of oracle/make_golden.py, in a separate
interpreter, because its package name `transformercvn` is the drop-in package's name too."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "transformercvn")), reason="reference tree not present")


def test_oracle_matches_live_reference_on_a_fresh_case():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "live_reference_check.py")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, script], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    print(res)
    assert res["event_logits"] <= 2e-5 and res["prong_logits"] <= 2e-5 and res["loss"] <= 2e-5
    for k, v in res.items():
        if k.startswith("grad:"):
            assert v < 5e-3, (k, v)
