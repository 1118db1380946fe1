"""A short run of the randomised parity campaign (tools/fuzz_parity.py): random shapes, build modes, flags, ef and
batch sizes, GPU search against the CPU restatement on the graph that was built."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_random_cases_agree_with_the_restatement():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "25", "--seed", "11"],
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '"failures": 0' in r.stdout
