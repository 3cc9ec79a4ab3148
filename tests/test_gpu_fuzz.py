"""A slice of tools/fuzz.py in the suite: random genomes, guide shapes (3' / 5' / no PAM, IUPAC, auxiliary PAMs), limits, costs
and window sizes -- calitas_search_hits in one pass and in lanes against the oracle, every column.  (tools/fuzz.py itself was run
over 6 000 configurations without a mismatch.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 4])
def test_random_configurations_against_the_oracle(seed, monkeypatch):
    spec = importlib.util.spec_from_file_location("calitas_fuzz", os.path.join(ROOT, "tools", "fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setenv("CALITAS_CHUNKS", "1")          # run() sets it per call; restored afterwards
    assert fuzz.run(60, seed) == 0
