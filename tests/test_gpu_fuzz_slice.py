"""A bounded slice of tools/fuzz_parity.py inside the `-m gpu` suite: fixed seeds, 25 cases (random transform sizes, six
sample formats, batch splits, client slices at the edges / empty / widest, mode switches, paused clients, waterfall
windows, the post chain at several audio rates), every frame against the oracle exactly as the tool compares it."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("psdr_fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed,cases", [(1, 9), (20260929, 8), (77, 8)])
def test_fuzz_slice_against_the_oracle(seed, cases):
    fz = _fuzz()
    rng = np.random.default_rng(seed)
    for c in range(cases):
        fz.one_case(rng, c)  # raises AssertionError with the case's description on a mismatch
