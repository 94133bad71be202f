"""Shared helpers for the test-suite."""
import functools
import os

import numpy as np

from oracle.network import OracleSSD3D
from tests.golden import detinit

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def oracle_model(input_channels=1, input_size=(64, 64, 64), seed=1234, n_classes=2, feature_ids=(3, 5, 7)):
    """Oracle network carrying the deterministic golden weights (tests/golden/detinit.py)."""
    m = OracleSSD3D(n_classes=n_classes, input_channels=input_channels, input_size=input_size,
                    emulate_reference_init=False, feature_ids=feature_ids)
    m.load_state_dict(detinit.fill_state_dict(m.state_dict(), seed))
    return m
