"""Shared test helpers: synthetic inputs (BASELINE.md section 3 recipe) and golden loading."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth(seed, Np, L=1.0, lognormal_density=True):
    """Must stay identical to tests/golden/make_goldens.py:synth."""
    rng = np.random.default_rng(seed)
    pos = rng.random((Np, 3), dtype=np.float32) * np.float32(L)
    vel = rng.standard_normal((Np, 3), dtype=np.float32)
    mass = np.ones(Np, dtype=np.float32)
    if lognormal_density:
        dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    else:
        dens = np.ones(Np, dtype=np.float32)
    return pos, vel, mass, dens
