"""Synthetic particle sets of the BASELINE configs: uniform
positions in [0,L)^3, unit-variance Gaussian velocities, unit masses, log-normal (or unit)
densities; then the reference's preprocessing -- shift to the origin and remove the
mass-weighted bulk velocity (scripts/parallel_optimized.py:280-291)."""
from __future__ import annotations

import numpy as np


def particles(seed, Np, L=1.0, lognormal_density=True, preprocess=True, chunk=10_000_000):
    """(pos f32 [Np,3], vel f32 [Np,3], mass f32 [Np], density f32 [Np]) on the host."""
    pos = np.empty((Np, 3), dtype=np.float32)
    vel = np.empty((Np, 3), dtype=np.float32)
    dens = np.ones(Np, dtype=np.float32)
    mass = np.ones(Np, dtype=np.float32)
    for c, s in enumerate(range(0, Np, chunk)):
        rng = np.random.default_rng([seed, c])
        n = min(chunk, Np - s)
        pos[s:s + n] = rng.random((n, 3), dtype=np.float32) * np.float32(L)
        vel[s:s + n] = rng.standard_normal((n, 3), dtype=np.float32)
        if lognormal_density:
            dens[s:s + n] = np.exp(0.5 * rng.standard_normal(n, dtype=np.float32))
    if preprocess:
        for a in range(3):
            pos[:, a] -= np.min(pos[:, a])
        M = np.sum(mass, dtype=np.float64)
        for a in range(3):
            vel[:, a] -= np.float32(np.sum(mass * vel[:, a], dtype=np.float64) / M)
    return pos, vel, mass, dens


CONFIGS = {
    # name: (N, Np, seed offset)   -- BASELINE.json configs[0..4]
    "C1": (128, 100_000, 1),
    "C2": (512, 10_000_000, 2),
    "C3": (1024, 50_000_000, 3),
    "C4": (2048, 100_000_000, 4),
    "C5": (4096, 1_000_000_000, 5),
}
BASE_SEED = 20240415
