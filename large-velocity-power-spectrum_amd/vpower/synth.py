"""Synthetic particle sets of the BASELINE configs: uniform
positions in [0,L)^3, unit-variance Gaussian velocities, unit masses, log-normal (or unit)
densities; then the reference's preprocessing -- shift to the origin and remove the
mass-weighted bulk velocity (scripts/parallel_optimized.py:280-291).

Two generators with the same distributions: `particles` (numpy, host; every size the CPU
oracle can follow) and `particles_device` (torch Philox on the GPU in 10^7-particle chunks
keyed by (seed, chunk), preprocessing by the library's own device reductions) for the 10^8
and 10^9 particle configs, where a host copy would cost minutes and tens of GB."""
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


def particles_device(kernels, seed, Np, L=1.0, lognormal_density=True, preprocess=True, chunk=10_000_000):
    """(pos, vel, density) float32 tensors on kernels.device; unit masses are implied.
    Every rank of a job calls this with the same seed and gets the same replicated set."""
    import torch
    dev = kernels.device
    pos = torch.empty((Np, 3), dtype=torch.float32, device=dev)
    vel = torch.empty((Np, 3), dtype=torch.float32, device=dev)
    dens = torch.ones((Np,), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    for c, s in enumerate(range(0, Np, chunk)):
        n = min(chunk, Np - s)
        gen.manual_seed(int(seed) * 1_000_003 + c)
        pos[s:s + n].uniform_(0.0, 1.0, generator=gen)
        pos[s:s + n].mul_(float(L)).clamp_(max=float(np.nextafter(np.float32(L), np.float32(0))))
        vel[s:s + n].normal_(generator=gen)
        if lognormal_density:
            dens[s:s + n].normal_(generator=gen).mul_(0.5).exp_()
    if preprocess:
        mass = torch.ones((Np,), dtype=torch.float32, device=dev)
        kernels.preprocess(pos, vel, mass, True, True)
        del mass
    return pos, vel, dens


CONFIGS = {
    # name: (N, Np, seed offset)   -- BASELINE.json configs[0..4]
    "C1": (128, 100_000, 1),
    "C2": (512, 10_000_000, 2),
    "C3": (1024, 50_000_000, 3),
    "C4": (2048, 100_000_000, 4),
    "C5": (4096, 1_000_000_000, 5),
}
# what each config computes: (route, quantities, binning flavour); routes: "ngp" = nearest-grid-point
# deposit of [rho v, rho] (interp.py:996-1015 + :272-273), "nn" = exact-NN resampling on the library
# lattice (interp.py:246-277), "script" = exact NN on the script lattice, raw velocities
# (scripts/parallel_optimized.py:337-358)
WORKLOADS = {
    "C1": ("script", ("velocity",), "script"),
    "C2": ("ngp", ("velocity",), "library"),
    "C3": ("nn", ("momentum",), "library"),
    "C4": ("ngp", ("velocity", "momentum", "energy"), "library"),
    "C5": ("ngp", ("energy",), "library"),
}
BASE_SEED = 20240415
