#!/usr/bin/env python3
"""All-host-cores form of the CPU oracle's NGP step -- TEST / BASELINE INFRASTRUCTURE ONLY (like the rest of oracle/:
nothing under large-velocity-power-spectrum_amd/ imports it; bench.py runs it as a child process for the
`cpu_baseline_allcores` leg, tests/ check it against the one-core oracle).

What it is: the SAME arithmetic as oracle/vps_oracle.py (deposit_to_grid -> v, m -> FFT power -> pair -> histogram -> table;
reference vpower/interp.py:996-1015, 272-273, 1372-1421, 1440-1482) with every stage spread over the host's cores the way the
reference spreads its work over MPI ranks (scripts/parallel_optimized.py:201-491 under `mpiexec -n R`: every rank owns a part
of the volume): one worker PROCESS per x-slab for gridding, field algebra, |F|^2 and the two histograms (fork: inputs shared
copy-on-write, grids in shared memory, only (nbins,) partial histograms travel), and pocketfft's own threads for the 3-D
transforms (scipy.fft `workers`).  Partial histograms are added in slab order, so shell COUNTS equal the one-core oracle's
exactly and shell SUMS to float64 rounding (different summation order).

    python oracle/allcores.py --grid 512 --particles 1562500 --workers 64 --quantities velocity,momentum,energy
prints one JSON line {"seconds": total, "stage_seconds": {...}, "workers": W, "nsample_sum": ..., "psum_head": [...]}.
Never touches a GPU and imports no torch."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import vps_oracle as orc  # noqa: E402

_G = {}   # what the forked workers read (set in the parent BEFORE the pool of a stage is created)


def _shared(shape):
    n = int(np.prod(shape))
    raw = mp.RawArray("d", n)
    return np.frombuffer(raw, dtype=np.float64, count=n).reshape(shape)


def _slabs(N, workers):
    edges = np.linspace(0, N, min(workers, N) + 1).astype(int)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _grid_slab(sl):
    """NGP deposit of [rho v, rho] into x-rows [i0, i1) (interp.py:996-1015 on the slab's particles), then v = rho v / rho with
    empty cells set to 0 (interp.py:329-331) and m = rho Lcell^3 (interp.py:272-273)."""
    i0, i1 = sl
    N, ix, flat_yz, vec = _G["N"], _G["ix"], _G["flat_yz"], _G["vec"]
    sel = np.nonzero((ix >= i0) & (ix < i1))[0]
    flat = (ix[sel] - i0) * (N * N) + flat_yz[sel]
    n = (i1 - i0) * N * N
    rho = np.bincount(flat, weights=vec[sel, 3], minlength=n)
    m = _G["fields"][3]
    m[i0:i1] = (rho * _G["Lcell"] ** 3).reshape(i1 - i0, N, N)
    with np.errstate(invalid="ignore", divide="ignore"):
        for c in range(3):
            q = np.bincount(flat, weights=vec[sel, c], minlength=n) / rho
            _G["fields"][c][i0:i1] = np.where(np.isnan(q), 0.0, q).reshape(i1 - i0, N, N)
    return len(sel)


def _hist_slab(sl):
    """P = 0.5 sum_c |a F_c|^2 on x-rows [i0, i1) of the transforms (interp.py:1372-1387), k = sqrt(kx^2 + ky^2 + kz^2)
    (interp.py:1449-1456), the two histograms of interp.py:1474-1477 restricted to the slab."""
    i0, i1 = sl
    ks, edges, a = _G["ks"], _G["edges"], _G["a"]
    P = None
    for F in _G["F"]:
        t = np.abs(F[i0:i1] * a) ** 2
        P = t if P is None else P + t
    P *= 0.5
    kx = ks[i0:i1, None, None]
    ky = ks[None, :, None]
    kz = ks[None, None, :]
    k = np.sqrt(kx * kx + ky * ky + kz * kz).ravel()
    psum, _ = np.histogram(k, bins=edges, weights=P.ravel())
    nsam, _ = np.histogram(k, bins=edges)
    return psum, nsam


def ngp_tables(quantities, flavour, N, L, pos, vel, dens, workers):
    """{quantity: (nbins, 4) table}, {stage: seconds} -- oracle.vps_oracle on `workers` cores (module docstring)."""
    t = {}
    t0 = time.perf_counter()
    ctx = mp.get_context("fork")
    sl = _slabs(N, workers)
    Lcell = L / N
    idx = orc.cell_index(pos, N, L)
    _G.clear()
    _G.update(N=N, Lcell=Lcell, ix=idx[:, 0], flat_yz=idx[:, 1] * N + idx[:, 2],
              vec=orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64)),
              fields=[_shared((N, N, N)) for _ in range(4)])
    with ctx.Pool(len(sl)) as pool:
        assert sum(pool.map(_grid_slab, sl)) == len(pos)
    vx, vy, vz, m = _G["fields"]
    t["gridding"] = time.perf_counter() - t0
    kmin, kmax, kres = orc.default_k_range(L, N)
    centers, edges = (orc.edges_library if flavour == "library" else orc.edges_script)(kmin, kmax, kres)
    out = {}
    orc.set_fft_workers(workers)
    try:
        for q in quantities:
            t1 = time.perf_counter()
            if q == "velocity":
                comps = (vx, vy, vz)
            elif q == "momentum":
                comps = orc.momentum_fields(vx, vy, vz, m)
            elif q == "energy":
                comps = (orc.kinetic_energy_field(vx, vy, vz, m),)
            else:
                raise Exception("Unrecognized physical quantity name.")
            _G.update(F=[orc._fftn(f) for f in comps], ks=orc.k_axis(L, N), edges=edges, a=orc.power_const(L, N))
            with ctx.Pool(len(sl)) as pool:
                parts = pool.map(_hist_slab, sl)
            _G["F"] = None
            psum = np.sum([p[0] for p in parts], axis=0)
            nsam = np.sum([p[1] for p in parts], axis=0)
            with np.errstate(invalid="ignore", divide="ignore"):
                P = psum / nsam
            if flavour == "library":
                P[nsam == 0] = 0
            tab = np.column_stack((centers, P, psum, nsam))
            tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2
            out[q] = tab
            t["spctrm_" + q] = time.perf_counter() - t1
    finally:
        orc.set_fft_workers(1)
        _G.clear()
    t["total"] = time.perf_counter() - t0
    return out, t


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, required=True)
    ap.add_argument("--particles", type=int, required=True)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--quantities", default="velocity")
    ap.add_argument("--flavour", default="library")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lognormal", type=int, default=1)
    args = ap.parse_args(argv)
    sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd", "vpower"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("vps_synth", os.path.join(ROOT, "large-velocity-power-spectrum_amd", "vpower", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)       # (the generator only: numpy, no torch, no library)
    pos, vel, _, dens = synth.particles(args.seed, args.particles, 1.0, bool(args.lognormal))
    tabs, t = ngp_tables(tuple(args.quantities.split(",")), args.flavour, args.grid, 1.0, pos, vel, dens, args.workers)
    first = tabs[args.quantities.split(",")[0]]
    print(json.dumps({"seconds": t["total"], "stage_seconds": t, "workers": args.workers,
                      "nsample_sum": float(first[:, 3].sum()), "psum_head": [float(x) for x in first[:4, 2]]}), flush=True)


if __name__ == "__main__":
    main()
