"""Exact-NN indices against the oracle over particle densities from far sparser to far denser than the lattice, uniform and
clumped, float32 and float64 positions (GPU box): python tools/stress_nn_density.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np
from vpower import interp
from oracle import vps_oracle as orc
bad = 0
for N in (24, 32):
    for per_cell in (0.003, 0.05, 0.4, 2.0, 6.0, 11.0, 14.0, 40.0):
        for clump in (False, True):
            for dtype in (np.float32, np.float64):
                Np = max(3, int(per_cell * N ** 3))
                rng = np.random.default_rng(int(per_cell * 1000) + N + clump)
                pos = rng.random((Np, 3)).astype(dtype)
                if clump:
                    pos[: Np // 4] = (0.5 + 0.02 * rng.standard_normal((Np // 4, 3))).astype(dtype)
                ax = orc.lattice_axes_library(1.0, N)
                t0 = time.time()
                idx = interp.nn_index(pos, (ax, ax, ax))
                t1 = time.time()
                ref = orc.exact_nn_lattice(pos, ax, ax, ax)
                ok = np.array_equal(idx.ravel(), ref)
                bad += not ok
                print("N=%d per_cell=%-6g clump=%d %s Np=%-8d %s  (%.2f s)" % (N, per_cell, clump, dtype.__name__, Np, "ok" if ok else "MISMATCH", t1 - t0), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
