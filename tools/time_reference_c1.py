#!/usr/bin/env python3
"""Time the REFERENCE's own main() on BASELINE config 1 (128^3 grid, 1e5 synthetic Gaussian-velocity
particles, velocity P(k), one MPI rank) in the build container -- SURVEY.md 8(d) / BASELINE.md section 3,
step 1: "reference-under-shims, 1 core".

    python tools/time_reference_c1.py [--ntot 128] [--np 100000] [--out BASELINE_c1_reference.json]

Build container only (needs /root/reference, which never travels to the GPU box).  The reference's
third-party dependencies that are not installed here are the `sys.modules` stand-ins of
tests/golden/make_goldens.py (pyfftw -> numpy.fft, mpi4py -> one-rank communicator, h5py -> dict), with
ONE difference: the Annoy stand-in answers from a scipy cKDTree (exact NN, one query per call, as the
reference calls it) instead of the brute-force search the fixtures use -- the brute force would make
the per-cell loop of scripts/parallel_optimized.py:337-358 hours long and say nothing about the
reference.  What is timed is therefore the reference's own Python (cell loop, phase fold,
FFTW_power call sites, pair_power, hist_sample, file output) around stand-in FFT and NN engines; stage
boundaries are the reference's own timestamped prints ("Build index", "Index built", "FFTW:", ...).
The resulting Pk.txt is compared with the oracle's table for the same particles (Nsample equal,
Psum within 2e-5) so that the timed run is known to have computed the right thing.
"""
import argparse
import contextlib
import datetime
import io
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntot", type=int, default=128)
    ap.add_argument("--np", type=int, default=100_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_reference_c1_under_shims.json"))
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", "1")

    import make_goldens as mg
    from scipy.spatial import cKDTree
    from vpower import synth
    from oracle import vps_oracle as orc

    h5_store = {}
    mg.install_shims(h5_store)

    class KDAnnoy:
        """exact-NN stand-in for annoy.AnnoyIndex(3, 'euclidean') with 1 tree (parity-unpinned: SURVEY.md 8c)"""
        def __init__(self, dim, metric):
            self.items = []
        def add_item(self, i, v):
            self.items.append(np.asarray(v, dtype=np.float32))
        def build(self, n_trees, n_jobs=-1):
            self.tree = cKDTree(np.stack(self.items).astype(np.float64))
        def save(self, fn):
            pass
        def load(self, fn):
            raise RuntimeError("no index cache")
        def get_nns_by_vector(self, q, n=1, search_k=-1, include_distances=False):
            return [int(self.tree.query(np.asarray(q, dtype=np.float64), k=1)[1])]
    sys.modules["annoy"].AnnoyIndex = KDAnnoy

    N, Np, L = args.ntot, args.np, 1
    pos, vel, mass, dens = synth.particles(synth.BASE_SEED + 1, Np, float(L), lognormal_density=False, preprocess=False)
    h5_store["PartType0"] = {"Coordinates": pos.copy(), "Masses": mass.copy(), "Velocities": vel.copy()}
    tmp = tempfile.mkdtemp()
    snap = os.path.join(tmp, "snap.hdf5")
    open(snap, "w").close()
    sys.argv = ["parallel_optimized.py", "-i", snap, "-o", tmp, "-N", str(N), "-M", str(N), "-l", str(L), "-f"]
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
        script = mg.load_module("ref_script", os.path.join(mg.REF, "scripts", "parallel_optimized.py"))
        t0 = time.perf_counter()
        rc = script.main()
        wall = time.perf_counter() - t0
    assert rc == 0
    # the reference's own timestamped stage prints
    stamps = []
    for line in sink.getvalue().splitlines():
        m = re.match(r"\[(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d\.\d+)\]\s*(.*)", line)
        if m:
            stamps.append((datetime.datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S.%f"), m.group(2)[:60]))
    stages = [{"from": a[1], "to": b[1], "seconds": (b[0] - a[0]).total_seconds()} for a, b in zip(stamps, stamps[1:])]
    pk = np.loadtxt(os.path.join(tmp, "Pk.txt"))
    cpos, cvel = orc.preprocess_script(pos.copy(), mass, vel.copy())
    ref, _ = orc.script_pipeline(cpos, cvel, N, L)
    ok = pk[:, 3] > 0
    nsample_equal = bool(np.array_equal(pk[:, 3], np.asarray(ref, dtype=np.float64)[:, 3]))
    psum_rel = float(np.max(np.abs(pk[ok, 2] - ref[ok, 2]) / np.abs(ref[ok, 2])))
    cpu = "unknown"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    res = {"what": "reference scripts/parallel_optimized.py main() (script:201-491) under sys.modules stand-ins, 1 emulated MPI rank, 1 core",
           "grid": N, "particles": Np, "wall_seconds": wall, "stages": stages,
           "cells_components_per_s": N ** 3 * 3 / wall,
           "stand_ins": "pyfftw -> numpy.fft; annoy -> scipy cKDTree (exact NN; Annoy's approximate answers are parity-unpinned); "
                        "mpi4py -> one-rank communicator; h5py -> dict",
           "check_vs_oracle": {"nsample_equal": nsample_equal, "psum_max_rel": psum_rel},
           "cpu_model": cpu, "container_cores": os.cpu_count(), "threads_used": 1}
    print(json.dumps(res, indent=1))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
