#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels to the
GPU box):   python tests/golden/make_goldens.py

The reference's third-party dependencies that are not installed here (pyfftw,
mpi4py, annoy, h5py, pyann, voxelize, memory_profiler) are replaced by small
`sys.modules` stand-ins defined below:
  pyfftw  -> numpy.fft          (DFT is mathematically defined; tolerance parity)
  annoy   -> exact NN (float32 coordinates, float64 distances)  [Annoy's own
             approximate answers are parity-unpinned, SURVEY.md section 8c]
  pyann   -> exact NN, 1-based indices as pyann returns them
  mpi4py  -> one-rank communicator; for the multi-rank fixtures a thread-backed one
             (every emulated rank is a thread calling the reference's main(); allgather /
             Reduce / Barrier meet at a threading.Barrier, gather order = rank order)
  h5py    -> dict-backed File (datasets hand out copies, as reading a file does)
Only inputs-by-seed and the reference's OUTPUTS are written (npz, no pickles);
no reference source text is stored.
"""
import os
import sys
import types
import tempfile
import importlib.util

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ----------------------------------------------------------------------------
# stand-ins
# ----------------------------------------------------------------------------
def _exact_nn(data, queries):
    """argmin of float64 squared distance ((dx^2+dy^2)+dz^2), lowest index on ties."""
    P = np.asarray(data, dtype=np.float64)
    Q = np.asarray(queries, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(Q), dtype=np.int64)
    for s in range(0, len(Q), 2048):
        q = Q[s:s + 2048]
        d = (q[:, None, 0] - P[None, :, 0]) ** 2
        d = d + (q[:, None, 1] - P[None, :, 1]) ** 2
        d = d + (q[:, None, 2] - P[None, :, 2]) ** 2
        out[s:s + 2048] = np.argmin(d, axis=1)
    return out


def install_shims(h5_store):
    # pyfftw ---------------------------------------------------------------
    pyfftw = types.ModuleType("pyfftw")
    interfaces = types.ModuleType("pyfftw.interfaces")
    cache = types.ModuleType("pyfftw.interfaces.cache")
    cache.enable = lambda: None
    numpy_fft = types.ModuleType("pyfftw.interfaces.numpy_fft")

    def fftn(a, s=None, axes=None, threads=1, **kw):
        return np.fft.fftn(a, s=s, axes=axes)
    numpy_fft.fftn = fftn
    interfaces.cache = cache
    interfaces.numpy_fft = numpy_fft
    pyfftw.interfaces = interfaces
    pyfftw.empty_aligned = lambda shape, dtype="float64", **kw: np.empty(shape, dtype=dtype)

    class FFTW:
        def __init__(self, a, b, axes=(-1,), **kw):
            self.a, self.b, self.axes = a, b, axes

        def __call__(self, x=None):
            src = self.a if x is None else x
            self.b[...] = np.fft.fftn(src, axes=self.axes)
            return self.b
    pyfftw.FFTW = FFTW
    for name, mod in (("pyfftw", pyfftw), ("pyfftw.interfaces", interfaces),
                      ("pyfftw.interfaces.cache", cache),
                      ("pyfftw.interfaces.numpy_fft", numpy_fft)):
        sys.modules[name] = mod

    # mpi4py ---------------------------------------------------------------
    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")

    class Comm:
        """One rank, or `size` ranks emulated by threads (set_world(size); each thread calls
        bind(rank) first).  Semantics of the three collectives the reference uses
        (scripts/parallel_optimized.py:254, 365-368, 455-456)."""
        def __init__(self):
            self.set_world(1)

        def set_world(self, size):
            import threading
            self.size = size
            self.local = threading.local()
            self.barrier = threading.Barrier(size)
            self.slots = [None] * size

        def bind(self, rank): self.local.rank = rank
        def Get_rank(self): return getattr(self.local, "rank", 0)
        def Get_size(self): return self.size
        def Barrier(self): self.barrier.wait()

        def allgather(self, x):
            if self.size == 1:
                return [x]
            self.slots[self.Get_rank()] = x
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out

        def Reduce(self, sendbuf, recvbuf, op=None, root=0):
            if self.size == 1:
                recvbuf[...] = sendbuf
                return
            parts = self.allgather(np.array(sendbuf, copy=True))
            if self.Get_rank() == root:
                acc = parts[0].copy()
                for q in parts[1:]:          # rank order, in the send dtype (MPI_SUM on float32 buffers)
                    acc = acc + q
                recvbuf[...] = acc
    MPI.COMM_WORLD = Comm()
    MPI.SUM = "sum"
    mpi4py.MPI = MPI
    sys.modules["mpi4py"] = mpi4py
    sys.modules["mpi4py.MPI"] = MPI

    # annoy ----------------------------------------------------------------
    annoy = types.ModuleType("annoy")

    class AnnoyIndex:
        def __init__(self, dim, metric):
            self.items = {}
            self.data = None

        def add_item(self, i, v): self.items[i] = np.asarray(v, dtype=np.float32)
        def build(self, n_trees, n_jobs=-1):
            self.data = np.stack([self.items[i] for i in range(len(self.items))])
        def save(self, fn): pass
        def load(self, fn): raise RuntimeError("no index cache in golden generation")
        def get_nns_by_vector(self, q, n=1, search_k=-1, include_distances=False):
            return [int(_exact_nn(self.data, np.asarray(q, dtype=np.float32)[None])[0])]
    annoy.AnnoyIndex = AnnoyIndex
    sys.modules["annoy"] = annoy

    # h5py -----------------------------------------------------------------
    h5py = types.ModuleType("h5py")

    class Dataset:
        def __init__(self, a): self.a = a
        def __getitem__(self, k): return np.array(self.a[k], copy=True)

    class File:
        def __init__(self, name, mode="r"): self.store = h5_store
        def __getitem__(self, key):
            node = self.store
            for part in key.split("/"):
                node = node[part]
            return Dataset(node) if isinstance(node, np.ndarray) else node
        def close(self): pass
    h5py.File = File
    sys.modules["h5py"] = h5py

    # pyann ----------------------------------------------------------------
    pyann = types.ModuleType("pyann")

    def nn2(data, query, k=1, eps=0.0, treetype="kd", searchtype="standard"):
        idx = _exact_nn(np.asarray(data), np.asarray(query)) + 1  # 1-based
        return types.SimpleNamespace(nn_idx=np.matrix(idx).T)
    pyann.nn2 = nn2
    sys.modules["pyann"] = pyann

    # voxelize, memory_profiler ---------------------------------------------
    vox = types.ModuleType("voxelize")

    class Voxelize:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): raise NotImplementedError
    vox.Voxelize = Voxelize
    sys.modules["voxelize"] = vox
    mp = types.ModuleType("memory_profiler")
    mp.profile = lambda f: f
    sys.modules["memory_profiler"] = mp


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------
# synthetic inputs (same recipe as BASELINE.md section 3)
# ----------------------------------------------------------------------------
def synth(seed, Np, L=1.0, lognormal_density=True):
    rng = np.random.default_rng(seed)
    pos = rng.random((Np, 3), dtype=np.float32) * np.float32(L)
    vel = rng.standard_normal((Np, 3), dtype=np.float32)
    mass = np.ones(Np, dtype=np.float32)
    if lognormal_density:
        dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    else:
        dens = np.ones(Np, dtype=np.float32)
    return pos, vel, mass, dens


def main():
    import matplotlib
    matplotlib.use("Agg")
    h5_store = {}
    install_shims(h5_store)
    sys.path.insert(0, os.path.join(REF, "vpower"))   # interp.py does `from spctrm import ...`
    import io
    import contextlib

    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        interp = load_module("interp", os.path.join(REF, "vpower", "interp.py"))
    spctrm = sys.modules["spctrm"]

    tmp = tempfile.mkdtemp()
    snap = os.path.join(tmp, "snap.hdf5")
    open(snap, "w").close()
    sys.argv = ["parallel_optimized.py", "-i", snap, "-o", tmp, "-N", "16", "-M", "16", "-f"]
    with contextlib.redirect_stdout(sink):
        script = load_module("ref_script", os.path.join(REF, "scripts", "parallel_optimized.py"))

    out = {}

    # (i) planner table -----------------------------------------------------
    plan_in = [(128, 1, 128, 1), (512, 1, 512, 1), (1000, 1, 500, 8), (1024, 1, 512, 8),
               (2048, 1, 512, 8), (4096, 1, 512, 8), (16, 1, 4, 8), (500, 1, 500, 8)]
    plan_out = [script.planner(*p) for p in plan_in]
    np.savez(os.path.join(HERE, "planner.npz"),
             inputs=np.array(plan_in, dtype=np.int64),
             outputs=np.array(plan_out, dtype=np.float64))

    # (ii) binning edges / counts for both flavours ---------------------------
    edge = {}
    for N in (16, 32, 64, 128, 500, 512, 1000, 1024, 2048, 4096):
        L = 1.0
        Lcell = L / N
        kmin, kmax = 2 * np.pi / L, np.pi / Lcell
        # drive the reference functions with a 2-sample pair array and read back
        # their centres; edges are recovered through np.histogram's return value
        probe = np.array([[kmin, 1.0], [kmax, 2.0]])
        tab_s = script.hist_sample(probe, kmin, kmax, kmin)
        tab_l = interp._hist_sample(probe, kmin, kmax, kmin)
        edge[f"script_centres_{N}"] = tab_s[:, 0]
        edge[f"library_centres_{N}"] = tab_l[:, 0]
        edge[f"script_probe_counts_{N}"] = tab_s[:, 3]
        edge[f"library_probe_counts_{N}"] = tab_l[:, 3]
    np.savez(os.path.join(HERE, "bin_edges.npz"), **edge)

    # (iii) Nsample lattice counts (input independent, integer exact) ---------
    ns = {}
    for N in (16, 32, 64, 128):
        L = 1.0
        P = np.ones((N, N, N))
        pk_l = interp._pair_power(P, L, N)
        pk_s = script.pair_power(P, L, N)
        kmin, kmax = 2 * np.pi / L, np.pi / (L / N)
        ns[f"library_{N}"] = interp._hist_sample(pk_l, kmin, kmax, kmin)[:, 3].astype(np.int64)
        ns[f"script_{N}"] = script.hist_sample(pk_s, kmin, kmax, kmin)[:, 3].astype(np.int64)
        if N <= 32:
            ns[f"pair_k_{N}"] = pk_l[:, 0]
    # a non power-of-two physical box to exercise the float64 edge arithmetic
    N, L = 32, 2.5
    pk_l = interp._pair_power(np.ones((N, N, N)), L, N)
    kmin, kmax = 2 * np.pi / L, np.pi / (L / N)
    ns["library_32_L2p5"] = interp._hist_sample(pk_l, kmin, kmax, kmin)[:, 3].astype(np.int64)
    np.savez(os.path.join(HERE, "nsample.npz"), **ns)

    # (iv) library pipeline: deposit / NN -> BoxField -> spctrm ---------------
    for tag, N, Np, seed in (("n16", 16, 3000, 101), ("n32", 32, 30000, 102)):
        L = 1.0
        pos, vel, mass, dens = synth(seed, Np, L)
        d = dict(seed=seed, N=N, Np=Np, L=L)
        # cell indices + deposit (float32 and float64 positions)
        Lcell = L / float(N)
        d["cell_f32"] = np.array((pos // Lcell) % N, dtype=int).astype(np.int32)
        pos64 = pos.astype(np.float64) * 1.000001
        d["cell_f64"] = np.array((pos64 // Lcell) % N, dtype=int).astype(np.int32)
        vec = np.stack((vel[:, 0] * dens, vel[:, 1] * dens, vel[:, 2] * dens, dens), axis=1)
        with contextlib.redirect_stdout(sink):
            gp = interp.GasParticles(pos.astype(np.float64), mass.astype(np.float64),
                                     dens.astype(np.float64), vel.astype(np.float64), L)
            vec_ref = gp.density_velocity_vector()
            grid = interp.deposit_to_grid(vec_ref, gp.pos, N, L)
            d["dvv_head"] = vec_ref[:8]
            d["deposit_grid"] = grid.astype(np.float64)
            d["deposit_scalar"] = interp.deposit_to_grid(dens.astype(np.float64), gp.pos, N, L)
            # composition following interp.py:272-275 with the NaN->0 rule of :329-331
            with np.errstate(invalid="ignore", divide="ignore"):
                v = grid[..., :3] / grid[..., 3, None]
            m = grid[..., 3] * Lcell ** 3
            v[np.isnan(v)] = 0
            m[np.isnan(m)] = 0
            bf = interp.BoxField(v, m, Lcell)
            for q in ("velocity", "momentum", "energy"):
                d[f"ngp_{q}"] = bf.spctrm(q).data()
            d["ngp_velocity_Pgrid"] = bf.velocity_power() if N <= 16 else np.zeros(1)
            # exact-NN flavour
            coords = interp.make_grid_coords(L, N)
            d["grid_coords_head"] = coords[: 2 * N]
            d["grid_coords_tail"] = coords[-2 * N:]
            vg = interp.ann_interpolate(gp.pos, coords, vec_ref, N, 0.0)
            d["nn_idx"] = _exact_nn(gp.pos, coords).astype(np.int32)
            d["nn_vec_grid_sum"] = vg.sum(axis=(0, 1, 2))
            bf2 = gp.ann_interp_to_field(N)
            d["nn_mass_head"] = bf2.mass[0, 0, :8]
            for q in ("velocity", "momentum", "energy"):
                d[f"nn_{q}"] = bf2.spctrm(q).data()
        np.savez_compressed(os.path.join(HERE, f"library_{tag}.npz"), **d)

    # (v) script main(), one rank ------------------------------------------
    for tag, N, Np, seed in (("n16", 16, 3000, 201), ("n32", 32, 20000, 202)):
        L = 1
        pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
        h5_store.clear()
        h5_store["PartType0"] = {"Coordinates": pos.copy(), "Masses": mass.copy(),
                                 "Velocities": vel.copy()}
        outdir = tempfile.mkdtemp()
        script.SNAPSHOT, script.SAVEDIR = snap, outdir
        script.NTOT, script.MAXNBOX, script.LTOT = N, N, L
        script.NBUFFER, script.FORCE = 5000, True
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            assert script.main() == 0
        pk = np.loadtxt(os.path.join(outdir, "Pk.txt"))
        np.savez_compressed(os.path.join(HERE, f"script_{tag}.npz"),
                            seed=seed, N=N, Np=Np, L=L, Pk=pk)

    # (v-b) script main() on 8 emulated MPI ranks: the fold + allgather + Reduce path
    # (scripts/parallel_optimized.py:322-391, 455-456) on the SAME particles as (v).
    #   n16_m2: -N 16 -M 8  -> 8 ranks x 8^3 boxes, 1 loop,  2-fold
    #   n16_m4: -N 16 -M 4  -> 8 ranks x 4^3 boxes, 8 loops, 4-fold (Pk.txt accumulates over the loops, :472-485)
    #   n32_m2: -N 32 -M 16 -> 8 ranks x 16^3 boxes, 1 loop, 2-fold
    import threading
    comm = sys.modules["mpi4py.MPI"].COMM_WORLD
    for tag, N, M, Np, seed in (("n16_m2", 16, 8, 3000, 201), ("n16_m4", 16, 4, 3000, 201), ("n32_m2", 32, 16, 20000, 202)):
        L = 1
        pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
        h5_store.clear()
        h5_store["PartType0"] = {"Coordinates": pos.copy(), "Masses": mass.copy(), "Velocities": vel.copy()}
        outdir = tempfile.mkdtemp()
        script.SNAPSHOT, script.SAVEDIR = snap, outdir
        script.NTOT, script.MAXNBOX, script.LTOT = N, M, L
        script.NBUFFER, script.FORCE = 5000, True
        plan = script.planner(N, L, M, 8)
        comm.set_world(8)
        errors = []

        def run_rank(r):
            comm.bind(r)
            try:
                assert script.main() == 0
            except BaseException as e:      # noqa: BLE001  (a dead rank must not leave the others at the barrier)
                errors.append((r, repr(e)))
                comm.barrier.abort()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            threads = [threading.Thread(target=run_rank, args=(r,)) for r in range(8)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        comm.set_world(1)
        assert not errors, errors
        pk = np.loadtxt(os.path.join(outdir, "Pk.txt"))
        np.savez_compressed(os.path.join(HERE, f"script_8rank_{tag}.npz"), seed=seed, N=N, M=M, Np=Np, L=L, ranks=8,
                            plan=np.array(plan, dtype=np.float64), Pk=pk)

    # (vi) FFT power known answers through the reference functions -----------
    rng = np.random.default_rng(7)
    N, L = 16, 3.0
    fx, fy, fz = (rng.standard_normal((N, N, N)) for _ in range(3))
    with contextlib.redirect_stdout(sink):
        Pv = interp._vector_power(fx, fy, fz, L, N)
        Ps = interp._scalar_power(fx, L, N)
        P32 = script.FFTW_power(fx.astype(np.complex64), L, N)
        P32v = script.FFTW_vector_power(fx.astype(np.complex64), fy.astype(np.complex64),
                                        fz.astype(np.complex64), L, N)
    np.savez_compressed(os.path.join(HERE, "fft_power.npz"), seed=7, N=N, L=L,
                        vector=Pv, scalar=Ps, script_scalar=P32, script_vector=P32v)

    # (vii) spectrum container behaviour -----------------------------------
    a = spctrm.PowerSpectrum(np.column_stack((np.arange(1., 6.), np.arange(5.) + 2,
                                              np.arange(5.) * 3 + 1, np.arange(5.) + 4)))
    b = spctrm.PowerSpectrum(np.column_stack((np.arange(1., 6.), np.arange(5.) + 1,
                                              np.arange(5.) * 2 + 1, np.arange(5.) + 1)))
    e0, kres = a.energy(), a.kres()
    c = a.copy()
    c.add(b)
    with contextlib.redirect_stdout(sink):
        rd = spctrm.relative_diff(a.copy(), b.copy(), mode="max")
        rdm = spctrm.relative_diff(a.copy(), b.copy(), mode="mean")
    np.savez(os.path.join(HERE, "spectrum_container.npz"), a=a.data(), b=b.data(),
             energy=e0, kres=kres, added=c.data(), reldiff_max=rd, reldiff_mean=rdm,
             beta_space_2=spctrm.init_beta_space(2))
    print("goldens written to", HERE)


if __name__ == "__main__":
    main()
