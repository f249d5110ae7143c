"""world_size-2 (and 4) gloo tests of the slab choreography on CPU: the all-to-all layout,
the segmented x-pass addressing and the shell all-reduce, with the oracle-backed kernel
stand-in in place of the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vps_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, L, seed, out_dir, chunks=None):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if chunks is not None:
        os.environ["VPS_A2A_CHUNKS"] = str(chunks)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vpower import device
        from oracle_kernels import OracleKernels
        rng = np.random.default_rng(seed)
        fields = [rng.standard_normal((N, N, N)).astype(np.float32) for _ in range(3)]
        pipe = device.PowerPipeline(N, L, kernels=OracleKernels(), comm=device.SlabComm())
        assert pipe.comm.world == world and pipe.nx == N // world and pipe.x0 == rank * (N // world)
        assert pipe.chunked and (chunks is None or pipe.nchunks == chunks)
        slabs = [torch.from_numpy(np.ascontiguousarray(f[pipe.x0: pipe.x0 + pipe.nx])) for f in fields]
        tab = pipe.spectrum(slabs)
        np.save(os.path.join(out_dir, f"tab_{rank}.npy"), tab)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 16), (2, 32), (4, 32)])
def test_slab_pipeline_matches_oracle(tmp_path, world, N):
    L, seed = 1.0, 11
    mp.spawn(_worker, args=(world, _free_port(), N, L, seed, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    fields = [rng.standard_normal((N, N, N)).astype(np.float32).astype(np.float64) for _ in range(3)]
    ref = orc.spectrum_table(orc.vector_power(*fields, L, N), L, N, "library")
    for r in range(world):
        tab = np.load(tmp_path / f"tab_{r}.npy")
        assert np.array_equal(tab[:, 3], ref[:, 3])          # shell counts bit exact on every rank
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-5)
        assert np.allclose(tab[:, 1], ref[:, 1], rtol=1e-5)
    assert np.array_equal(np.load(tmp_path / "tab_0.npy"), np.load(tmp_path / f"tab_{world - 1}.npy"))


@pytest.mark.parametrize("world,N,chunks", [(2, 32, 2), (2, 32, 4), (4, 32, 4), (4, 64, 2)])
def test_chunked_exchange_one_message_per_field(tmp_path, world, N, chunks):
    """The kz-chunked pipeline (y pass of a chunk -> all-to-all of that chunk -> x pass of the arrived chunk, the
    Nyquist-plane rows riding behind the last chunk of the same message) with more than one chunk per field."""
    L, seed = 2.5, 5
    mp.spawn(_worker, args=(world, _free_port(), N, L, seed, str(tmp_path), chunks), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    fields = [rng.standard_normal((N, N, N)).astype(np.float32).astype(np.float64) for _ in range(3)]
    ref = orc.spectrum_table(orc.vector_power(*fields, L, N), L, N, "library")
    for r in range(world):
        tab = np.load(tmp_path / f"tab_{r}.npy")
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-5)


def _pipelined_worker(rank, world, port, N, L, seed, out_dir, chunks, per_component):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VPS_A2A_CHUNKS"] = str(chunks)
    if per_component:
        os.environ["VPS_X_PER_COMPONENT"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vpower import device
        from oracle_kernels import OracleKernels
        K = OracleKernels()
        pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm())
        assert pipe.chunked and pipe.nchunks == chunks
        quantities = _quantity_fields(N, seed)
        order = []

        def producer(i, fields):
            def produce():
                order.append(("z", i))
                return [K.fft_z(torch.from_numpy(np.ascontiguousarray(f[pipe.x0: pipe.x0 + pipe.nx])), N, pipe.nx) for f in fields]
            return produce
        accs = [pipe.new_accumulators() for _ in quantities]
        # spy on the jobs (one per kz chunk and group of components): started in order, binned in the same order, never more
        # than `inflight_max` between start and x pass, and quantity q + 1's z images produced while q's last jobs travel
        real_start, real_finish = pipe._start_chunk, pipe._finish_chunk

        def spy_start(comps, c, packed):
            job = real_start(comps, c, packed)
            job["id"] = len([o for o in order if o[0] == "s"])
            order.append(("s", job["id"]))
            return job

        def spy_finish(job, *a, **kw):
            order.append(("x", job["id"]))
            return real_finish(job, *a, **kw)
        pipe._start_chunk, pipe._finish_chunk = spy_start, spy_finish
        pipe.pipelined_quantities([producer(i, f) for i, f in enumerate(quantities)], accs)
        group = 1 if per_component else 3
        njobs = [-(-len(f) // group) * chunks for f in quantities]
        maxin = pipe.inflight_max()
        assert maxin == 2
        assert [o[1] for o in order if o[0] == "s"] == list(range(sum(njobs)))
        assert [o[1] for o in order if o[0] == "x"] == list(range(sum(njobs)))
        live = 0
        for o in order:
            live += {"s": 1, "x": -1, "z": 0}[o[0]]
            assert 0 <= live <= maxin
        for q in (1, 2):        # quantity q's z pass is enqueued before the last jobs of quantity q - 1 are binned
            last_prev = sum(njobs[:q]) - 1
            assert order.index(("z", q)) < order.index(("x", last_prev))
            assert order.index(("z", q)) > order.index(("s", last_prev))
        tabs = np.stack([pipe.finish(*a) for a in accs])
        np.save(os.path.join(out_dir, f"tabs_{rank}.npy"), tabs)
    finally:
        dist.destroy_process_group()


def _quantity_fields(N, seed):
    """Three 'quantities' like C4's: two vector fields and one scalar field (float32 values)."""
    rng = np.random.default_rng(seed)
    mk = lambda n: [rng.standard_normal((N, N, N)).astype(np.float32) for _ in range(n)]
    return [mk(3), mk(3), mk(1)]


@pytest.mark.parametrize("world,N,chunks,per_component", [(4, 32, 2, False), (4, 64, 4, False), (2, 32, 2, True), (4, 32, 1, True)])
def test_quantity_pipelined_exchange_matches_oracle(tmp_path, world, N, chunks, per_component):
    """PowerPipeline.pipelined_quantities -- the branch `bench.py --gpus N` takes for C4: the kz chunks of all quantities form
    ONE sequence of jobs (y pass -> all-to-all -> x pass + shell sums) of which two are in flight at a time, so quantity
    q+1's deposit + z pass is enqueued while q's last chunks cross the node and only two chunks' packed send / receive buffers
    are alive; every quantity has its own accumulators.  Three quantities (3 + 3 + 1 components) on 2 / 4 gloo ranks, grouped
    (three components per binning launch) and per component, against the oracle's tables."""
    L, seed = 1.5, 23
    mp.spawn(_pipelined_worker, args=(world, _free_port(), N, L, seed, str(tmp_path), chunks, per_component),
             nprocs=world, join=True)
    refs = []
    for fields in _quantity_fields(N, seed):
        f64 = [f.astype(np.float64) for f in fields]
        P = orc.vector_power(*f64, L, N) if len(f64) == 3 else orc.scalar_power(f64[0], L, N)
        refs.append(orc.spectrum_table(P, L, N, "library"))
    for r in range(world):
        tabs = np.load(tmp_path / f"tabs_{r}.npy")
        for tab, ref in zip(tabs, refs):
            t4 = tab.copy()
            t4[:, 1] *= 4 * np.pi * t4[:, 0] ** 2
            assert np.array_equal(tab[:, 3], ref[:, 3])
            assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-5)
            assert np.allclose(t4[:, 1], ref[:, 1], rtol=1e-5)
    assert np.array_equal(np.load(tmp_path / "tabs_0.npy"), np.load(tmp_path / f"tabs_{world - 1}.npy"))


def _fields_worker(rank, world, port, N, L, seed, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vpower import device
        from oracle_kernels import OracleKernels
        comm = device.FieldComm()
        assert (comm.world, comm.rank, comm.field_world, comm.field_rank) == (1, 0, world, rank)
        quantities = ("velocity", "energy")
        units = device.FieldComm.units(quantities)
        assert units == [("velocity", 0), ("velocity", 1), ("velocity", 2), ("energy", None)]
        mine = comm.mine(quantities)
        base, extra = divmod(len(units), world)          # contiguous blocks, the first `extra` ranks one field more
        lo = rank * base + min(rank, extra)
        assert mine == units[lo: lo + base + (1 if rank < extra else 0)]
        rng = np.random.default_rng(seed)
        fields = {("velocity", c): rng.standard_normal((N, N, N)).astype(np.float32) for c in range(3)}
        fields[("energy", None)] = rng.standard_normal((N, N, N)).astype(np.float32)
        pipe = device.PowerPipeline(N, L, kernels=OracleKernels(), comm=comm)
        assert pipe.nx == N and pipe.x0 == 0 and not pipe.chunked          # whole grids on every rank
        tabs = []
        for q in quantities:
            psum, nsample = pipe.new_accumulators()
            first = True
            for u in mine:
                if u[0] == q:
                    pipe.accumulate([torch.from_numpy(fields[u])], psum, nsample, count=first)
                    first = False
            tabs.append(pipe.finish(psum, nsample))      # sums added, counts MAX-reduced (ranks without a field of q hold zeros)
        np.save(os.path.join(out_dir, f"ftab_{rank}.npy"), np.stack(tabs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 16), (3, 16), (4, 32), (5, 16)])
def test_field_parallel_ranks_match_oracle(tmp_path, world, N):
    """device.FieldComm: the scalar fields of a step dealt out over the ranks, whole grids, no exchange; shell sums added and
    shell counts MAX-reduced -- also with more ranks than fields (5 ranks, 4 fields: one rank idles through the reductions)."""
    L, seed = 1.5, 23
    mp.spawn(_fields_worker, args=(world, _free_port(), N, L, seed, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    v = [rng.standard_normal((N, N, N)).astype(np.float32).astype(np.float64) for _ in range(3)]
    e = rng.standard_normal((N, N, N)).astype(np.float32).astype(np.float64)
    refs = [orc.spectrum_table(orc.vector_power(*v, L, N), L, N, "library"),
            orc.spectrum_table(orc.scalar_power(e, L, N), L, N, "library")]
    for r in range(world):
        tabs = np.load(tmp_path / f"ftab_{r}.npy")
        for tab, ref in zip(tabs, refs):
            assert np.array_equal(tab[:, 3], ref[:, 3])
            assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-5)
            assert np.allclose(tab[:, 1] * 4 * np.pi * tab[:, 0] ** 2, ref[:, 1], rtol=1e-5)    # (finish() stops before 4 pi k^2)
