"""Two ranks sharing the one GPU of the test box: the real HIP kernels on x-slabs, the
exchange over gloo (staged through host memory by SlabComm), checked against the oracle.
RCCL itself needs one GPU per rank, so the nccl backend is exercised by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import vps_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, Np, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vpower import device, synth
        K = device.default_kernels(0)
        pos, vel, mass, dens = synth.particles(31, Np, 1.0)
        pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm())
        assert pipe.comm.world == world and pipe.x0 == rank * (N // world)
        fields = K.deposit_field(K.to_device(pos), K.to_device(vel), K.to_device(dens), N, 1.0,
                                 pipe.x0, pipe.nx, device.VELOCITY)
        tab = pipe.spectrum([fields[0], fields[1], fields[2]])
        np.save(os.path.join(out_dir, f"tab_{rank}.npy"), tab)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 64), (4, 128)])
def test_two_ranks_one_gpu_hip_kernels(tmp_path, world, N):
    import torch.multiprocessing as mp
    from vpower import synth
    Np = 200000
    mp.spawn(_worker, args=(world, _free_port(), N, Np, str(tmp_path)), nprocs=world, join=True)
    pos, vel, mass, dens = synth.particles(31, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, "velocity")
    for r in range(world):
        tab = np.load(tmp_path / f"tab_{r}.npy")
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-5, atol=0)


def _rccl_worker(rank, port, N, Np, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["VPS_FORCE_COLLECTIVES"] = "1"
    os.environ["VPS_A2A_CHUNKS"] = "4"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from vpower import device, synth
        K = device.default_kernels(0)
        pos, vel, mass, dens = synth.particles(33, Np, 1.0)
        comm = device.SlabComm()
        assert comm.backend == "nccl" and comm.force
        pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=comm)
        d = [K.to_device(a) for a in (pos, vel, dens)]
        assert pipe.chunked and pipe.nchunks == 4
        z = K.deposit_fft_z(d[0], d[1], d[2], N, 1.0, 0, N, device.VELOCITY)
        tab = pipe.finish(*pipe.accumulate_zimages([z[0], z[1], z[2]]))   # fused path: chunked complex all-to-alls + all-reduces over RCCL
        fields = K.deposit_field(d[0], d[1], d[2], N, 1.0, 0, N, device.VELOCITY)
        tab2 = pipe.finish(*pipe.accumulate([fields[0], fields[1], fields[2]]))
        # field-parallel reductions over RCCL: SUM of the float64 view, MAX of the int64 view of the one accumulator buffer
        # (a one-rank group, the early return for a single field rank switched off)
        fcomm = device.FieldComm()
        assert fcomm.backend == "nccl" and (fcomm.world, fcomm.field_world) == (1, 1)
        fcomm.field_world = 2                      # reductions are issued; over one rank they are the identity
        fpipe = device.PowerPipeline(N, 1.0, kernels=K, comm=fcomm)
        psum, nsample = fpipe.new_accumulators()
        with K.binning_only():
            spec, nyq = K.deposit_fft_zy(d[0], d[1], d[2], N, 1.0, 0, N, device.VELOCITY, component=(0, 1))
        fpipe.accumulate_spectra(spec, nyq, psum, nsample)
        with K.binning_only():
            spec, nyq = K.deposit_fft_zy(d[0], d[1], d[2], N, 1.0, 0, N, device.VELOCITY, component=2)
        fpipe.accumulate_spectra(spec, nyq, psum, nsample, count=False)
        tab3 = fpipe.finish(psum, nsample)
        np.save(os.path.join(out_dir, "tab_rccl.npy"), np.stack([tab, tab2, tab3]))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_single_rank(tmp_path):
    """The nccl (= RCCL) code path of SlabComm -- complex tensors as real views, asynchronous
    all_to_all_single, all-reduce of the float64 / int64 shell accumulators -- driven on the one GPU of
    the test box by forcing the collectives in a one-rank group."""
    import torch.multiprocessing as mp
    from vpower import synth
    N, Np = 64, 100000
    mp.spawn(_rccl_worker, args=(_free_port(), N, Np, str(tmp_path)), nprocs=1, join=True)
    pos, vel, mass, dens = synth.particles(33, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, "velocity")
    for tab in np.load(tmp_path / "tab_rccl.npy"):
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-5, atol=0)


def _library_rccl_worker(rank, port, N, Np, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["VPS_A2A_CHUNKS"] = "4"
    torch.cuda.set_device(0)
    from vpower import device, synth
    K = device.default_kernels(0)
    pos, vel, mass, dens = synth.particles(37, Np, 1.0)
    comm = device.LibraryComm(K, rank=0, world=1, uid=device.HipKernels.comm_unique_id())      # no torch.distributed at all
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=comm)
    assert pipe.chunked and pipe.nchunks == 4 and comm.backend == "library"
    d = [K.to_device(a) for a in (pos, vel, dens)]
    tabs = []
    for q in (device.VELOCITY, device.ENERGY):
        z = K.deposit_fft_z(d[0], d[1], d[2], N, 1.0, 0, N, q)
        tabs.append(pipe.finish(*pipe.accumulate_zimages([z[i] for i in range(z.shape[0])])))
        tabs.append(pipe.finish(*pipe.accumulate_zimages([z[i] for i in range(z.shape[0])])))    # buffers and events re-used
    np.save(os.path.join(out_dir, "tab_lib.npy"), np.stack(tabs))
    K.comm_destroy()


def test_library_rccl_exchange_single_rank(tmp_path):
    """The exchange behind the C ABI (vps_comm_create / vps_spectrum_zimages / vps_allreduce_shells: RCCL loaded by the
    library, grouped ncclSend / ncclRecv per kz chunk on its own stream, events against the context's stream, ncclAllReduce of
    the float64 / uint64 accumulators) on the one GPU of the test box: a one-rank communicator made from a unique id, no
    torch.distributed involved.  Velocity (three components per launch) and energy against the oracle, twice each."""
    import torch.multiprocessing as mp
    from vpower import synth
    N, Np = 128, 200000
    mp.spawn(_library_rccl_worker, args=(_free_port(), N, Np, str(tmp_path)), nprocs=1, join=True)
    pos, vel, mass, dens = synth.particles(37, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    tabs = np.load(tmp_path / "tab_lib.npy")
    for i, q in enumerate(("velocity", "velocity", "energy", "energy")):
        ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, q)
        ref[:, 1] /= np.where(ref[:, 0] > 0, 4 * np.pi * ref[:, 0] ** 2, 1)
        assert np.array_equal(tabs[i][:, 3], ref[:, 3])
        assert np.allclose(tabs[i][:, 2], ref[:, 2], rtol=2e-5, atol=0)


def _nccl_worker(rank, world, port, N, Np, out_dir, transport="torch"):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["VPS_A2A_CHUNKS"] = "2"
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from vpower import device, synth
        K = device.default_kernels(rank)
        pos, vel, mass, dens = synth.particles(35, Np, 1.0)
        comm = device.SlabComm() if transport == "torch" else device.LibraryComm(K)    # the latter: RCCL inside libvps_hip.so
        pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=comm)
        assert pipe.comm.backend == ("nccl" if transport == "torch" else "library") and pipe.comm.world == world and pipe.chunked
        d = [K.to_device(a) for a in (pos, vel, dens)]
        z = K.deposit_fft_z(d[0], d[1], d[2], N, 1.0, pipe.x0, pipe.nx, device.VELOCITY)
        tab = pipe.finish(*pipe.accumulate_zimages([z[0], z[1], z[2]]))
        np.save(os.path.join(out_dir, f"tab_nccl_{rank}.npy"), tab)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["torch", "library"])
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL cannot put two ranks on one device)")
def test_two_ranks_two_gpus_rccl(tmp_path, transport):
    """The slab path over RCCL between two real devices (runs wherever two GPUs are visible; the 1-GPU test box
    skips it): chunked all-to-all of the fused z images, x pass on the received blocks, all-reduce; every rank's
    table against the oracle."""
    import torch.multiprocessing as mp
    from vpower import synth
    N, Np, world = 128, 300000, 2
    mp.spawn(_nccl_worker, args=(world, _free_port(), N, Np, str(tmp_path), transport), nprocs=world, join=True)
    pos, vel, mass, dens = synth.particles(35, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, "velocity")
    for r in range(world):
        tab = np.load(tmp_path / f"tab_nccl_{r}.npy")
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-5, atol=0)


def _bench_pipelined_worker(rank, world, port, N, Np, chunks, per_component, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VPS_PIPELINE_QUANTITIES"] = "1"
    os.environ["VPS_A2A_CHUNKS"] = str(chunks)
    if per_component:
        os.environ["VPS_X_PER_COMPONENT"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from vpower import device, synth
        K = device.default_kernels(0)
        pos, vel, mass, dens = synth.particles(41, Np, 1.0)
        wl = bench.Workload(K, device.SlabComm(), N, 1.0, "ngp", ("velocity", "momentum", "energy"), "library",
                            K.to_device(pos), K.to_device(vel), K.to_device(dens))
        assert wl.pipelined and wl.pipe.chunked and wl.pipe.nchunks == chunks and wl.pipe.comm.world == world
        tabs = wl.step()
        tabs2 = wl.step()            # a second step reuses every buffer (z images, accumulators, sort workspace)
        for q in tabs:
            assert np.array_equal(tabs[q][:, 3], tabs2[q][:, 3]) and np.allclose(tabs[q][:, 2], tabs2[q][:, 2], rtol=1e-6)
        np.save(os.path.join(out_dir, f"tabs_{rank}.npy"), np.stack([tabs[q] for q in ("velocity", "momentum", "energy")]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,chunks,per_component", [(4, 128, 2, False), (4, 256, 4, False), (2, 128, 4, True)])
def test_bench_quantity_pipelined_step_on_shared_gpu(tmp_path, world, N, chunks, per_component):
    """The branch `bench.py --gpus 8` takes for C4 (bench.Workload.step with VPS_PIPELINE_QUANTITIES: fused deposit + z pass
    of quantity q+1 and its packed-row y passes / exchanges issued before quantity q's x passes, the bucket sort reused, per-
    quantity accumulators), with the real HIP kernels on 2 / 4 gloo ranks sharing the one GPU: every rank's three tables
    against the oracle."""
    import torch.multiprocessing as mp
    from vpower import synth
    Np = 400000
    mp.spawn(_bench_pipelined_worker, args=(world, _free_port(), N, Np, chunks, per_component, str(tmp_path)), nprocs=world, join=True)
    pos, vel, mass, dens = synth.particles(41, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    refs = [orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, q) for q in ("velocity", "momentum", "energy")]
    for r in range(world):
        tabs = np.load(tmp_path / f"tabs_{r}.npy")
        for tab, ref in zip(tabs, refs):
            assert np.array_equal(tab[:, 3], ref[:, 3])
            assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-5, atol=0)
            assert np.allclose(tab[:, 1], ref[:, 1], rtol=2e-5, atol=0)


def _bench_fields_worker(rank, world, port, N, Np, out_dir):
    for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from vpower import device, synth
        K = device.default_kernels(0)
        pos, vel, mass, dens = synth.particles(43, Np, 1.0)
        comm = device.FieldComm()
        assert (comm.world, comm.rank, comm.field_world, comm.field_rank) == (1, 0, world, rank)
        wl = bench.Workload(K, comm, N, 1.0, "ngp", ("velocity", "momentum", "energy"), "library",
                            K.to_device(pos), K.to_device(vel), K.to_device(dens))
        assert wl.fused and wl.nx == N and not wl.pipe.chunked
        units = device.FieldComm.units(("velocity", "momentum", "energy"))
        base, extra = divmod(len(units), world)
        lo = rank * base + min(rank, extra)
        assert wl.my_units == units[lo: lo + base + (1 if rank < extra else 0)]
        tabs = wl.step()
        tabs2 = wl.step()
        for q in tabs:
            assert np.array_equal(tabs[q][:, 3], tabs2[q][:, 3]) and np.allclose(tabs[q][:, 2], tabs2[q][:, 2], rtol=1e-6)
        np.save(os.path.join(out_dir, f"ftabs_{rank}.npy"), np.stack([tabs[q] for q in ("velocity", "momentum", "energy")]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 128), (4, 128), (3, 64)])
def test_bench_field_parallel_step_on_shared_gpu(tmp_path, world, N):
    """`bench.py --decomposition fields` (and the `alternative` leg of the slab run): the seven scalar fields of the C4 step dealt
    out over 2 / 3 / 4 gloo ranks sharing the one GPU, whole grids, one VPS_FLAG_COMPONENT launch per field; shell sums added and
    shell counts MAX-reduced over the ranks (device.FieldComm).  Every rank's three tables against the oracle."""
    import torch.multiprocessing as mp
    from vpower import synth
    Np = 300000
    mp.spawn(_bench_fields_worker, args=(world, _free_port(), N, Np, str(tmp_path)), nprocs=world, join=True)
    pos, vel, mass, dens = synth.particles(43, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    refs = [orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, q) for q in ("velocity", "momentum", "energy")]
    for r in range(world):
        tabs = np.load(tmp_path / f"ftabs_{r}.npy")
        for tab, ref in zip(tabs, refs):
            assert np.array_equal(tab[:, 3], ref[:, 3])
            assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-5, atol=0)
            assert np.allclose(tab[:, 1], ref[:, 1], rtol=2e-5, atol=0)
