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
    # workspace: two chunk slots, whatever the chunk count (it used to be every chunk of the field, send and receive)
    ncomp = 3
    per_chunk = [K.lib.vps_fft_y_chunk_elems(N, N, 1, 4, c) for c in range(4)]
    assert K.lib.vps_spectrum_zimages_workspace_bytes(N, N, 1, 4, ncomp) == 2 * 2 * ncomp * max(per_chunk) * 8
    assert K.lib.vps_spectrum_zimages_workspace_bytes(N, N, 1, 1, ncomp) == 2 * 1 * ncomp * K.lib.vps_fft_y_chunk_elems(N, N, 1, 1, 0) * 8
    # exchange intervals on the library's communication stream, and how long the context's stream waited for them
    z = K.deposit_fft_z(d[0], d[1], d[2], N, 1.0, 0, N, device.VELOCITY)
    K.timing(True)
    pipe.accumulate_zimages([z[i] for i in range(3)])
    tim = K.timing_get()
    K.timing(False)
    assert tim["exchange"][0] == 4 and tim["exchange_wait"][0] == 4 and tim["exchange"][1] > 0 and tim["exchange_wait"][1] >= 0
    # error path: the 5th ncclSend from now on (second chunk, second component) fails inside an open group -- the call
    # reports it, the group is closed (csrc/comm_group.h), both streams are drained, and the NEXT call works and is right
    from vpower import _ffi
    _ffi.set_option("comm_fail_send", 5)
    try:
        failed = False
        try:
            pipe.accumulate_zimages([z[i] for i in range(3)])
        except Exception as e:
            failed = "ncclSend" in str(e) and "group closed" in str(e)
        assert failed, "the injected send failure must surface as an error"
    finally:
        _ffi.set_option("comm_fail_send", None)
    tabs.append(pipe.finish(*pipe.accumulate_zimages([z[i] for i in range(3)])))
    np.save(os.path.join(out_dir, "tab_lib.npy"), np.stack(tabs))
    K.comm_destroy()


def test_library_rccl_exchange_single_rank(tmp_path):
    """The exchange behind the C ABI (vps_comm_create / vps_spectrum_zimages / vps_allreduce_shells: RCCL loaded by the
    library, grouped ncclSend / ncclRecv per kz chunk on its own stream, events against the context's stream, ncclAllReduce of
    the float64 / uint64 accumulators) on the one GPU of the test box: a one-rank communicator made from a unique id, no
    torch.distributed involved.  Velocity (three components per launch) and energy against the oracle, twice each; the
    two-slot workspace formula; the exchange timing kinds; and an injected ncclSend failure in the middle of a group (error
    reported, group closed, the next call correct)."""
    import torch.multiprocessing as mp
    from vpower import synth
    N, Np = 128, 200000
    mp.spawn(_library_rccl_worker, args=(_free_port(), N, Np, str(tmp_path)), nprocs=1, join=True)
    pos, vel, mass, dens = synth.particles(37, Np, 1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    tabs = np.load(tmp_path / "tab_lib.npy")
    for i, q in enumerate(("velocity", "velocity", "energy", "energy", "velocity")):     # (the last one: after the injected failure)
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


@pytest.mark.parametrize("world,transport", [(2, "torch"), (4, "torch"), (1, "library")])
def test_bench_main_several_ranks_rehearsal(tmp_path, world, transport):
    """`python bench.py --gpus N` as the driver starts it (bench.main's own self-spawn: a torch.distributed.run child), at a
    rehearsal grid on the one GPU of the test box: 2 and 4 gloo ranks sharing the device (torch transport: all_to_all_single
    staged through the host), and the library transport (RCCL inside libvps_hip.so; RCCL cannot put two ranks on one device,
    so one rank with forced collectives).  The line must report the SLAB decomposition as `value` -- what BASELINE.json's C4
    names -- with, inside `config`, the transport, chunk count, row fraction, both decompositions' step times and the
    exchange diagnosis (kernel_ms, exchange_ms, exposed_exchange_ms: max over ranks); the field-parallel split under
    `alternative`, its tables equal to the slab run's; parity against the oracle asserted inside."""
    import json
    import subprocess
    env = dict(os.environ, VPS_BENCH_GRID="128", VPS_BENCH_PARTICLES="300000", VPS_A2A_CHUNKS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if transport == "library":
        env.update(VPS_BENCH_TRANSPORT="library", VPS_FORCE_COLLECTIVES="1")
    else:
        env.update(VPS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-other-configs", "--profile-steps", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "strong" and "REHEARSAL" in cfg["deviation"]
    assert cfg["decomposition"] == "slab" and cfg["chunks"] == 4 and cfg["chunks_in_flight"] == 2
    assert ("gloo" in cfg["transport"]) if transport == "torch" else ("libvps_hip.so" in cfg["transport"])
    assert abs(cfg["ms_per_step_slab"] - d["ms_per_step"]) < 1e-9
    ex = cfg["slab_exchange"]
    assert ex == d["slab_exchange"]
    for key in ("kernel_ms", "exchange_ms", "exposed_exchange_ms"):
        assert ex[key] >= 0.0
    assert ex["kernel_ms"] > 0 and ex["exchange_ms"] > 0
    assert d["full_size_check"]["nsample_exact"]
    if world > 1:
        assert 0 < cfg["exchange_row_fraction"] <= 1.0 and "max over the %d ranks" % world in ex["over"]
        assert d["parity"]["nsample_equal"] and d["parity"]["psum_max_rel"] < 2e-5
        alt = d["alternative"]
        assert alt["decomposition"] == "fields" and alt["vs_own_tables"]["nsample_equal"] and alt["vs_own_tables"]["psum_max_rel"] < 2e-5
        assert abs(cfg["ms_per_step_fields"] - alt["ms_per_step"]) < 1e-9
