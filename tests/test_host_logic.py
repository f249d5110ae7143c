"""CPU tests of the host side: C-ABI surface, binning tables, containers, script helpers."""
import ctypes
import importlib.util
import os
import re

import numpy as np
import pytest

from helpers import golden, synth
from oracle import vps_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from vpower import _ffi
    hdr = open(os.path.join(ROOT, "include", "vps_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vps_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = _ffi.lib()                      # loads the in-tree .so (no GPU needed to load)
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vps_version() == _ffi.ABI_VERSION == int(re.search(r"#define VPS_ABI_VERSION (\d+)", hdr).group(1))
    # option switches: explicit, process-wide, unknown names refused (the library never reads the environment)
    assert lib.vps_set_option(b"nn_kappa", 1.3) == 0 and lib.vps_get_option(b"nn_kappa", 0.0) == 1.3
    assert lib.vps_set_option(b"nn_kappa", float("nan")) == 0 and lib.vps_get_option(b"nn_kappa", 1.15) == 1.15
    assert lib.vps_set_option(b"no_such_switch", 1.0) != 0
    assert lib.vps_fft_supported(512) == 1 and lib.vps_fft_supported(500) == 1 and lib.vps_fft_supported(768) == 1 and lib.vps_fft_supported(2000) == 1 and lib.vps_fft_supported(600) == 0
    assert lib.vps_fft_workspace_bytes(512, 512) == (512 * 256 * 512 + 512 * 512) * 8
    assert lib.vps_nn_workspace_bytes(10 ** 6, 0, 64 ** 3) > 10 ** 6 * 16 + 4 * 64 ** 3


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from vpower import _ffi, interp
    with pytest.raises(_ffi.VpsError):
        interp.deposit_to_grid(np.ones(4), np.zeros((4, 3)), 16, 1.0)
    # the raw ABI reports the failure as a status code + message, never a crash
    lib = _ffi.lib()
    h = ctypes.c_void_p()
    rc = lib.vps_create(ctypes.byref(h), 0)
    assert rc < 0 and h.value is None
    assert b"HIP" in lib.vps_last_error(None) or b"device" in lib.vps_last_error(None)


def test_library_does_not_read_the_environment():
    """A stray VPS_* variable in a user's job must not change a code path: no getenv in the shipped sources."""
    csrc = os.path.join(ROOT, "large-velocity-power-spectrum_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), fn


def test_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "large-velocity-power-spectrum_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "vps_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, fn


@pytest.mark.parametrize("N", [16, 128, 500, 512, 1000, 1024, 2048, 4096])
@pytest.mark.parametrize("flavour", ["library", "script"])
def test_bin_edges_match_reference_centres(N, flavour):
    from vpower import device
    g = golden("bin_edges.npz")
    kmin, kmax, kres = orc.default_k_range(1.0, N)
    c, e = device.bin_edges(kmin, kmax, kres, flavour)
    assert np.array_equal(c, g[f"{flavour}_centres_{N}"])
    co, eo = (orc.edges_library if flavour == "library" else orc.edges_script)(kmin, kmax, kres)
    assert np.array_equal(e, eo)


@pytest.mark.parametrize("N,L,flavour", [(16, 1.0, "library"), (64, 1.0, "script"), (32, 2.5, "library"),
                                         (128, 1.0, "library"), (1024, 1.0, "script")])
def test_sqrt_thresholds_reproduce_histogram_rule(N, L, flavour):
    """thr[i] <= s < thr[i+1]  <=>  numpy.histogram puts sqrt(s) into bin i."""
    from vpower import device
    kmin, kmax, kres = orc.default_k_range(L, N)
    _, e = device.bin_edges(kmin, kmax, kres, flavour)
    thr = device.sqrt_thresholds(e)
    assert np.all(np.diff(thr) > 0)
    rng = np.random.default_rng(N)
    # all lattice values of s for a sample of modes, plus values hugging every threshold
    ks = device.k_axis(L, N)
    k2 = ks * ks
    i, j, l = (rng.integers(0, N, 200000) for _ in range(3))
    s = (k2[i] + k2[j]) + k2[l]
    hug = np.concatenate([np.nextafter(thr, -np.inf), thr, np.nextafter(thr, np.inf)])
    s = np.concatenate([s, hug, [0.0]])
    b = np.searchsorted(thr, s, side="right") - 1
    ok = (b >= 0) & (b < len(e) - 1)
    mine = np.bincount(b[ok], minlength=len(e) - 1)
    ref, _ = np.histogram(np.sqrt(s), bins=e)
    assert np.array_equal(mine, ref)


def test_threshold_binning_equals_reference_nsample():
    """Half-spectrum + multiplicity + thresholds == the reference's full-spectrum counts."""
    import torch
    from vpower import device
    from oracle_kernels import OracleKernels
    g = golden("nsample.npz")
    for N, key in ((16, "library_16"), (32, "library_32"), (64, "library_64")):
        k = OracleKernels()
        pipe = device.PowerPipeline(N, 1.0, kernels=k, comm=device.SlabComm(enabled=False))
        psum, ns = pipe.accumulate([torch.ones((N, N, N), dtype=torch.float32)])
        tab = pipe.finish(psum, ns)
        assert np.array_equal(tab[:, 3].astype(np.int64), g[key])
    k = OracleKernels()
    pipe = device.PowerPipeline(32, 2.5, kernels=k, comm=device.SlabComm(enabled=False))
    psum, ns = pipe.accumulate([torch.ones((32, 32, 32), dtype=torch.float32)])
    assert np.array_equal(pipe.finish(psum, ns)[:, 3].astype(np.int64), g["library_32_L2p5"])


def test_pipeline_host_math_against_oracle_single_rank():
    import torch
    from vpower import device
    from oracle_kernels import OracleKernels
    rng = np.random.default_rng(3)
    N, L = 32, 1.0
    f = [rng.standard_normal((N, N, N)).astype(np.float32) for _ in range(3)]
    pipe = device.PowerPipeline(N, L, kernels=OracleKernels(), comm=device.SlabComm(enabled=False))
    tab = pipe.spectrum([torch.from_numpy(x) for x in f])
    P = orc.vector_power(*[x.astype(np.float64) for x in f], L, N)
    ref = orc.spectrum_table(P, L, N, "library")
    assert np.array_equal(tab[:, 3], ref[:, 3])
    assert np.array_equal(tab[:, 0], ref[:, 0])
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-5)
    assert np.allclose(tab[:, 1], ref[:, 1], rtol=1e-5)


def test_spectrum_containers_match_reference():
    from vpower import spctrm
    g = golden("spectrum_container.npz")
    a, b = spctrm.PowerSpectrum(g["a"].copy()), spctrm.PowerSpectrum(g["b"].copy())
    assert a.energy() == float(g["energy"]) and a.kres() == float(g["kres"])
    c = a.copy()
    c.add(b)
    assert np.array_equal(c.data(), g["added"])
    assert spctrm.relative_diff(a.copy(), b.copy(), "max") == float(g["reldiff_max"])
    assert spctrm.relative_diff(a.copy(), b.copy(), "mean") == float(g["reldiff_mean"])
    assert np.array_equal(spctrm.init_beta_space(2), g["beta_space_2"])
    with pytest.raises(Exception):
        a.add(spctrm.PowerSpectrum(g["a"][:3].copy()))
    with pytest.raises(Exception):
        spctrm.relative_diff(a, spctrm.PowerSpectrum(g["a"][:3].copy()))
    lst = spctrm.SpectrumList([spctrm.PowerSpectrum(g["a"].copy(), m=2, beta=np.array([0, 0, 0])),
                               spctrm.PowerSpectrum(g["b"].copy(), m=2, beta=np.array([1, 0, 0]))])
    comb = lst.combine_all()
    assert np.array_equal(comb.Psum, g["a"][:, 2] + g["b"][:, 2])
    assert np.array_equal(lst[np.array([1, 0, 0])].Psum, g["b"][:, 2])
    e = spctrm.empty_spectrum_like(a)
    assert np.all(e.P == 0) and np.array_equal(e.k, a.k)


def test_spectrum_roundtrip_files(tmp_path):
    from vpower import spctrm
    g = golden("spectrum_container.npz")
    a = spctrm.PowerSpectrum(g["a"].copy())
    a.save(str(tmp_path))
    b = spctrm.PowerSpectrum.load(str(tmp_path))
    assert np.array_equal(a.data(), b.data())
    a.savetxt(str(tmp_path / "Pk.txt"))
    assert np.allclose(spctrm.PowerSpectrum.loadtxt(str(tmp_path / "Pk.txt")).data(), a.data())


def _load_script():
    path = os.path.join(ROOT, "large-velocity-power-spectrum_amd", "scripts", "parallel_optimized.py")
    spec = importlib.util.spec_from_file_location("vps_script", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_script_surface_and_planner():
    s = _load_script()
    for name in ("planner", "FFTW_power", "FFTW_vector_power", "pair_power", "hist_sample", "main"):
        assert callable(getattr(s, name))
    g = golden("planner.npz")
    for inp, out in zip(g["inputs"], g["outputs"]):
        assert tuple(float(x) for x in s.planner(*[int(v) for v in inp])) == tuple(out)
    with pytest.raises(AssertionError):
        s.planner(128, 1, 128, 5)          # not a cube (parallel_optimized.py:72)
    p = s.build_parser().parse_args(["-i", "x", "-o", "y", "-N", "64", "-M", "32", "-l", "2", "-b", "10", "-f"])
    assert (p.input, p.output, p.ntot, p.maxnbox, p.ltot, p.nbuffer, p.f) == ("x", "y", 64, 32, 2, 10, True)


def test_synth_is_deterministic_and_preprocessed():
    from vpower import synth as sy
    a = sy.particles(5, 1000)
    b = sy.particles(5, 1000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    pos, vel, mass, dens = a
    assert pos.dtype == np.float32 and pos.min(axis=0).max() == 0.0
    assert abs(np.sum(mass * vel[:, 0]) / np.sum(mass)) < 1e-6
    assert sy.CONFIGS["C2"][:2] == (512, 10_000_000)


def test_lattice_recovery_from_grid_coords():
    from vpower import interp
    q = interp.make_grid_coords(1.0, 8)
    assert np.array_equal(q, orc.make_grid_coords(1.0, 8))
    ax, ay, az = interp._axes_of_lattice(q, 8)
    assert np.array_equal(ax, orc.lattice_axes_library(1.0, 8)) and np.array_equal(ay, ax) and np.array_equal(az, ax)
    with pytest.raises(Exception):
        interp._axes_of_lattice(np.random.default_rng(1).permutation(q), 8)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher spawns N ranks (torch.distributed.run child,
    gloo here) before importing torch, relays rank 0's line, and fails when a rank fails."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 1.0
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    # under a launcher (WORLD_SIZE set) it must not spawn again, and a wrong --gpus is an error
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode == 2


@pytest.mark.parametrize("n", [200_000, 1_000_000])
def test_residency_fingerprint_sees_sparse_in_place_edits(n):
    """GasParticles keeps device copies keyed by a hash of EVERY byte of the host array (vpower/interp.py `_fingerprint`): the
    edits a 257-value strided sample of an (n, 3) array misses -- a whole column, a short row range, one value -- all change it
    (the reference edits these arrays in place, interp.py:400-402)."""
    from vpower import interp
    rng = np.random.default_rng(11)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    rho = np.exp(rng.standard_normal(n)).astype(np.float32)
    f0 = interp._fingerprint(v)
    assert interp._fingerprint(v) == f0 and interp._fingerprint(v.copy()) == f0
    for edit in (lambda a: a.__setitem__((slice(None), 1), a[:, 1] * 2), lambda a: a.__setitem__((slice(None), 2), a[:, 2] + 1),
                 lambda a: a.__setitem__(slice(5, 50), 0.0), lambda a: a.__setitem__((n - 1, 0), 3.0)):
        w = v.copy()
        edit(w)
        assert interp._fingerprint(w) != f0
    r0 = interp._fingerprint(rho)
    rho[n // 3] *= 2
    assert interp._fingerprint(rho) != r0
    # a non-contiguous view hashes what it shows
    assert interp._fingerprint(v[:, 1]) == interp._fingerprint(np.ascontiguousarray(v[:, 1]))


@pytest.mark.parametrize("bulk,shift", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_load_snapshot_npz_branch_matches_oracle_preprocessing(tmp_path, bulk, shift, dtype):
    """load_snapshot (reference interp.py:84-131) through its `.npz` branch -- h5py is not on this host, the four GIZMO keys
    are the same -- with every combination of the two preprocessing flags, against the oracle's restatement of
    interp.py:169-182 / script:280-291.  Without a GPU the numpy route runs: the same float arithmetic, so equality is exact."""
    from vpower import interp
    rng = np.random.default_rng(5)
    n, L = 5000, 3.0
    c = (0.4 + rng.random((n, 3)) * (L - 0.4)).astype(dtype)
    m = (0.5 + rng.random(n)).astype(dtype)
    d = np.exp(0.3 * rng.standard_normal(n)).astype(dtype)
    v = (rng.standard_normal((n, 3)) + np.array([2.0, -1.0, 0.5])).astype(dtype)
    f = tmp_path / "snap.npz"
    np.savez(f, Coordinates=c, Masses=m, Density=d, Velocities=v)
    gp = interp.load_snapshot(str(f), Lbox=L, remove_bulk_velocity=bulk, shift_to_origin=shift)
    assert isinstance(gp, interp.GasParticles) and len(gp) == n and gp.Lbox == L
    ref_pos, ref_vel = orc.preprocess_script(c, m, v, remove_bulk=bulk)
    if not shift:
        ref_pos = c
    import torch
    exact = not torch.cuda.is_available()      # on a GPU box the float32 arrays go through vps_preprocess (float64 bulk sums)
    if exact or dtype == np.float64:
        assert np.array_equal(gp.pos, ref_pos) and np.array_equal(gp.v, ref_vel)
    else:
        assert np.array_equal(gp.pos, ref_pos) and np.allclose(gp.v, ref_vel, rtol=0, atol=2e-6)
    assert np.array_equal(gp.mass, m) and np.array_equal(gp.density, d)
    assert gp.v is gp.velocity or np.shares_memory(gp.v, gp.velocity)     # one array under both names, as in the reference
    assert np.allclose(gp.r, ((3 * m / d) / (4 * np.pi)) ** (1 / 3))
    with pytest.raises(Exception):
        interp.load_snapshot(str(tmp_path / "missing.hdf5"))


def test_exchange_group_is_closed_on_every_error_path(tmp_path):
    """csrc/comm_group.h -- one chunk of the slab exchange (what replaces the reference's comm.allgather calls,
    scripts/parallel_optimized.py:365-368) as ONE RCCL group -- driven by a recording stand-in for the RCCL table on the host:
    the n-th send / receive / GroupStart / GroupEnd fails, the group is closed on every path, nothing is issued after the first
    failure and the first failure is what the caller sees (tests/native/test_comm_group.cpp)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "test_comm_group")
    src = os.path.join(ROOT, "tests", "native", "test_comm_group.cpp")
    inc = os.path.join(ROOT, "large-velocity-power-spectrum_amd", "csrc")
    subprocess.run([cxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-I", inc, src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


# ------------------------------------------------------------ the NN kernel's hand-made LDS pipeline ----
def test_asm_pipeline_checker_accepts_a_clean_step_and_rejects_touched_registers(tmp_path):
    """vpower/_asmcheck.py on synthetic listings: the request/wait pair with arithmetic in between passes; a copy of a
    destination register, a spill of one, or control flow before the wait are reported."""
    from vpower import _asmcheck
    head = "_ZN12_GLOBAL__N_116nn_column_kernelIfLi4EEEvPKT_NS_15NnScatterParamsE:\n"
    req = "\t;;#ASMSTART\n\tds_read_b128 v[6:9], v5\n\tds_read_u16 v25, v23 offset:16\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
    work = "\tv_sub_f32_e32 v3, s36, v4\n\tv_fma_f32 v3, v3, v3, v2\n\tv_min_u32_e32 v107, v107, v3\n"

    def run(body):
        f = tmp_path / "k.s"
        f.write_text(head + body + "\ts_endpgm\n")
        return _asmcheck.check(str(f))

    assert run(req + work + wait) == (1, [])
    n, bad = run(req + work + "\tv_mov_b32_e32 v30, v7\n" + wait)                   # copy of a destination register
    assert n == 1 and len(bad) == 1 and "touches" in bad[0][1]
    n, bad = run(req + "\tscratch_store_dwordx4 off, v[6:9], off offset:16\n" + work + wait)   # spill
    assert len(bad) == 1
    n, bad = run(req + work + "\tv_mov_b32_e32 v25, v1\n" + wait)                   # the slot register overwritten
    assert len(bad) == 1
    n, bad = run(req + work + "\ts_cbranch_scc1 .LBB0_1\n" + wait)
    assert len(bad) == 1 and "control flow" in bad[0][1]
    # a listing without the kernel's requests is not a pass either (the entry point reports n == 0)
    assert run(work)[0] == 0


def test_built_library_passed_the_pipeline_check():
    """The build runs the check on the device assembly of nn.hip and leaves its verdict next to the objects: it must be
    about the sources in the tree and hold no violation (vpower/_ffi.py:_check_nn_pipeline refuses to link otherwise)."""
    import json
    from vpower import _ffi
    _ffi.build()
    rep = json.load(open(_ffi.NN_CHECK_REPORT))
    assert rep["violations"] == 0 and rep["requests"] >= 75, rep       # 25 classes x 3 step bodies per instantiation
    assert rep["source_sha16"] == _ffi._source_sha(os.path.join(_ffi.CSRC, "nn.hip"))
