"""Parity of the HIP path against the oracle / the reference's golden fixtures.
Run on an MI355X:  python -m pytest tests -m gpu -x -q
Every test calls through the C ABI (vpower._ffi -> libvps_hip.so)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from helpers import golden, synth
from oracle import vps_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Stated floating-point tolerances (DESIGN.md section "Parity"):
PSUM_RTOL = 2e-5      # shell sums: float32 FFT + float32 grids vs the float64 oracle
FFT_RTOL = 3e-6       # per-mode spectrum error relative to the rms mode amplitude


@pytest.fixture(scope="module")
def K():
    from vpower import device
    return device.default_kernels()


# ------------------------------------------------------------------ stage A1 ----
def _edge_positions(N, L, dtype):
    rng = np.random.default_rng(42)
    lc = L / N
    base = np.concatenate([
        rng.random(20000) * L,
        np.arange(0, N + 1) * lc,                                   # exact cell faces
        np.nextafter(np.arange(0, N + 1) * lc, -np.inf),
        np.nextafter(np.arange(0, N + 1) * lc, np.inf),
        rng.random(2000) * 3 * L - L,                                # outside the box, negative
        [0.0, -0.0, L, 2 * L, -L, 1e-30, -1e-30, 7.3 * L, 1e10, -1e10, 123456.7 * L, 3e7 * L, -2.9e8 * lc],
    ])
    base = base.astype(dtype)
    n = (len(base) // 3) * 3
    return base[:n].reshape(-1, 3)


@pytest.mark.parametrize("N,L", [(16, 1.0), (128, 1.0), (512, 1.0), (500, 1.0), (1000, 1.0), (96, 2.5), (2048, 1.0)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cell_index_bit_exact(K, N, L, dtype):
    pos = _edge_positions(N, L, dtype)
    got = K.cell_index(K.to_device(pos), N, L).cpu().numpy()
    ref = orc.cell_index(pos, N, L)
    assert np.array_equal(got, ref.astype(np.int32))


@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_cell_index_golden(K, tag):
    g = golden(f"library_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), float(g["L"]), int(g["seed"])
    pos = synth(seed, Np, L)[0]
    assert np.array_equal(K.cell_index(K.to_device(pos), N, L).cpu().numpy(), g["cell_f32"])
    pos64 = pos.astype(np.float64) * 1.000001
    assert np.array_equal(K.cell_index(K.to_device(pos64), N, L).cpu().numpy(), g["cell_f64"])


@pytest.mark.parametrize("C", [1, 3, 4])
def test_deposit_matches_oracle(K, C):
    rng = np.random.default_rng(C)
    N, L, Np = 32, 1.0, 50000
    pos = (rng.random((Np, 3)) * L).astype(np.float32)
    f = rng.standard_normal((Np, C)).astype(np.float32)
    grid = K.deposit(K.to_device(pos), K.to_device(f), N, L, 0, N).cpu().numpy()     # [C,N,N,N]
    ref = orc.deposit_to_grid(f.astype(np.float64), pos, N, L)                       # [N,N,N,C]
    assert np.allclose(grid.transpose(1, 2, 3, 0), ref, rtol=1e-5, atol=1e-5)
    # slabs tile the full grid
    parts = [K.deposit(K.to_device(pos), K.to_device(f), N, L, x0, 8).cpu().numpy() for x0 in range(0, N, 8)]
    assert np.allclose(np.concatenate(parts, axis=1), grid, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("opts", [{"sort_atomic": 1}, {"sort_staged": 0}, {"sort_groups": 7}, {}])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_deposit_sort_variants_are_exact_on_integers(K, opts, dtype):
    """Every bucket-sort flavour (two-level staged / two-level direct / one atomic per particle) feeds the same
    records to the accumulation: small-integer payloads make the float32 sums exact, so results must be EQUAL."""
    import contextlib
    from vpower import _ffi
    with contextlib.ExitStack() as stack:
        for k, v in opts.items():
            stack.enter_context(_ffi.option(k, v))
        _deposit_sort_variant_body(K, dtype)


def _deposit_sort_variant_body(K, dtype):
    rng = np.random.default_rng(31)
    N, L, Np = 96, 2.0, 300000                       # N not a power of two: bricks with a ragged edge
    pos = (rng.random((Np, 3)) * L).astype(dtype)
    pos[: Np // 4] = pos[: Np // 4] * 0.05 + 0.9     # a clump: long buckets
    pos[-1000:] = L * 3.5                            # wraps around (periodic)
    f = rng.integers(1, 4, (Np, 3)).astype(np.float32)
    ref = orc.deposit_to_grid(f.astype(np.float64), pos, N, L)
    grid = K.deposit(K.to_device(pos), K.to_device(f), N, L, 0, N).cpu().numpy()
    assert np.array_equal(grid.transpose(1, 2, 3, 0), ref)
    part = K.deposit(K.to_device(pos), K.to_device(f), N, L, 40, 24).cpu().numpy()   # a slab: most particles outside
    assert np.array_equal(part, grid[:, 40:64])


def test_deposit_api_and_integer_conservation():
    from vpower import interp
    rng = np.random.default_rng(9)
    N, L, Np = 64, 1.0, 200000
    pos = rng.random((Np, 3)) * L                 # float64 positions
    f = rng.integers(1, 5, Np).astype(np.float64)  # small integers: float32 sums are exact
    grid = interp.deposit_to_grid(f, pos, N, L)
    ref = orc.deposit_to_grid(f, pos, N, L)
    assert grid.dtype == np.float64 and grid.shape == (N, N, N)
    assert np.array_equal(grid, ref)               # bit exact in the integer regime
    f5 = rng.integers(0, 3, (Np, 5)).astype(np.float64)   # 5 channels: split 4+1 internally
    assert np.array_equal(interp.deposit_to_grid(f5, pos, N, L), orc.deposit_to_grid(f5, pos, N, L))
    # empty input
    assert np.all(interp.deposit_to_grid(np.zeros(0), np.zeros((0, 3)), N, L) == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_preprocess_matches_reference_rules(K, dtype):
    rng = np.random.default_rng(21)
    Np = 300001
    pos = (rng.random((Np, 3)) * 3.0 + 0.25).astype(dtype)
    vel = (rng.standard_normal((Np, 3)) + 0.3).astype(np.float32)
    mass = np.exp(rng.standard_normal(Np)).astype(np.float32)
    dp, dv = K.to_device(pos), K.to_device(vel)
    mn, bv = K.preprocess(dp, dv, K.to_device(mass))
    rp, rv = orc.preprocess_script(pos, mass, vel)
    assert np.array_equal(dp.cpu().numpy(), rp)                       # min and subtraction are exact
    assert np.array_equal(mn, pos.min(axis=0).astype(np.float64))
    assert np.allclose(dv.cpu().numpy(), rv, rtol=0, atol=2e-6)        # float32 pairwise sum vs float64
    ref_bulk = (mass[:, None].astype(np.float64) * vel).sum(0) / mass.astype(np.float64).sum()
    assert np.allclose(bv, ref_bulk, rtol=1e-6)


# ------------------------------------------------------------------ stage A2 ----
@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_nn_index_golden_library_lattice(tag):
    from vpower import interp
    g = golden(f"library_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), float(g["L"]), int(g["seed"])
    pos = synth(seed, Np, L)[0].astype(np.float64)
    ax = orc.lattice_axes_library(L, N)
    idx = interp.nn_index(pos, (ax, ax, ax))
    assert np.array_equal(idx.ravel(), g["nn_idx"])


@pytest.mark.parametrize("Np,N,dtype", [(1, 8, np.float32), (7, 16, np.float64), (3000, 24, np.float32),
                                        (200000, 40, np.float32), (200000, 33, np.float64)])
def test_nn_index_matches_oracle(Np, N, dtype):
    from vpower import interp
    rng = np.random.default_rng(Np + N)
    pos = rng.random((Np, 3)).astype(dtype)
    pos[: Np // 10] *= 0.05                        # a dense clump: uneven cell occupancy
    ax = np.linspace(-0.1, 1.1, N)                 # queries outside the particle box too
    ay = np.linspace(0.0, 1.0, N)
    az = (np.arange(N) / N).astype(np.float32).astype(np.float64)
    idx = interp.nn_index(pos, (ax, ay, az))
    ref = orc.exact_nn_lattice(pos, ax, ay, az)
    assert np.array_equal(idx.ravel(), ref)


@pytest.mark.parametrize("kernel", ["column", "scatter"])
def test_nn_particles_far_denser_than_the_lattice(kernel):
    """~100 particles per lattice cell: the cells of the particle list are a quarter of a lattice step wide, and a 16 x 16 tile
    alone covers more cell columns than a workgroup can stage at ANY search radius.  The tile kernels then give up (their
    radius-shrinking loop has an exit since round 4 -- it used to spin forever here) and leave the tile to the exact
    fallback: indices bit exact all the same, for the column-register kernel and (forced) the scatter kernel."""
    from vpower import interp, _ffi
    Np, N = 400000, 16
    rng = np.random.default_rng(99)
    pos = rng.random((Np, 3)).astype(np.float32)
    ax = orc.lattice_axes_library(1.0, N)
    _ffi.set_option("nn_column", 1 if kernel == "column" else 0)
    try:
        idx = interp.nn_index(pos, (ax, ax, ax))
    finally:
        _ffi.set_option("nn_column", None)
    assert np.array_equal(idx.ravel(), orc.exact_nn_lattice(pos, ax, ax, ax))


def test_nn_ties_lowest_index_and_duplicates():
    from vpower import interp
    pos = np.array([[0.25, 0.5, 0.5], [0.75, 0.5, 0.5], [0.75, 0.5, 0.5], [0.25, 0.5, 0.5]], dtype=np.float64)
    ax = np.array([0.5, 0.2, 0.8])
    idx = interp.nn_index(pos, (ax, np.array([0.5]), np.array([0.5])))
    assert idx.ravel().tolist() == [0, 0, 1]       # equidistant -> lowest index; duplicates -> lowest


def test_ann_interpolate_api():
    from vpower import interp
    rng = np.random.default_rng(2)
    N, Np = 12, 5000
    pos = rng.random((Np, 3))
    q = interp.make_grid_coords(1.0, N)
    f4 = rng.standard_normal((Np, 4))
    out = interp.ann_interpolate(pos, q, f4, N, 0.0)
    ax = orc.lattice_axes_library(1.0, N)
    ref, idx = orc.ann_interpolate(pos, (ax, ax, ax), f4, N)
    assert out.shape == (N, N, N, 4) and np.array_equal(out, ref)     # float64 payload: exact gather
    f1 = rng.standard_normal(Np).astype(np.float32)
    assert np.array_equal(interp.ann_interpolate(pos, q, f1, N, 0.0), f1[idx].reshape(N, N, N))
    with pytest.raises(Exception):
        interp.ann_interpolate(pos, q, f4, N, 0.1)
    with pytest.raises(Exception):
        interp.ann_interpolate(pos, q, np.zeros((Np, 2, 2)), N, 0.0)


# ------------------------------------------------------------------ stage B ----
@pytest.mark.parametrize("N", [16, 32, 64, 128, 256, 250, 96, 192])
def test_rfft3_matches_numpy(K, N):
    rng = np.random.default_rng(N)
    f = rng.standard_normal((N, N, N)).astype(np.float32)
    got = K.rfft3(K.to_device(f), N).cpu().numpy()                  # [kz,ky,kx]
    ref = np.fft.rfftn(f.astype(np.float64)).transpose(2, 1, 0)     # [kz<=N/2, ky, kx]
    scale = np.sqrt(np.mean(np.abs(ref) ** 2))
    assert np.max(np.abs(got - ref)) / scale < FFT_RTOL


def test_radix5_grid_sizes_against_the_oracle(K):
    """N = 250 (the reference's own benchmark size, buffer_test.log): lines of 125 packed-real and 250 complex
    points through the radix-5 / radix-10 plans -- NGP deposit on a non-power-of-two grid, spectrum with the
    library binning: Nsample bit exact, Psum within the float32 bar.  N = 500 / 1000 and the 3 * 2^a sizes 384 / 768
    (radix 3 / 6 / 12): plane-wave known answer."""
    from vpower import device, interp
    N, L, Np = 250, 1.0, 400000
    pos, vel, mass, dens = synth(250, Np, L)
    gp = interp.GasParticles(pos, mass, dens, vel, L)
    sp = gp.deposit_to_field(N).spctrm("velocity")
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, L), L / N, zero_empty=True)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, L / N, "velocity")
    assert len(sp.k) == 125 and np.array_equal(sp.Nsample, ref[:, 3])
    assert np.allclose(sp.Psum, ref[:, 2], rtol=PSUM_RTOL, atol=0)
    for n, (a, b, c) in ((500, (3, 41, 7)), (1000, (11, 2, 333)), (384, (5, 150, 9)), (768, (300, 1, 77))):
        x = np.arange(n, dtype=np.float64)
        ph = 2 * np.pi * ((a * x)[:, None, None] + (b * x)[None, :, None] + (c * x)[None, None, :]) / n
        f = K.to_device(np.cos(ph).astype(np.float32))
        del ph
        mag = K.rfft3(f, n).abs()
        peak = float(mag[c, b, a])
        assert abs(peak - n ** 3 / 2) < 2e-3 * n ** 3
        mag[c, b, a] = 0
        assert float(mag.max()) < 2e-4 * n ** 3
        del f, mag
        torch.cuda.empty_cache()


def test_rfft3_plane_wave_known_answer(K):
    N = 64
    x = np.arange(N)
    f = np.cos(2 * np.pi * (3 * x[:, None, None] + 5 * x[None, :, None] + 7 * x[None, None, :]) / N).astype(np.float32)
    got = K.rfft3(K.to_device(f), N).cpu().numpy()
    mag = np.abs(got)
    assert abs(mag[7, 5, 3] - N ** 3 / 2) < 1e-3 * N ** 3
    mag[7, 5, 3] = 0
    assert mag.max() < 1e-3 * N ** 3 / 2


@pytest.mark.parametrize("N", [16, 64])
def test_vector_and_scalar_power_api(N):
    from vpower import interp
    rng = np.random.default_rng(7)
    L = 3.0
    fx, fy, fz = (rng.standard_normal((N, N, N)) for _ in range(3))
    P = interp._vector_power(fx, fy, fz, L, N)
    ref = orc.vector_power(fx, fy, fz, L, N)
    assert P.shape == (N, N, N) and np.allclose(P, ref, rtol=2e-4, atol=1e-5 * ref.mean())
    Ps = interp._scalar_power(fx, L, N)
    assert np.allclose(Ps, orc.scalar_power(fx, L, N), rtol=2e-4, atol=1e-5 * ref.mean())
    # Parseval normalisation stated at interp.py:1377-1378
    assert abs(np.sum(P) * (2 * np.pi / L) ** 3 - 0.5 * np.mean(fx ** 2 + fy ** 2 + fz ** 2)) < 1e-5
    if N == 16:
        g = golden("fft_power.npz")
        r = np.random.default_rng(int(g["seed"]))
        gx, gy, gz = (r.standard_normal((N, N, N)) for _ in range(3))
        assert np.allclose(interp._vector_power(gx, gy, gz, L, N), g["vector"], rtol=2e-4, atol=1e-5 * g["vector"].mean())


# ------------------------------------------------------------------ stage C ----
@pytest.mark.parametrize("N,L,flavour", [(16, 1.0, "library"), (32, 2.5, "library"), (64, 1.0, "script"),
                                         (128, 1.0, "library"), (256, 1.0, "script")])
def test_fused_binning_matches_oracle(K, N, L, flavour):
    from vpower import device
    rng = np.random.default_rng(N)
    fields = [rng.standard_normal((N, N, N)).astype(np.float32) for _ in range(3)]
    pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False), flavour=flavour)
    tab = pipe.spectrum([K.to_device(f) for f in fields])
    ref = orc.spectrum_table(orc.vector_power(*[f.astype(np.float64) for f in fields], L, N), L, N, flavour)
    assert np.array_equal(tab[:, 0], ref[:, 0])
    assert np.array_equal(tab[:, 3], ref[:, 3])                       # Nsample: bit exact
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)
    ok = ref[:, 3] > 0
    assert np.allclose(tab[ok, 1], ref[ok, 1], rtol=PSUM_RTOL, atol=0)


@pytest.mark.parametrize("N,L", [(32, 1.0), (64, 2.5), (256, 1.0)])
def test_general_binning_path_matches_fast_path(K, N, L):
    """The branch-free mirrored-kx path and the general shell walk give the same counts
    (bit for bit) and sums; custom narrow bins exercise several shells per kx step."""
    from vpower import device, _ffi
    rng = np.random.default_rng(N + 1)
    fields = [K.to_device(rng.standard_normal((N, N, N)).astype(np.float32)) for _ in range(2)]
    out = {}
    for kres in (None, 0.37 * 2 * np.pi / L):
        for general in (False, True, "nopair"):
            _ffi.set_option("no_fast_binning", 1 if general is True else None)
            _ffi.set_option("no_pair_binning", 1 if general == "nopair" else None)
            try:
                pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False), kres=kres)
                out[general] = pipe.finish(*pipe.accumulate(fields))
            finally:
                _ffi.set_option("no_fast_binning", None)
                _ffi.set_option("no_pair_binning", None)
        for other in (True, "nopair"):
            assert np.array_equal(out[False][:, 3], out[other][:, 3])
            assert np.allclose(out[False][:, 2], out[other][:, 2], rtol=1e-6, atol=0)
        ref = orc.spectrum_table(orc.vector_power(fields[0].cpu().numpy().astype(np.float64),
                                                  fields[1].cpu().numpy().astype(np.float64),
                                                  np.zeros((N, N, N)), L, N), L, N, "library", kres=kres)
        ref[:, 1] /= np.where(ref[:, 0] > 0, 4 * np.pi * ref[:, 0] ** 2, 1)
        assert np.array_equal(out[False][:, 3], ref[:, 3])
        assert np.allclose(out[False][:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)


@pytest.mark.parametrize("N,L", [(32, 1.0), (64, 2.5), (256, 1.0), (512, 1.0)])
def test_integer_shells_equal_float64_shells(K, N, L):
    """Mirrored-kx binning with INTEGER shells (fft.hip FASTMODE 2: ix^2 + iy^2 + iz^2 against ceil(thr / k2[1])) -- taken
    wherever vps_set_binning finds no threshold on an integer multiple of k2[1], which covers both reference flavours (edges at
    half-integer multiples of 2 pi / L, interp.py:1472-1473 / script:178-180) -- decides every mode exactly like the float64
    comparison of k^2 sums: same counts bit for bit as the float64 path AND as the oracle's np.histogram.  Edges ON integer
    multiples (modes sitting exactly on shell edges: float64 rounding decides) must fall back to the float64 path."""
    from vpower import device, _ffi
    rng = np.random.default_rng(N + 7)
    fields = [K.to_device(rng.standard_normal((N, N, N)).astype(np.float32)) for _ in range(3)]
    f64 = [f.cpu().numpy().astype(np.float64) for f in fields]
    P = orc.vector_power(*f64, L, N)
    kf = 2 * np.pi / L
    cases = [("library", None, None, None, 2), ("script", None, None, None, 2), ("library", None, None, 0.37 * kf, 2),
             ("library", 1.5 * kf, None, kf, 1),        # edges at 1, 2, 3 ... kf: thresholds ON integers
             ("library", 2.0 * kf, 0.4 * np.pi * N / L, 2.0 * kf, 1)]
    for flavour, kmin, kmax, kres, want_mode in cases:
        tabs = {}
        for no_int in (None, 1):
            _ffi.set_option("no_int_binning", no_int)
            try:
                pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False), flavour=flavour,
                                            kmin=kmin, kmax=kmax, kres=kres)
                pipe.prepare()
                mode = K.binning_mode()
                assert mode == (1 if no_int else want_mode), (flavour, kmin, kres, mode)
                tabs[no_int] = pipe.finish(*pipe.accumulate(fields))
            finally:
                _ffi.set_option("no_int_binning", None)
        ref = orc.spectrum_table(P, L, N, flavour, kmin=kmin, kmax=kmax, kres=kres)
        for t in tabs.values():
            assert np.array_equal(t[:, 3], ref[:, 3])                        # Nsample: bit exact, both paths
            assert np.allclose(t[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)
        assert np.array_equal(tabs[None][:, 3], tabs[1][:, 3])
        assert np.allclose(tabs[None][:, 2], tabs[1][:, 2], rtol=1e-6, atol=0)


@pytest.mark.parametrize("N", [16, 32, 64, 128])
def test_nsample_golden(K, N):
    from vpower import device
    g = golden("nsample.npz")
    for flavour in ("library", "script"):
        pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False), flavour=flavour)
        psum, ns = pipe.accumulate([K.zeros((N, N, N), torch.float32)])
        assert np.array_equal(ns.cpu().numpy(), g[f"{flavour}_{N}"])


def test_custom_k_range_and_unfused_api():
    from vpower import interp
    rng = np.random.default_rng(4)
    N, L = 32, 1.0
    v = rng.standard_normal((N, N, N, 3))
    m = np.exp(rng.standard_normal((N, N, N)))
    bf = interp.BoxField(v, m, L / N)
    kmin, kmax, kres = 4 * np.pi / L, 20 * np.pi / L, 3 * np.pi / L
    s = bf.spctrm("velocity", kmin=kmin, kmax=kmax, kres=kres)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, L / N, "velocity", kmin=kmin, kmax=kmax, kres=kres)
    assert np.array_equal(s.k, ref[:, 0]) and np.array_equal(s.Nsample, ref[:, 3])
    assert np.allclose(s.Psum, ref[:, 2], rtol=PSUM_RTOL, atol=0)
    # the reference's two-step API: _pair_power then _hist_sample
    P = orc.vector_power(v[..., 0], v[..., 1], v[..., 2], L, N)
    pair = interp._pair_power(P, L, N)
    assert np.array_equal(pair, orc.pair_power(P, L, N))             # float64 k, bit exact
    tab = interp._hist_sample(pair, 2 * np.pi / L, np.pi * N / L, 2 * np.pi / L)
    rt = orc.hist_sample(orc.pair_power(P, L, N), 2 * np.pi / L, np.pi * N / L, 2 * np.pi / L, "library")
    assert np.array_equal(tab[:, 3], rt[:, 3]) and np.allclose(tab[:, 2], rt[:, 2], rtol=1e-12)


# ------------------------------------------------------- whole pipeline, goldens ----
@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_library_pipeline_against_reference_golden(tag):
    from vpower import interp
    g = golden(f"library_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), float(g["L"]), int(g["seed"])
    pos, vel, mass, dens = synth(seed, Np, L)
    gp = interp.GasParticles(pos.astype(np.float64), mass.astype(np.float64), dens.astype(np.float64),
                             vel.astype(np.float64), L)
    interp.REFERENCE_COMPAT["momentum_bug"] = True       # the golden tables carry quirk Q1
    try:
        for maker, prefix in ((gp.deposit_to_field, "ngp"), (gp.ann_interp_to_field, "nn")):
            bf = maker(N)
            assert bf.Nsize == N and abs(bf.Lbox - L) < 1e-12
            for q in ("velocity", "momentum", "energy"):
                s = bf.spctrm(q)
                ref = g[f"{prefix}_{q}"]
                assert np.array_equal(s.k, ref[:, 0])
                assert np.array_equal(s.Nsample, ref[:, 3])
                assert np.allclose(s.Psum, ref[:, 2], rtol=PSUM_RTOL, atol=0)
                assert np.allclose(s.P, ref[:, 1], rtol=PSUM_RTOL, atol=0)
        assert np.allclose(gp.ann_interp_to_field(N).mass[0, 0, :8], g["nn_mass_head"], rtol=1e-6)
        grid = interp.deposit_to_grid(gp.density_velocity_vector(), gp.pos, N, L)
        assert np.allclose(grid, g["deposit_grid"], rtol=1e-5, atol=1e-5)
    finally:
        interp.REFERENCE_COMPAT["momentum_bug"] = False
    # default (physically correct) momentum differs from the quirk
    s = gp.deposit_to_field(N).spctrm("momentum")
    assert not np.allclose(s.Psum, g["ngp_momentum"][:, 2], rtol=1e-3, atol=0)
    with pytest.raises(Exception):
        gp.deposit_to_field(N).spctrm("vorticity")


def _load_script():
    path = os.path.join(ROOT, "large-velocity-power-spectrum_amd", "scripts", "parallel_optimized.py")
    spec = importlib.util.spec_from_file_location("vps_script", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_script_main_against_reference_golden(tmp_path, tag):
    g = golden(f"script_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), int(g["L"]), int(g["seed"])
    pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
    snap = tmp_path / "snap.npz"
    np.savez(snap, Coordinates=pos, Masses=mass, Velocities=vel)
    s = _load_script()
    assert s.main(["-i", str(snap), "-o", str(tmp_path), "-N", str(N), "-M", str(N), "-l", str(L), "-f"]) == 0
    pk = np.loadtxt(tmp_path / "Pk.txt")
    ref = g["Pk"]
    assert pk.shape == ref.shape
    assert np.array_equal(pk[:, 3], ref[:, 3])
    assert np.allclose(pk[:, 0], ref[:, 0], rtol=1e-7)
    assert np.allclose(pk[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)
    assert np.allclose(pk[:, 1], ref[:, 1], rtol=PSUM_RTOL, atol=0)


def test_config1_full_size_against_oracle():
    """BASELINE config 1 (128^3, 1e5 particles, velocity P(k), script flavour) end to end."""
    from vpower import synth as sy
    s = _load_script()
    N, Np, off = sy.CONFIGS["C1"]
    pos, vel, mass, dens = sy.particles(sy.BASE_SEED + off, Np, 1.0, lognormal_density=False, preprocess=False)
    tab = s.velocity_spectrum(pos, mass, vel, N, 1)
    c, v = orc.preprocess_script(pos, mass, vel)
    ref, _ = orc.script_pipeline(c, v, N, 1)
    assert np.array_equal(tab[:, 3], ref[:, 3])
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)
    assert np.allclose(tab[:, 1], ref[:, 1], rtol=PSUM_RTOL, atol=0)


# ------------------------------------------------ fused deposit -> z pass (pencils) ----
@pytest.mark.parametrize("N,nx,x0", [(64, 64, 0), (128, 128, 0), (128, 32, 64), (256, 256, 0), (512, 64, 448), (1024, 16, 480),
                                     (2048, 4, 1000), (4096, 2, 3001), (192, 192, 0), (384, 48, 96), (768, 16, 752), (1536, 8, 8)])
@pytest.mark.parametrize("quantity,flags", [("velocity", 0), ("momentum", 0), ("momentum", 1)])
def test_fused_deposit_fft_matches_unfused(K, N, nx, x0, quantity, flags):
    from vpower import device
    q = device.QUANTITY[quantity]
    assert K.fused_supported(N, q) and K.fused_supported(N, device.ENERGY) and K.fused_supported(4096, q) and not K.fused_supported(1000, q)
    rng = np.random.default_rng(N + nx)
    Np = 150000
    pos = rng.random((Np, 3)).astype(np.float32)
    pos[: Np // 5] = pos[: Np // 5] * 0.1 + 0.45            # a clump: some very full pencils
    dpos = K.to_device(pos)
    dvel = K.to_device(rng.standard_normal((Np, 3)).astype(np.float32))
    drho = K.to_device(np.exp(rng.standard_normal(Np)).astype(np.float32))
    fields = K.deposit_field(dpos, dvel, drho, N, 1.0, x0, nx, q, flags)
    spec, nyq = K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, x0, nx, q, flags)
    for c in range(3):
        s_ref, n_ref = K.fft_zy(fields[c], N, nx)
        scale = float(s_ref.abs().pow(2).mean().sqrt())
        assert float((spec[c] - s_ref).abs().max()) / scale < 2e-5
        assert float((nyq[c] - n_ref).abs().max()) / scale < 2e-5


def test_component_summed_binning_equals_per_component(K, monkeypatch):
    """One x-pass launch that sums |F|^2 over the three components before binning (default) against
    three launches that bin every component on its own: same counts, sums equal to float32 rounding."""
    from vpower import device
    N, L = 128, 1.0
    rng = np.random.default_rng(5)
    fields = [K.to_device(rng.standard_normal((N, N, N)).astype(np.float32)) for _ in range(3)]
    pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False))
    a = pipe.finish(*pipe.accumulate(fields))
    monkeypatch.setenv("VPS_X_PER_COMPONENT", "1")
    b = pipe.finish(*pipe.accumulate(fields))
    assert np.array_equal(a[:, 3], b[:, 3])
    assert np.allclose(a[:, 2], b[:, 2], rtol=1e-6, atol=0)
    two = pipe.finish(*pipe.accumulate(fields[:2]))          # ncomp = 2
    monkeypatch.delenv("VPS_X_PER_COMPONENT")
    assert np.allclose(pipe.finish(*pipe.accumulate(fields[:2]))[:, 2], two[:, 2], rtol=1e-6, atol=0)


@pytest.mark.parametrize("N,nx,x0", [(64, 64, 0), (256, 64, 64), (1024, 16, 480)])
def test_fused_energy_matches_unfused(K, N, nx, x0):
    from vpower import device
    rng = np.random.default_rng(N + 3)
    Np = 200000
    pos = rng.random((Np, 3)).astype(np.float32)
    pos[: Np // 5] = pos[: Np // 5] * 0.1 + 0.45
    d = [K.to_device(pos), K.to_device(rng.standard_normal((Np, 3)).astype(np.float32)),
         K.to_device(np.exp(rng.standard_normal(Np)).astype(np.float32))]
    field = K.deposit_field(d[0], d[1], d[2], N, 1.0, x0, nx, device.ENERGY)
    spec, nyq = K.deposit_fft_zy(d[0], d[1], d[2], N, 1.0, x0, nx, device.ENERGY)
    assert spec.shape[0] == 1
    s_ref, n_ref = K.fft_zy(field[0], N, nx)
    scale = float(s_ref.abs().pow(2).mean().sqrt())
    assert float((spec[0] - s_ref).abs().max()) / scale < 2e-5
    assert float((nyq[0] - n_ref).abs().max()) / scale < 2e-5


def test_boxfield_energy_after_momentum_shares_the_deposit(K):
    """box.spctrm('momentum') then box.spctrm('energy') of a particle-backed field: the second call runs no deposit launch (the
    momentum launch left the energy field's z image behind, VPS_FLAG_SHARE_ENERGY) -- and both tables match the oracle."""
    from vpower import interp
    N, L, Np = 128, 1.0, 300000
    pos, vel, mass, dens = synth(77, Np, L)
    gp = interp.GasParticles(pos, mass, dens, vel, L)
    box = gp.deposit_to_field(N)
    got_m = box.spctrm("momentum")
    K.timing(True)
    got_e = box.spctrm("energy")
    z_launches = len(K.timing_list("fft_z"))
    K.timing(False)
    assert z_launches == 0
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid_fast(vec, pos, N, L), L / N, zero_empty=True)
    for got, q in ((got_m, "momentum"), (got_e, "energy")):
        ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, L / N, q)
        assert np.array_equal(got.Nsample, ref[:, 3])
        assert np.allclose(got.Psum, ref[:, 2], rtol=PSUM_RTOL, atol=0)


def test_particle_backed_boxfield_takes_the_fused_path_and_matches_the_grid_path():
    """gp.deposit_to_field(N).spctrm(q) goes from the particles to P(k) without a grid; touching a grid
    attribute afterwards materialises the field, and the classic grid path gives the same spectrum."""
    from vpower import interp
    N, L, Np = 64, 1.0, 80000
    pos, vel, mass, dens = synth(91, Np, L)
    gp = interp.GasParticles(pos, mass, dens, vel, L)
    for q in ("velocity", "momentum", "energy"):
        box = gp.deposit_to_field(N)
        assert box._src is not None and box._chans is None
        fast = box.spctrm(q)
        assert box._src is not None and box._chans is None          # still no grid
        _ = box.mass                                                # materialise
        assert box._src is None and box._chans is not None
        slow = box.spctrm(q)
        assert np.array_equal(fast.Nsample, slow.Nsample)
        assert np.allclose(fast.Psum, slow.Psum, rtol=2e-5, atol=0)


def test_reuse_sort_is_safe_and_equal(K):
    """Second quantity of the same particle tensors with the first call's token: same spectra as a fresh sort;
    a stale token (other tensors, a modified tensor, another call in between) silently sorts again."""
    from vpower import device
    N, Np = 128, 300000
    rng = np.random.default_rng(12)
    mk = lambda: [K.to_device(rng.random((Np, 3)).astype(np.float32)), K.to_device(rng.standard_normal((Np, 3)).astype(np.float32)),
                  K.to_device(np.exp(rng.standard_normal(Np)).astype(np.float32))]
    a, b = mk(), mk()
    ref_m = [t.clone() for t in K.deposit_fft_zy(a[0], a[1], a[2], N, 1.0, 0, N, device.MOMENTUM)]
    ref_b = [t.clone() for t in K.deposit_fft_zy(b[0], b[1], b[2], N, 1.0, 0, N, device.MOMENTUM)]
    K.deposit_fft_zy(a[0], a[1], a[2], N, 1.0, 0, N, device.VELOCITY)
    tok = K.fused_token()
    got = K.deposit_fft_zy(a[0], a[1], a[2], N, 1.0, 0, N, device.MOMENTUM, reuse_sort=tok)       # reused
    assert all(float((g - r).abs().max()) <= 2e-5 * float(r.abs().max()) for g, r in zip(got, ref_m))
    got = K.deposit_fft_zy(b[0], b[1], b[2], N, 1.0, 0, N, device.MOMENTUM, reuse_sort=tok)       # other tensors: re-sorted
    assert all(float((g - r).abs().max()) <= 2e-5 * float(r.abs().max()) for g, r in zip(got, ref_b))
    K.deposit_fft_zy(a[0], a[1], a[2], N, 1.0, 0, N, device.VELOCITY)
    tok = K.fused_token()
    a[0].mul_(0.5)                                                                                 # modified in place: re-sorted
    got = K.deposit_fft_zy(a[0], a[1], a[2], N, 1.0, 0, N, device.MOMENTUM, reuse_sort=tok)
    fresh = K.deposit_fft_zy(a[0].clone(), a[1], a[2], N, 1.0, 0, N, device.MOMENTUM)
    assert all(float((g - r).abs().max()) <= 2e-5 * float(r.abs().max()) for g, r in zip(got, [t.clone() for t in fresh]))


def test_weighted_z_pass_equals_algebra_then_z_pass(K):
    """fft_zy(field, weight=mass) against the out-of-place momentum algebra followed by the plain z/y passes."""
    from vpower import device
    N, nx = 128, 32
    rng = np.random.default_rng(21)
    ch = K.to_device(rng.standard_normal((4, nx, N, N)).astype(np.float32))
    p = K.field_algebra_out(ch, device.MOMENTUM, device.FLAG_INPUT_IS_VM, 1.0)       # p_c = v_c * mass (Lcell = 1)
    for c in range(3):
        s_ref, n_ref = [t.clone() for t in K.fft_zy(p[c], N, nx)]
        s_w, n_w = K.fft_zy(ch[c], N, nx, weight=ch[3])
        scale = float(s_ref.abs().pow(2).mean().sqrt())
        assert float((s_w - s_ref).abs().max()) / scale < 1e-6 and float((n_w - n_ref).abs().max()) / scale < 1e-6


def test_fused_pipeline_against_oracle(K):
    from vpower import device
    N, L, Np = 64, 1.0, 60000
    pos, vel, mass, dens = synth(77, Np, L)
    spec, nyq = K.deposit_fft_zy(K.to_device(pos), K.to_device(vel), K.to_device(dens), N, L, 0, N, device.VELOCITY)
    pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False))
    tab = pipe.finish(*pipe.accumulate_spectra(spec, nyq))
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid(vec, pos, N, L), L / N, zero_empty=True)
    ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, L / N, "velocity")
    assert np.array_equal(tab[:, 3], ref[:, 3])
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)


# ------------------------------------------------------------------- edge cases ----
def test_edge_cases_empty_single_and_errors(K):
    from vpower import device, interp, _ffi
    N, L = 16, 1.0
    # no particles: every cell is written (zeros), spectrum is identically zero with full counts
    empty = K.deposit_field(K.empty((0, 3), torch.float32), K.empty((0, 3), torch.float32),
                            K.empty((0,), torch.float32), N, L, 0, N, device.VELOCITY)
    assert empty.shape == (3, N, N, N) and float(empty.abs().max()) == 0.0
    pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False))
    tab = pipe.finish(*pipe.accumulate([empty[0], empty[1], empty[2]]))
    assert np.all(tab[:, 2] == 0) and list(tab[:, 3].astype(int)) == [18, 62, 98, 210, 350, 450, 602, 687]
    # one particle exactly on the upper box face wraps to cell 0 (periodic rule of interp.py:1011)
    g = interp.deposit_to_grid(np.array([2.0]), np.array([[1.0, 0.5, 0.999999]]), N, L)
    assert g[0, 8, 15] == 2.0 and g.sum() == 2.0
    # many particles in ONE cell (bucket far larger than a workgroup; float sums of small integers are exact)
    pos = np.full((100000, 3), 0.53, dtype=np.float32)
    g = interp.deposit_to_grid(np.ones(100000), pos, N, L)
    assert g[8, 8, 8] == 100000.0 and g.sum() == 100000.0
    # a single particle is everybody's nearest neighbour
    idx = interp.nn_index(np.array([[0.3, 0.3, 0.3]]), (np.linspace(0, 1, 5),) * 3)
    assert idx.shape == (5, 5, 5) and not idx.any()
    # unsupported sizes and bad arguments fail loudly, with the library's message
    with pytest.raises(Exception):
        device.PowerPipeline(600, L, kernels=K, comm=device.SlabComm(enabled=False))
    with pytest.raises(_ffi.VpsError, match="slab"):
        K.deposit(K.zeros((4, 3), torch.float32), K.zeros((4, 1), torch.float32), N, L, 8, 16)
    with pytest.raises(_ffi.VpsError, match="contiguous|float32"):
        K.fft_zy(K.zeros((N, N, N), torch.float64), N, N)
    with pytest.raises(_ffi.VpsError, match="C=2"):
        K.deposit(K.zeros((4, 3), torch.float32), K.zeros((4, 2), torch.float32), N, L, 0, N)


def test_float64_positions_through_fused_deposit(K):
    from vpower import device
    rng = np.random.default_rng(12)
    N, L, Np = 32, 2.5, 40000
    pos = rng.random((Np, 3)) * L
    vel = rng.standard_normal((Np, 3)).astype(np.float32)
    rho = np.exp(rng.standard_normal(Np)).astype(np.float32)
    out = K.deposit_field(K.to_device(pos), K.to_device(vel), K.to_device(rho), N, L, 0, N, device.VM).cpu().numpy()
    vec = orc.density_velocity_vector(vel.astype(np.float64), rho.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid(vec, pos, N, L), L / N, zero_empty=True)
    assert np.allclose(out[:3].transpose(1, 2, 3, 0), v, rtol=2e-5, atol=1e-6)
    assert np.allclose(out[3], m, rtol=2e-5, atol=0)
    e = K.deposit_field(K.to_device(pos), K.to_device(vel), K.to_device(rho), N, L, 0, N, device.ENERGY).cpu().numpy()
    assert np.allclose(e[0], orc.kinetic_energy_field(v[..., 0], v[..., 1], v[..., 2], m), rtol=1e-4, atol=1e-9)


# ------------------------------------------------ slab / segment layout, long lines ----
@pytest.mark.parametrize("N,G", [(64, 2), (128, 4), (256, 8), (250, 5), (500, 2)])
def test_emulated_slab_ranks_on_one_gpu(K, N, G):
    """The multi-GPU data path with the real kernels: per-slab z/y passes, the all-to-all
    emulated by assembling each rank's receive buffer, segmented x pass with kz/ky offsets."""
    from vpower import device
    rng = np.random.default_rng(N + G)
    f = K.to_device(rng.standard_normal((N, N, N)).astype(np.float32))
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
    ref = pipe.finish(*pipe.accumulate([f]))
    nx, nkz, nky = N // G, N // 2 // G, N // G
    specs, nyqs = [], []
    for r in range(G):
        s_, q_ = K.fft_zy(f[r * nx:(r + 1) * nx].contiguous(), N, nx)
        specs.append(s_)
        nyqs.append(q_)
    psum, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
    K.set_binning(*pipe._binning)
    for h in range(G):
        recv = torch.cat([specs[g][h * nkz:(h + 1) * nkz].reshape(-1) for g in range(G)])
        recvn = torch.cat([nyqs[g][h * nky:(h + 1) * nky].reshape(-1) for g in range(G)])
        K.fft_x_bin(recv, N, nkz * N, 0, h * nkz, G, nkz * N * nx, psum, ns)
        K.fft_x_bin(recvn, N, nky, h * nky, N // 2, G, nky * nx, psum, ns)
    tab = pipe.finish(psum, ns)
    assert np.array_equal(tab[:, 3], ref[:, 3])
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-6, atol=0)


def _plane_waves(N, modes, x0, nx, device):
    """f(x,y,z) = sum_j A_j cos(2 pi a_j.x / N) on the slab x in [x0, x0+nx) (test-side torch)."""
    x = torch.arange(x0, x0 + nx, device=device, dtype=torch.float64)[:, None, None]
    y = torch.arange(N, device=device, dtype=torch.float64)[None, :, None]
    z = torch.arange(N, device=device, dtype=torch.float64)[None, None, :]
    f = torch.zeros((nx, N, N), device=device, dtype=torch.float32)
    for (ax, ay, az), A in modes:
        f += (A * torch.cos(2 * np.pi * ((ax * x + ay * y + az * z) % N) / N)).to(torch.float32)
    return f


@pytest.mark.parametrize("N,G", [(1024, 1), (1024, 4), (2048, 8), (1536, 8), (2000, 8)])
def test_large_grid_plane_wave_known_answer(K, N, G):
    """Long-line kernels (N = 1024, 2048; 1536 = 8*8*24 and 2000 = 10*10*20) through the slab layout: a few plane waves must land
    in exactly their shells with power 2 (A N^3 / 2)^2, everything else ~ 0."""
    from vpower import device
    modes = [((3, 5, 7), 1.0), ((N // 2 - 1, 11, 2), 0.5), ((17, N - 9, N // 4), 2.0), ((0, 0, 40), 1.5)]
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
    nx, nkz, nky = N // G, N // 2 // G, N // G
    K.set_binning(*pipe._binning)
    psum, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
    specs, nyqs = [], []
    for r in range(G):
        f = _plane_waves(N, modes, r * nx, nx, K.device)
        s_, q_ = K.fft_zy(f, N, nx)
        specs.append(s_)
        nyqs.append(q_)
        del f
    for h in range(G):
        recv = torch.cat([specs[g][h * nkz:(h + 1) * nkz].reshape(-1) for g in range(G)]) if G > 1 else specs[0]
        recvn = torch.cat([nyqs[g][h * nky:(h + 1) * nky].reshape(-1) for g in range(G)]) if G > 1 else nyqs[0]
        K.fft_x_bin(recv, N, nkz * N, 0, h * nkz, G, nkz * N * nx, psum, ns)
        K.fft_x_bin(recvn, N, nky, h * nky, N // 2, G, nky * nx, psum, ns)
        del recv, recvn
    raw = psum.cpu().numpy()
    expect = np.zeros(pipe.nbins)
    ks = device.k_axis(1.0, N)
    for (ax, ay, az), A in modes:
        k = np.sqrt(ks[ax] ** 2 + ks[ay] ** 2 + ks[az] ** 2)
        b = np.searchsorted(pipe.edges, k, side="right") - 1
        assert 0 <= b < pipe.nbins
        expect[b] += 2 * (A * N ** 3 / 2) ** 2
    assert np.allclose(raw, expect, rtol=2e-5, atol=1e-9 * expect.max())
    # shell counts: every mode inside the edges exactly once (checked against the analytic total
    # of the full sphere shells that fit in the box: bins up to N/2 - 1 are complete spheres)
    assert int(ns.sum().item()) > 0 and int(ns.min().item()) > 0


@pytest.mark.parametrize("N", [2048, 4096])
def test_longest_lines_against_numpy(K, N):
    """z/y/x line kernels at N = 2048 and 4096 on a thin slab (nx = 16), against numpy."""
    rng = np.random.default_rng(N)
    nx = 16
    f = rng.standard_normal((nx, 4, N)).astype(np.float32)
    field = np.zeros((nx, N, N), dtype=np.float32)
    field[:, :4, :] = f                        # only four y rows are non-zero
    spec, nyq = K.fft_zy(K.to_device(field), N, nx)
    ref = np.fft.fft(np.fft.rfft(field.astype(np.float64), axis=2)[:, :, :], axis=1)   # [x, ky, kz]
    got = spec.cpu().numpy()                   # [kz, ky, x]
    scale = np.sqrt(np.mean(np.abs(ref) ** 2))
    for kz in (0, 1, 7, N // 4, N // 2 - 1):
        assert np.max(np.abs(got[kz].T - ref[:, :, kz])) / scale < 2e-5
    assert np.max(np.abs(nyq.cpu().numpy().T - ref[:, :, N // 2])) / scale < 2e-5
    # x lines of length N
    lines = (rng.standard_normal((8, N)) + 1j * rng.standard_normal((8, N))).astype(np.complex64)
    out = K.empty((8, N), torch.complex64)
    K.fft_x_write(K.to_device(lines), N, 8, 1, 0, out)
    refx = np.fft.fft(lines.astype(np.complex128), axis=1)
    assert np.max(np.abs(out.cpu().numpy() - refx)) / np.sqrt(np.mean(np.abs(refx) ** 2)) < 2e-5


# -------------------------------------------- full-size properties (config 2) ----
def test_config2_properties(K):
    """512^3 / 1e7 particles: properties that need no CPU reference at this size --
    mass conservation of the deposit, Parseval of FFT+binning, linearity, and the exact
    number of modes falling inside the binned shell range."""
    from vpower import device, synth as sy
    N, Np, off = sy.CONFIGS["C2"]
    L = 1.0
    pos, vel, mass, dens = sy.particles(sy.BASE_SEED + off, Np, L)
    dpos = K.to_device(pos)
    payload = K.density_velocity_vector(K.to_device(vel), K.to_device(dens))
    grid = K.deposit(dpos, payload, N, L, 0, N)
    tot = grid.sum(dim=(1, 2, 3), dtype=torch.float64).cpu().numpy()
    ref = (vel.astype(np.float64) * dens[:, None]).sum(0)
    assert np.allclose(tot[:3], ref, rtol=1e-6, atol=1e-3 * np.sqrt(Np))
    assert abs(tot[3] - dens.astype(np.float64).sum()) < 1e-6 * Np
    K.field_algebra(grid, device.VELOCITY, 0, L / N)
    # the fused deposit+algebra path gives the same three velocity fields
    fused = K.deposit_field(dpos, K.to_device(vel), K.to_device(dens), N, L, 0, N, device.VELOCITY)
    assert torch.allclose(fused, grid[:3], rtol=1e-5, atol=1e-6)
    del fused
    pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False))
    fields = [grid[0], grid[1], grid[2]]
    tab = pipe.finish(*pipe.accumulate(fields))
    # every mode except k=0 and the corners beyond kmax+kres/2 is counted exactly once
    ks = device.k_axis(L, N)
    k2 = ks * ks
    inside = 0
    thr = pipe.thr
    for i in range(N):                       # exact count, one x-plane at a time
        s = (k2[i] + k2[:, None]) + k2[None, :]
        inside += int(np.count_nonzero((s >= thr[0]) & (s < thr[-1])))
    assert int(tab[:, 3].sum()) == inside
    # Parseval over the binned range: sum Psum (2pi/L)^3 = 0.5 <|v|^2> minus the excluded modes
    v2 = sum((f.double() ** 2).sum().item() for f in fields) / N ** 3
    total = tab[:, 2].sum() * (2 * np.pi / L) ** 3
    assert 0.5 * v2 * 0.45 < total <= 0.5 * v2 * (1 + 1e-5)      # corners hold < 55% of white noise
    # linearity: P(2 f) = 4 P(f), bit-for-bit equal Nsample
    tab2 = pipe.finish(*pipe.accumulate([2 * fields[0]]))
    tab1 = pipe.finish(*pipe.accumulate([fields[0]]))
    assert np.array_equal(tab1[:, 3], tab2[:, 3])
    assert np.allclose(tab2[:, 2], 4 * tab1[:, 2], rtol=1e-6)
    del grid, fields
    K._work.clear()
    torch.cuda.empty_cache()
    # the step function bench.py times for C2 (fused deposit + z pass, y passes in a binning-only scope, component-summed
    # binning x pass) at FULL size against the oracle at full size (float64 numpy: ~1 min on one core)
    import bench
    wl = bench.Workload(K, device.SlabComm(enabled=False), N, L, "ngp", ("velocity",), "library", dpos, K.to_device(vel),
                        K.to_device(dens))
    assert wl.fused and not wl.pipe.chunked
    t_step = wl.step()["velocity"]
    ora, _ = bench.oracle_tables("ngp", ("velocity",), "library", N, L, pos, vel, dens)
    ref2 = np.asarray(ora["velocity"], dtype=np.float64)
    assert np.array_equal(t_step[:, 3], ref2[:, 3]) and np.array_equal(t_step[:, 3], tab[:, 3])
    assert np.allclose(t_step[:, 2], ref2[:, 2], rtol=PSUM_RTOL, atol=0)
    assert np.allclose(t_step[:, 1], ref2[:, 1], rtol=PSUM_RTOL, atol=0)


# ------------------------------------------------ particle data stays resident across library calls ----
def test_particles_stay_resident_across_library_calls(K):
    """interp.py:84-131, 169-182, 246-277 on the device: the particle arrays are uploaded once per GasParticles object;
    shift_to_origin / remove_bulk_velocity run vps_preprocess on the resident copies (and update the host arrays); a second
    deposit_to_field(N).spctrm(q) or ann_interp_to_field on the same object performs no host-to-device copy; assigning an
    attribute or editing an array in place drops exactly the copies that depend on it."""
    from vpower import interp
    rng = np.random.default_rng(3)
    Np, N, L = 200_000, 64, 2.0
    pos = (0.3 + rng.random((Np, 3)) * (L - 0.3)).astype(np.float32)
    vel = (rng.standard_normal((Np, 3)) + np.array([0.5, -1.0, 0.2])).astype(np.float32)
    dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    mass = np.ones(Np, dtype=np.float32)
    ref_pos, ref_vel = orc.preprocess_script(pos, mass, vel)
    gp = interp.GasParticles(pos.copy(), mass, dens, vel.copy(), L)
    gp.remove_bulk_velocity()
    gp.shift_to_origin()
    assert np.array_equal(gp.pos, ref_pos)                               # float32 positions: the same subtraction
    assert np.allclose(gp.v, ref_vel, rtol=0, atol=2e-6)                 # bulk velocity summed in float64 on the device
    n0 = K.h2d_copies
    sp1 = gp.deposit_to_field(N).spctrm("velocity")
    n1 = K.h2d_copies
    assert n1 - n0 == 1                                                  # the densities (positions and velocities were already there)
    sp2 = gp.deposit_to_field(N).spctrm("momentum")
    box = gp.ann_interp_to_field(N)
    sp3 = box.spctrm("energy")
    assert K.h2d_copies == n1                                            # nothing uploaded: positions, velocities, densities resident
    assert np.isfinite(sp2.Psum).all() and np.isfinite(sp3.Psum).all()
    # same numbers as a fresh object built from the preprocessed host arrays
    sp1b = interp.GasParticles(gp.pos.copy(), mass, dens, gp.v.copy(), L).deposit_to_field(N).spctrm("velocity")
    assert np.array_equal(sp1.Nsample, sp1b.Nsample) and np.allclose(sp1.Psum, sp1b.Psum, rtol=1e-6)
    # an in-place edit is noticed (hash of every byte), an assignment drops the copy
    n2 = K.h2d_copies
    gp.v[:, 0] *= 2.0
    sp4 = gp.deposit_to_field(N).spctrm("velocity")
    assert K.h2d_copies == n2 + 1 and not np.allclose(sp4.Psum, sp1.Psum, rtol=1e-3)
    gp.density = dens * 2
    gp.deposit_to_field(N).spctrm("momentum")
    assert K.h2d_copies == n2 + 2
    # SPARSE in-place edits (the reference edits these arrays in place, interp.py:400-402): one column that a strided
    # sample of an (n, 3) array never visits, a short row range, ONE particle's density, one coordinate -- each is seen as
    # exactly one upload of the edited array and a different spectrum
    last = sp4
    for edit, quantity in ((lambda: gp.v.__setitem__((slice(None), 1), gp.v[:, 1] * 2.0), "velocity"),
                           (lambda: gp.v.__setitem__(slice(5, 50), 0.0), "velocity"),
                           (lambda: gp.density.__setitem__(12345, gp.density[12345] * 50.0), "momentum"),
                           (lambda: gp.pos.__setitem__((777, 2), 0.5 * L - gp.pos[777, 2] * 0.5), "velocity")):
        ref_sp = gp.deposit_to_field(N).spctrm(quantity)
        n3 = K.h2d_copies
        edit()
        sp = gp.deposit_to_field(N).spctrm(quantity)
        assert K.h2d_copies == n3 + 1, "a sparse in-place edit must cost exactly one upload"
        assert not np.array_equal(sp.Psum, ref_sp.Psum), "the edit must reach the device"
        fresh = interp.GasParticles(gp.pos.copy(), mass, gp.density.copy(), gp.v.copy(), L).deposit_to_field(N).spctrm(quantity)
        assert np.array_equal(sp.Nsample, fresh.Nsample) and np.allclose(sp.Psum, fresh.Psum, rtol=1e-6)
    # a lazy field made BEFORE a preprocessing call keeps the particles as they were then (the reference's field is eager)
    gp2 = interp.GasParticles(pos.copy(), mass, dens, vel.copy(), L)
    before = interp.GasParticles(pos.copy(), mass, dens, vel.copy(), L).deposit_to_field(N).spctrm("velocity")
    box_lazy = gp2.deposit_to_field(N)
    gp2.remove_bulk_velocity()
    gp2.shift_to_origin()
    sp_lazy = box_lazy.spctrm("velocity")
    assert np.array_equal(sp_lazy.Nsample, before.Nsample) and np.allclose(sp_lazy.Psum, before.Psum, rtol=1e-6)


def test_load_snapshot_npz_to_spectrum_against_oracle(K, tmp_path):
    """load_snapshot (reference interp.py:84-131; `.npz` branch, h5py is absent here) -> vps_preprocess on the resident copies ->
    deposit_to_field(N).spctrm(q): against the oracle run on the oracle's own preprocessing of the same file contents."""
    from vpower import interp
    rng = np.random.default_rng(21)
    n, N, L = 60_000, 32, 1.5
    c = (0.2 + rng.random((n, 3)) * (L - 0.2) * 0.999).astype(np.float32)
    m = np.ones(n, dtype=np.float32)
    d = np.exp(0.4 * rng.standard_normal(n)).astype(np.float32)
    v = (rng.standard_normal((n, 3)) + np.array([1.0, 0.0, -2.0])).astype(np.float32)
    f = tmp_path / "snap.npz"
    np.savez(f, Coordinates=c, Masses=m, Density=d, Velocities=v)
    n0 = K.h2d_copies
    gp = interp.load_snapshot(str(f), Lbox=L)
    ref_pos, ref_vel = orc.preprocess_script(c, m, v)
    assert np.array_equal(gp.pos, ref_pos) and np.allclose(gp.v, ref_vel, rtol=0, atol=2e-6)
    for q in ("velocity", "energy"):
        sp = gp.deposit_to_field(N).spctrm(q)
        grid = orc.deposit_to_grid_fast(orc.density_velocity_vector(ref_vel.astype(np.float64), d.astype(np.float64)),
                                        ref_pos, N, L)
        vv, mm = orc.vm_from_vec_grid(grid, L / N, zero_empty=True)
        ref = orc.box_spctrm(vv[..., 0], vv[..., 1], vv[..., 2], mm, L / N, quantity=q)
        assert np.array_equal(sp.Nsample, ref[:, 3])
        assert np.allclose(sp.Psum, ref[:, 2], rtol=PSUM_RTOL, atol=0)
    assert K.h2d_copies - n0 == 4      # positions, velocities, masses (bulk velocity), densities: once each


@pytest.mark.parametrize("N,quantity,flags", [(128, "velocity", 0), (256, "momentum", 0), (128, "momentum", 1)])
def test_single_component_launch_equals_the_vector_launch(K, N, quantity, flags):
    """VPS_FLAG_COMPONENT(c): the fused deposit + z (+ y) pass of ONE component of a vector quantity equals component c of the
    three-component launch (same records, same kernel, ncomp = 1) up to the order in which the LDS float atomics of a cell
    with several particles happen to add -- 1e-5 of a mode + 1e-6 of the rms; energy has no components: error."""
    from vpower import device, synth
    pos, vel, mass, dens = synth.particles(77, 200000, 1.0)
    dpos, dvel, drho = K.to_device(pos), K.to_device(vel), K.to_device(dens)
    q = device.QUANTITY[quantity]

    def same(a, b):
        rms = float(torch.sqrt(torch.mean(torch.abs(b) ** 2)))
        return bool(torch.allclose(torch.view_as_real(a), torch.view_as_real(b), rtol=1e-5, atol=1e-6 * rms))
    spec3, nyq3 = K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, q, flags=flags)
    z3 = K.deposit_fft_z(dpos, dvel, drho, N, 1.0, 0, N, q, flags=flags).clone()
    tok = None
    for c in range(3):
        s1, n1 = K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, q, flags=flags, component=c, reuse_sort=tok)
        tok = K.fused_token()
        assert s1.shape[0] == 1 and same(s1[0], spec3[c]) and same(n1[0], nyq3[c])
    for c in range(3):
        z1 = K.deposit_fft_z(dpos, dvel, drho, N, 1.0, 0, N, q, flags=flags, component=c)
        assert z1.shape[0] == 1 and same(z1[0], z3[c])
    # two components in one launch: in ascending order at the start of the output
    s2, n2 = K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, q, flags=flags, component=(2, 0))
    assert s2.shape[0] == 2 and same(s2[0], spec3[0]) and same(s2[1], spec3[2]) and same(n2[1], nyq3[2])
    with pytest.raises(Exception):
        K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, device.ENERGY, component=0)
    with pytest.raises(Exception):
        K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, q, component=(1, 1))
    with pytest.raises(Exception):     # the library refuses the flag for energy as well
        K.deposit_fft_zy(dpos, dvel, drho, N, 1.0, 0, N, device.ENERGY, flags=(1 << 4))


def test_position_arrays_off_a_16_byte_boundary(K):
    """Coordinate arrays that start 12 bytes into an allocation (a slice handed in through the C ABI): the passes that read
    positions with 16-byte loads (NN bounding box, slab filter) must not care.  NN indices, slab counts and the slab filter's
    z image against the same particles in an aligned copy."""
    from vpower import device, synth
    N, Np = 64, 150001
    pos, vel, mass, dens = synth.particles(91, Np + 1, 1.0)
    dp, dv, dr = K.to_device(pos), K.to_device(vel), K.to_device(dens)
    p1, v1, r1 = dp[1:], dv[1:], dr[1:]                      # 12 bytes (positions, velocities) / 4 bytes (densities) in
    assert p1.data_ptr() % 16 != 0 and p1.is_contiguous()
    p0, v0, r0 = p1.clone(), v1.clone(), r1.clone()
    assert p0.data_ptr() % 16 == 0
    ax = np.linspace(0.5 / N, 1.0 + 0.5 / N, N)
    pay = K.density_velocity_vector(v0, r0)
    g0, i0 = K.nn_resample(p0, pay, (ax, ax, ax), 0, N, want_index=True)
    g1, i1 = K.nn_resample(p1, K.density_velocity_vector(v1, r1), (ax, ax, ax), 0, N, want_index=True)
    assert torch.equal(i0, i1) and torch.equal(g0, g1)
    ref = orc.exact_nn_lattice(pos[1:], ax, ax, ax)
    assert np.array_equal(i1.cpu().numpy().ravel(), ref)
    x0, nx = 16, 16
    c0, c1 = K.count_in_slab(p0, N, 1.0, x0, nx), K.count_in_slab(p1, N, 1.0, x0, nx)
    assert c0 == c1 == int(np.sum((orc.cell_index(pos[1:, 0], N, 1.0) >= x0) & (orc.cell_index(pos[1:, 0], N, 1.0) < x0 + nx)))
    z0 = K.deposit_fft_z(p0, v0, r0, N, 1.0, x0, nx, device.MOMENTUM, slab_particles=c0).clone()
    z1 = K.deposit_fft_z(p1, v1, r1, N, 1.0, x0, nx, device.MOMENTUM, slab_particles=c1)
    rms = float(torch.sqrt(torch.mean(torch.abs(z0) ** 2)))
    assert torch.allclose(torch.view_as_real(z1), torch.view_as_real(z0), rtol=1e-5, atol=1e-6 * rms)


@pytest.mark.parametrize("N,Np", [(64, 50000), (128, 3000)])
def test_nn_resample_quantity_forms(K, N, Np):
    """vps_nn_resample_quantity: the fields a spectrum transforms, written by the search's epilogue (column kernel, scalar
    epilogue, float64 fallback alike) -- against the BoxField form (v, mass) of the same search and the field algebra on it;
    the neighbour indices are the same; and the library path `ann_interp_to_field(N).spctrm(q)` that uses it against the oracle
    (first quantity: direct form; second: grids built once)."""
    from vpower import device, interp, synth
    pos, vel, mass, dens = synth.particles(57, Np, 1.0)
    dp, dv, dr = K.to_device(pos), K.to_device(vel), K.to_device(dens)
    pay = K.density_velocity_vector(dv, dr)
    ax = orc.lattice_axes_library(1.0, N)
    Lcell = 1.0 / N
    vm, i0 = K.nn_resample_field(dp, pay, (ax, ax, ax), 0, N, Lcell, want_index=True)
    for q, flags in ((device.VELOCITY, 0), (device.MOMENTUM, 0), (device.MOMENTUM, device.FLAG_REFERENCE_MOMENTUM_BUG),
                     (device.ENERGY, 0), (device.VM, 0)):
        f, i1 = K.nn_resample_quantity(dp, pay, (ax, ax, ax), 0, N, Lcell, q, flags, want_index=True)
        assert torch.equal(i0, i1)
        if q == device.VM:
            assert torch.equal(f, vm)
            continue
        want = K.field_algebra_out(vm, q, device.FLAG_INPUT_IS_VM | flags, Lcell) if q != device.VELOCITY else vm[:3]
        assert f.shape[0] == want.shape[0] == (1 if q == device.ENERGY else 3)
        assert torch.allclose(f, want, rtol=2e-6, atol=0)
    # the library surface on top of it
    gp = interp.GasParticles(pos, mass, dens, vel, Lbox=1.0)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    grid, _ = orc.ann_interpolate(pos, (ax, ax, ax), vec, N)
    v, m = orc.vm_from_vec_grid(grid, Lcell)
    box = gp.ann_interp_to_field(N)
    for q in ("momentum", "energy", "velocity"):            # first: direct form; then the materialised grids
        sp = box.spctrm(q)
        ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, Lcell, q)
        assert np.array_equal(sp.Nsample, ref[:, 3]) and np.allclose(sp.Psum, ref[:, 2], rtol=2e-5, atol=0)
    assert np.allclose(box.mass, m, rtol=1e-6)              # the grids exist now


@pytest.mark.parametrize("argv,grid", [(["--no-other-configs"], "128"), (["--emulate-ranks", "4", "--no-other-configs"], "64"),
                                       (["--config", "C3"], "64"), (["--config", "C5", "--emulate-ranks", "8"], "128"),
                                       (["--config", "C1"], "64"), (["--config", "C2"], "128"), (["--unfused", "--no-other-configs"], "64")])
def test_bench_command_line_paths(argv, grid, monkeypatch, capsys):
    """bench.py's own control flow on one GPU at a rehearsal size (VPS_BENCH_GRID): the default line with its parity and
    full-size checks, one emulated rank's share, the NN config, C5's emulated share -- each must print ONE JSON line with the
    contract's keys (the checks inside assert parity themselves)."""
    import json
    import bench
    monkeypatch.setenv("VPS_BENCH_GRID", grid)
    monkeypatch.setenv("VPS_BENCH_PARTICLES", "200000")
    bench.main(argv + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["value"] > 0 and "REHEARSAL" in d["config"]["deviation"]
    if "--emulate-ranks" not in argv:
        assert d["parity"]["nsample_equal"] and d["parity"]["psum_max_rel"] < 2e-5 and d["full_size_check"]["nsample_exact"]
