"""Pins oracle/vps_oracle.py against fixtures produced by the reference itself
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import vps_oracle as orc
from helpers import golden, synth


def test_planner_matches_reference():
    g = golden("planner.npz")
    for inp, out in zip(g["inputs"], g["outputs"]):
        got = orc.planner(int(inp[0]), int(inp[1]), int(inp[2]), int(inp[3]))
        assert tuple(float(x) for x in got) == tuple(out)


@pytest.mark.parametrize("N", [16, 32, 64, 128, 500, 512, 1000, 1024, 2048, 4096])
def test_bin_centres_both_flavours(N):
    g = golden("bin_edges.npz")
    kmin, kmax, kres = orc.default_k_range(1.0, N)
    cs, es = orc.edges_script(kmin, kmax, kres)
    cl, el = orc.edges_library(kmin, kmax, kres)
    assert np.array_equal(cs, g[f"script_centres_{N}"])
    assert np.array_equal(cl, g[f"library_centres_{N}"])
    assert len(es) == len(cs) + 1 and len(el) == len(cl) + 1
    probe = np.array([[kmin, 1.0], [kmax, 2.0]])
    assert np.array_equal(orc.hist_sample(probe, kmin, kmax, kres, "script")[:, 3],
                          g[f"script_probe_counts_{N}"])
    assert np.array_equal(orc.hist_sample(probe, kmin, kmax, kres, "library")[:, 3],
                          g[f"library_probe_counts_{N}"])


def test_q2_quirk_bins():
    # script flavour truncates to 511 bins at N=1024, 499 at N=1000
    for N, nb in ((1024, 511), (1000, 499), (512, 256), (2048, 1024)):
        c, _ = orc.edges_script(*orc.default_k_range(1.0, N))
        assert len(c) == nb
    for N in (1000, 1024):
        c, _ = orc.edges_library(*orc.default_k_range(1.0, N))
        assert len(c) == N // 2


@pytest.mark.parametrize("N", [16, 32, 64, 128])
def test_nsample_lattice_counts(N):
    g = golden("nsample.npz")
    P = np.ones((N, N, N))
    kmin, kmax, kres = orc.default_k_range(1.0, N)
    pk = orc.pair_power(P, 1.0, N)
    if N <= 32:
        assert np.array_equal(pk[:, 0], g[f"pair_k_{N}"])
    nl = orc.hist_sample(pk, kmin, kmax, kres, "library")[:, 3].astype(np.int64)
    ns = orc.hist_sample(pk, kmin, kmax, kres, "script")[:, 3].astype(np.int64)
    assert np.array_equal(nl, g[f"library_{N}"])
    assert np.array_equal(ns, g[f"script_{N}"])
    # half-spectrum restatement reproduces the full-spectrum counts bit for bit
    _, edges = orc.edges_library(kmin, kmax, kres)
    _, nh = orc.bin_half_spectrum(np.ones((N, N, N // 2 + 1)), 1.0, N, edges)
    assert np.array_equal(nh, g[f"library_{N}"])


def test_nsample_known_answer_n16():
    # lattice-count identity quoted in SURVEY.md section 4
    g = golden("nsample.npz")
    assert list(g["library_16"]) == [18, 62, 98, 210, 350, 450, 602, 687]


def test_nsample_nonunit_box():
    g = golden("nsample.npz")
    N, L = 32, 2.5
    kmin, kmax, kres = orc.default_k_range(L, N)
    pk = orc.pair_power(np.ones((N, N, N)), L, N)
    assert np.array_equal(orc.hist_sample(pk, kmin, kmax, kres, "library")[:, 3].astype(np.int64),
                          g["library_32_L2p5"])


@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_library_pipeline(tag):
    g = golden(f"library_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), float(g["L"]), int(g["seed"])
    pos, vel, mass, dens = synth(seed, Np, L)
    Lcell = L / N
    # cell indices, float32 and float64 positions: bit exact
    assert np.array_equal(orc.cell_index(pos, N, L).astype(np.int32), g["cell_f32"])
    pos64 = pos.astype(np.float64) * 1.000001
    assert np.array_equal(orc.cell_index(pos64, N, L).astype(np.int32), g["cell_f64"])
    p64, v64, d64 = pos.astype(np.float64), vel.astype(np.float64), dens.astype(np.float64)
    vec = orc.density_velocity_vector(v64, d64)
    assert np.array_equal(vec[:8], g["dvv_head"])
    grid = orc.deposit_to_grid(vec, p64, N, L)
    assert np.array_equal(grid, g["deposit_grid"])
    assert np.allclose(orc.deposit_to_grid_fast(vec, p64, N, L), g["deposit_grid"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(orc.deposit_to_grid(d64, p64, N, L), g["deposit_scalar"])
    v, m = orc.vm_from_vec_grid(grid, Lcell, zero_empty=True)
    for q in ("velocity", "momentum", "energy"):
        tab = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, Lcell, q, reference_compat=True)
        ref = g[f"ngp_{q}"]
        assert np.array_equal(tab[:, 3], ref[:, 3])          # Nsample bit exact
        assert np.array_equal(tab[:, 0], ref[:, 0])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-12)
        assert np.allclose(tab[:, 1], ref[:, 1], rtol=1e-12)
    if N <= 16:
        assert np.allclose(orc.vector_power(v[..., 0], v[..., 1], v[..., 2], L, N),
                           g["ngp_velocity_Pgrid"], rtol=1e-12, atol=1e-18)
    # NN flavour
    coords = orc.make_grid_coords(L, N)
    assert np.array_equal(coords[: 2 * N], g["grid_coords_head"])
    assert np.array_equal(coords[-2 * N:], g["grid_coords_tail"])
    ax = orc.lattice_axes_library(L, N)
    vg, idx = orc.ann_interpolate(p64, (ax, ax, ax), vec, N)
    assert np.array_equal(idx.astype(np.int32), g["nn_idx"])
    assert np.allclose(vg.sum(axis=(0, 1, 2)), g["nn_vec_grid_sum"], rtol=1e-12)
    v2, m2 = orc.vm_from_vec_grid(vg, Lcell)
    assert np.allclose(m2[0, 0, :8], g["nn_mass_head"], rtol=1e-14)
    for q in ("velocity", "momentum", "energy"):
        tab = orc.box_spctrm(v2[..., 0], v2[..., 1], v2[..., 2], m2, Lcell, q, reference_compat=True)
        ref = g[f"nn_{q}"]
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-12)
        assert np.allclose(tab[:, 1], ref[:, 1], rtol=1e-12)


def test_exact_nn_kdtree_branch_equals_bruteforce():
    rng = np.random.default_rng(5)
    P = rng.random((3000, 3))
    ax = np.linspace(0.01, 0.99, 12)
    brute = orc.exact_nn_lattice(P[:500], ax, ax, ax)            # brute-force branch
    d = ((np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 1, 3) - P[None, :500]) ** 2).sum(-1)
    assert np.array_equal(brute, d.argmin(1))
    big = orc.exact_nn_lattice(np.tile(P, (1, 1)), np.linspace(0, 1, 40), ax, ax)   # kd-tree branch
    qq = np.stack(np.meshgrid(np.linspace(0, 1, 40), ax, ax, indexing="ij"), -1).reshape(-1, 3)
    ref = np.array([np.argmin(((q - P) ** 2).sum(1)) for q in qq])
    assert np.array_equal(big, ref)


@pytest.mark.parametrize("tag", ["n16", "n32"])
def test_script_main_pk_table(tag):
    g = golden(f"script_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), int(g["L"]), int(g["seed"])
    pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
    coords, velocity = orc.preprocess_script(pos, mass, vel)
    tab, _ = orc.script_pipeline(coords, velocity, N, L)
    ref = g["Pk"]
    assert tab.shape == ref.shape
    assert np.array_equal(tab[:, 3].astype(np.float64), ref[:, 3])     # Nsample
    assert np.allclose(tab[:, 0], ref[:, 0], rtol=1e-7)
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-6)                # complex64 FFT rounding
    assert np.allclose(tab[:, 1], ref[:, 1], rtol=2e-6)


def test_fft_power_known_answers():
    g = golden("fft_power.npz")
    rng = np.random.default_rng(int(g["seed"]))
    N, L = int(g["N"]), float(g["L"])
    fx, fy, fz = (rng.standard_normal((N, N, N)) for _ in range(3))
    assert np.allclose(orc.vector_power(fx, fy, fz, L, N), g["vector"], rtol=1e-12, atol=1e-20)
    assert np.allclose(orc.scalar_power(fx, L, N), g["scalar"], rtol=1e-12, atol=1e-20)
    s = orc.fftw_power_c64(fx.astype(np.complex64), L, N)
    assert s.dtype == np.float32
    assert np.allclose(s, g["script_scalar"], rtol=1e-4, atol=1e-9)
    # Parseval normalisation stated at interp.py:1377-1378
    lhs = np.sum(g["vector"]) * (2 * np.pi / L) ** 3
    rhs = 0.5 * np.mean(fx ** 2 + fy ** 2 + fz ** 2)
    assert abs(lhs - rhs) < 1e-12 * rhs


@pytest.mark.parametrize("tag", ["n16_m2", "n16_m4", "n32_m2"])
def test_unfolded_oracle_equals_reference_on_8_mpi_ranks(tag):
    """The reference's fold + allgather + Reduce path (scripts/parallel_optimized.py:322-391, 455-456)
    on 8 emulated ranks -- 2-fold in one loop, 4-fold in 8 loops accumulated through Pk.txt -- is one
    full N^3 transform: the un-folded one-rank restatement reproduces its table."""
    g = golden(f"script_8rank_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), int(g["L"]), int(g["seed"])
    assert int(g["ranks"]) == 8 and tuple(g["plan"][:3]) == orc.planner(N, L, int(g["M"]), 8)[:3]
    pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
    coords, velocity = orc.preprocess_script(pos, mass, vel)
    tab, _ = orc.script_pipeline(coords, velocity, N, L)
    ref = g["Pk"]
    assert tab.shape == ref.shape
    assert np.array_equal(tab[:, 3].astype(np.float64), ref[:, 3])     # Nsample: bit exact
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=2e-6)                # fold phases in complex128, sums in float32
    assert np.allclose(tab[:, 1], ref[:, 1], rtol=2e-6)
    one = golden("script_%s.npz" % tag.split("_")[0])["Pk"]            # the reference's own 1-rank run
    assert np.array_equal(one[:, 3], ref[:, 3]) and np.allclose(one[:, 2], ref[:, 2], rtol=1e-6)


@pytest.mark.parametrize("N,Np,workers", [(32, 30000, 3), (48, 50000, 8)])
def test_allcores_oracle_equals_the_one_core_oracle(N, Np, workers):
    """oracle/allcores.py -- the oracle's NGP step spread over the host's cores (one process per x-slab for gridding, |F|^2 and
    the two histograms; threaded transforms), what bench.py times as `cpu_baseline_allcores` -- gives the one-core oracle's
    tables: shell counts exactly, shell sums to float64 rounding (partial histograms are added in slab order)."""
    from oracle import allcores
    from oracle import vps_oracle as orc
    pos, vel, mass, dens = synth(7, Np, 1.0)
    quantities = ("velocity", "momentum", "energy")
    tabs, t = allcores.ngp_tables(quantities, "library", N, 1.0, pos, vel, dens, workers)
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    v, m = orc.vm_from_vec_grid(orc.deposit_to_grid(vec, pos, N, 1.0), 1.0 / N, zero_empty=True)
    for q in quantities:
        ref = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, 1.0 / N, q)
        assert np.array_equal(tabs[q][:, 0], ref[:, 0]) and np.array_equal(tabs[q][:, 3], ref[:, 3])
        assert np.allclose(tabs[q][:, 2], ref[:, 2], rtol=1e-11, atol=0) and np.allclose(tabs[q][:, 1], ref[:, 1], rtol=1e-11, atol=0)
    assert set(t) == {"gridding", "spctrm_velocity", "spctrm_momentum", "spctrm_energy", "total"}
