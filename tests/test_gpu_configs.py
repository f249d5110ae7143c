"""GPU tests at the sizes of the BASELINE configs (C2..C5), of the script-surface free
functions, of the fold elimination (8-rank reference fixtures) and of the INTEGRATION.md
binding.  Everything goes through the C ABI of libvps_hip.so; the oracle and torch/numpy
appear only as checkers.

Where the oracle cannot follow at full size (2048^3 float64 grids do not fit a host), the test
either works on a thin x-slab of the full-size grid -- the kernels' line lengths, tile shapes and
index arithmetic are those of the full problem -- or checks size-independent properties on the
full problem: exact shell counts per bin, Parseval over ALL modes (k range widened to the box
corners), conservation of the deposited sums."""
import ctypes as C
import importlib.util
import os
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from helpers import golden, synth  # noqa: E402
from oracle import vps_oracle as orc  # noqa: E402
from oracle import gpu_checks as chk  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PSUM_RTOL = 2e-5   # Psum per non-empty bin vs the float64 oracle / reference


@pytest.fixture(scope="module")
def K():
    from vpower import device
    return device.default_kernels()


def _free(K):
    K._work.clear()
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ helpers (checkers) ----
def _oracle_slab_fields(pos, vel, dens, N, L, x0, nx, quantity):
    """float64 fields [ncomp][nx][N][N] of `quantity` on the x-slab from the oracle's NGP rule:
    orc.cell_index (interp.py:1010-1011), scatter-add of [rho v, rho] (interp.py:1013),
    v = rho v / rho with empty cells 0 (interp.py:272, 329-331), m = rho Lcell^3 (:273)."""
    idx = orc.cell_index(pos, N, L)
    keep = (idx[:, 0] >= x0) & (idx[:, 0] < x0 + nx)
    idx = idx[keep]
    flat = ((idx[:, 0] - x0) * N + idx[:, 1]) * N + idx[:, 2]
    v64, d64 = vel[keep].astype(np.float64), dens[keep].astype(np.float64)
    n3 = nx * N * N
    rho = np.bincount(flat, weights=d64, minlength=n3)
    with np.errstate(invalid="ignore", divide="ignore"):
        v = [np.where(rho > 0, np.bincount(flat, weights=v64[:, c] * d64, minlength=n3) / rho, 0.0) for c in range(3)]
    m = rho * (L / N) ** 3
    if quantity == "velocity":
        out = v
    elif quantity == "momentum":
        out = [v[c] * m for c in range(3)]
    else:
        out = [m * (v[0] ** 2 + v[1] ** 2 + v[2] ** 2)]
    return [f.reshape(nx, N, N) for f in out]


def _shell_counts_exact(K, pipe):
    """Exact number of modes per bin (oracle/gpu_checks.py: torch on the GPU is only the calculator)."""
    return chk.shell_counts_exact(K.device, pipe.N, pipe.k2, pipe.thr)


def _all_mode_pipeline(K, N, L):
    """A pipeline whose bins reach the box corners: every mode except k = 0 is binned, so that
    sum(Psum) obeys Parseval exactly."""
    from vpower import device
    kmin, kmax, kres = chk.all_mode_k_range(N, L)
    return device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False), flavour="script",
                                kmin=kmin, kmax=kmax, kres=kres)


def _sum_and_sumsq(f, planes=32):
    """float64 sum and sum of squares of a big float32 tensor, a few x-planes at a time."""
    s = q = 0.0
    for i in range(0, f.shape[0], planes):
        d = f[i:i + planes].double()
        s += float(d.sum().item())
        q += float((d * d).sum().item())
    return s, q


def _ngp_moments_float64(K, dpos, dvel, drho, N, L, quantities, rows=128):
    """Oracle-side float64 statistics of the NGP fields (oracle/gpu_checks.py); no library call."""
    return chk.ngp_moments_float64(dpos, dvel, drho, N, L, quantities, rows)


# ------------------------------------------- fused deposit + z/y passes, full-size lines ----
@pytest.mark.parametrize("N,nx,x0,quantity", [(512, 16, 96, "velocity"), (512, 16, 496, "momentum"), (1024, 16, 512, "velocity"),
                                               (1024, 16, 0, "energy"), (2048, 16, 1200, "velocity"),
                                               (2048, 16, 2032, "momentum"), (2048, 16, 16, "energy"),
                                               (4096, 16, 2064, "energy"), (4096, 16, 4080, "velocity")])
def test_fused_deposit_fft_zy_thin_slab_against_oracle(K, N, nx, x0, quantity):
    """vps_deposit_fft_zy (the kernel pair bench.py times) against numpy's rfft/fft of the
    ORACLE-deposited slab -- not against another HIP path -- at the line lengths of C2, C3, C4, C5."""
    from vpower import device
    L = 1.0
    rng = np.random.default_rng(N + x0)
    Np = 600_000
    pos = rng.random((Np, 3), dtype=np.float32)
    # half of the particles inside the slab (several per cell in places), half anywhere in the box
    pos[: Np // 2, 0] = (x0 + rng.random(Np // 2, dtype=np.float32) * nx) / N
    pos[: Np // 8, 1:] *= 0.05                                      # a crowded corner
    vel = rng.standard_normal((Np, 3), dtype=np.float32)
    dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    spec, nyq = K.deposit_fft_zy(K.to_device(pos), K.to_device(vel), K.to_device(dens), N, L, x0, nx,
                                 device.QUANTITY[quantity])
    fields = _oracle_slab_fields(pos, vel, dens, N, L, x0, nx, quantity)
    assert spec.shape[0] == len(fields)
    for c, f in enumerate(fields):
        ref = np.fft.fft(np.fft.rfft(f, axis=2), axis=1)            # [x, ky, kz <= N/2]
        scale = np.sqrt(np.mean(np.abs(ref) ** 2))
        got = spec[c].cpu().numpy()                                  # [kz, ky, x]
        err = np.max(np.abs(got.transpose(2, 1, 0) - ref[:, :, : N // 2]))
        errn = np.max(np.abs(nyq[c].cpu().numpy().T - ref[:, :, N // 2]))
        assert err / scale < 1e-5 and errn / scale < 1e-5, (c, err / scale, errn / scale)
    _free(K)


@pytest.mark.parametrize("N,nx,x0", [(512, 16, 40), (2048, 16, 1000), (4096, 8, 4088)])
def test_momentum_launch_shares_the_energy_field(K, N, nx, x0):
    """VPS_FLAG_SHARE_ENERGY: the momentum launch of a step leaves the z image of E = mass |v|^2 behind as a fourth component
    (made of the cell totals its own rounds accumulate); the energy call that follows runs no deposit launch at all.  Against the
    ORACLE-deposited slab (momentum and energy), against the energy launch of its own, and: any other order of calls falls back
    to that launch."""
    from vpower import device
    L = 1.0
    rng = np.random.default_rng(N + x0 + 1)
    Np = 500_000
    pos = rng.random((Np, 3), dtype=np.float32)
    pos[: Np // 2, 0] = (x0 + rng.random(Np // 2, dtype=np.float32) * nx) / N
    pos[: Np // 8, 1:] *= 0.05                                      # a crowded corner: pencils that outgrow their registers
    vel = rng.standard_normal((Np, 3), dtype=np.float32)
    dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    dpos, dvel, drho = K.to_device(pos), K.to_device(vel), K.to_device(dens)

    def check(spec, nyq, quantity, tol=1e-5):
        fields = _oracle_slab_fields(pos, vel, dens, N, L, x0, nx, quantity)
        for c, f in enumerate(fields):
            ref = np.fft.fft(np.fft.rfft(f, axis=2), axis=1)            # [x, ky, kz <= N/2]
            scale = np.sqrt(np.mean(np.abs(ref) ** 2))
            err = np.max(np.abs(spec[c].cpu().numpy().transpose(2, 1, 0) - ref[:, :, : N // 2]))
            errn = np.max(np.abs(nyq[c].cpu().numpy().T - ref[:, :, N // 2]))
            assert err / scale < tol and errn / scale < tol, (quantity, c, err / scale, errn / scale)

    own, own_n = K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.ENERGY)           # the energy launch of its own
    own, own_n = own.clone(), own_n.clone()
    K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.VELOCITY, share_energy=True)
    tok = K.fused_token()
    sm, nm = K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.MOMENTUM, reuse_sort=tok, share_energy=True)
    check(sm, nm, "momentum")
    tok = K.fused_token()
    K.timing(True)
    se, ne = K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.ENERGY, reuse_sort=tok, share_energy=True)
    launches = K.timing_list("fft_z")
    K.timing(False)
    assert len(launches) == 0                                       # no deposit + z launch: only the y pass ran
    check(se, ne, "energy")
    scale = float(own.abs().pow(2).mean().sqrt())
    assert float((se[0] - own[0]).abs().max()) / scale < 2e-5 and float((ne[0] - own_n[0]).abs().max()) / scale < 2e-5
    # energy asked for after something else than the sharing momentum call: the launch of its own, same numbers
    K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.VELOCITY, share_energy=True)
    tok = K.fused_token()
    K.timing(True)
    se2, ne2 = K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.ENERGY, reuse_sort=tok, share_energy=True)
    assert len(K.timing_list("fft_z")) == 1
    K.timing(False)
    assert float((se2[0] - own[0]).abs().max()) / scale < 2e-5
    # the C ABI refuses the flag where it cannot be honoured
    with pytest.raises(Exception):
        K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.VELOCITY, flags=device.FLAG_SHARE_ENERGY)
    _free(K)


# ---------------------------------------------------------------- C3: exact NN + momentum ----
def test_config3_nn_slab_against_oracle(K):
    """The C3 lattice (1024 points per axis, interp.py:1063) on a slab of 4 x-rows, 3e6 particles,
    against the oracle's kd-tree branch: neighbour indices bit exact."""
    N, L, nx, x0 = 1024, 1.0, 4, 700
    rng = np.random.default_rng(33)
    Np = 3_000_000
    pos = rng.random((Np, 3), dtype=np.float32)
    pos[: Np // 2, 0] = (x0 - 2 + rng.random(Np // 2, dtype=np.float32) * (nx + 4)) / N    # dense around the slab
    pos[: Np // 16, 1:] = 0.3 + 0.01 * rng.standard_normal((Np // 16, 2)).astype(np.float32)  # a clump
    ax = orc.lattice_axes_library(L, N)
    payload = K.zeros((Np, 1), torch.float32)
    _, idx = K.nn_resample(K.to_device(pos), payload, (ax, ax, ax), x0, nx, want_index=True)
    ref = orc.exact_nn_lattice(pos, ax[x0:x0 + nx], ax, ax)
    assert np.array_equal(idx.cpu().numpy().ravel(), ref)
    _free(K)


def test_config3_full_size_momentum_properties(K):
    """C3 at full size through the step function bench.py times (exact-NN resampling of 5e7
    particles onto 1024^3, momentum P(k)): shell counts per bin exact, Parseval over all modes
    against the resampled grid itself, and the grid equals a gather of the particle payload."""
    import bench
    from vpower import device, synth as sy
    N, Np, off = sy.CONFIGS["C3"]
    L = 1.0
    dpos, dvel, drho = sy.particles_device(K, sy.BASE_SEED + off, Np, L)
    wl = bench.Workload(K, device.SlabComm(enabled=False), N, L, "nn", ("momentum",), "library", dpos, dvel, drho)
    tab = wl.step()["momentum"]
    assert np.array_equal(tab[:, 3], _shell_counts_exact(K, wl.pipe))
    assert np.isfinite(tab[:, 2]).all() and (tab[:, 2] > 0).all()
    g = wl.grid[:3]                                 # px, py, pz: the search's epilogue wrote the momentum field itself
    # the BoxField form of the same search: every cell holds the payload of SOME particle (mass / Lcell^3 is one of the
    # particle densities), and the momentum written directly equals v * mass of that form
    vm, _ = K.nn_resample_field(dpos, K.density_velocity_vector(dvel, drho), wl.axes, 0, N, L / N)
    rho_cells = (vm[3, ::97, ::89, ::83] / (L / N) ** 3).reshape(-1)
    srt = torch.sort(drho).values
    pos_ = torch.searchsorted(srt, rho_cells).clamp(1, Np - 1)
    near = torch.minimum((srt[pos_] - rho_cells).abs(), (srt[pos_ - 1] - rho_cells).abs())
    assert float((near / rho_cells).max().item()) < 1e-6
    for c in range(3):
        for i in range(0, N, 128):
            want_p = vm[c, i:i + 128] * vm[3, i:i + 128]
            assert torch.allclose(g[c, i:i + 128], want_p, rtol=1e-6, atol=0)
    del vm
    # Parseval, all modes binned: sum Psum (2 pi/L)^3 = 0.5 sum_c (<p_c^2> - <p_c>^2)
    pall = _all_mode_pipeline(K, N, L)
    t2 = pall.finish(*pall.accumulate([g[0], g[1], g[2]]))
    want = 0.0
    for c in range(3):
        s = q = 0.0
        for i in range(0, N, 32):
            d = g[c, i:i + 32].double()
            s += float(d.sum().item())
            q += float((d * d).sum().item())
        want += 0.5 * (q / N ** 3 - (s / N ** 3) ** 2)
    got = t2[:, 2].sum() * (2 * np.pi / L) ** 3
    assert abs(got - want) / want < 2e-5
    del wl, g
    _free(K)


# ------------------------------------------------- C4: velocity + momentum + KE, 2048^3 ----
def test_config4_full_size_properties(K):
    """C4 at full size on one GPU through bench.Workload (fused deposit + z pass at N = 2048,
    reuse of the bucket sort across the three quantities): per-bin shell counts exact for every
    quantity, and Parseval over all modes for each quantity against float64 statistics of the NGP
    fields computed oracle-side (_ngp_moments_float64: a torch restatement of the reference's rule,
    no library deposit)."""
    import bench
    from vpower import device, synth as sy
    N, Np, off = sy.CONFIGS["C4"]
    L = 1.0
    dpos, dvel, drho = sy.particles_device(K, sy.BASE_SEED + off, Np, L)
    # Parseval targets from the oracle-side float64 restatement of the NGP fields (no HIP deposit involved)
    mom = _ngp_moments_float64(K, dpos, dvel, drho, N, L, ("velocity", "momentum", "energy"))
    want = chk.parseval_targets(mom, N)
    _free(K)
    wl = bench.Workload(K, device.SlabComm(enabled=False), N, L, "ngp", ("velocity", "momentum", "energy"), "library",
                        dpos, dvel, drho)
    assert wl.fused
    tabs = wl.step()
    counts = _shell_counts_exact(K, wl.pipe)
    for q, tab in tabs.items():
        assert np.array_equal(tab[:, 3], counts), q
        assert np.isfinite(tab[:, 2]).all() and (tab[:, 2] > 0).all(), q
    # same spectra, bins widened to the corners: Parseval
    pall = _all_mode_pipeline(K, N, L)
    token = None
    for q in ("velocity", "momentum", "energy"):
        nc = bench.NCOMP[q]
        spec, nyq = K.deposit_fft_zy(dpos, dvel, drho, N, L, 0, N, device.QUANTITY[q], spec=wl.spec[:nc], nyq=wl.nyq[:nc],
                                     reuse_sort=token)
        token = K.fused_token()
        t2 = pall.finish(*pall.accumulate_spectra(spec, nyq))
        got = t2[:, 2].sum() * (2 * np.pi / L) ** 3
        assert abs(got - want[q]) / want[q] < 2e-5, (q, got, want[q])
    del wl
    _free(K)


# ------------------------------------------------------ C5: one rank's share of 4096^3 ----
def test_config5_one_rank_share(K):
    """C5 as rank 3 of 8 sees it: 1e9 replicated particles, x-slab of 512 rows of the 4096^3 grid,
    kinetic-energy field through the un-fused deposit and the 4096-point z / y / x line kernels, and through the fused
    deposit + z pass on a slab-sized sort workspace (the route of bench.py).
    Checks: the deposited density of the slab sums to that of the particles whose bit-exact cell
    index (vps_cell_index) falls in the slab; Parseval of the z+y passes; Parseval of
    the segmented x pass + binning over all modes (on the un-exchanged local buffer: any data
    obeys it)."""
    from vpower import device, synth as sy
    N, Np, off = sy.CONFIGS["C5"]
    L, G, r = 1.0, 8, 3
    nx, x0 = N // G, r * (N // G)
    dpos, dvel, drho = sy.particles_device(K, sy.BASE_SEED + off, Np, L)
    rho_grid = K.deposit(dpos, drho[:, None].contiguous(), N, L, x0, nx)      # [1, nx, N, N]
    mass_grid, _ = _sum_and_sumsq(rho_grid[0])
    del rho_grid
    _free(K)
    inside_sum = 0.0
    for s in range(0, Np, 50_000_000):                                         # bit-exact cell index of x (vps_cell_index)
        e = min(Np, s + 50_000_000)
        cx = K.cell_index(dpos[s:e], N, L)[:, 0]
        inside_sum += float(drho[s:e][(cx >= x0) & (cx < x0 + nx)].double().sum().item())
        del cx
    assert abs(mass_grid - inside_sum) / inside_sum < 1e-6
    e_field = K.deposit_field(dpos, dvel, drho, N, L, x0, nx, device.ENERGY)[0]   # [nx, N, N]
    _free(K)
    _, e2 = _sum_and_sumsq(e_field)
    spec, nyq = K.fft_zy(e_field, N, nx)
    del e_field
    _free(K)
    # Parseval of the two local passes: sum_x sum_{ky,kz} |F|^2 (Hermitian weights) = N^2 sum f^2
    tot = 0.0
    for kz0 in range(0, N // 2, 64):
        a = torch.view_as_real(spec[kz0:kz0 + 64]).double()
        w = torch.full((a.shape[0], 1, 1, 1), 2.0, dtype=torch.float64, device=K.device)
        if kz0 == 0:
            w[0] = 1.0
        tot += float((a * a * w).sum().item())
    tot += float((torch.view_as_real(nyq).double() ** 2).sum().item())
    assert abs(tot - float(N) ** 2 * e2) / (float(N) ** 2 * e2) < 2e-5
    # x pass + binning on the rank's kz share of its OWN buffer, segment layout of G slabs
    pall = _all_mode_pipeline(K, N, L)
    K.set_binning(*pall._binning)
    psum, ns = K.zeros((pall.nbins,), torch.float64), K.zeros((pall.nbins,), torch.int64)
    nkz = N // 2 // G
    lines = spec.reshape(-1)[: G * nkz * N * (N // G)]      # G segments of nkz*N lines of N/G points
    K.fft_x_bin(lines, N, nkz * N, 0, r * nkz, G, nkz * N * (N // G), psum, ns)
    a = torch.view_as_real(lines)
    insq = 0.0
    for s in range(0, a.shape[0], 1 << 26):
        d = a[s:s + (1 << 26)].double()
        insq += float((d * d).sum().item())
    # all modes of these kz planes are binned (kz >= 1 here, so no k = 0): sum = 2 * N * sum |in|^2
    got = float(psum.sum().item())
    assert abs(got - 2.0 * N * insq) / (2.0 * N * insq) < 2e-5
    assert int(ns.sum().item()) == 2 * nkz * N * N
    del lines
    # the FUSED route bench.py takes for C5 on several ranks: particles counted into the slab (bit-exact rule), a sort
    # workspace sized for them (not for the 1e9 replicated ones), fused deposit + z pass at 4096-cell lines, y pass --
    # against the un-fused spectra above, plane by plane
    inside = K.count_in_slab(dpos, N, L, x0, nx)
    n_ref = 0
    for s in range(0, Np, 50_000_000):
        e = min(Np, s + 50_000_000)
        cx = K.cell_index(dpos[s:e], N, L)[:, 0]
        n_ref += int(((cx >= x0) & (cx < x0 + nx)).sum().item())
        del cx
    assert inside == n_ref
    assert K.fused_supported(N, device.ENERGY)
    wbytes = int(K.lib.vps_deposit_fft_z_workspace_bytes_slab(Np, inside, N, nx))
    assert wbytes <= 10 * 2 ** 30 and wbytes < int(K.lib.vps_deposit_fft_z_workspace_bytes(Np, N, nx)) // 4
    z = K.deposit_fft_z(dpos, dvel, drho, N, L, x0, nx, device.ENERGY, slab_particles=inside)
    out = K.fft_y_chunk(z[0], N, nx, 1, 1, 0)               # one rank, one chunk: [spec | nyq]
    del z
    scale = float(np.sqrt(tot / (N * (N // 2 + 1.0) * nx)))            # rms amplitude of the spectrum
    ns_ = spec.numel()
    worst = 0.0
    step = 64 * N * nx
    # per mode: 2e-5 of the rms amplitude + 1e-5 of the mode's own (the energy field is positive: its low modes are
    # thousands of times the rms; both routes are float32 -- E = vol sum q^2 / rho here, m |rho v / rho|^2 there -- summed in
    # different orders; each is held to 1e-5 of the rms against the float64 oracle by the thin-slab test above)
    for o in range(0, ns_, step):
        ref_ = spec.reshape(-1)[o:o + step]
        worst = max(worst, float(((out[o:o + step] - ref_).abs() - 1e-5 * ref_.abs()).max().item()))
    worst = max(worst, float(((out[ns_:] - nyq.reshape(-1)).abs() - 1e-5 * nyq.reshape(-1).abs()).max().item()))
    assert worst / scale < 2e-5, worst / scale
    del spec, nyq, out
    _free(K)


def test_slab_sized_sort_workspace(K):
    """vps_count_in_slab + vps_deposit_fft_z_slab: a rank that holds a replicated particle set filters the particles of its
    slab into compact arrays (slab_compact_kernel: block reservations, invalid keys in the block tails) and sorts only those --
    same z images as the plain call, from a workspace sized for the slab.  3e7 particles, an eighth of them in the slab."""
    from vpower import device
    N, Np, L, x0, nx = 512, 30_000_000, 1.0, 192, 64
    gen = torch.Generator(device=K.device)
    gen.manual_seed(11)
    dpos = torch.rand((Np, 3), dtype=torch.float32, device=K.device, generator=gen)
    dpos[:1000] = float("nan")                              # never inside any slab
    dvel = torch.randn((Np, 3), dtype=torch.float32, device=K.device, generator=gen)
    drho = torch.rand((Np,), dtype=torch.float32, device=K.device, generator=gen) + 0.5
    inside = K.count_in_slab(dpos, N, L, x0, nx)
    cx = K.cell_index(dpos[1000:], N, L)[:, 0]              # (bit-exact against the oracle: test_cell_index_bit_exact)
    assert inside == int(((cx >= x0) & (cx < x0 + nx)).sum().item())
    del cx
    plain = int(K.lib.vps_deposit_fft_z_workspace_bytes(Np, N, nx))
    slab = int(K.lib.vps_deposit_fft_z_workspace_bytes_slab(Np, inside, N, nx))
    assert slab < plain // 2
    for q in (device.VELOCITY, device.ENERGY):
        a = K.deposit_fft_z(dpos, dvel, drho, N, L, x0, nx, q)
        K._work.clear()
        b = K.deposit_fft_z(dpos, dvel, drho, N, L, x0, nx, q, slab_particles=inside)
        tol = 1e-5 * float(a.abs().square().mean().sqrt().item())
        assert float((a - b).abs().max().item()) < tol
        # a generous bound changes nothing; a second quantity re-uses the compacted sort
        tok = K.fused_token()
        c = K.deposit_fft_z(dpos, dvel, drho, N, L, x0, nx, q, slab_particles=inside, reuse_sort=tok)
        assert float((a - c).abs().max().item()) < tol
        K._work.clear()
    # small inputs (nothing to gain) take the plain route under the same entry point
    pos, vel, mass, dens = synth(8, 400_000, L)
    d = [K.to_device(x_) for x_ in (pos, vel, dens)]
    ins = K.count_in_slab(d[0], 256, L, 96, 32)
    assert ins == int(np.count_nonzero((lambda c_: (c_ >= 96) & (c_ < 128))(orc.cell_index(pos, 256, L)[:, 0])))
    a = K.deposit_fft_z(d[0], d[1], d[2], 256, L, 96, 32, device.VELOCITY)
    b = K.deposit_fft_z(d[0], d[1], d[2], 256, L, 96, 32, device.VELOCITY, slab_particles=ins)
    assert float((a - b).abs().max().item()) < 1e-5 * float(a.abs().square().mean().sqrt().item())
    _free(K)


# --------------------------------------------------------- script-surface free functions ----
def _load_script():
    path = os.path.join(ROOT, "large-velocity-power-spectrum_amd", "scripts", "parallel_optimized.py")
    spec = importlib.util.spec_from_file_location("vps_script", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_script_fftw_power_against_reference_golden():
    """FFTW_power / FFTW_vector_power (scripts/parallel_optimized.py:92-141) against the outputs
    of the reference's own functions (fft_power.npz: script_scalar, script_vector)."""
    s = _load_script()
    g = golden("fft_power.npz")
    N, L = int(g["N"]), float(g["L"])
    rng = np.random.default_rng(int(g["seed"]))
    fx, fy, fz = (rng.standard_normal((N, N, N)) for _ in range(3))
    P = s.FFTW_power(fx.astype(np.complex64), L, N)
    Pv = s.FFTW_vector_power(fx.astype(np.complex64), fy.astype(np.complex64), fz.astype(np.complex64), L, N)
    assert P.dtype == np.float32 and Pv.dtype == np.float32 and P.shape == (N, N, N)
    tol = 2e-5 * float(g["script_vector"].max())
    assert np.allclose(P, g["script_scalar"], rtol=2e-5, atol=tol)
    assert np.allclose(Pv, g["script_vector"], rtol=2e-5, atol=tol)
    # Parseval as the reference states it (interp.py:1377-1378)
    assert abs(Pv.astype(np.float64).sum() * (2 * np.pi / L) ** 3 - 0.5 * np.mean(fx ** 2 + fy ** 2 + fz ** 2)) < 1e-5


def test_script_pair_power_and_hist_sample():
    """pair_power with the script's shift convention (subtracted where != 0, script:158-163) and
    hist_sample with np.linspace edges / NaN in empty bins (script:176-190), against numpy."""
    s = _load_script()
    N, L = 32, 1.0
    rng = np.random.default_rng(5)
    P = rng.random((N, N, N))
    ks = orc.k_axis(L, N)
    for shift in (np.array([0.0, 0.0, 0.0]), np.array([-2 * np.pi / L * 0.5, 0.0, 2 * np.pi / L * 0.25])):
        pair = s.pair_power(P, L, N, shift)
        ax = [ks - shift[i] if shift[i] != 0 else ks for i in range(3)]
        kx, ky, kz = np.meshgrid(*ax, indexing="ij")
        assert np.array_equal(pair[:, 0], np.sqrt(kx * kx + ky * ky + kz * kz).ravel())
        assert np.array_equal(pair[:, 1], P.ravel())
    kmin, kmax = 2 * np.pi / L, np.pi * N / L
    tab = s.hist_sample(pair, kmin, kmax, kmin)
    ref = orc.hist_sample(pair, kmin, kmax, kmin, "script")
    assert np.array_equal(tab[:, 0], ref[:, 0]) and np.array_equal(tab[:, 3], ref[:, 3])
    assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-12)
    # sparse input: empty bins are NaN in the script flavour
    few = np.array([[kmin, 1.0], [3.2 * kmin, 2.0]])
    t2 = s.hist_sample(few, kmin, kmax, kmin)
    r2 = orc.hist_sample(few, kmin, kmax, kmin, "script")
    assert np.array_equal(np.isnan(t2[:, 1]), np.isnan(r2[:, 1])) and np.isnan(t2[:, 1]).any()


# ------------------------------------------------- fold elimination: 8-rank reference ----
@pytest.mark.parametrize("tag", ["n16_m2", "n16_m4", "n32_m2"])
def test_single_fft_reproduces_the_reference_on_8_mpi_ranks(tmp_path, tag):
    """The reference's own fold + allgather + Reduce path (scripts/parallel_optimized.py:322-391,
    455-456) run on 8 emulated MPI ranks (tests/golden/make_goldens.py, thread-backed communicator;
    2-fold with one loop, 4-fold with 8 loops accumulated through Pk.txt) against ONE full-size
    transform of the script clone on the same particles."""
    g = golden(f"script_8rank_{tag}.npz")
    N, Np, L, seed = int(g["N"]), int(g["Np"]), int(g["L"]), int(g["seed"])
    pos, vel, mass, dens = synth(seed, Np, float(L), lognormal_density=False)
    snap = tmp_path / "snap.npz"
    np.savez(snap, Coordinates=pos, Masses=mass, Velocities=vel)
    s = _load_script()
    assert s.main(["-i", str(snap), "-o", str(tmp_path), "-N", str(N), "-M", str(int(g["M"])), "-l", str(L), "-f"]) == 0
    pk = np.loadtxt(tmp_path / "Pk.txt")
    ref = g["Pk"]
    assert pk.shape == ref.shape
    assert np.array_equal(pk[:, 3], ref[:, 3])
    assert np.allclose(pk[:, 2], ref[:, 2], rtol=PSUM_RTOL, atol=0)
    assert np.allclose(pk[:, 1], ref[:, 1], rtol=PSUM_RTOL, atol=0)


# --------------------------------------------------------- INTEGRATION.md section B stub ----
def test_integration_md_ctypes_stub_runs_verbatim(K):
    """Executes the reference-side binding printed in INTEGRATION.md section B exactly as written
    (torch-free: vps_malloc / vps_memcpy_*), then checks its deposit_to_grid against the oracle."""
    from vpower import _ffi
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# vpower/_vps\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md lost its section B stub"
    code = m.group(1).replace('C.CDLL("libvps_hip.so")', "C.CDLL(%r)" % _ffi.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md:B", "exec"), ns)      # noqa: S102  (our own documentation)
    rng = np.random.default_rng(9)
    N, Np, L = 32, 20000, 1.0
    pos = rng.random((Np, 3)).astype(np.float32)
    f = np.rint(rng.random((Np, 3)) * 8).astype(np.float64)
    grid = ns["deposit_to_grid"](f, pos, N, L)
    assert grid.shape == (N, N, N, 3) and grid.dtype == np.float64
    assert np.array_equal(grid, orc.deposit_to_grid(f, pos, N, L))        # integer payloads: exact
    g1 = ns["deposit_to_grid"](f[:, 0], pos.astype(np.float64), N, L)
    assert np.array_equal(g1, orc.deposit_to_grid(f[:, 0], pos.astype(np.float64), N, L))


# ----------------------------------------------------------- conservation diagnostics ----
def test_check_conservation_device_reductions(K):
    """check_conservation (interp.py:1269-1319) with the BoxField totals computed by one float64 device
    reduction (vps_totals): NGP deposition of [rho v, rho] with m_p = rho_p Lcell^3 conserves mass and
    momentum exactly (ratios 1 to float32 rounding) and can only lose kinetic energy."""
    from vpower import interp
    rng = np.random.default_rng(12)
    N, Np, L = 64, 400_000, 2.0
    pos = (rng.random((Np, 3)) * L).astype(np.float32)
    vel = (rng.standard_normal((Np, 3)) + np.array([1.5, -0.7, 0.4])).astype(np.float32)
    dens = np.exp(0.5 * rng.standard_normal(Np)).astype(np.float32)
    mass = dens.astype(np.float64) * (L / N) ** 3
    gp = interp.GasParticles(pos, mass, dens, vel.astype(np.float64), L)
    box = gp.deposit_to_field(N)
    t = box._totals()                                   # device reduction (the field is particle-backed -> materialised in HBM)
    assert box._chans is not None and not box._host     # nothing was downloaded for it
    m, mom, en, sp = interp.check_conservation(gp, box)
    assert abs(m - 1) < 2e-6 and np.all(np.abs(mom - 1) < 2e-5)
    assert 0.2 < en <= 1 + 1e-6 and 0.2 < sp <= 1 + 1e-5
    # the same totals from the downloaded arrays (numpy, float64)
    ref = np.array([box.mass.sum(), (box.mass * box.vx).sum(), (box.mass * box.vy).sum(), (box.mass * box.vz).sum(),
                    (box.mass * (box.vx ** 2 + box.vy ** 2 + box.vz ** 2)).sum()])
    assert np.allclose(t, ref, rtol=1e-12)
    assert np.allclose(K.particle_totals(K.to_device(vel), K.to_device(mass.astype(np.float32))),
                       [mass.sum(), *(mass[:, None] * vel).sum(0), (mass * (vel.astype(np.float64) ** 2).sum(1)).sum()], rtol=1e-6)
    # a field built from host arrays takes the numpy route and agrees
    box2 = interp.BoxField(box.get_v(), box.mass, L / N)
    assert np.allclose(box2._totals(), t, rtol=1e-12)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices")
def test_context_device_need_not_be_current():
    """A context created for device 1 works while device 0 is current (every entry point selects
    ctx->device for its duration and restores the caller's)."""
    from vpower import device
    torch.cuda.set_device(0)
    K1 = device.HipKernels(1)
    rng = np.random.default_rng(2)
    f = torch.as_tensor(rng.standard_normal((32, 32, 32)).astype(np.float32)).to("cuda:1")
    with torch.cuda.device(0):
        out = K1.rfft3(f, 32)
        torch.cuda.synchronize(1)
    assert torch.cuda.current_device() == 0
    ref = np.fft.fftn(f.cpu().numpy().astype(np.float64))[:, :, :17].transpose(2, 1, 0)               # [kz <= N/2, ky, kx]
    assert out.device.index == 1 and np.allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    K1.close()


# ------------------------------------------------- the two exact-NN search kernels ----
def test_nn_scatter_and_query_centric_kernels_agree(K):
    """Uniform lattices take the particle-centric scatter kernel, anything else the query-centric ring search;
    both are exact, so they agree with each other and with the oracle -- including clumps, voids larger than
    the scatter radius (finished by the fallback search), duplicated particles and a descending axis."""
    rng = np.random.default_rng(77)
    Np, N, L = 40000, 48, 1.0
    pos = rng.random((Np, 3)).astype(np.float32)
    pos[:8000] = (0.2 + 0.004 * rng.standard_normal((8000, 3))).astype(np.float32)   # clump
    pos[8000:30000, 0] *= 0.4                                                         # a void at x > 0.4 for most of them
    pos[100:110] = pos[100]                                                           # duplicates: lowest index wins
    ax = orc.lattice_axes_library(L, N)
    payload = K.zeros((Np, 1), torch.float32)
    dpos = K.to_device(pos)
    ref = orc.exact_nn_lattice(pos, ax, ax, ax)
    _, i1 = K.nn_resample(dpos, payload, (ax, ax, ax), 0, N, want_index=True)
    from vpower import _ffi
    with _ffi.option("nn_query_centric", 1):
        _, i2 = K.nn_resample(dpos, payload, (ax, ax, ax), 0, N, want_index=True)
    assert np.array_equal(i1.cpu().numpy().ravel(), ref) and np.array_equal(i2.cpu().numpy().ravel(), ref)
    # the cell list by the counting sort with global atomics instead of the two-level LDS bucket sort (Np >= 16384 takes the latter)
    with _ffi.option("nn_build_atomic", 1):
        _, i1a = K.nn_resample(dpos, payload, (ax, ax, ax), 0, N, want_index=True)
    assert np.array_equal(i1a.cpu().numpy().ravel(), ref)
    # the two uniform-lattice kernels, each forced: column-register search (lanes own z-columns, no atomics) and
    # particle-centric scatter (LDS minima); with a payload, so that the fused epilogues are compared too
    pay4 = K.to_device(rng.standard_normal((Np, 4)).astype(np.float32))
    outs = []
    for col in (1, 0):
        with _ffi.option("nn_column", col):
            o, ii = K.nn_resample(dpos, pay4, (ax, ax, ax), 0, N, want_index=True)
        assert np.array_equal(ii.cpu().numpy().ravel(), ref), col
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0].reshape(4, -1), pay4[torch.as_tensor(ref, device=K.device).long()].T)
    # very few particles on a fine lattice: by default the scatter kernel; the column kernel forced -- almost every point is
    # beyond its capped radius and goes to the exact fallback, and with more tiles than cells the radii are computed in the kernel
    few = rng.random((50, 3)).astype(np.float32)
    axf = orc.lattice_axes_library(L, 64)
    reff = orc.exact_nn_lattice(few, axf, axf, axf)
    for col in (None, 1):
        with _ffi.option("nn_column", col):
            _, ifew = K.nn_resample(K.to_device(few), K.zeros((50, 1), torch.float32), (axf, axf, axf), 0, 64, want_index=True)
        assert np.array_equal(ifew.cpu().numpy().ravel(), reff), col
    # descending z axis, x-slab [8, 24): uniform, so still one of the two lattice kernels
    axr = ax[::-1].copy()
    _, i3 = K.nn_resample(dpos, payload, (ax, ax, axr), 8, 16, want_index=True)
    assert np.array_equal(i3.cpu().numpy().ravel(), orc.exact_nn_lattice(pos, ax[8:24], ax, axr))
    # non-uniform axes: the ring search
    ay = np.sort(rng.random(20)) * L
    az = np.cumsum(np.exp(rng.standard_normal(24))) / 30
    _, i4 = K.nn_resample(dpos, payload, (ax[:12], ay, az), 0, 12, want_index=True)
    assert np.array_equal(i4.cpu().numpy().ravel(), orc.exact_nn_lattice(pos, ax[:12], ay, az))
    # float64 positions through the scatter kernel
    pos64 = pos.astype(np.float64) * (1 + 1e-9)
    _, i5 = K.nn_resample(K.to_device(pos64), payload, (ax, ax, ax), 0, N, want_index=True)
    assert np.array_equal(i5.cpu().numpy().ravel(), orc.exact_nn_lattice(pos64, ax, ax, ax))


# --------------------------------------------------- chunked exchange layout (one GPU) ----
@pytest.mark.parametrize("N,G,C", [(64, 2, 1), (128, 4, 2), (128, 4, 4), (256, 8, 2), (500, 5, 2), (250, 5, 1),
                                   (1536, 8, 2), (2048, 8, 4)])   # the last two: the wide y pass (16 lines per workgroup), C4's 8-rank layout
def test_chunked_y_pass_layout_emulated_ranks(K, N, G, C):
    """vps_fft_z + vps_fft_y with the real kernels: every emulated rank produces its send buffers chunk by chunk, the
    all-to-all is played by slicing them, the x pass reads the received blocks (Nyquist rows behind the last chunk) --
    the result must equal the one-rank transform of the same field.  Twice: with all rows in place, and inside a
    binning-only scope, where the blocks carry only the rows that can still reach a shell (packed rows)."""
    import contextlib
    from vpower import device
    if N >= 1024:   # generated on the device: a host array of this size would take minutes
        gen = torch.Generator(device=K.device)
        gen.manual_seed(N + G + C)
        f = torch.randn((N, N, N), dtype=torch.float32, device=K.device, generator=gen)
    else:
        rng = np.random.default_rng(N + G + C)
        f = K.to_device(rng.standard_normal((N, N, N)).astype(np.float32))
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
    ref = pipe.finish(*pipe.accumulate([f]))
    nx, nky = N // G, N // G
    nkc = N // 2 // G // C
    zimgs = [K.fft_z(f[g * nx:(g + 1) * nx].contiguous(), N, nx) for g in range(G)]
    del f
    sent = {}
    for packed in (False, True):
        psum, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
        pipe.prepare()
        total = 0
        for c in range(C):
            with (K.binning_only() if packed else contextlib.nullcontext()):
                is_packed = K.y_packed(N)
                blk = K.chunk_block(N, nx, G, C, c, is_packed)
                # NaN-filled send buffers: a row the x pass reads without the y pass having written it would poison the sums
                sends = [K.fft_y_chunk(zimgs[g], N, nx, G, C, c,
                                       out=torch.full((G * blk,), complex(float("nan"), float("nan")), dtype=torch.complex64,
                                                      device=K.device)) for g in range(G)]
            assert is_packed == (packed and N >= 128 and N % 2 == 0)
            assert blk <= nkc * N * nx + (nky * nx if c == C - 1 else 0)
            assert all(s_.numel() == G * blk for s_ in sends)
            total += G * blk
            for h in range(G):
                recv = torch.cat([sends[g][h * blk:(h + 1) * blk] for g in range(G)])
                K.fft_x_bin_chunk([recv], N, nx, G, C, c, h, is_packed, psum, ns)
            del sends
        sent[packed] = total
        tab = pipe.finish(psum, ns)
        assert np.isfinite(tab[:, 2]).all()
        assert np.array_equal(tab[:, 3], ref[:, 3])
        assert np.allclose(tab[:, 2], ref[:, 2], rtol=1e-6, atol=0)
    assert sent[True] <= sent[False]
    if N >= 256 and N % 16 == 0:
        assert sent[True] < 0.9 * sent[False]           # the default k range leaves a fifth of the rows unbinned


def _slot_rows(pipe, N, G, C, c, packed):
    """Host restatement of the packed block layout (include/vps_hip.h: vps_fft_y): per slot j of chunk c the pair
    (kc, rows) -- kc = largest row cut of the slot's G planes (cut of a plane = last |ky| with fl(ky^2 + kz^2) < thr[nbins],
    rounded up to 16 | 15), rows = 2 kc + 1 or all N -- and per PLANE its own cut."""
    nkc = N // 2 // G // C
    k2h = pipe.k2[: N // 2 + 1]

    def cut(kz):
        ok = np.nonzero(~((k2h + pipe.k2[kz]) >= pipe.thr[-1]))[0]
        return -1 if len(ok) == 0 else min(int(ok[-1]) | 15, N // 2)
    slots = []
    for j in range(nkc):
        kc = -1
        if packed:
            kc = max(0, max(cut(c * G * nkc + j * G + h) for h in range(G)))
            if 2 * kc + 1 >= N:
                kc = -1
        slots.append((kc, N if kc < 0 else 2 * kc + 1))
    return slots, cut


def test_chunked_packed_exchange_at_4096_one_sender_slab(K):
    """C5's multi-rank path at its real line length: the chunked + packed vps_fft_y -> vps_fft_x_bin_chunk at N = 4096,
    G = 8 ranks, 4 chunks.  ONE sender's x-slab (512 rows of random data, its z image 34 GB) stands for all eight -- the
    global field is that slab repeated along x --, so every receiver's buffer is the sender's block for it, eight times.
    Checked: the size and row layout of every block against the documented packing; whole planes of the send buffers
    (first / middle / last slot, several destinations, the Nyquist rows) against numpy's FFT of the z image; the buffers are
    pre-filled with NaN, so any row that is read without having been written poisons the sums; shell counts exact; shell
    sums of the packed exchange equal to those of the unpacked one."""
    import contextlib
    from vpower import device
    N, G, C = 4096, 8, 4
    nx, NH, nky = N // G, N // 2, N // G
    nkc = NH // G // C
    gen = torch.Generator(device=K.device)
    gen.manual_seed(4096)
    f = torch.randn((nx, N, N), dtype=torch.float32, device=K.device, generator=gen)
    z = K.fft_z(f, N, nx)
    del f
    _free(K)
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
    pipe.prepare()
    counts = _shell_counts_exact(K, pipe)
    B = z[: nx * NH * N].view(nx, NH, N)
    BN = z[nx * NH * N:].view(nx, N)
    nan = complex(float("nan"), float("nan"))
    tabs, sent = {}, {}
    for packed in (False, True):
        psum, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
        sent[packed] = 0
        for c in range(C):
            slots, cut = _slot_rows(pipe, N, G, C, c, packed)
            rows_total = sum(r for _, r in slots)
            with (K.binning_only() if packed else contextlib.nullcontext()):
                assert K.y_packed(N) == packed
                blk = K.chunk_block(N, nx, G, C, c, packed)
                assert blk == rows_total * nx + (nky * nx if c == C - 1 else 0)
                out = torch.full((G * blk,), nan, dtype=torch.complex64, device=K.device)
                K.fft_y_chunk(z, N, nx, G, C, c, out=out)
            sent[packed] += G * blk
            # whole planes of the send buffer against numpy
            for h, j in ((0, 0), (3, nkc // 2), (7, nkc - 1), (5, 1)):
                kz = c * G * nkc + j * G + h
                ref = np.fft.fft(B[:, kz, :].cpu().numpy().astype(np.complex128), axis=1).T      # [ky][x]
                scale = np.sqrt(np.mean(np.abs(ref) ** 2))
                kc, rows = slots[j]
                r0 = sum(r for _, r in slots[:j])
                got = out[h * blk + r0 * nx: h * blk + (r0 + rows) * nx].view(rows, nx).cpu().numpy()
                own = cut(kz) if packed else N          # rows this plane itself can still bin (<= the slot's kc)
                for pos_ in range(rows):
                    ky = pos_ if (kc < 0 or pos_ <= kc) else N - (2 * kc + 1 - pos_)
                    aky = min(ky, N - ky)
                    if aky <= own:
                        assert np.max(np.abs(got[pos_] - ref[ky])) / scale < 1e-5, (packed, c, h, j, ky)
                    else:      # between the plane's own cut and the slot's: untouched (NaN) or written, never garbage
                        assert np.isnan(got[pos_]).all() or np.max(np.abs(got[pos_] - ref[ky])) / scale < 1e-5
            if c == C - 1:
                refn = np.fft.fft(BN.cpu().numpy().astype(np.complex128), axis=1).T              # [ky][x]
                scale = np.sqrt(np.mean(np.abs(refn) ** 2))
                own = cut(NH) if packed else N
                for h in (0, 2, 7):
                    got = out[h * blk + rows_total * nx: (h + 1) * blk].view(nky, nx).cpu().numpy()
                    for i in range(nky):
                        ky = h * nky + i
                        if min(ky, N - ky) <= own:
                            assert np.max(np.abs(got[i] - refn[ky])) / scale < 1e-5, (packed, "nyq", ky)
                        else:
                            assert np.isnan(got[i]).all() or np.max(np.abs(got[i] - refn[ky])) / scale < 1e-5
            for h in range(G):
                recv = out[h * blk:(h + 1) * blk].repeat(G)          # all eight senders hold the same slab
                K.fft_x_bin_chunk([recv], N, nx, G, C, c, h, packed, psum, ns)
                del recv
            del out
        tabs[packed] = pipe.finish(psum, ns)
        assert np.isfinite(tabs[packed][:, 2]).all()                  # no NaN row reached a shell sum
        assert np.array_equal(tabs[packed][:, 3], counts)
    assert np.allclose(tabs[True][:, 2], tabs[False][:, 2], rtol=1e-6, atol=0)
    assert sent[True] < 0.85 * sent[False]
    del z, B, BN
    _free(K)


def test_fused_z_images_equal_the_fused_zy_path(K):
    """vps_deposit_fft_z (+ vps_fft_y with one rank, one chunk) writes what vps_deposit_fft_zy writes (two separate
    LDS accumulations: the float32 sums may differ in the order of their additions, nothing else)."""
    from vpower import device
    N, Np, L = 256, 300_000, 1.0
    pos, vel, mass, dens = synth(5, Np, L)
    d = [K.to_device(a) for a in (pos, vel, dens)]
    for q in (device.VELOCITY, device.ENERGY):
        spec, nyq = K.deposit_fft_zy(d[0], d[1], d[2], N, L, 64, 32, q)
        z = K.deposit_fft_z(d[0], d[1], d[2], N, L, 64, 32, q)
        for c in range(z.shape[0]):
            out = K.fft_y_chunk(z[c], N, 32, 1, 1, 0)
            tol = 1e-5 * float(spec[c].abs().square().mean().sqrt().item())
            assert float((out[: spec[c].numel()] - spec[c].reshape(-1)).abs().max().item()) < tol
            assert float((out[spec[c].numel():] - nyq[c].reshape(-1)).abs().max().item()) < tol
    _free(K)


# ------------------------------------------------ CIC / TSC assignment + window deconvolution ----
@pytest.mark.parametrize("assignment", ["cic", "tsc"])
def test_higher_order_assignment_and_deconvolution(K, assignment):
    """Extension beyond the reference: cloud-in-cell / triangular-shaped-cloud deposits against the
    oracle's textbook statement (float64), exact conservation of the deposited sums, and the binning x pass with the
    assignment window divided out against the oracle's P(k) / W(k)^2 binned with numpy -- shell counts bit exact."""
    from vpower import device, interp
    rng = np.random.default_rng(21)
    N, Np, L = 32, 60000, 2.0
    pos = (rng.random((Np, 3)) * L).astype(np.float32)
    pos[:100] = np.array([L - 1e-4, 1e-5, L / 2], dtype=np.float32)          # periodic wrap on two faces
    f = rng.standard_normal((Np, 4)).astype(np.float32)
    grid = interp.deposit_to_grid(f, pos, N, L, assignment=assignment)
    ref = orc.deposit_assign(f.astype(np.float64), pos, N, L, assignment)
    assert grid.shape == (N, N, N, 4)
    assert np.allclose(grid, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    assert np.allclose(grid.sum(axis=(0, 1, 2)), f.astype(np.float64).sum(0), rtol=0, atol=1e-3)
    g1 = interp.deposit_to_grid(f[:, 0], pos.astype(np.float64), N, L, assignment=assignment)
    assert np.allclose(g1, orc.deposit_assign(f[:, 0].astype(np.float64), pos.astype(np.float64), N, L, assignment),
                       rtol=0, atol=2e-5 * np.abs(ref).max())
    # deconvolution in the x pass: momentum spectrum of the assigned field, with and without the window
    vel = rng.standard_normal((Np, 3))
    dens = np.exp(0.3 * rng.standard_normal(Np))
    gp = interp.GasParticles(pos, dens * (L / N) ** 3, dens, vel, L)
    box = gp.deposit_to_field(N, assignment=assignment)
    assert box.assignment == assignment
    sp0 = box.spctrm("momentum")
    sp1 = box.spctrm("momentum", deconvolve=True)
    vx, vy, vz, m = box.vx, box.vy, box.vz, box.mass
    P = orc.vector_power(vx * m, vy * m, vz * m, L, N)
    r0 = orc.spectrum_table(P, L, N, "library")
    r1 = orc.spectrum_table(P * orc.window_inv2(N, assignment), L, N, "library")
    for sp, r in ((sp0, r0), (sp1, r1)):
        assert np.array_equal(sp.Nsample, r[:, 3])
        assert np.allclose(sp.Psum, r[:, 2], rtol=PSUM_RTOL, atol=0)
    assert np.all(sp1.Psum > sp0.Psum)
    # the window is switched off again for the next caller
    sp2 = box.spctrm("momentum")
    assert np.allclose(sp2.Psum, sp0.Psum, rtol=1e-6)
    # the general (non mirrored) binning path applies the same factors
    from vpower import _ffi
    with _ffi.option("no_fast_binning", 1):
        sp3 = box.spctrm("momentum", deconvolve=True)
    assert np.array_equal(sp3.Nsample, r1[:, 3]) and np.allclose(sp3.Psum, r1[:, 2], rtol=PSUM_RTOL, atol=0)


# ------------------------------------------------- rows nobody bins are not moved ----
@pytest.mark.parametrize("N", [64, 96, 128, 256, 250])
def test_binning_only_scope_skips_exactly_the_unbinned_rows(K, N):
    """Inside a `binning_only` scope the y passes leave every row (ky, kz) with fl(ky^2 + kz^2) >= thr[nbins] untouched
    (all its modes lie beyond the last edge np.histogram keeps) and write every other row as before; the binning x pass
    gives bit-identical counts and the same sums with or without the scope; rfft3 / power_grid are not affected by it."""
    from vpower import device
    rng = np.random.default_rng(N)
    f = K.to_device(rng.standard_normal((N, N, N)).astype(np.float32))
    # (N = 96: np.arange's rounding gives the library flavour one edge too many there, as it would the reference)
    pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False), flavour="script" if N == 96 else "library")
    pipe.prepare()
    full_s, full_n = K.fft_zy(f, N, N)
    sentinel = complex(1234.5, -6789.0)
    spec = torch.full_like(full_s, sentinel)
    nyq = torch.full_like(full_n, sentinel)
    with K.binning_only():
        K.fft_zy(f, N, N, spec=spec, nyq=nyq)
        ref3 = K.rfft3(f, N)                     # asks for every mode: must ignore the scope
    k2, thr_last = pipe.k2, pipe.thr[-1]
    beyond = (k2[None, : N] + k2[: N // 2 + 1, None]) >= thr_last                # [kz <= N/2][ky]
    kept = ~beyond
    got = torch.cat([spec, nyq[None]], dim=0)                                      # [kz <= N/2][ky][x]
    want = torch.cat([full_s, full_n[None]], dim=0)
    untouched = (got == sentinel).all(dim=2).cpu().numpy()
    written = (got == want).all(dim=2).cpu().numpy()
    assert written[kept].all()                       # every row that can reach a shell is there, bit for bit
    assert (untouched | written).all()               # a row is either written in full or not at all
    assert not untouched[kept].any()
    if N % 2 == 0 and N >= 128 and (N & (N - 1)) == 0:
        assert untouched[beyond].mean() > 0.6        # most of the corner rows were skipped (16-row rounding keeps a few)
    assert torch.equal(ref3, K.rfft3(f, N))
    # shell sums: identical counts, sums equal to rounding of the summation order
    a = pipe.finish(*pipe.accumulate([f]))
    K.set_binning(*pipe._binning)
    ps, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
    K.fft_x_bin(full_s, N, (N // 2) * N, 0, 0, 1, 0, ps, ns)
    K.fft_x_bin(full_n, N, N, 0, N // 2, 1, 0, ps, ns)
    b = pipe.finish(ps, ns)
    assert np.array_equal(a[:, 3], b[:, 3]) and np.allclose(a[:, 2], b[:, 2], rtol=1e-12, atol=0)
    # the rows the scope leaves unwritten may still be READ by a pair tile of the x pass (tiles of 32 |ky| on the small grids,
    # of 6 on 3 2^a do not line up with the 16-row rounding of the cut): whatever they hold must stay out of every sum
    nan = complex(float("nan"), float("nan"))
    spec, nyq = torch.full_like(full_s, nan), torch.full_like(full_n, nan)
    with K.binning_only():
        K.fft_zy(f, N, N, spec=spec, nyq=nyq)
    ps, ns = K.zeros((pipe.nbins,), torch.float64), K.zeros((pipe.nbins,), torch.int64)
    K.fft_x_bin(spec, N, (N // 2) * N, 0, 0, 1, 0, ps, ns)
    K.fft_x_bin(nyq, N, N, 0, N // 2, 1, 0, ps, ns)
    c = pipe.finish(ps, ns)
    assert np.array_equal(c[:, 3], b[:, 3]) and np.allclose(c[:, 2], b[:, 2], rtol=1e-12, atol=0)
