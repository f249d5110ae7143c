"""Device pipeline: particles -> slab grid -> 3-D R2C FFT -> |f|^2 -> shell bins.

Host-side orchestration only.  All arithmetic on grids and particles is done by the
HIP kernels of libvps_hip.so through `HipKernels`; torch supplies device buffers,
streams and (for more than one GPU) `torch.distributed` collectives.

Multi-GPU: 1-D slab decomposition along x, one process per GPU.  Particles are
replicated (as the reference replicates the snapshot on every MPI rank,
scripts/parallel_optimized.py:272-276); each rank deposits / resamples its own x-slab,
runs the z pass locally, then per kz chunk the y pass straight into the send buffer of an
all-to-all (ONE message per scalar field and pair of ranks, the Nyquist-plane rows riding
behind the last chunk; the chunks only pipeline it against the passes), runs the x pass
with fused binning on the arrived chunks and finally all-reduces the (nbins,) shell sums --
the step that replaces the two comm.Reduce calls at scripts/parallel_optimized.py:455-456.

The kernel set is injectable so that the distributed choreography can be exercised on
CPU tensors by the test-suite's oracle-backed stand-in; the product default is
`HipKernels` and there is no fallback from it.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _ffi

VELOCITY, MOMENTUM, ENERGY, VM = 0, 1, 2, 3
QUANTITY = {"velocity": VELOCITY, "momentum": MOMENTUM, "energy": ENERGY}
FLAG_REFERENCE_MOMENTUM_BUG = 1
FLAG_INPUT_IS_VM = 2
FLAG_REUSE_SORT = 4
FLAG_SHARE_ENERGY = 8


# --------------------------------------------------------------------------- #
# host-side tables for the binning kernel
# --------------------------------------------------------------------------- #
def k_axis(Lbox, Nsize):
    """2*pi*fftfreq(N, Lcell), as interp.py:1448-1449 / parallel_optimized.py:153-154."""
    Lcell = Lbox / float(Nsize)
    return 2 * np.pi * np.fft.fftfreq(Nsize, Lcell)


def bin_edges(kmin, kmax, spacing, flavour="library"):
    """Bin centres and edges with the reference's own numpy expressions:
    library -> np.arange (interp.py:1472-1473); script -> np.linspace with
    n_bins=int((kmax-kmin)/spacing)+1 (parallel_optimized.py:178-180)."""
    if flavour == "library":
        centers = np.arange(kmin, kmax + spacing, spacing)
        edges = np.arange(kmin - spacing / 2, kmax + 3 * spacing / 2, spacing)
    elif flavour == "script":
        n_bins = int((kmax - kmin) / spacing) + 1
        centers = np.linspace(kmin, kmax, n_bins)
        edges = np.linspace(kmin - spacing / 2, kmax + spacing / 2, n_bins + 1)
    else:
        raise Exception("flavour must be 'library' or 'script'")
    if len(edges) != len(centers) + 1:
        raise Exception("bin centres and edges are inconsistent (%d, %d)" % (len(centers), len(edges)))
    return centers, edges


ASSIGNMENT_ORDER = {"ngp": 1, "cic": 2, "tsc": 3}


def window_inv2_axis(Nsize, assignment):
    """1 / W(k)^2 of one axis for the mass-assignment window W(k) = sinc(pi k / (2 k_Nyquist))^p, p = 1 (NGP),
    2 (CIC), 3 (TSC), indexed like fftfreq (float32).  The three axis factors multiply."""
    p = ASSIGNMENT_ORDER[assignment]
    n = np.fft.fftfreq(Nsize, 1.0 / Nsize)          # integer mode numbers
    x = np.pi * n / Nsize
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.where(n == 0, 1.0, np.sin(x) / x)
    return (w ** (-2.0 * p)).astype(np.float32)


def sqrt_thresholds(edges):
    """thr[i] = smallest float64 t with sqrt(t) >= edges[i] (i < nbins) and
    thr[nbins] = smallest t with sqrt(t) > edges[nbins], so that comparing
    s = kx^2+ky^2+kz^2 against thr reproduces numpy.histogram's comparison of
    sqrt(s) against the edges (left-closed bins, last bin right-closed) bit for bit
    without a square root on the device."""
    e = np.asarray(edges, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        t = e * e
        t[e <= 0] = 0.0
        pos = e > 0
        lo = t.copy()
        # lower thresholds: walk down while the predecessor still satisfies sqrt >= e
        for _ in range(8):
            prev = np.nextafter(lo, -np.inf)
            ok = pos & (prev >= 0) & (np.sqrt(np.maximum(prev, 0)) >= e)
            if not ok.any():
                break
            lo = np.where(ok, prev, lo)
        for _ in range(8):
            bad = pos & (np.sqrt(lo) < e)
            if not bad.any():
                break
            lo = np.where(bad, np.nextafter(lo, np.inf), lo)
        # last edge: first t with sqrt(t) > e
        hi = lo[-1]
        while np.sqrt(hi) <= e[-1]:
            hi = np.nextafter(hi, np.inf)
        while hi > 0 and np.sqrt(np.nextafter(hi, -np.inf)) > e[-1]:
            hi = np.nextafter(hi, -np.inf)
    thr = lo.copy()
    thr[-1] = hi
    return thr


# --------------------------------------------------------------------------- #
# kernel set backed by libvps_hip.so
# --------------------------------------------------------------------------- #
class HipKernels:
    """Thin typed wrapper of the C ABI working on torch CUDA tensors."""

    name = "hip"

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise _ffi.VpsError("no HIP device is visible to this process; the vpower device path "
                                "needs an MI355X (there is no CPU fallback)")
        self.lib = _ffi.lib()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        h = C.c_void_p()
        rc = self.lib.vps_create(C.byref(h), self.device.index)
        if rc != 0:
            raise _ffi.VpsError("vps_create failed (%d): %s" % (rc, self.lib.vps_last_error(None).decode()))
        self.ctx = h
        self._work = {}
        info = (C.c_int64 * 4)()
        self._chk(self.lib.vps_device_info(self.ctx, info))
        self.num_cu, self.lds_per_cu, self.wave, self.hbm_mib = (int(x) for x in info)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vps_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise _ffi.VpsError("libvps_hip call failed (%d): %s" % (rc, self.lib.vps_last_error(self.ctx).decode()))

    def _stream(self):
        self._chk(self.lib.vps_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _ptr(self, t, dtype=None):
        if t is None:
            return None
        if not t.is_cuda or t.device != self.device:
            raise _ffi.VpsError("tensor must live on %s (got %s)" % (self.device, t.device))
        if not t.is_contiguous():
            raise _ffi.VpsError("tensor must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise _ffi.VpsError("tensor must be %s (got %s)" % (dtype, t.dtype))
        return C.c_void_p(t.data_ptr())

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    h2d_copies = 0      # host -> device uploads made through to_device (tests assert residency with it)
    h2d_bytes = 0

    def to_device(self, arr, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        self.h2d_copies += 1
        self.h2d_bytes += t.numel() * t.element_size()
        return t.to(self.device)

    def workspace(self, key, nbytes):
        w = self._work.get(key)
        if w is None or w.numel() < nbytes:
            self._work[key] = None
            w = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._work[key] = w
        return w

    def sync(self):
        self._chk(self.lib.vps_sync(self.ctx))

    # -- timing ---------------------------------------------------------------
    def timing(self, on):
        self._chk(self.lib.vps_timing_enable(self.ctx, 1 if on else 0))
        self._chk(self.lib.vps_timing_reset(self.ctx))

    def timing_get(self):
        out = {}
        for name, kind in _ffi.KERNEL_KINDS.items():
            n, ms = C.c_int64(), C.c_double()
            self._chk(self.lib.vps_timing_get(self.ctx, kind, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def timing_list(self, name):
        kind = _ffi.KERNEL_KINDS[name]
        n = C.c_int64()
        self._chk(self.lib.vps_timing_list(self.ctx, kind, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.float64)
        self._chk(self.lib.vps_timing_list(self.ctx, kind, _ffi.as_dp(buf), n.value, C.byref(n)))
        return buf[: n.value]

    # -- stage A --------------------------------------------------------------
    def preprocess(self, pos, vel, mass, shift_to_origin=True, remove_bulk_velocity=True):
        """In place: shift positions to the origin, remove the mass-weighted bulk velocity.
        Returns (min[3], bulk[3]) that were subtracted."""
        self._stream()
        mn, bv = np.zeros(3), np.zeros(3)
        self._chk(self.lib.vps_preprocess(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                          self._ptr(vel, torch.float32) if vel is not None else None,
                                          self._ptr(mass, torch.float32) if mass is not None else None,
                                          pos.shape[0], 1 if shift_to_origin else 0,
                                          1 if remove_bulk_velocity else 0, _ffi.as_dp(mn), _ffi.as_dp(bv)))
        return mn, bv

    def totals(self, v, mass, elem_stride, comp_stride, n):
        """float64 [sum m, sum m vx, sum m vy, sum m vz, sum m |v|^2] (vps_totals)."""
        self._stream()
        out = np.zeros(5)
        self._chk(self.lib.vps_totals(self.ctx, self._ptr(v, torch.float32), int(elem_stride), int(comp_stride),
                                      self._ptr(mass, torch.float32), int(n), _ffi.as_dp(out)))
        return out

    def particle_totals(self, vel, mass):
        return self.totals(vel, mass, 3, 1, vel.shape[0])

    def field_totals(self, chans):
        """chans [4, ...] = vx, vy, vz, mass of a gridded field."""
        n = chans[0].numel()
        return self.totals(chans, chans[3], 1, n, n)

    @staticmethod
    def _pos_kind(pos):
        if pos.dtype == torch.float32:
            return 0
        if pos.dtype == torch.float64:
            return 1
        raise _ffi.VpsError("positions must be float32 or float64")

    def cell_index(self, pos, N, Lbox):
        self._stream()
        out = self.empty((pos.shape[0], 3), torch.int32)
        self._chk(self.lib.vps_cell_index(self.ctx, self._ptr(pos), self._pos_kind(pos), pos.shape[0], N,
                                          float(Lbox), self._ptr(out)))
        return out

    def density_velocity_vector(self, vel, rho):
        self._stream()
        out = self.empty((vel.shape[0], 4), torch.float32)
        self._chk(self.lib.vps_density_velocity_vector(self.ctx, self._ptr(vel, torch.float32),
                                                       self._ptr(rho, torch.float32), vel.shape[0],
                                                       self._ptr(out)))
        return out

    def deposit(self, pos, payload, N, Lbox, x0, nx, out=None):
        """payload [np, C] float32 -> grid [C, nx, N, N] float32 (every cell written)."""
        self._stream()
        C_ = payload.shape[1]
        if out is None:
            out = self.empty((C_, nx, N, N), torch.float32)
        work = self.workspace("deposit", self.lib.vps_deposit_workspace_bytes(pos.shape[0], C_, N, nx))
        self._chk(self.lib.vps_deposit_ngp(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                           self._ptr(payload, torch.float32), pos.shape[0], C_, N,
                                           float(Lbox), x0, nx, self._ptr(out, torch.float32), self._ptr(work)))
        return out

    def deposit_field(self, pos, vel, rho, N, Lbox, x0, nx, quantity, flags=0, out=None):
        """Fused deposit of [rho v, rho] + field algebra -> [ncomp, nx, N, N] float32."""
        self._stream()
        ncomp = {VELOCITY: 3, MOMENTUM: 3, ENERGY: 1, VM: 4}[quantity]
        if out is None:
            out = self.empty((ncomp, nx, N, N), torch.float32)
        work = self.workspace("deposit", self.lib.vps_deposit_workspace_bytes(pos.shape[0], 4, N, nx))
        self._chk(self.lib.vps_deposit_field(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                             self._ptr(vel, torch.float32), self._ptr(rho, torch.float32),
                                             pos.shape[0], N, float(Lbox), x0, nx, quantity, flags,
                                             self._ptr(out, torch.float32), self._ptr(work)))
        return out

    def fused_supported(self, N, quantity):
        return bool(self.lib.vps_deposit_fft_zy_supported(self.ctx, int(N), int(quantity)))

    def deposit_fft_zy(self, pos, vel, rho, N, Lbox, x0, nx, quantity, flags=0, spec=None, nyq=None, reuse_sort=None,
                       component=None, share_energy=False):
        """Fused deposit + field algebra + z/y passes:
        -> spec [ncomp, N/2, N, nx], nyq [ncomp, N, nx] (complex64); ncomp = 1 for ENERGY, else 3.
        component = 0..2, or a collection of them: only those components of a velocity / momentum field, in ascending order
        (ncomp = their number; VPS_FLAG_COMPONENTS).
        share_energy (pass it on EVERY call of a step that asks for momentum and, later, kinetic energy): the workspace gets room
        for a fourth z image; the momentum launch -- whose rounds accumulate the very cell totals of rho v_c the energy field is
        made of -- leaves that field's z image there as well (VPS_FLAG_SHARE_ENERGY), and the energy call that follows it with a
        valid reuse_sort token only runs its y pass.  Any other order of calls simply launches the energy kernel as usual."""
        self._stream()
        ncomp = 1 if quantity == ENERGY else 3
        if component is not None:
            mask = self._component_mask(quantity, component)
            flags |= mask << 4
            ncomp = bin(mask).count("1")
        if spec is None:
            spec = self.empty((ncomp, N // 2, N, nx), torch.complex64)
        if nyq is None:
            nyq = self.empty((ncomp, N, nx), torch.complex64)
        size_of = self.lib.vps_deposit_fft_zy_workspace_bytes_shared if share_energy else self.lib.vps_deposit_fft_zy_workspace_bytes
        work = self.workspace("fused", size_of(pos.shape[0], N, nx))
        flags |= self._reuse_flag(reuse_sort, pos, vel, rho, N, Lbox, x0, nx, work)
        shared_by = getattr(self, "_energy_shared_by", None)     # token of the momentum call that left the energy z image behind
        self._energy_shared_by = None
        if share_energy and component is None:
            if quantity == MOMENTUM and not (flags & FLAG_REFERENCE_MOMENTUM_BUG):
                flags |= FLAG_SHARE_ENERGY
                self._energy_shared_by = self._fused_token
            elif quantity == ENERGY and (flags & FLAG_REUSE_SORT) and shared_by is not None and reuse_sort is shared_by:
                flags |= FLAG_SHARE_ENERGY
        self._chk(self.lib.vps_deposit_fft_zy(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                              self._ptr(vel, torch.float32), self._ptr(rho, torch.float32),
                                              pos.shape[0], N, float(Lbox), x0, nx, quantity, flags,
                                              self._ptr(spec), self._ptr(nyq), self._ptr(work)))
        return spec, nyq

    def share_energy_fits(self, N, nx):
        """True where the fourth z image of share_energy (4 N^2 nx bytes more: 29 N^2 nx bytes in all for a step with
        three-component quantities) still leaves a tenth of the device free."""
        total = torch.cuda.get_device_properties(self.device).total_memory
        return 30.5 * float(N) ** 2 * nx < 0.88 * total

    @staticmethod
    def _component_mask(quantity, component):
        """component: 0..2 or a collection of them -> the bit mask of VPS_FLAG_COMPONENTS."""
        comps = (component,) if isinstance(component, (int, np.integer)) else tuple(component)
        if quantity == ENERGY or not comps or any(not 0 <= int(c) <= 2 for c in comps) or len(set(comps)) != len(comps):
            raise Exception("component(s) = distinct values 0..2 of a velocity or momentum field")
        mask = 0
        for c in comps:
            mask |= 1 << int(c)
        return mask

    def _reuse_flag(self, reuse_sort, pos, vel, rho, N, Lbox, x0, nx, work, cap=None):
        """FLAG_REUSE_SORT if `reuse_sort` (a token returned by an earlier fused call) proves that the bucketed records
        of that call are still in `work`: several quantities of the SAME particle tensors then sort only once.  The
        token holds WEAK references to the tensors (a freed tensor whose address is recycled cannot pass for the old
        one, and nothing is kept alive by the library) and their in-place modification counters; any mismatch silently
        sorts again.  Records the token of THIS call."""
        # (the library options that decide the record layout are part of the state: a sort made under one value of
        #  sort_atomic / sort_staged / sort_groups cannot be reused under another)
        layout_opts = tuple(sorted((k, v) for k, v in _ffi.OPTIONS.items() if k.startswith("sort_")))
        state = (weakref.ref(pos), weakref.ref(vel), weakref.ref(rho), pos._version, vel._version, rho._version,
                 int(N), float(Lbox), int(x0), int(nx), work.data_ptr(), cap, layout_opts)
        last = getattr(self, "_fused_token", None)
        ok = (reuse_sort is not None and reuse_sort is last
              and all(r() is t for r, t in zip(last[:3], (pos, vel, rho))) and last[3:] == state[3:])
        self._fused_token = state
        return FLAG_REUSE_SORT if ok else 0

    def fused_token(self):
        """Token of the last deposit_fft_zy call (pass it as reuse_sort= to the next one)."""
        return getattr(self, "_fused_token", None)

    def nn_resample(self, pos, payload, axes, x0, nx, want_index=False, out=None):
        """Exact NN of every lattice point axes[0][x0:x0+nx] x axes[1] x axes[2]."""
        self._stream()
        ax = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        C_ = payload.shape[1]
        if out is None:
            out = self.empty((C_, nx, len(ax[1]), len(ax[2])), torch.float32)
        idx = self.empty((nx, len(ax[1]), len(ax[2])), torch.int32) if want_index else None
        kind = self._pos_kind(pos)
        work = self.workspace("nn", self.lib.vps_nn_workspace_bytes(pos.shape[0], kind, nx * len(ax[1]) * len(ax[2])))
        self._chk(self.lib.vps_nn_resample(self.ctx, self._ptr(pos), kind, self._ptr(payload, torch.float32),
                                           pos.shape[0], C_, _ffi.as_dp(ax[0]), len(ax[0]),
                                           _ffi.as_dp(ax[1]), len(ax[1]), _ffi.as_dp(ax[2]), len(ax[2]),
                                           x0, nx, self._ptr(out), self._ptr(idx), self._ptr(work)))
        return out, idx

    def nn_resample_field(self, pos, rhov, axes, x0, nx, Lcell, out=None, want_index=False):
        """Exact-NN resampling of [rho v, rho] with v = rho v / rho, mass = rho Lcell^3 formed in the search's
        epilogue: -> [4, nx, ny, nz] float32 = vx, vy, vz, mass (interp.py:246-277 without a pass over the grid)."""
        self._stream()
        ax = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        if out is None:
            out = self.empty((4, nx, len(ax[1]), len(ax[2])), torch.float32)
        idx = self.empty((nx, len(ax[1]), len(ax[2])), torch.int32) if want_index else None
        kind = self._pos_kind(pos)
        work = self.workspace("nn", self.lib.vps_nn_workspace_bytes(pos.shape[0], kind, nx * len(ax[1]) * len(ax[2])))
        self._chk(self.lib.vps_nn_resample_field(self.ctx, self._ptr(pos), kind, self._ptr(rhov, torch.float32),
                                                 pos.shape[0], _ffi.as_dp(ax[0]), len(ax[0]), _ffi.as_dp(ax[1]), len(ax[1]),
                                                 _ffi.as_dp(ax[2]), len(ax[2]), x0, nx, float(Lcell), self._ptr(out),
                                                 self._ptr(idx), self._ptr(work)))
        return out, idx

    def nn_resample_quantity(self, pos, rhov, axes, x0, nx, Lcell, quantity, flags=0, out=None, want_index=False):
        """Exact-NN resampling of [rho v, rho] straight into the fields the spectrum of `quantity` transforms
        (vps_nn_resample_quantity): VELOCITY -> [3, ...] v; MOMENTUM -> [3, ...] p = v * mass; ENERGY -> [1, ...] mass |v|^2;
        VM -> [4, ...] (= nn_resample_field)."""
        self._stream()
        ax = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        nout = 1 if quantity == ENERGY else (4 if quantity == VM else 3)
        if out is None:
            out = self.empty((nout, nx, len(ax[1]), len(ax[2])), torch.float32)
        idx = self.empty((nx, len(ax[1]), len(ax[2])), torch.int32) if want_index else None
        kind = self._pos_kind(pos)
        work = self.workspace("nn", self.lib.vps_nn_workspace_bytes(pos.shape[0], kind, nx * len(ax[1]) * len(ax[2])))
        self._chk(self.lib.vps_nn_resample_quantity(self.ctx, self._ptr(pos), kind, self._ptr(rhov, torch.float32),
                                                    pos.shape[0], _ffi.as_dp(ax[0]), len(ax[0]), _ffi.as_dp(ax[1]), len(ax[1]),
                                                    _ffi.as_dp(ax[2]), len(ax[2]), x0, nx, float(Lcell), int(quantity), int(flags),
                                                    self._ptr(out), self._ptr(idx), self._ptr(work)))
        return out, idx

    def field_algebra(self, chans, quantity, flags, Lcell):
        self._stream()
        ncell = chans[0].numel()
        self._chk(self.lib.vps_field_algebra(self.ctx, quantity, flags, float(Lcell),
                                             self._ptr(chans, torch.float32), ncell))

    def field_algebra_out(self, chans, quantity, flags, Lcell):
        """Fields of `quantity` from the four channels, which are left untouched: -> [3 | 1 | 4, ...] float32."""
        self._stream()
        ncell = chans[0].numel()
        nout = 1 if quantity == ENERGY else (4 if quantity == VM else 3)
        out = self.empty((nout,) + tuple(chans.shape[1:]), torch.float32)
        self._chk(self.lib.vps_field_algebra_out(self.ctx, quantity, flags, float(Lcell),
                                                 self._ptr(chans, torch.float32), ncell, self._ptr(out)))
        return out

    # -- stage B + C ------------------------------------------------------------
    def fft_supported(self, N):
        return bool(self.lib.vps_fft_supported(int(N)))

    def set_binning(self, N, k2, thr, edge0, inv_spacing):
        k2 = np.ascontiguousarray(k2, dtype=np.float64)
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        self._stream()
        self._chk(self.lib.vps_set_binning(self.ctx, N, _ffi.as_dp(k2), _ffi.as_dp(thr), len(thr) - 1,
                                           float(edge0), float(inv_spacing)))

    def binning_mode(self):
        """How the binning x pass decides shells for the tables set last (vps_binning_mode): 0 general shell walk, 1 mirrored kx
        with float64 k^2 sums, 2 mirrored kx with integer ix^2 + iy^2 + iz^2 (the same shells, checked by vps_set_binning)."""
        return int(self.lib.vps_binning_mode(self.ctx))

    def binning_only(self):
        """Context manager: inside it the y passes skip rows whose modes all lie beyond the last shell edge of the
        binning tables set last (vps_set_bin_only) -- for outputs that go straight into the binning x pass."""
        k = self

        class _Scope:
            def __enter__(self_inner):
                k._chk(k.lib.vps_set_bin_only(k.ctx, 1))

            def __exit__(self_inner, *exc):
                k._chk(k.lib.vps_set_bin_only(k.ctx, 0))
                return False
        return _Scope()

    def set_window(self, N, table):
        """1/W^2 axis table (window_inv2_axis) for the binning x pass, or None to switch deconvolution off."""
        self._stream()
        if table is None:
            self._chk(self.lib.vps_set_window(self.ctx, int(N), None))
            return
        t = np.ascontiguousarray(table, dtype=np.float32)
        if t.shape != (N,):
            raise _ffi.VpsError("window table must have N entries")
        self._chk(self.lib.vps_set_window(self.ctx, int(N), t.ctypes.data_as(C.c_void_p)))

    def assign_expand(self, pos, payload, N, Lbox, assignment):
        """CIC / TSC: (pos [np,3], payload [np,C]) -> weighted sub-particles at cell centres
        (pos' float32 [np*S,3], payload' [np*S,C]); deposit them with `deposit`."""
        self._stream()
        order = ASSIGNMENT_ORDER[assignment]
        if order == 1:
            return pos, payload
        S = order ** 3
        n, C_ = payload.shape
        pos_o = self.empty((n * S, 3), torch.float32)
        pay_o = self.empty((n * S, C_), torch.float32)
        self._chk(self.lib.vps_assign_expand(self.ctx, self._ptr(pos), self._pos_kind(pos), self._ptr(payload, torch.float32),
                                             n, C_, int(N), float(Lbox), order, self._ptr(pos_o), self._ptr(pay_o)))
        return pos_o, pay_o

    def fft_zy(self, field, N, nx, spec=None, nyq=None, weight=None):
        """field [nx,N,N] float32 (times `weight`, same shape, if given) -> spec [N/2,N,nx], nyq [N,nx] (complex64)."""
        self._stream()
        if spec is None:
            spec = self.empty((N // 2, N, nx), torch.complex64)
        if nyq is None:
            nyq = self.empty((N, nx), torch.complex64)
        work = self.workspace("fft", self.lib.vps_fft_workspace_bytes(N, nx))
        self._chk(self.lib.vps_fft_zy_weighted(self.ctx, N, nx, self._ptr(field, torch.float32),
                                               self._ptr(weight, torch.float32), self._ptr(spec), self._ptr(nyq),
                                               self._ptr(work)))
        return spec, nyq

    # -- split form for the chunked slab exchange --------------------------------
    def zimage_elems(self, N, nx):
        return int(self.lib.vps_fft_zimage_bytes(int(N), int(nx))) // 8

    def fft_z(self, field, N, nx, weight=None, zimg=None):
        """z pass only: field [nx,N,N] float32 (* weight) -> z image (flat complex64: B[x][kz][y] | BN[x][y])."""
        self._stream()
        if zimg is None:
            zimg = self.empty((self.zimage_elems(N, nx),), torch.complex64)
        self._chk(self.lib.vps_fft_z(self.ctx, N, nx, self._ptr(field, torch.float32), self._ptr(weight, torch.float32),
                                     self._ptr(zimg, torch.complex64)))
        return zimg

    def count_in_slab(self, pos, N, Lbox, x0, nx):
        """Particles whose (bit-exact) cell lies in the x-slab [x0, x0 + nx) (vps_count_in_slab)."""
        self._stream()
        n = int(self.lib.vps_count_in_slab(self.ctx, self._ptr(pos), self._pos_kind(pos), pos.shape[0], int(N), float(Lbox),
                                           int(x0), int(nx)))
        if n < 0:
            self._chk(n)
        return n

    def deposit_fft_z(self, pos, vel, rho, N, Lbox, x0, nx, quantity, flags=0, zimg=None, reuse_sort=None, slab_particles=None,
                      component=None):
        """Fused deposit + field algebra + z pass -> z images [ncomp, zimage_elems] (ncomp = 1 for ENERGY, else 3).
        slab_particles: a bound on the particles inside the slab (count_in_slab): the sort workspace is then sized for it,
        not for all of a replicated particle set.  component: as in deposit_fft_zy."""
        self._stream()
        ncomp = 1 if quantity == ENERGY else 3
        if component is not None:
            mask = self._component_mask(quantity, component)
            flags |= mask << 4
            ncomp = bin(mask).count("1")
        if zimg is None:
            zimg = self.empty((ncomp, self.zimage_elems(N, nx)), torch.complex64)
        if slab_particles is None:
            work = self.workspace("fused_z", self.lib.vps_deposit_fft_z_workspace_bytes(pos.shape[0], N, nx))
        else:
            work = self.workspace("fused_z", self.lib.vps_deposit_fft_z_workspace_bytes_slab(pos.shape[0], int(slab_particles), N, nx))
        flags |= self._reuse_flag(reuse_sort, pos, vel, rho, N, Lbox, x0, nx, work, slab_particles)
        if slab_particles is None:
            self._chk(self.lib.vps_deposit_fft_z(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                                 self._ptr(vel, torch.float32), self._ptr(rho, torch.float32),
                                                 pos.shape[0], N, float(Lbox), x0, nx, quantity, flags,
                                                 self._ptr(zimg, torch.complex64), self._ptr(work)))
        else:
            self._chk(self.lib.vps_deposit_fft_z_slab(self.ctx, self._ptr(pos), self._pos_kind(pos),
                                                      self._ptr(vel, torch.float32), self._ptr(rho, torch.float32),
                                                      pos.shape[0], int(slab_particles), N, float(Lbox), x0, nx, quantity, flags,
                                                      self._ptr(zimg, torch.complex64), self._ptr(work)))
        return zimg

    def y_chunk_elems(self, N, nx, G, nchunks, chunk):
        n = int(self.lib.vps_fft_y_chunk_elems(int(N), int(nx), int(G), int(nchunks), int(chunk)))
        if n < 0:
            raise _ffi.VpsError("%d ranks x %d chunks do not divide N/2 = %d" % (G, nchunks, N // 2))
        return n

    def y_packed(self, N):
        """True when fft_y_chunk packs rows now: inside a `binning_only` scope whose tables have a row cut for N."""
        return bool(self.lib.vps_fft_y_packed(self.ctx, int(N)))

    def chunk_block(self, N, nx, G, nchunks, chunk, packed):
        """Elements of ONE destination's block of chunk `chunk` (include/vps_hip.h: vps_fft_y)."""
        n = int(self.lib.vps_fft_y_chunk_block(self.ctx, int(N), int(nx), int(G), int(nchunks), int(chunk), int(bool(packed))))
        if n < 0:
            why = self.lib.vps_last_error(self.ctx).decode() if n < -1 else "bad arguments"
            raise _ffi.VpsError("vps_fft_y_chunk_block(N=%d, nx=%d, G=%d, nchunks=%d, chunk=%d, packed=%d) failed (%d): %s"
                                % (N, nx, G, nchunks, chunk, bool(packed), n, why))
        return n

    def fft_y_chunk(self, zimg, N, nx, G, nchunks, chunk, out=None):
        """y pass of one kz chunk of a z image -> the send buffer of an equal-split all-to-all over G ranks:
        [h][ F_zy[kz in h's planes of the chunk][row][x] | (last chunk) Nyquist rows of h ] (flat complex64),
        G * chunk_block(..., packed=y_packed(N)) elements (ask in the same scope)."""
        self._stream()
        n = G * self.chunk_block(N, nx, G, nchunks, chunk, self.y_packed(N))
        if out is None:
            out = self.empty((n,), torch.complex64)
        self._chk(self.lib.vps_fft_y(self.ctx, N, nx, self._ptr(zimg, torch.complex64), G, nchunks, chunk,
                                     self._ptr(out, torch.complex64)))
        return out

    def fft_x_bin(self, lines, N, nlines, line0, kz0, nseg, seg_stride, psum, nsample, count=True):
        """x pass + shell sums into psum; with `count` also the shell counts into nsample."""
        self._stream()
        self._chk(self.lib.vps_fft_x(self.ctx, N, nlines, line0, kz0, self._ptr(lines, torch.complex64), nseg,
                                     seg_stride, 0 if count else 3, self._ptr(psum, torch.float64),
                                     self._ptr(nsample, torch.int64), None))

    def fft_x_bin_multi(self, comps, N, nlines, line0, kz0, nseg, seg_stride, psum, nsample, count=True):
        """x pass of up to three component spectra, |F|^2 summed over the components, binned once."""
        self._stream()
        ptrs = (C.c_void_p * len(comps))(*[self._ptr(c, torch.complex64).value for c in comps])
        self._chk(self.lib.vps_fft_x_bin(self.ctx, N, nlines, line0, kz0, ptrs, len(comps), nseg, seg_stride,
                                         1 if count else 0, self._ptr(psum, torch.float64),
                                         self._ptr(nsample, torch.int64)))

    def fft_x_bin_chunk(self, comps, N, nx, G, nchunks, chunk, rank, packed, psum, nsample, count=True):
        """Binning x pass of one RECEIVED chunk of the slab exchange (the G blocks per component, fft_y_chunk's layout; the
        last chunk's Nyquist-plane rows included), up to three components summed before the shell search."""
        self._stream()
        ptrs = (C.c_void_p * len(comps))(*[self._ptr(c, torch.complex64).value for c in comps])
        self._chk(self.lib.vps_fft_x_bin_chunk(self.ctx, int(N), int(nx), int(G), int(nchunks), int(chunk), int(rank),
                                               int(bool(packed)), ptrs, len(comps), 1 if count else 0,
                                               self._ptr(psum, torch.float64), self._ptr(nsample, torch.int64)))

    # -- the slab exchange inside the library (RCCL behind the C ABI) -------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId: made on one rank, handed to all (the host moves them)."""
        buf = C.create_string_buffer(128)
        lib = _ffi.lib()
        rc = lib.vps_comm_unique_id(buf)
        if rc != 0:
            raise _ffi.VpsError("vps_comm_unique_id failed (%d): %s" % (rc, lib.vps_last_error(None).decode()))
        return buf.raw

    def comm_create(self, rank, world, uid):
        """Collective over the job's ranks: an RCCL communicator on this context's device."""
        self._chk(self.lib.vps_comm_create(self.ctx, int(rank), int(world), bytes(uid)))
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        self._chk(self.lib.vps_comm_destroy(self.ctx))

    def spectrum_zimages(self, zimgs, N, nx, nchunks, psum, nsample, count=True):
        """vps_spectrum_zimages: chunked y pass -> grouped ncclSend / ncclRecv -> binning x pass of up to three z images,
        all inside the library (its own communication stream and events)."""
        self._stream()
        G = self.comm_world
        ptrs = (C.c_void_p * len(zimgs))(*[self._ptr(z, torch.complex64).value for z in zimgs])
        work = self.workspace("exchange", self.lib.vps_spectrum_zimages_workspace_bytes(int(N), int(nx), G, int(nchunks), len(zimgs)))
        self._chk(self.lib.vps_spectrum_zimages(self.ctx, int(N), int(nx), ptrs, len(zimgs), int(nchunks), self._ptr(work),
                                                1 if count else 0, self._ptr(psum, torch.float64), self._ptr(nsample, torch.int64)))

    def allreduce_shells(self, psum, nsample):
        self._stream()
        self._chk(self.lib.vps_allreduce_shells(self.ctx, self._ptr(psum, torch.float64), self._ptr(nsample, torch.int64),
                                                int(psum.numel())))

    def fft_x_write(self, lines, N, nlines, nseg, seg_stride, out):
        self._stream()
        self._chk(self.lib.vps_fft_x(self.ctx, N, nlines, 0, 0, self._ptr(lines, torch.complex64), nseg,
                                     seg_stride, 1, None, None, self._ptr(out, torch.complex64)))

    def rfft3(self, field, N):
        """Half spectrum F[kz][ky][kx], kz <= N/2, of a full [N,N,N] float32 field."""
        self._stream()
        out = self.empty((N // 2 + 1, N, N), torch.complex64)
        work = self.workspace("power", self.lib.vps_power_workspace_bytes(N))
        self._chk(self.lib.vps_rfft3(self.ctx, N, self._ptr(field, torch.float32), self._ptr(work),
                                     self._ptr(out)))
        return out

    def power_grid(self, fields, N):
        """sum over fields of |F[kz][ky][kx]|^2, kz <= N/2, float32."""
        self._stream()
        out = self.zeros((N // 2 + 1, N, N), torch.float32)
        work = self.workspace("power", self.lib.vps_power_workspace_bytes(N))
        for f in fields:
            self._chk(self.lib.vps_power_grid(self.ctx, N, self._ptr(f, torch.float32), self._ptr(work),
                                              self._ptr(out)))
        return out

    def pair_k(self, kx, ky, kz):
        self._stream()
        ax = [np.ascontiguousarray(a, dtype=np.float64) for a in (kx, ky, kz)]
        N = len(ax[0])
        out = self.empty((N * N * N,), torch.float64)
        self._chk(self.lib.vps_pair_k(self.ctx, N, _ffi.as_dp(ax[0]), _ffi.as_dp(ax[1]), _ffi.as_dp(ax[2]),
                                      self._ptr(out)))
        return out

    def hist_pairs(self, k, w, edges):
        self._stream()
        edges = np.ascontiguousarray(edges, dtype=np.float64)
        nb = len(edges) - 1
        psum = self.zeros((nb,), torch.float64)
        ns = self.zeros((nb,), torch.int64)
        self._chk(self.lib.vps_hist_pairs(self.ctx, self._ptr(k, torch.float64),
                                          self._ptr(w, torch.float64) if w is not None else None,
                                          k.numel(), _ffi.as_dp(edges), nb, self._ptr(psum), self._ptr(ns)))
        return psum, ns


_default_kernels = {}


def default_kernels(device=None):
    """Process-wide HipKernels for a device (created on first use)."""
    if device is None:
        if not torch.cuda.is_available():
            raise _ffi.VpsError("no HIP device is visible to this process; the vpower device path "
                                "needs an MI355X (there is no CPU fallback)")
        device = torch.cuda.current_device()
    idx = device if isinstance(device, int) else torch.device(device).index
    k = _default_kernels.get(idx)
    if k is None:
        k = HipKernels(idx)
        _default_kernels[idx] = k
    return k


# --------------------------------------------------------------------------- #
# the pipeline
# --------------------------------------------------------------------------- #
class SlabComm:
    """Rank / world bookkeeping and the two collectives the path needs."""

    def __init__(self, group=None, enabled=None):
        import torch.distributed as dist
        self.dist = dist
        if enabled is None:
            enabled = dist.is_available() and dist.is_initialized()
        self.enabled = bool(enabled)
        self.group = group
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.backend = dist.get_backend(group) if self.enabled else None
        # VPS_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (exercises the RCCL
        # plumbing -- complex views, async work handles -- on a single GPU; tests only)
        self.force = self.enabled and os.environ.get("VPS_FORCE_COLLECTIVES") == "1"

    def all_to_all_start(self, send):
        """Begin an equal-split all-to-all along dim 0 of a contiguous tensor.  Returns
        (recv, work); `all_to_all_finish` makes `recv` usable on the current stream.  With
        RCCL the exchange runs on the communicator's stream, so kernels issued in between
        (the z/y passes of the next field, the x pass of the previous one) overlap it."""
        if self.world == 1 and not self.force:
            return send, None
        if send.is_cuda and self.backend != "nccl":
            # gloo rehearsal on a GPU box: stage through host memory (synchronous)
            h = send.cpu()
            r = torch.empty_like(h)
            self.dist.all_to_all_single(r, h, group=self.group)
            return r.to(send.device), None
        if send.is_complex():
            s = torch.view_as_real(send)
            recv = torch.empty_like(send)
            r = torch.view_as_real(recv)
            if not send.is_cuda:           # gloo wants plain contiguous real tensors
                s = s.contiguous()
                r = torch.empty_like(s)
                recv = torch.view_as_complex(r)
            work = self.dist.all_to_all_single(r, s, group=self.group, async_op=True)
            return recv, (work, s)        # keep the send view alive until the wait
        recv = torch.empty_like(send)
        work = self.dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return recv, (work, send)

    @staticmethod
    def all_to_all_finish(handle):
        recv, work = handle
        if work is not None:
            work[0].wait()
        return recv

    def all_to_all(self, send):
        """Blocking form of all_to_all_start / all_to_all_finish."""
        return self.all_to_all_finish(self.all_to_all_start(send))

    def all_reduce_sum(self, t):
        if self.world == 1 and not self.force:
            return t
        if t.is_cuda and self.backend != "nccl":
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
            return t
        self.dist.all_reduce(t, group=self.group)
        return t


class FieldComm(SlabComm):
    """Field-parallel ranks: every rank holds WHOLE grids and transforms a share of the scalar fields of a step (the three
    components of a vector quantity, the energy field, several quantities); nothing but the shell tables crosses the node.
    For grids that fit one GPU's 288 GB this is the decomposition that suits point-to-point xGMI: the slab decomposition sends
    the whole half spectrum through the links once per field (at two ranks through ONE link), this one sends nbins numbers.
    To the pipeline the rank is a one-rank slab (world = 1: nx = N, no exchange); `field_rank` / `field_world` say which
    fields are this rank's (`mine`), and the closing reductions ADD the shell sums of all ranks and take the shell counts from
    whichever ranks counted (MAX: they depend on the mode lattice alone, so every rank that counted holds the same numbers)."""

    def __init__(self, group=None, enabled=None):
        super().__init__(group=group, enabled=enabled)
        self.field_rank, self.field_world = self.rank, self.world
        self.rank, self.world, self.force = 0, 1, False

    @staticmethod
    def units(quantities):
        """The scalar fields of a step, in dealing order: (quantity, component) -- component None for the energy field."""
        return [(q, c) for q in quantities for c in ((None,) if q == "energy" else (0, 1, 2))]

    def mine(self, quantities):
        """This rank's fields: CONTIGUOUS blocks of the dealing order (the first n mod W ranks take one more), so that a rank's
        fields mostly belong to one quantity and go through one multi-component launch (shared rho round, one shell search)."""
        u = self.units(quantities)
        W, r = self.field_world, self.field_rank
        base, extra = divmod(len(u), W)
        lo = r * base + min(r, extra)
        return u[lo: lo + base + (1 if r < extra else 0)]

    def all_reduce_sum(self, t):
        if self.field_world == 1:
            return t
        op = self.dist.ReduceOp.SUM if t.is_floating_point() else self.dist.ReduceOp.MAX
        if t.is_cuda and self.backend != "nccl":
            h = t.cpu()
            self.dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return t
        self.dist.all_reduce(t, op=op, group=self.group)
        return t


class LibraryComm(SlabComm):
    """The same two collectives, run INSIDE libvps_hip.so over RCCL (vps_comm_create / vps_spectrum_zimages /
    vps_allreduce_shells): what a host without torch.distributed uses.  `PowerPipeline` hands whole groups of z images to
    the library, which owns the chunk pipeline (communication stream, events).  rank / world / uid come from the host --
    here: from an initialised torch.distributed group (only to move the 128-byte id), or explicitly."""

    def __init__(self, kernels, rank=None, world=None, uid=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        if rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                rank, world = 0, 1
            else:
                rank, world = dist.get_rank(group), dist.get_world_size(group)
        if uid is None:
            box = [HipKernels.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
        self.rank, self.world = int(rank), int(world)
        self.enabled, self.force, self.backend = True, True, "library"
        self.k = kernels
        kernels.comm_create(self.rank, self.world, uid)

    def all_reduce_shells(self, psum, nsample):
        self.k.allreduce_shells(psum, nsample)


class PowerPipeline:
    """P(k) of real fields on an N^3 grid of box length Lbox, slab-decomposed over the
    ranks of `comm` (default: the initialised torch.distributed world, else 1 rank)."""

    def __init__(self, Nsize, Lbox, kernels=None, comm=None, flavour="library",
                 kmin=None, kmax=None, kres=None, deconvolve=None):
        self.N = int(Nsize)
        self.Lbox = float(Lbox)
        self.Lcell = self.Lbox / self.N
        self.k = kernels if kernels is not None else default_kernels()
        self.comm = comm if comm is not None else SlabComm()
        G = self.comm.world
        if not self.k.fft_supported(self.N):
            raise Exception("Nsize=%d is not supported by the device FFT (powers of two 16..4096, 96, 192, 384, 768, 1536, 250, 500, 1000, 2000)" % self.N)
        if self.N % G or (self.N // 2) % G:
            raise Exception("Nsize/2=%d must be divisible by the number of ranks %d" % (self.N // 2, G))
        self.nx = self.N // G
        self.x0 = self.comm.rank * self.nx
        self.flavour = flavour
        # defaults of BoxField.spctrm (interp.py:565-570) / main() (parallel_optimized.py:430)
        self.kmin = 2 * np.pi / self.Lbox if kmin is None else kmin
        self.kmax = np.pi / self.Lcell if kmax is None else kmax
        self.kres = self.kmin if kres is None else kres
        self.centers, self.edges = bin_edges(self.kmin, self.kmax, self.kres, flavour)
        self.nbins = len(self.centers)
        ks = k_axis(self.Lbox, self.N)
        self.k2 = ks * ks
        self.thr = sqrt_thresholds(self.edges)
        spacing = (self.edges[-1] - self.edges[0]) / self.nbins
        self._binning = (self.N, self.k2, self.thr, float(self.edges[0]), 1.0 / spacing)
        self.const = (self.Lbox / (2 * np.pi)) ** 1.5 / self.N ** 3   # interp.py:1381
        # deconvolve = "ngp" | "cic" | "tsc": divide every |F(k)|^2 by the assignment window W(k)^2 while binning
        self.window = None if deconvolve is None else window_inv2_axis(self.N, deconvolve)
        # Several ranks: every scalar field crosses the node as ONE message per pair of ranks (its kz rows with the
        # Nyquist-plane rows riding behind them), cut into `nchunks` kz chunks so that the exchange of a chunk
        # overlaps the y pass of the next and the x pass of the previous one.
        self.chunked = G > 1 or self.comm.force or os.environ.get("VPS_CHUNKED") == "1"
        self.nchunks = self._pick_chunks()

    def _pick_chunks(self):
        """kz chunks per field: VPS_A2A_CHUNKS (default 4), lowered until it divides the rank's kz rows and a
        chunk's pair message stays above 2 MiB (smaller messages are latency-bound on xGMI)."""
        G = self.comm.world
        nkz = self.N // 2 // G
        try:
            c = max(1, int(os.environ.get("VPS_A2A_CHUNKS", "4")))
        except ValueError:
            c = 4
        c = min(c, nkz)
        while c > 1 and (nkz % c or (os.environ.get("VPS_A2A_CHUNKS") is None
                                     and nkz // c * self.N * self.nx * 8 < (2 << 20))):
            c -= 1
        return max(c, 1)

    def kept_row_fraction(self, kz_lo=0, kz_hi=None):
        """Fraction of the (ky, kz) rows of kz planes [kz_lo, kz_hi) that the binning passes keep: a row whose modes all
        lie beyond the last shell edge -- fl(ky^2 + kz^2) >= thr[nbins] -- is neither stored by the y pass nor read by
        the x pass (same rule and 16-row rounding as vps_set_binning's cut table; 1.0 for N < 128)."""
        N = self.N
        kz_hi = N // 2 if kz_hi is None else kz_hi
        if N < 128:
            return 1.0
        k2h = self.k2[: N // 2 + 1]
        kept = 0
        for kz in range(kz_lo, kz_hi):
            ok = np.nonzero(~((k2h + self.k2[kz]) >= self.thr[-1]))[0]
            if len(ok):
                kc = min(int(ok[-1]) | 15, N // 2)
                kept += min(2 * kc + 1, N)
        return kept / float(N * max(kz_hi - kz_lo, 1))

    def prepare(self):
        """Upload this pipeline's binning (and window) tables; `self.k.binning_only()` scopes refer to them."""
        self.k.set_binning(*self._binning)
        self._set_window()

    def _bin_scope(self):
        import contextlib
        return self.k.binning_only() if hasattr(self.k, "binning_only") else contextlib.nullcontext()

    # -- the chunk pipeline of the slab exchange ------------------------------------------------------------------------
    # A JOB is one kz chunk of up to three z images (the components of a vector quantity share one binning launch): y pass
    # into fresh send buffers -> all-to-all (asynchronous) -> x pass + shell sums of the received blocks.  Jobs are started
    # in order and at most `inflight_max` of them (VPS_A2A_INFLIGHT, default 2) are between their start and their x pass:
    # job j + 1 crosses the node while job j is binned, and only TWO chunks' send / receive buffers are alive, whatever the
    # chunk count -- not the whole field twice over.  The jobs of several quantities form ONE sequence: the next quantity's
    # deposit + z pass is enqueued while the last chunks of the previous one are still travelling.
    def inflight_max(self):
        try:
            return max(1, int(os.environ.get("VPS_A2A_INFLIGHT", "2")))
        except ValueError:
            return 2

    def _stamp(self):
        """Timing event on the current stream (instrumented steps only: `self.instr` is a list)."""
        if getattr(self, "instr", None) is None or not torch.cuda.is_available():
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _start_chunk(self, comps, c, packed):
        N, nx, G = self.N, self.nx, self.comm.world
        sends = [self.k.fft_y_chunk(z, N, nx, G, self.nchunks, c) for z in comps]
        t0 = self._stamp()
        return {"handles": [self.comm.all_to_all_start(s_) for s_ in sends], "chunk": c, "packed": packed, "t0": t0}

    def _finish_chunk(self, job, psum, nsample, count):
        N, nx, G, r = self.N, self.nx, self.comm.world, self.comm.rank
        t1 = self._stamp()
        recvs = [self.comm.all_to_all_finish(h) for h in job["handles"]]
        t2 = self._stamp()
        if t1 is not None:
            self.instr.append((job["t0"], t1, t2))      # exchange started | x pass ready to go | blocks there
        job["handles"] = None
        self.k.fft_x_bin_chunk(recvs, N, nx, G, self.nchunks, job["chunk"], r, job["packed"], psum, nsample, count=count)

    def exchange_times(self):
        """(exposed_ms, span_ms) of the instrumented jobs so far: the time the compute stream stood still waiting for blocks,
        and -- meaningful when the jobs ran one at a time (inflight_max = 1) -- the time from a job's exchange start to its
        arrival.  Synchronises."""
        torch.cuda.synchronize()
        ex = sum(a.elapsed_time(b) for _, a, b in self.instr)
        span = sum(a.elapsed_time(b) for a, _, b in self.instr)
        return ex, span

    def pipelined_quantities(self, producers, accumulators, counts=None):
        """The jobs of several quantities of ONE particle set as one bounded pipeline: producers[i]() enqueues quantity i's
        deposit + z pass and returns its z images (one per component, which may live in the SAME buffer for every quantity:
        by then all y passes of quantity i - 1 have been enqueued); its chunks are binned into accumulators[i] =
        (psum, nsample).  counts[i] (default True): whether quantity i's shell counts are accumulated (first group only).
        What `bench.py` runs on several ranks (torch transport); one quantity: `accumulate_zimages`."""
        import collections
        group = 1 if os.environ.get("VPS_X_PER_COMPONENT") == "1" else 3
        k, C_, maxin = self.k, self.nchunks, self.inflight_max()
        inflight = collections.deque()

        def finish_oldest():
            job, qi, cnt = inflight.popleft()
            self._finish_chunk(job, accumulators[qi][0], accumulators[qi][1], cnt)
        self.prepare()
        for qi, produce in enumerate(producers):
            zimgs = produce()
            cnt = True if counts is None else bool(counts[qi])
            for i in range(0, len(zimgs), group):
                for c in range(C_):
                    while len(inflight) >= maxin:
                        finish_oldest()
                    with self._bin_scope():   # the blocks carry only the rows a shell can reach
                        packed = k.y_packed(self.N) if hasattr(k, "y_packed") else False
                        inflight.append((self._start_chunk(zimgs[i:i + group], c, packed), qi, cnt and i == 0))
        while inflight:
            finish_oldest()
        return accumulators

    def start_zimages(self, zimgs):
        """Every chunk of every group of `zimgs` started at once (y pass + all-to-all); `finish_zimages` bins them.  All
        send / receive buffers are alive in between: the bounded form is `accumulate_zimages` / `pipelined_quantities`."""
        group = 1 if os.environ.get("VPS_X_PER_COMPONENT") == "1" else 3
        self.prepare()
        groups = []
        with self._bin_scope():
            packed = self.k.y_packed(self.N) if hasattr(self.k, "y_packed") else False
            for i in range(0, len(zimgs), group):
                groups.append([self._start_chunk(zimgs[i:i + group], c, packed) for c in range(self.nchunks)])
        return {"packed": packed, "groups": groups}

    def finish_zimages(self, started, psum=None, nsample=None, count=True):
        self.prepare()
        if psum is None:
            psum, nsample = self.new_accumulators()
        for i, jobs in enumerate(started["groups"]):
            for c in range(self.nchunks):
                self._finish_chunk(jobs[c], psum, nsample, count and i == 0)
                jobs[c] = None
        return psum, nsample

    def accumulate_zimages(self, zimgs, psum=None, nsample=None, count=True):
        """x-side of the transform for z images (HipKernels.fft_z / deposit_fft_z): per kz chunk, y pass into the
        send buffer, all-to-all, x pass + shell sums, two chunks in flight (`pipelined_quantities`)."""
        group = 1 if os.environ.get("VPS_X_PER_COMPONENT") == "1" else 3
        if psum is None:
            psum, nsample = self.new_accumulators()
        if isinstance(self.comm, LibraryComm):     # the whole chunk pipeline runs inside the library (RCCL, two chunk slots)
            self.prepare()
            for i in range(0, len(zimgs), group):
                self.k.spectrum_zimages(zimgs[i:i + group], self.N, self.nx, self.nchunks, psum, nsample, count=count and i == 0)
            return psum, nsample
        self.pipelined_quantities([lambda: zimgs], [(psum, nsample)], counts=[count])
        return psum, nsample

    # -- stage B + C on one or more real fields of this rank's slab ---------------
    def accumulate(self, fields, psum=None, nsample=None, count=True, weight=None):
        """Add sum_w |F|^2 of every field ([nx,N,N] float32) into the shell sums, and (with
        `count`) the number of modes per shell into nsample -- once, on the first field: the
        reference histograms the component-summed P grid once (interp.py:1474-1477).
        Returns LOCAL (this rank's) accumulators; call `finish` to reduce them."""
        N, nx, G, r = self.N, self.nx, self.comm.world, self.comm.rank
        k = self.k
        k.set_binning(*self._binning)
        self._set_window()
        if psum is None:
            psum, nsample = self.new_accumulators()
        if self.chunked:
            zimgs = [k.fft_z(f, N, nx, weight=weight) for f in fields]
            return self.accumulate_zimages(zimgs, psum, nsample, count)
        nkz = N // 2 // G      # kz rows per rank after the exchange
        nky = N // G           # Nyquist-plane ky rows per rank
        # one rank: no exchange -- z/y passes of every field, then the x passes
        pending = []
        with self._bin_scope():
            for f in fields:
                spec, nyq = k.fft_zy(f, N, nx) if weight is None else k.fft_zy(f, N, nx, weight=weight)
                pending.append((self.comm.all_to_all_start(spec), self.comm.all_to_all_start(nyq)))
        self._bin_exchanged(pending, psum, nsample, count)
        return psum, nsample

    def _bin_exchanged(self, pending, psum, nsample, count):
        """x pass + shell sums of exchanged spectra.  Up to three components at a time go through ONE
        launch that sums their |F|^2 before the shell search (what the reference does on the grid,
        interp.py:1386, 1474-1477); VPS_X_PER_COMPONENT=1 bins every component on its own instead."""
        N, nx, G, r = self.N, self.nx, self.comm.world, self.comm.rank
        k = self.k
        nkz, nky = N // 2 // G, N // G
        done = [(self.comm.all_to_all_finish(hs), self.comm.all_to_all_finish(hq)) for hs, hq in pending]
        group = 1 if os.environ.get("VPS_X_PER_COMPONENT") == "1" else 3
        for i in range(0, len(done), group):
            c = count and i == 0
            specs = [d[0] for d in done[i:i + group]]
            nyqs = [d[1] for d in done[i:i + group]]
            k.fft_x_bin_multi(specs, N, nkz * N, 0, r * nkz, G, nkz * N * nx, psum, nsample, count=c)
            k.fft_x_bin_multi(nyqs, N, nky, r * nky, N // 2, G, nky * nx, psum, nsample, count=c)

    def accumulate_spectra(self, spec, nyq, psum=None, nsample=None, count=True):
        """Like `accumulate`, for fields that already went through the z and y passes
        (spec [ncomp, N/2, N, nx], nyq [ncomp, N, nx], e.g. from HipKernels.deposit_fft_zy)."""
        N, nx, G, r = self.N, self.nx, self.comm.world, self.comm.rank
        if self.chunked:
            raise Exception("several ranks exchange z images chunk by chunk: use HipKernels.deposit_fft_z / fft_z "
                            "and PowerPipeline.accumulate_zimages")
        k = self.k
        k.set_binning(*self._binning)
        self._set_window()
        if psum is None:
            psum, nsample = self.new_accumulators()
        nkz, nky = N // 2 // G, N // G
        pending = [(self.comm.all_to_all_start(spec[i]), self.comm.all_to_all_start(nyq[i]))
                   for i in range(spec.shape[0])]
        self._bin_exchanged(pending, psum, nsample, count)
        return psum, nsample

    def _set_window(self):
        if hasattr(self.k, "set_window"):
            self.k.set_window(self.N, self.window)
        elif self.window is not None:
            raise Exception("this kernel set has no window deconvolution")

    def new_accumulators(self):
        """Zeroed shell sums (float64) and shell counts (int64) as two views of ONE device buffer, so that a
        step clears them with one fill and `finish` brings them to the host with one copy."""
        buf = self.k.zeros((2 * self.nbins,), torch.float64)
        self._acc_buf = buf
        return buf[: self.nbins], buf[self.nbins:].view(torch.int64)

    def finish(self, psum, nsample, buf=None):
        """Reduce over ranks and build the reference's (nbins,4) table
        [centre, P, Psum, Nsample] (interp.py:1478-1480 / parallel_optimized.py:185-188),
        before the 4 pi k^2 factor.  buf: the ONE float64 buffer of 2 nbins words that psum and nsample are the two halves
        of (as `new_accumulators` lays them out), if the caller keeps several such pairs: one device-to-host copy."""
        if isinstance(self.comm, LibraryComm):
            self.comm.all_reduce_shells(psum, nsample)
        else:
            self.comm.all_reduce_sum(psum)
            self.comm.all_reduce_sum(nsample)
        if buf is None:
            buf = getattr(self, "_acc_buf", None)
        if (buf is not None and psum.data_ptr() == buf.data_ptr()
                and nsample.data_ptr() == buf.data_ptr() + 8 * self.nbins):
            host = buf.cpu()                         # one device-to-host copy for both accumulators
            ps = host[: self.nbins].numpy() * (0.5 * self.const ** 2)
            ns = host[self.nbins:].view(torch.int64).numpy()
        else:
            ps = psum.cpu().numpy() * (0.5 * self.const ** 2)
            ns = nsample.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            P = ps / ns
        if self.flavour == "library":
            P[ns == 0] = 0                      # interp.py:1479
        return np.column_stack((self.centers, P, ps, ns.astype(np.float64)))

    def spectrum(self, fields, weight=None):
        """P(k) table of the fields (each multiplied cell by cell by `weight` if given), times 4 pi k^2."""
        psum, nsample = self.accumulate(fields, weight=weight)
        tab = self.finish(psum, nsample)
        tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2   # interp.py:590 / parallel_optimized.py:434
        return tab
