"""vpower.interp on MI355X: the function surface of the reference's vpower/interp.py for
the particles -> grid -> FFT -> P(k) path, executed by the HIP kernels of libvps_hip.so.

Names, argument meaning and error behaviour follow the reference (file:line cited per
function) so that `from vpower.interp import *` user code moves over unchanged: numpy
arrays in, numpy arrays / `BoxField` / `PowerSpectrum` out.  Grids stay resident in HBM
between the steps of a pipeline (a `BoxField` made by `ann_interp_to_field` or
`deposit_to_field` carries its device buffers), numpy views are materialised lazily.

Arithmetic is float32 on the device (the reference library is float64; its MPI script is
float32/complex64): spectra agree with the reference within the tolerances stated in
DESIGN.md, cell indices, neighbour indices and shell counts agree bit for bit.

Out of scope here: Voxelize smoothing, brick decomposition, the ANN
subprocess path, plotting, and the folding classes -- folding is mathematically a full
N^3 transform, which is what the device path computes directly.
"""
from __future__ import annotations

import numpy as np
import torch

try:  # `from vpower.interp import *` and the reference's bare `import interp` both work
    from . import device as _dev
    from .spctrm import PowerSpectrum
except ImportError:  # pragma: no cover
    import device as _dev
    from spctrm import PowerSpectrum

__all__ = [
    "load_snapshot", "GasParticles", "BoxField", "deposit_to_grid", "ann_interpolate",
    "make_grid_coords", "check_conservation", "_vector_power", "_scalar_power", "_pair_power",
    "_hist_sample", "PowerSpectrum", "REFERENCE_COMPAT",
]

# Reproduce reference defects that change numbers:
#   momentum_bug: BoxField.momentum_power uses vx for all three components (interp.py:523-525)
REFERENCE_COMPAT = {"momentum_bug": False}


def _kernels():
    return _dev.default_kernels()


def _pos_tensor(k, pos):
    pos = np.asarray(pos)
    if pos.dtype != np.float32:
        pos = pos.astype(np.float64, copy=False)
    if pos.ndim != 2 or pos.shape[1] != 3:
        raise Exception("positions must have shape (Nparticles, 3)")
    return k.to_device(pos)


# --------------------------------------------------------------------------- #
# particles
# --------------------------------------------------------------------------- #
def load_snapshot(file, Lbox=1.0, remove_bulk_velocity=True, shift_to_origin=True):
    """PartType0/{Coordinates,Masses,Density,Velocities} of a GIZMO/Gadget HDF5 snapshot
    (interp.py:84-131).  Needs h5py on the host; `.npz` files with the same four keys
    (`Coordinates`, `Masses`, `Density`, `Velocities`) are accepted as well."""
    if str(file).endswith(".npz"):
        z = np.load(file)
        c, m, d, v = z["Coordinates"], z["Masses"], z["Density"], z["Velocities"]
    else:
        try:
            import h5py
        except ImportError as e:
            raise Exception("h5py is required to read HDF5 snapshots") from e
        with h5py.File(file, "r") as f:
            c = f["PartType0"]["Coordinates"][:]
            m = f["PartType0"]["Masses"][:]
            d = f["PartType0"]["Density"][:]
            v = f["PartType0"]["Velocities"][:]
    gp = GasParticles(c, m, d, v, Lbox=Lbox)
    if remove_bulk_velocity:
        gp.remove_bulk_velocity()
    if shift_to_origin:
        gp.shift_to_origin()
    return gp


try:   # a 10 GB/s hash where the host has it; zlib (stdlib) otherwise
    from xxhash import xxh3_128_digest as _digest
except ImportError:  # pragma: no cover
    import zlib

    def _digest(buf):
        return zlib.crc32(buf).to_bytes(4, "little") + zlib.adler32(buf).to_bytes(4, "little")


def _fingerprint(a):
    """Identity of a host array's CONTENT: its layout and a hash of EVERY byte, so that no in-place edit -- one particle's
    density, one column of the velocities, a few rows -- can go unnoticed between two device calls.  (A strided sample of
    the values, which this used to be, misses exactly those.)  Hashing runs at memory speed, several times faster than the
    dtype conversion + pageable host-to-device copy it decides about."""
    a = np.asarray(a)
    c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, _digest(memoryview(c).cast("B")))


class GasParticles:
    """interp.py:135-450 (I/O-free part).

    The particle arrays are numpy arrays, as in the reference; their device copies (positions as given, float32
    velocities and densities, the [rho v, rho] payload) are made on first use and KEPT: a second `ann_interp_to_field` /
    `deposit_to_field(N).spctrm(q)` on the same object uploads nothing.  A copy is dropped when its attribute is assigned
    again or when the hash of the array's bytes changed (any in-place edit); `invalidate_device()` drops them all."""

    _DEVICE_ATTRS = {"pos": ("pos",), "mass": ("mass",), "density": ("rho", "payload"), "velocity": ("vel", "payload"),
                     "v": ("vel", "payload")}

    def __init__(self, pos, mass, density, velocity, Lbox) -> None:
        object.__setattr__(self, "_devcache", {})
        self.pos = pos
        self.mass = mass
        self.density = density
        self.velocity = velocity
        self.Lbox = Lbox
        self.r = self.h()
        self.v = self.velocity

    def __setattr__(self, name, value):
        for key in self._DEVICE_ATTRS.get(name, ()):
            self._devcache.pop(key, None)
        object.__setattr__(self, name, value)

    def invalidate_device(self):
        self._devcache.clear()

    def _cached(self, key, sources, make):
        """Device tensor `key`, rebuilt by make() when one of the host arrays it was made from changed."""
        fp = tuple(_fingerprint(a) for a in sources)
        hit = self._devcache.get(key)
        if hit is not None and hit[0] == fp:
            return hit[1]
        t = make()
        self._devcache[key] = (fp, t)
        return t

    def _device_pos(self, k):
        return self._cached("pos", (self.pos,), lambda: _pos_tensor(k, self.pos))

    def _device_vel(self, k):
        return self._cached("vel", (self.v,), lambda: k.to_device(np.asarray(self.v), torch.float32))

    def _device_rho(self, k):
        return self._cached("rho", (self.density,), lambda: k.to_device(np.asarray(self.density), torch.float32))

    def _device_mass(self, k):
        return self._cached("mass", (self.mass,), lambda: k.to_device(np.asarray(self.mass), torch.float32))

    def __len__(self) -> int:
        return len(self.pos)

    def __getitem__(self, index):
        return GasParticles(self.pos[index], self.mass[index], self.density[index], self.v[index], self.Lbox)

    def _preprocess_on_device(self, shift, bulk):
        """vps_preprocess on the resident copies (two device reductions + one elementwise pass, interp.py:169-182), the host
        arrays then updated from them -- when those are float32 (positions may be float64), so that nothing is rounded that
        the reference would not round.  Returns False when the host arrays have to take numpy's route."""
        pos, v = np.asarray(self.pos), np.asarray(self.v)
        if not torch.cuda.is_available() or pos.dtype not in (np.float32, np.float64) or v.dtype != np.float32 \
                or not pos.flags.writeable or not v.flags.writeable:
            return False
        k = _kernels()
        # fresh tensors: a lazy BoxField made earlier (deposit_to_field / ann_interp_to_field) holds the resident copies it
        # was made from and must keep seeing the particles as they were then, like the reference's eager field would
        dpos = self._device_pos(k).clone() if shift else self._device_pos(k)
        dvel = self._device_vel(k).clone() if bulk else None
        k.preprocess(dpos, dvel, self._device_mass(k) if bulk else None, shift, bulk)
        if shift:
            pos[...] = dpos.cpu().numpy()
            self._devcache["pos"] = ((_fingerprint(self.pos),), dpos)
        if bulk:
            v[...] = dvel.cpu().numpy()
            self._devcache["vel"] = ((_fingerprint(self.v),), dvel)
            self._devcache.pop("payload", None)
        return True

    def shift_to_origin(self) -> None:
        if self._preprocess_on_device(True, False):
            return
        for a in range(3):
            self.pos[:, a] -= np.min(self.pos[:, a])

    def remove_bulk_velocity(self) -> None:
        if self._preprocess_on_device(False, True):
            return
        M = np.sum(self.mass)
        for a in range(3):
            self.v[:, a] -= np.sum(self.mass * self.v[:, a]) / M

    def rho(self, smoothing_rate=1.0):
        return self.density / smoothing_rate ** 3

    def h(self, smoothing_rate=1.0):
        V = self.mass / (self.density / smoothing_rate ** 3)
        return ((3 * V) / (4 * np.pi)) ** (1 / 3)

    def density_velocity_vector(self):
        """[vx*rho, vy*rho, vz*rho, rho], shape (Np,4) (interp.py:199-213)."""
        return np.stack((self.v[:, 0] * self.density, self.v[:, 1] * self.density,
                         self.v[:, 2] * self.density, self.density), axis=1)

    def _device_payload(self, k):
        return self._cached("payload", (self.v, self.density),
                            lambda: k.density_velocity_vector(self._device_vel(k), self._device_rho(k)))

    def ann_interp_to_field(self, Nsize, eps=0.0, treetype="kd", searchtype="standard"):
        """Exact-NN resampling onto the library lattice, then v=rho v/rho, m=rho*Lcell^3
        (interp.py:246-277).  `eps` must be 0 (exact search is what runs); `treetype` and
        `searchtype` are accepted and ignored, as the reference ignores them (:268-269)."""
        if eps != 0.0:
            raise Exception("only the exact search (eps=0.0) is implemented on the device")
        k = _kernels()
        Lcell = self.Lbox / Nsize
        ax = _lattice_axis(self.Lbox, Nsize)
        # The four grids are built on first use: `ann_interp_to_field(N).spctrm(q)`, the usual composition, resamples
        # straight into the fields of q (vps_nn_resample_quantity: no mass channel written, no weighted z pass).
        return BoxField._from_neighbours((self._device_pos(k), self._device_payload(k), ax), Nsize, Lcell)

    def deposit_to_field(self, Nsize, assignment="ngp"):
        """NGP composition the reference leaves to the caller: deposit_to_grid of
        density_velocity_vector (interp.py:996-1015), then v=rho v/rho with empty cells set
        to 0 (the rule of interp.py:329-331) and m=rho*Lcell^3 (interp.py:272-273).
        `assignment` = "cic" / "tsc" (extension; the reference offers NGP, NN and Voxelize) spreads
        [rho v, rho] over 8 / 27 cells; `BoxField.spctrm(..., deconvolve=True)` then divides the
        spectrum by the assignment window."""
        k = _kernels()
        vel, rho = self._device_vel(k), self._device_rho(k)
        if assignment != "ngp":
            if assignment not in _dev.ASSIGNMENT_ORDER:
                raise Exception("assignment must be 'ngp', 'cic' or 'tsc'")
            pos_e, pay_e = k.assign_expand(self._device_pos(k), self._device_payload(k), Nsize, self.Lbox, assignment)
            grid = k.deposit(pos_e, pay_e, Nsize, self.Lbox, 0, Nsize)
            k.field_algebra(grid, _dev.VM, 0, self.Lbox / Nsize)
            box = BoxField._from_device(grid, self.Lbox / Nsize)
            box.assignment = assignment
            return box
        # The grid itself is built on first use: `deposit_to_field(N).spctrm(...)`, the usual composition, goes
        # from the particles to P(k) through the fused deposit + z-pass kernel and never writes a grid.
        return BoxField._from_particles((self._device_pos(k), vel, rho), Nsize, self.Lbox)

    def total_mass(self) -> float:
        return np.sum(self.mass)

    def total_momentum(self) -> np.ndarray:
        return np.array([np.sum(self.mass * self.v[:, a]) for a in range(3)])

    def total_kinetic_energy(self) -> float:
        return 0.5 * np.sum(self.mass * (self.v[:, 0] ** 2 + self.v[:, 1] ** 2 + self.v[:, 2] ** 2))

    def specific_kinetic_energy(self) -> float:
        return self.total_kinetic_energy() / self.total_mass()


# --------------------------------------------------------------------------- #
# gridded field
# --------------------------------------------------------------------------- #
class BoxField:
    """vx, vy, vz, mass on an Nsize^3 grid of cell length Lcell (interp.py:456-471).

    Built from numpy arrays exactly like the reference (`BoxField(v, mass, Lcell)` with v of
    shape (N,N,N,3)), or internally from device buffers; `.vx/.vy/.vz/.mass` are numpy arrays
    either way (downloaded on first access when the field lives on the device)."""

    def __init__(self, v, mass, Lcell) -> None:
        self.Lcell = Lcell
        self._host = {"vx": v[..., 0], "vy": v[..., 1], "vz": v[..., 2], "mass": mass}
        self._chans = None
        self.Nsize = len(mass)
        self.Lbox = self.Nsize * self.Lcell

    @classmethod
    def _from_device(cls, chans, Lcell):
        self = cls.__new__(cls)
        self.Lcell = Lcell
        self._host = {}
        self._chans = chans            # [4, N, N, N] float32: vx, vy, vz, mass
        self.Nsize = chans.shape[1]
        self.Lbox = self.Nsize * Lcell
        return self

    @classmethod
    def _from_particles(cls, src, Nsize, Lbox):
        """Lazy NGP field of the device particle arrays src = (pos, vel, rho)."""
        self = cls.__new__(cls)
        self.Lcell = Lbox / Nsize
        self._host = {}
        self._chans = None
        self._src = src
        self.Nsize = Nsize
        self.Lbox = Lbox               # as given (Nsize * Lcell may differ in the last bit)
        return self

    @classmethod
    def _from_neighbours(cls, nn_src, Nsize, Lcell):
        """Lazy exact-NN field of nn_src = (device positions, device [rho v, rho], lattice axis)."""
        self = cls.__new__(cls)
        self.Lcell = Lcell
        self._host = {}
        self._chans = None
        self._nn_src = nn_src
        self._nn_spectra = 0
        self.Nsize = Nsize
        self.Lbox = Nsize * Lcell
        return self

    def _materialise(self):
        nn = getattr(self, "_nn_src", None)
        if nn is not None and self._chans is None and not self._host:
            self._chans, _ = _kernels().nn_resample_field(nn[0], nn[1], (nn[2], nn[2], nn[2]), 0, self.Nsize, self.Lcell)
        self._nn_src = None
        src = getattr(self, "_src", None)
        if src is not None and self._chans is None and not self._host:
            self._chans = _kernels().deposit_field(src[0], src[1], src[2], self.Nsize, self.Lbox, 0, self.Nsize, _dev.VM)
        self._src = None

    def _get(self, name, ch):
        self._materialise()
        if name not in self._host:
            self._host[name] = self._chans[ch].cpu().numpy().astype(np.float64)
        return self._host[name]

    def _set(self, name, value):
        self._materialise()
        self._host[name] = value
        if self._chans is not None:       # host copy is now authoritative
            for n, c in (("vx", 0), ("vy", 1), ("vz", 2), ("mass", 3)):
                self._get(n, c)
            self._chans = None

    vx = property(lambda s: s._get("vx", 0), lambda s, v: s._set("vx", v))
    vy = property(lambda s: s._get("vy", 1), lambda s, v: s._set("vy", v))
    vz = property(lambda s: s._get("vz", 2), lambda s, v: s._set("vz", v))
    mass = property(lambda s: s._get("mass", 3), lambda s, v: s._set("mass", v))

    def _device_chans(self, k):
        self._materialise()
        if self._chans is None:
            host = np.stack([np.asarray(self._host[n], dtype=np.float32) for n in ("vx", "vy", "vz", "mass")])
            self._chans = k.to_device(host)
        return self._chans

    def __getitem__(self, index):
        return BoxField(self.get_v()[index], self.mass[index], self.Lcell)

    def get_v(self) -> np.ndarray:
        return np.stack((self.vx, self.vy, self.vz), axis=3)

    def get_density(self) -> np.ndarray:
        return self.mass / self.Lcell ** 3

    def get_data(self) -> np.ndarray:
        return np.stack((self.vx, self.vy, self.vz, self.mass), axis=3)

    # -- quantities -> real fields on the device -------------------------------------
    def _fields(self, k, quantity):
        ch = self._device_chans(k)
        if quantity == "velocity":
            return [ch[0], ch[1], ch[2]]
        flags = _dev.FLAG_INPUT_IS_VM
        if quantity == "momentum":
            if REFERENCE_COMPAT["momentum_bug"]:
                flags |= _dev.FLAG_REFERENCE_MOMENTUM_BUG
            work = k.field_algebra_out(ch, _dev.MOMENTUM, flags, self.Lcell)     # (no copy of the four channels)
            return [work[0], work[1], work[2]]
        if quantity == "energy":
            work = k.field_algebra_out(ch, _dev.ENERGY, flags, self.Lcell)
            return [work[0]]
        raise Exception("""Unrecognized physical quantity name.
        Supported: 'velocity', 'momentum', 'energy'.""")

    def _power(self, quantity):
        k = _kernels()
        return _expand_half_power(k.power_grid(self._fields(k, quantity), self.Nsize), self.Lbox, self.Nsize)

    def velocity_power(self) -> np.ndarray:
        """(N,N,N) float64 P = 0.5*sum_c |a F v_c|^2 (interp.py:501-518)."""
        return self._power("velocity")

    def momentum_power(self) -> np.ndarray:
        """interp.py:521-541 with the correct vy, vz components unless
        REFERENCE_COMPAT['momentum_bug'] is set."""
        return self._power("momentum")

    def kinetic_energy_power(self) -> np.ndarray:
        """P of E = mass*(vx^2+vy^2+vz^2) (interp.py:544-557)."""
        return self._power("energy")

    def spctrm(self, quantity="velocity", kmin=None, kmax=None, kres=None, deconvolve=False) -> PowerSpectrum:
        """Binned spectrum, P multiplied by 4 pi k^2 (interp.py:560-595): z/y/x FFT passes
        with |F|^2 and the shell histogram fused into the last pass.  `deconvolve` (extension): divide
        |F(k)|^2 by the window W(k)^2 of the field's mass assignment (`deposit_to_field(N, assignment=...)`;
        exact for the linearly assigned momentum density, a customary approximation for v = rho v / rho)."""
        k = _kernels()
        pipe = _dev.PowerPipeline(self.Nsize, self.Lbox, kernels=k, comm=_dev.SlabComm(enabled=False),
                                  flavour="library", kmin=kmin, kmax=kmax, kres=kres,
                                  deconvolve=(getattr(self, "assignment", "ngp") if deconvolve else None))
        src = getattr(self, "_src", None)
        if quantity not in _dev.QUANTITY:
            raise Exception("""Unrecognized physical quantity name.
        Supported: 'velocity', 'momentum', 'energy'.""")
        if src is not None and k.fused_supported(self.Nsize, _dev.QUANTITY[quantity]):
            # particle-backed field: particles -> z/y-transformed spectra in one go (no grid in HBM)
            flags = _dev.FLAG_REFERENCE_MOMENTUM_BUG if (quantity == "momentum" and REFERENCE_COMPAT["momentum_bug"]) else 0
            pipe.prepare()
            with k.binning_only():      # the spectra go straight into the binning pass
                # (share_energy: spctrm('momentum') leaves the energy field's z image behind; spctrm('energy') right after it
                #  then launches no deposit of its own -- vpower/device.py)
                spec, nyq = k.deposit_fft_zy(src[0], src[1], src[2], self.Nsize, self.Lbox, 0, self.Nsize,
                                             _dev.QUANTITY[quantity], flags, reuse_sort=getattr(self, "_sort_token", None),
                                             share_energy=k.share_energy_fits(self.Nsize, self.Nsize))
            self._sort_token = k.fused_token()       # a second quantity of this field skips the particle sort
            tab = pipe.finish(*pipe.accumulate_spectra(spec, nyq))
            tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2
            return PowerSpectrum(tab)
        nn = getattr(self, "_nn_src", None)
        if nn is not None and self._nn_spectra == 0:
            # neighbour-backed field, first spectrum: the search writes the fields of this quantity directly.  (A second
            # quantity builds the four grids once -- one more search -- and every later one is served from them.)
            self._nn_spectra = 1
            flags = _dev.FLAG_REFERENCE_MOMENTUM_BUG if (quantity == "momentum" and REFERENCE_COMPAT["momentum_bug"]) else 0
            f, _ = k.nn_resample_quantity(nn[0], nn[1], (nn[2], nn[2], nn[2]), 0, self.Nsize, self.Lcell,
                                          _dev.QUANTITY[quantity], flags)
            return PowerSpectrum(pipe.spectrum([f[i] for i in range(f.shape[0])]))
        if quantity == "momentum":
            # p_c = v_c * mass is formed inside the z pass (two reads per line instead of an algebra pass)
            ch = self._device_chans(k)
            comps = [ch[0], ch[0], ch[0]] if REFERENCE_COMPAT["momentum_bug"] else [ch[0], ch[1], ch[2]]
            return PowerSpectrum(pipe.spectrum(comps, weight=ch[3]))
        fields = self._fields(k, quantity)
        return PowerSpectrum(pipe.spectrum(fields))

    # -- diagnostics (interp.py:639-666) ----------------------------------------------
    def _totals(self):
        """[sum m, sum m vx, sum m vy, sum m vz, sum m |v|^2]: one float64 device reduction over the
        four channels while the field lives in HBM (no download); numpy if it was built from / has been
        replaced by host arrays."""
        self._materialise()
        if self._chans is not None:
            return _kernels().field_totals(self._chans)
        m, vx, vy, vz = self.mass, self.vx, self.vy, self.vz
        return np.array([np.sum(m), np.sum(m * vx), np.sum(m * vy), np.sum(m * vz),
                         np.sum(m * (vx ** 2 + vy ** 2 + vz ** 2))], dtype=np.float64)

    def mean_kinetic_energy(self) -> float:
        return 0.5 * self._totals()[4] / float(self.Nsize) ** 3

    def total_kinetic_energy(self) -> float:
        return 0.5 * self._totals()[4]

    def total_mass(self) -> float:
        return self._totals()[0]

    def specific_kinetic_energy(self) -> float:
        t = self._totals()
        return 0.5 * t[4] / t[0]

    def total_momentum(self) -> np.ndarray:
        return self._totals()[1:4].copy()


def check_conservation(gasParticles, boxField) -> tuple:
    """Ratios of gridded to particle mass / momentum / kinetic energy / specific energy
    (interp.py:1269-1319), without the printing."""
    mass = boxField.total_mass() / gasParticles.total_mass()
    mom = boxField.total_momentum() / gasParticles.total_momentum()
    en = boxField.total_kinetic_energy() / gasParticles.total_kinetic_energy()
    sp = boxField.specific_kinetic_energy() / gasParticles.specific_kinetic_energy()
    return mass, mom, en, sp


# --------------------------------------------------------------------------- #
# free functions
# --------------------------------------------------------------------------- #
def deposit_to_grid(f, pos, Nsize, Lbox, assignment="ngp"):
    """Nearest-grid-point scatter-add of f (Np,) or (Np,C) into an (N,N,N[,C]) float64 grid;
    cell = int((pos // Lcell) % Nsize), periodic (interp.py:996-1015).  `assignment` = "cic" / "tsc"
    (extension) spreads every particle over 8 / 27 cells."""
    k = _kernels()
    f = np.asarray(f)
    f2 = f[:, None] if f.ndim == 1 else f
    if f2.ndim != 2:
        raise Exception("Unsupported data shape.")
    post = _pos_tensor(k, pos)
    cols = []
    c = 0
    while c < f2.shape[1]:      # the kernel takes 1, 3 or 4 channels per launch
        w = 4 if f2.shape[1] - c >= 4 else (3 if f2.shape[1] - c == 3 else 1)
        pe, fe = k.assign_expand(post, k.to_device(f2[:, c:c + w], torch.float32), Nsize, Lbox, assignment)
        g = k.deposit(pe, fe, Nsize, Lbox, 0, Nsize)
        cols.append(g)
        c += w
    grid = torch.cat(cols, dim=0).permute(1, 2, 3, 0).cpu().numpy().astype(np.float64)
    return grid[..., 0] if f.ndim == 1 else grid


def _lattice_axis(Lbox, Nsize):
    Lcell = Lbox / Nsize
    return np.linspace(Lcell / 2, Lbox + Lcell / 2, Nsize)


def make_grid_coords(Lbox, Nsize) -> np.ndarray:
    """(Nsize^3, 3) float64 query lattice linspace(Lcell/2, Lbox+Lcell/2, Nsize)^3, C order
    (interp.py:1060-1069; note the spacing Lbox/(Nsize-1))."""
    xs = _lattice_axis(Lbox, Nsize)
    return np.reshape(np.meshgrid(xs, xs, xs, indexing="ij"), (3, Nsize ** 3)).T


def _axes_of_lattice(query_pos, Nsize):
    q = np.asarray(query_pos)
    if q.shape != (Nsize ** 3, 3):
        raise Exception("query_pos must have shape (Nsize**3, 3)")
    ax = q[:: Nsize * Nsize, 0].copy()
    ay = q[: Nsize * Nsize: Nsize, 1].copy()
    az = q[:Nsize, 2].copy()
    rng = np.random.default_rng(0)
    probe = rng.integers(0, Nsize ** 3, size=min(4096, Nsize ** 3))
    i, j, l = probe // (Nsize * Nsize), (probe // Nsize) % Nsize, probe % Nsize
    if not (np.array_equal(q[probe, 0], ax[i]) and np.array_equal(q[probe, 1], ay[j])
            and np.array_equal(q[probe, 2], az[l])):
        raise Exception("query_pos must be a C-ordered lattice (e.g. make_grid_coords)")
    return ax, ay, az


def ann_interpolate(data_pos, query_pos, f, Nsize, eps, treetype="kd", searchtype="standard"):
    """f of the exact nearest particle at every query lattice point, reshaped to
    (N,N,N[,C]) (interp.py:1018-1049).  `query_pos` must be a C-ordered Nsize^3 lattice
    such as make_grid_coords returns; only eps=0 (exact) is implemented."""
    if eps != 0.0:
        raise Exception("only the exact search (eps=0.0) is implemented on the device")
    f = np.asarray(f)
    if f.ndim not in (1, 2):
        raise Exception("Unsupported data shape.")
    k = _kernels()
    axes = _axes_of_lattice(query_pos, Nsize)
    post = _pos_tensor(k, data_pos)
    idx_only = f.dtype != np.float32 or (f.ndim == 2 and f.shape[1] not in (1, 3, 4))
    if idx_only:
        # gather on the host in f's own dtype from the device-found indices
        _, idx = k.nn_resample(post, k.zeros((len(f), 1), torch.float32), axes, 0, Nsize, want_index=True)
        idx = idx.cpu().numpy().ravel()
        out = f[idx] if f.ndim == 1 else f[idx, :]
        return out.reshape((Nsize, Nsize, Nsize) + f.shape[1:])
    f2 = f[:, None] if f.ndim == 1 else f
    grid, _ = k.nn_resample(post, k.to_device(f2, torch.float32), axes, 0, Nsize)
    out = grid.permute(1, 2, 3, 0).cpu().numpy()
    return out[..., 0] if f.ndim == 1 else out


def nn_index(data_pos, axes):
    """Index of the exact nearest particle for every point of the lattice axes[0] x axes[1] x
    axes[2] (int32, C order)."""
    k = _kernels()
    post = _pos_tensor(k, data_pos)
    _, idx = k.nn_resample(post, k.zeros((post.shape[0], 1), torch.float32), axes, 0, len(axes[0]),
                           want_index=True)
    return idx.cpu().numpy()


def _expand_half_power(half, Lbox, Nsize):
    """(N,N,N) float64 P[kx,ky,kz] = 0.5*a^2*|F|^2 from the device half spectrum
    half[kz,ky,kx] (kz <= N/2) using |F(-k)| = |F(k)| for real input."""
    N = Nsize
    a = (Lbox / (2 * np.pi)) ** 1.5 / N ** 3          # interp.py:1381
    h = half.permute(2, 1, 0).cpu().numpy().astype(np.float64)   # [kx,ky,kz<=N/2]
    P = np.empty((N, N, N))
    P[:, :, : N // 2 + 1] = h
    neg = (-np.arange(N)) % N
    P[:, :, N // 2 + 1:] = h[neg][:, neg][:, :, N // 2 - 1: 0: -1]
    return P * (0.5 * a * a)


def _as_field(k, f, Nsize):
    f = np.asarray(f)
    if np.iscomplexobj(f):
        if np.any(f.imag != 0):
            raise Exception("the device transform takes real fields (un-folded input)")
        f = f.real
    if f.shape != (Nsize, Nsize, Nsize):
        raise Exception("field must have shape (Nsize, Nsize, Nsize)")
    return k.to_device(f, torch.float32)


def _vector_power(fx, fy, fz, Lbox, Nsize):
    """0.5*(|a F fx|^2+|a F fy|^2+|a F fz|^2) on the full (N,N,N) k grid, a=(L/2pi)^1.5/N^3
    (interp.py:1372-1387); normalised so that sum(P)*(2pi/L)^3 = 0.5*mean(|f|^2)."""
    k = _kernels()
    return _expand_half_power(k.power_grid([_as_field(k, f, Nsize) for f in (fx, fy, fz)], Nsize), Lbox, Nsize)


def _scalar_power(f, Lbox, Nsize):
    """0.5*|a F f|^2 (interp.py:1408-1421)."""
    k = _kernels()
    return _expand_half_power(k.power_grid([_as_field(k, f, Nsize)], Nsize), Lbox, Nsize)


def _pair_power(Pk, Lbox, Nsize, shift=np.array([0, 0, 0])):
    """(N^3,2) float64 [|k|, P] with k = 2 pi fftfreq(N, Lcell) per axis, each axis shifted
    by +shift[i] where shift[i] > 0 (interp.py:1440-1467)."""
    k = _kernels()
    ks = _dev.k_axis(Lbox, Nsize)
    axes = [ks + shift[i] if shift[i] > 0 else ks for i in range(3)]
    kk = k.pair_k(*axes).cpu().numpy()
    return np.stack((kk, np.ravel(Pk))).T


def _hist_sample(Pk_pair, kmin, kmax, spacing):
    """(nbins,4) [centre, P, Psum, Nsample]; np.arange centres/edges, P=0 in empty bins
    (interp.py:1470-1482)."""
    k = _kernels()
    centers, edges = _dev.bin_edges(kmin, kmax, spacing, "library")
    pair = np.asarray(Pk_pair, dtype=np.float64)
    psum, ns = k.hist_pairs(k.to_device(pair[:, 0]), k.to_device(pair[:, 1]), edges)
    psum, ns = psum.cpu().numpy(), ns.cpu().numpy().astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        P = psum / ns
    P[ns == 0] = 0
    return np.column_stack((centers, P, psum, ns))
