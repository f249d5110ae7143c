"""Power-spectrum result containers: the host-side surface of vpower/spctrm.py.

Same class and function names, argument meaning and error behaviour as the reference
(`PowerSpectrum` spctrm.py:55-246, `SpectrumList` :252-316, helpers :321-380) so that
user code and saved `Pk.txt` tables move over unchanged.  Pure host bookkeeping on
(nbins,) columns; the arithmetic that produces the columns runs on the GPU
(vpower.device).  Known reference defects that crash or silently do nothing
 are not reproduced; see each docstring.
"""
from __future__ import annotations

import os
import pickle

import numpy as np


class PowerSpectrum:
    """Columns k, P, Psum, Nsample of a binned spectrum (spctrm.py:55-66)."""

    def __init__(self, Pk, m=0, beta=np.array([-1, -1, -1])) -> None:
        Pk = np.asarray(Pk)
        self.k = Pk[:, 0]
        self.P = Pk[:, 1]
        self.Psum = Pk[:, 2]
        self.Nsample = Pk[:, 3]
        self.m = m
        self.beta = beta
        self.check_alignment()

    def data(self):
        return np.stack([self.k, self.P, self.Psum, self.Nsample], axis=1)

    def subtract_shot_noise(self, Lbox, Np) -> None:
        self.P -= Lbox ** 3 / Np
        self.P[self.P < 0] = 0

    def __len__(self):
        n = len(self.k)
        for name in ("P", "Psum", "Nsample"):
            if len(getattr(self, name)) != n:
                raise Exception("k and %s have different length." % name)
        return n

    check_alignment = __len__

    def kmin(self) -> float:
        return np.min(self.k)

    def kmax(self) -> float:
        return np.max(self.k)

    def kres(self) -> float:
        return (self.kmax() - self.kmin()) / (len(self) - 1)

    def Lbox(self) -> float:
        return 2 * np.pi / self.kmin()

    def energy(self) -> float:
        """Rectangle-rule integral of P dk (spctrm.py:108-113)."""
        return np.sum(self.P[:-1] * (self.k[1:] - self.k[:-1]))

    def copy(self):
        return PowerSpectrum(self.data(), self.m, self.beta)

    def _recompute(self):
        with np.errstate(invalid="ignore", divide="ignore"):
            self.P = self.Psum / self.Nsample * (4 * np.pi * self.k ** 2)

    def add(self, spctrm) -> None:
        """Accumulate shell sums and counts, then P = Psum/Nsample*4 pi k^2 (spctrm.py:118-126)."""
        if len(self) != len(spctrm):
            raise Exception("Spectra has different length therefore cannot be combined directly.")
        self.Psum = self.Psum + spctrm.Psum
        self.Nsample = self.Nsample + spctrm.Nsample
        self._recompute()

    def remove(self, spctrm) -> None:
        if len(self) != len(spctrm):
            raise Exception("Spectra has different length therefore cannot be combined directly.")
        self.Psum = self.Psum - spctrm.Psum
        self.Nsample = self.Nsample - spctrm.Nsample
        if (self.Nsample < 0).any():
            raise ValueError("Nsample is less than zero.")
        if (self.Psum < 0).any():
            raise ValueError("Psum is less than zero.")
        self._recompute()

    def append(self, spctrm) -> None:
        """Splice a coarser, higher-k spectrum onto this one (spctrm.py:142-166).  The
        reference rebinds the local name `self` and so discards its result; here the
        spliced columns replace this object's columns."""
        kspacing2 = spctrm.kres()
        keep = self.k < spctrm.k[0]
        full = PowerSpectrum(np.concatenate((self.data()[keep], spctrm.data())))
        for k in spctrm.k[spctrm.k < self.k[-1]]:
            sel = ((k - kspacing2 / 2) <= self.k) & (self.k < (k + kspacing2 / 2))
            at = np.where(full.k == k)
            full.Psum[at] += np.sum(self.Psum[sel])
            full.Nsample[at] += np.sum(self.Nsample[sel])
        nz = full.Psum > 0
        full.P[nz] = full.Psum[nz] / full.Nsample[nz] * (4 * np.pi * full.k[nz] ** 2)
        self.k, self.P, self.Psum, self.Nsample = full.k, full.P, full.Psum, full.Nsample
        self.check_alignment()

    def index(self) -> float:
        """Slope of log10 P against log10 k (spctrm.py:168-174)."""
        sel = self.P > 0
        power, _ = np.polyfit(np.log10(self.k[sel]), np.log10(self.P[sel]), 1)
        return power

    def plot(self, ax=None, remove_zero_power=True, **kwargs):
        import matplotlib.pyplot as plt
        if ax is None:
            _, ax = plt.subplots()
        sel = self.P > 0 if remove_zero_power else slice(None)
        ax.loglog(self.k[sel], self.P[sel], **kwargs)
        ax.set_xlabel(r"$k\,\mathrm{(kpc^{-1})}$")
        ax.set_ylabel(r"$P(k)\,\mathrm{(km^2\,s^{-2}\,kpc^{-1})}$")
        ax.grid(True)
        return ax

    def peek(self, fit_title=True, remove_zero_power=True) -> None:
        import matplotlib.pyplot as plt
        ax = self.plot(remove_zero_power=remove_zero_power)
        if fit_title:
            ax.set_title("$P(k) = k^{%.2f}$" % self.index())
        plt.show()

    def _filename(self, run_output_dir, beta):
        if beta is None or (np.asarray(beta) == np.array([-1, -1, -1])).all():
            return os.path.join(run_output_dir, "full_spctrm.pkl")
        return os.path.join(run_output_dir, "sub_spctrm_b{}{}{}.pkl".format(*beta))

    def save(self, run_output_dir) -> None:
        with open(self._filename(run_output_dir, self.beta), "wb") as fh:
            pickle.dump(self, fh)

    @staticmethod
    def load(run_output_dir, beta=None):
        name = (os.path.join(run_output_dir, "full_spctrm.pkl") if beta is None else
                os.path.join(run_output_dir, "sub_spctrm_b{}{}{}.pkl".format(*beta)))
        with open(name, "rb") as fh:
            return pickle.load(fh)

    # text form written by scripts/parallel_optimized.py:473
    def savetxt(self, filename) -> None:
        np.savetxt(filename, self.data())

    @staticmethod
    def loadtxt(filename):
        return PowerSpectrum(np.loadtxt(filename))


class SpectrumList:
    """Sub-spectra keyed by their beta vector (spctrm.py:252-316)."""

    def __init__(self, spctrm_list):
        self.list = spctrm_list
        self.m = spctrm_list[0].m

    def __len__(self):
        return len(self.list)

    def __getitem__(self, beta) -> PowerSpectrum:
        for s in self.list:
            if (np.asarray(s.beta) == np.asarray(beta)).all():
                return s
        raise Exception("No spectrum in the list with beta = {}".format(beta))

    def __setitem__(self, beta, spctrm) -> None:
        """Replace the entry with this beta or append (the reference's loop variable
        shadows the argument, spctrm.py:266-271; here the argument is what is stored)."""
        for i, s in enumerate(self.list):
            if (np.asarray(s.beta) == np.asarray(beta)).all():
                self.list[i] = spctrm
                return
        self.list.append(spctrm)

    def __iter__(self):
        return iter(self.list)

    def combine_all(self) -> PowerSpectrum:
        combined = empty_spectrum_like(self.list[0])
        for s in self.list:
            combined.add(s)
        return combined

    def combine_from_beta_sequence(self, beta_sequence=None) -> PowerSpectrum:
        if beta_sequence is None:
            beta_sequence = init_beta_space(m=self.m)
        combined = empty_spectrum_like(self.list[0])
        for beta in beta_sequence:
            combined.add(self[beta])
        return combined

    def append(self, spctrm) -> None:
        self.list.append(spctrm)

    def save(self, run_output_dir) -> None:
        with open(os.path.join(run_output_dir, "spctrm_list.pkl"), "wb") as fh:
            pickle.dump(self, fh)

    @staticmethod
    def load(run_output_dir):
        found = []
        for name in sorted(os.listdir(run_output_dir)):
            if name.startswith("sub_spctrm_b"):
                beta = np.array([int(c) for c in name[-7:-4]])
                found.append(PowerSpectrum.load(run_output_dir, beta=beta))
        if not found:
            raise Exception("No sub-spectrum files in {}".format(run_output_dir))
        return SpectrumList(found)


def relative_diff(spctrm1, spctrm2, mode="max") -> float:
    """The reference's spectrum distance (spctrm.py:321-346): NaN -> 0, zeros of P1 -> 1e-10,
    then max |P1-P2|/P1, or the rms / root-sum-square of (P1-P2)/P1."""
    if len(spctrm1) != len(spctrm2):
        raise Exception("Spectra has different length therefore cannot be compared directly.")
    P1 = spctrm1.P
    P1[np.isnan(P1)] = 0
    P1[P1 == 0] = 1e-10
    P2 = spctrm2.P
    P2[np.isnan(P2)] = 0
    if mode == "mean":
        return np.mean(((P1 - P2) / P1) ** 2) ** 0.5
    if mode == "max":
        return np.max(abs(P1 - P2) / P1)
    if mode == "sum":
        return np.sum(((P1 - P2) / P1) ** 2) ** 0.5
    raise Exception("Mode not recognized. Use 'mean' or 'max'.")


def empty_spectrum_like(spctrm, keep_m=False, keep_beta=False) -> PowerSpectrum:
    zeros = np.zeros_like(spctrm.k)
    return PowerSpectrum(np.column_stack((spctrm.k, zeros, zeros, zeros)),
                         m=spctrm.m if keep_m else 0,
                         beta=spctrm.beta if keep_beta else np.array([-1, -1, -1]))


def load_spectrum(filename) -> PowerSpectrum:
    with open(filename, "rb") as fh:
        return pickle.load(fh)


def init_beta_space(m):
    """All beta in [0,m)^3, shape (m^3, 3), first component fastest (spctrm.py:366-372)."""
    b = np.arange(0, m)
    return np.array(np.meshgrid(b, b, b, indexing="ij")).T.reshape(-1, 3)


def random_beta_sequence(m, seed=1):
    """A seeded permutation of the beta space (the reference discards its permutation,
    spctrm.py:375-380; this one returns it)."""
    rng = np.random.RandomState(seed)
    return rng.permutation(init_beta_space(m))
