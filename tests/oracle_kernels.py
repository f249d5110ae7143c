"""CPU stand-in for vpower.device.HipKernels, built on numpy, used ONLY by the tests that
exercise the host-side slab choreography (all-to-all layout, segment addressing, shell
reduction) without a GPU.  It implements the kernels' documented contracts
(include/vps_hip.h) -- it is test infrastructure, never imported by the package."""
import numpy as np
import torch


class OracleKernels:
    name = "oracle-cpu"

    def __init__(self):
        self.binning = None
        self._bin_only = False

    def fft_supported(self, N):
        return N >= 4 and (N & (N - 1)) == 0

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def to_device(self, arr, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(arr))
        return t.to(dtype) if dtype is not None else t

    def set_binning(self, N, k2, thr, edge0, inv_spacing):
        self.binning = (N, np.asarray(k2), np.asarray(thr))

    def fft_zy(self, field, N, nx, weight=None):
        f = field.numpy().astype(np.float64)
        if weight is not None:
            f = f * weight.numpy().astype(np.float64)
        assert f.shape == (nx, N, N)
        F = np.fft.fft(np.fft.rfft(f, axis=2), axis=1)       # [x, ky, kz<=N/2]
        F = np.ascontiguousarray(F.transpose(2, 1, 0)).astype(np.complex64)   # [kz, ky, x]
        return torch.from_numpy(F[: N // 2].copy()), torch.from_numpy(F[N // 2].copy())

    # -- split form: z image = [B[x][kz][y] (kz < N/2) | BN[x][y]] flat complex64 (include/vps_hip.h: vps_fft_z) --
    def fft_z(self, field, N, nx, weight=None, zimg=None):
        f = field.numpy().astype(np.float64)
        if weight is not None:
            f = f * weight.numpy().astype(np.float64)
        Fz = np.fft.rfft(f, axis=2)                                   # [x, y, kz <= N/2]
        B = np.ascontiguousarray(Fz[:, :, : N // 2].transpose(0, 2, 1)).astype(np.complex64)   # [x, kz, y]
        BN = np.ascontiguousarray(Fz[:, :, N // 2]).astype(np.complex64)                        # [x, y]
        return torch.from_numpy(np.concatenate((B.ravel(), BN.ravel())))

    # -- chunk layout of the slab exchange (include/vps_hip.h: vps_fft_y, vps_fft_y_chunk_block, vps_fft_x_bin_chunk) --
    def binning_only(self):
        """Scope in which fft_y_chunk packs the rows that can still reach a shell (vps_set_bin_only)."""
        k = self

        class _Scope:
            def __enter__(self_inner):
                k._bin_only = True

            def __exit__(self_inner, *exc):
                k._bin_only = False
                return False
        return _Scope()

    def y_packed(self, N):
        return bool(self._bin_only and self.binning is not None and N >= 16)

    def _slots(self, N, G, nchunks, chunk, packed):
        """Per slot j of the chunk: (kc, rows) -- the rows |ky| <= kc that the slot's G planes chunk*G*nkc + j*G + h can still
        contribute to a shell (kc = -1: all N rows).  Any kc >= the exact cut is valid; the library rounds up to 16."""
        nkc = N // 2 // G // nchunks
        assert nkc * G * nchunks == N // 2
        out = []
        for j in range(nkc):
            kc = -1
            if packed:
                Nb, k2, thr = self.binning
                kc = 0
                for h in range(G):
                    ok = np.nonzero(~((k2[: N // 2 + 1] + k2[chunk * G * nkc + j * G + h]) >= thr[-1]))[0]
                    kc = max(kc, int(ok.max()) if ok.size else 0)
                kc = min(kc | 3, N // 2)
                if 2 * kc + 1 >= N:
                    kc = -1
            out.append((kc, N if kc < 0 else 2 * kc + 1))
        return out

    @staticmethod
    def _keep(N, kc):
        return np.arange(N) if kc < 0 else np.concatenate((np.arange(kc + 1), np.arange(N - kc, N)))

    def chunk_block(self, N, nx, G, nchunks, chunk, packed):
        rows = sum(r for _, r in self._slots(N, G, nchunks, chunk, packed))
        return rows * nx + (N // G * nx if chunk == nchunks - 1 else 0)

    def fft_y_chunk(self, zimg, N, nx, G, nchunks, chunk, out=None):
        """vps_fft_y: [h][ slot 0 rows | slot 1 rows | ... | (last chunk) Nyquist rows ky in h's range ]."""
        NH = N // 2
        nky = N // G
        nkc = NH // G // nchunks
        slots = self._slots(N, G, nchunks, chunk, self.y_packed(N))
        z = zimg.numpy()
        B = z[: nx * NH * N].reshape(nx, NH, N).astype(np.complex128)
        BN = z[nx * NH * N:].reshape(nx, N).astype(np.complex128)
        Cy = np.fft.fft(B, axis=2).transpose(1, 2, 0)                 # [kz, ky, x]
        CN = np.fft.fft(BN, axis=1).T                                 # [ky, x]
        parts = []
        for h in range(G):
            for j, (kc, _) in enumerate(slots):
                parts.append(Cy[chunk * G * nkc + j * G + h][self._keep(N, kc)].ravel())
            if chunk == nchunks - 1:
                parts.append(CN[h * nky:(h + 1) * nky].ravel())
        return torch.from_numpy(np.concatenate(parts).astype(np.complex64))

    def fft_x_bin_chunk(self, comps, N, nx, G, nchunks, chunk, rank, packed, psum, nsample, count=True):
        """vps_fft_x_bin_chunk: the G received blocks per component -> planes in place, binned plane by plane (+ Nyquist rows)."""
        nkc, nky = N // 2 // G // nchunks, N // G
        slots = self._slots(N, G, nchunks, chunk, packed)
        blk = self.chunk_block(N, nx, G, nchunks, chunk, packed)
        rows_total = sum(r for _, r in slots)
        for i, lines in enumerate(comps):
            flat = lines.numpy().reshape(-1)
            assert flat.size == G * blk
            r0 = 0
            for j, (kc, rows) in enumerate(slots):
                plane = np.zeros((N, N), dtype=np.complex64)          # [ky][x]; rows that were not sent hold nothing binned
                keep = self._keep(N, kc)
                for g in range(G):
                    plane[keep, g * nx:(g + 1) * nx] = flat[g * blk + r0 * nx: g * blk + (r0 + rows) * nx].reshape(rows, nx)
                self.fft_x_bin(torch.from_numpy(plane.reshape(-1)), N, N, 0, chunk * G * nkc + j * G + rank, 1, N * N, psum, nsample,
                               count=count and i == 0)
                r0 += rows
            if chunk == nchunks - 1:
                nyq = np.concatenate([flat[g * blk + rows_total * nx: (g + 1) * blk].reshape(nky, nx) for g in range(G)], axis=1)
                self.fft_x_bin(torch.from_numpy(np.ascontiguousarray(nyq).reshape(-1)), N, nky, rank * nky, N // 2, 1, nky * N, psum,
                               nsample, count=count and i == 0)

    def _lines(self, lines, N, nlines, nseg, seg_stride):
        flat = lines.numpy().reshape(-1)
        seglen = N // nseg
        segs = [flat[g * seg_stride: g * seg_stride + nlines * seglen].reshape(nlines, seglen)
                for g in range(nseg)]
        return np.concatenate(segs, axis=1)

    def fft_x_bin_multi(self, comps, N, nlines, line0, kz0, nseg, seg_stride, psum, nsample, count=True):
        for i, lines in enumerate(comps):
            self.fft_x_bin(lines, N, nlines, line0, kz0, nseg, seg_stride, psum, nsample, count=count and i == 0)

    def fft_x_bin(self, lines, N, nlines, line0, kz0, nseg, seg_stride, psum, nsample, count=True):
        Nb, k2, thr = self.binning
        assert Nb == N
        L = np.fft.fft(self._lines(lines, N, nlines, nseg, seg_stride).astype(np.complex128), axis=1)
        pw = (L.real ** 2 + L.imag ** 2)
        g = line0 + np.arange(nlines)
        ky, kz = g % N, kz0 + g // N
        assert kz.max() <= N // 2
        s = (k2[None, :] + k2[ky][:, None]) + k2[kz][:, None]
        w = np.where((kz == 0) | (2 * kz == N), 1, 2)[:, None] * np.ones((1, N), dtype=np.int64)
        b = np.searchsorted(thr, s, side="right") - 1
        ok = (b >= 0) & (b < len(thr) - 1)
        nb = len(thr) - 1
        ps = np.bincount(b[ok], weights=(pw * w)[ok], minlength=nb)
        ns = np.bincount(b[ok], weights=w[ok], minlength=nb)
        psum += torch.from_numpy(ps)
        if count:
            nsample += torch.from_numpy(np.rint(ns).astype(np.int64))
