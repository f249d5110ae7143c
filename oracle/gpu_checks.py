"""Checkers for sizes no host array can follow (2048^3 float64 grids do not fit a host): size-independent properties
computed with torch float64 ON THE GPU AS A CALCULATOR -- no library call is involved in them.

Test infrastructure like the rest of oracle/: only tests/, bench.py's parity / check legs and smoke() import it; the
product path never does.  Each function restates the reference rule it follows (file:line under /root/reference).
"""
import numpy as np
import torch


def shell_counts_exact(device, N, k2_axis, thr):
    """Number of modes of the full N^3 spectrum per bin: np.histogram's rule thr[b] <= s < thr[b+1] applied to
    s = (k2x + k2y) + k2z in float64 with numpy's association (vpower/interp.py:1449-1462, 1474-1477; thr[] are the squared
    edges, vpower.device.sqrt_thresholds), counted on the |k| octant with multiplicities."""
    h = N // 2
    nb = len(thr) - 1
    k2 = torch.as_tensor(np.asarray(k2_axis)[: h + 1].copy(), dtype=torch.float64, device=device)
    t = torch.as_tensor(np.asarray(thr), dtype=torch.float64, device=device)
    w = torch.full((h + 1,), 2, dtype=torch.int64, device=device)
    w[0] = 1
    w[h] = 1
    counts = torch.zeros(nb + 2, dtype=torch.int64, device=device)
    wyz = (w[:, None] * w[None, :]).reshape(-1)
    for i in range(h + 1):
        s = ((k2[i] + k2[:, None]) + k2[None, :]).reshape(-1)
        b = torch.bucketize(s, t, right=True)            # 0: below thr[0]; nbins+1: >= thr[nbins]
        counts.index_add_(0, b, wyz * int(w[i]))
    return counts[1: nb + 1].cpu().numpy()


def ngp_moments_float64(dpos, dvel, drho, N, L, quantities, rows=128):
    """Per quantity the float64 sum and sum of squares of every component field of the NGP fields: a restatement of
    interp.py:1010-1013 (cell = (pos // Lcell) % N, scatter-add of [rho v, rho]) + 272-273 (v = rho v / rho, m = rho Lcell^3;
    empty cells 0, the rule of :329-331) + 523-525 (p = v m) + 546 (E = m |v|^2), in torch float64, one x-slab of `rows` planes
    at a time.  -> {quantity: [[sum, sum of squares] per component]}."""
    Lcell = L / N
    vol = Lcell ** 3
    lc = torch.tensor(Lcell, dtype=dpos.dtype, device=dpos.device)
    cx = (torch.floor_divide(dpos[:, 0], lc) % N).to(torch.int64)
    out = {q: [[0.0, 0.0] for _ in range(1 if q == "energy" else 3)] for q in quantities}
    for x0 in range(0, N, rows):
        sel = torch.nonzero((cx >= x0) & (cx < x0 + rows)).squeeze(1)
        p = dpos[sel]
        flat = ((cx[sel] - x0) * N + (torch.floor_divide(p[:, 1], lc) % N).to(torch.int64)) * N \
            + (torch.floor_divide(p[:, 2], lc) % N).to(torch.int64)
        del p
        d = drho[sel].double()
        n3 = rows * N * N
        rho = torch.zeros(n3, dtype=torch.float64, device=dpos.device).index_add_(0, flat, d)
        inv = torch.where(rho > 0, 1.0 / rho, torch.zeros_like(rho))
        v = []
        for c in range(3):
            a = torch.zeros(n3, dtype=torch.float64, device=dpos.device).index_add_(0, flat, d * dvel[sel, c].double())
            v.append(a * inv)
            del a
        m = rho * vol
        del rho, inv, flat, d, sel
        for q in quantities:
            if q == "velocity":
                fs = v
            elif q == "momentum":
                fs = [v[c] * m for c in range(3)]
            else:
                fs = [m * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2])]
            for c, f in enumerate(fs):
                out[q][c][0] += float(f.sum().item())
                out[q][c][1] += float((f * f).sum().item())
            del fs
        del v, m
    return out


def parseval_targets(moments, N):
    """sum over ALL modes except k = 0 of 0.5 |a F|^2 (2 pi / L)^3 = 0.5 sum_c (<f_c^2> - <f_c>^2): the Parseval statement of
    interp.py:1377-1378 / 1413-1414 with the k = 0 mode (which no shell holds) taken out."""
    return {q: sum(0.5 * (sq / N ** 3 - (s_ / N ** 3) ** 2) for s_, sq in comps) for q, comps in moments.items()}


def all_mode_k_range(N, L):
    """(kmin, kmax, kres) of a script-flavour binning whose shells reach the corners of the k cube: every mode but k = 0."""
    kmin = 2 * np.pi / L
    return kmin, (int(np.ceil(np.sqrt(3.0) * N / 2)) + 1) * kmin, kmin
