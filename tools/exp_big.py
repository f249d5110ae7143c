"""Plane-wave known answers and a slab rfft check for the long non-power-of-two lines (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
for N, nx in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]:
    # slab of nx planes: z and y passes, then compare a few lines of the x-less spectrum with numpy
    rng = np.random.default_rng(N)
    f = rng.standard_normal((nx, N, N)).astype(np.float32)
    spec, nyq = K.fft_zy(K.to_device(f), N, nx)                      # [kz][ky][x], [ky][x]
    ref = np.fft.fft(np.fft.rfft(f.astype(np.float64), axis=2), axis=1)      # [x][ky][kz]
    got = spec.cpu().numpy(); gq = nyq.cpu().numpy()
    scale = np.sqrt((np.abs(ref) ** 2).mean())
    e1 = np.abs(got - ref[:, :, : N // 2].transpose(2, 1, 0)).max() / scale
    e2 = np.abs(gq - ref[:, :, N // 2].T).max() / scale
    print("N=%d nx=%d: z,y passes max err/rms %.2e (nyquist plane %.2e)" % (N, nx, e1, e2), flush=True)
    assert e1 < 5e-6 and e2 < 5e-6
    # x pass (write mode) on the same data against numpy
    out = K.empty((N // 2, N, N), torch.complex64) if nx == N else None
    if nx == N:
        K.fft_x_write(spec, N, N // 2 * N, 1, 0, out)
        r3 = np.fft.fft(ref[:, :, : N // 2].transpose(2, 1, 0), axis=2)
        e3 = np.abs(out.cpu().numpy() - r3).max() / np.sqrt((np.abs(r3) ** 2).mean())
        print("   x pass max err/rms %.2e" % e3, flush=True); assert e3 < 5e-6
    else:
        # x pass on lines of N from a buffer of N/nx-fold repeated slabs is not meaningful; transform the local lines as if N = nx is wrong.
        pass
