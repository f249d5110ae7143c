# timing of a pathological input: 20 % of the particles in a handful of cells
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
N, Np, L = 512, 10_000_000, 1.0
rng = np.random.default_rng(1)
pos = rng.random((Np, 3)).astype(np.float32)
hot = rng.integers(0, 8, Np // 5)
pos[: Np // 5] = (np.array([[0.3, 0.4, 0.5]]) + hot[:, None] * 0.05 + rng.random((Np // 5, 3)) * 1e-4).astype(np.float32)
vel = rng.standard_normal((Np, 3)).astype(np.float32); rho = np.exp(rng.standard_normal(Np)).astype(np.float32)
d = [K.to_device(a) for a in (pos, vel, rho)]
for fused in (True, False):
    for _ in range(2):
        (K.deposit_fft_zy if fused else K.deposit_field)(d[0], d[1], d[2], N, L, 0, N, device.VELOCITY)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        (K.deposit_fft_zy if fused else K.deposit_field)(d[0], d[1], d[2], N, L, 0, N, device.VELOCITY)
    torch.cuda.synchronize(); print("hot cells, fused" if fused else "hot cells, bricks", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms", flush=True)
