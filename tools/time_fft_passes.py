"""Per-pass timing on an x-slab: python tools/time_fft_passes.py N nx   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
N = int(sys.argv[1]); nx = int(sys.argv[2]); G = N // nx
pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
K.set_binning(*pipe._binning)
f = torch.randn((nx, N, N), dtype=torch.float32, device="cuda")
spec = K.empty((N // 2, N, nx), torch.complex64); nyq = K.empty((N, nx), torch.complex64)
psum = K.zeros((pipe.nbins,), torch.float64); ns = K.zeros((pipe.nbins,), torch.int64)
nkz = N // 2 // G
import contextlib
def run():
    with (K.binning_only() if os.environ.get("BIN_ONLY", "1") == "1" else contextlib.nullcontext()):
        K.fft_zy(f, N, nx, spec=spec, nyq=nyq)
    # treat the local spec as the post-exchange buffer (same sizes and segment layout)
    K.fft_x_bin(spec, N, nkz * N, 0, 0, G, nkz * N * nx, psum, ns, count=True)
for _ in range(3): run()
K.timing(True)
for _ in range(5): run()
per = {k: K.timing_list(k) for k in ("fft_z", "fft_y", "fft_x")}
K.timing(False)
NH = N // 2
bytes_ = {"fft_z": 4.0 * nx * N * N + 8.0 * nx * N * (NH + 1), "fft_y": 16.0 * nx * N * NH, "fft_x": 8.0 * nkz * N * N}
for k in per:
    v = per[k] if k != "fft_y" else per[k][0::2]
    ms = float(np.mean(v))
    print("N=%d nx=%d %-6s %.3f ms  %.0f GB/s  (%.1f%% of 8 TB/s)" % (N, nx, k, ms, bytes_[k] / ms / 1e6, bytes_[k] / ms / 1e6 / 80), flush=True)
