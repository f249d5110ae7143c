"""Per-pass timing of the FFT stage on one 512^3 (or given N) field (GPU box)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
K.set_binning(*pipe._binning)
f = torch.randn((N, N, N), dtype=torch.float32, device="cuda")
spec = K.empty((N // 2, N, N), torch.complex64); nyq = K.empty((N, N), torch.complex64)
psum = K.zeros((pipe.nbins,), torch.float64); ns = K.zeros((pipe.nbins,), torch.int64)
def xmode(mode):
    K._stream()
    K._chk(K.lib.vps_fft_x(K.ctx, N, N // 2 * N, 0, 0, K._ptr(spec), 1, 0, mode, K._ptr(psum), K._ptr(ns),
                           K._ptr(spec) if mode in (1, 2) else None))
for name, fn in (("zy", lambda: K.fft_zy(f, N, N, spec=spec, nyq=nyq)), ("x mode0 (sum+count)", lambda: xmode(0)),
                 ("x mode3 (sum)", lambda: xmode(3)), ("x mode4 (fft only)", lambda: xmode(4))):
    for _ in range(3): fn()
    K.timing(True)
    for _ in range(10): fn()
    t = K.timing_get(); K.timing(False)
    print(name, {k: round(v[1] / 10, 4) for k, v in t.items() if v[0]}, flush=True)
