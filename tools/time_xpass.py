"""x-pass cost split at N (full slab): python tools/time_xpass.py N   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
if os.environ.get("VPS_X_WG"):      # occupancy experiment: persistent workgroups per CU
    K.lib.vps_set_option(b"x_wg_per_cu", float(os.environ["VPS_X_WG"]))
N = int(sys.argv[1]); ncomp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
pipe.prepare()
specs = [torch.view_as_complex(torch.randn((N // 2, N, N, 2), dtype=torch.float32, device="cuda")) for _ in range(ncomp)]
psum = K.zeros((pipe.nbins,), torch.float64); ns = K.zeros((pipe.nbins,), torch.int64)
import inspect
print(inspect.signature(K.fft_x_bin))
for count in (True, False):
    def run():
        if ncomp > 1: K.fft_x_bin_multi(specs, N, (N // 2) * N, 0, 0, 1, 0, psum, ns, count=count)
        else: K.fft_x_bin(specs[0], N, (N // 2) * N, 0, 0, 1, 0, psum, ns, count=count)
    for _ in range(2): run()
    K.timing(True)
    for _ in range(5): run()
    v = K.timing_list("fft_x"); K.timing(False)
    ms = float(np.mean(v))
    print("N=%d ncomp=%d count=%s  %.3f ms  %.0f GB/s of full rows" % (N, ncomp, count, ms, ncomp * 8.0 * (N // 2) * N * N / ms / 1e6), flush=True)
