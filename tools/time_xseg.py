"""Segmented x pass (the layout a rank receives from the slab exchange) against the contiguous one, same lines:
python tools/time_xseg.py N planes   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device
K = device.default_kernels()
N = int(sys.argv[1]); planes = int(sys.argv[2]); ncomp = 3
pipe = device.PowerPipeline(N, 1.0, kernels=K, comm=device.SlabComm(enabled=False))
pipe.prepare()
nlines = planes * N
specs = [torch.view_as_complex(torch.randn((nlines * N, 2), dtype=torch.float32, device="cuda")) for _ in range(ncomp)]
psum = K.zeros((pipe.nbins,), torch.float64); ns = K.zeros((pipe.nbins,), torch.int64)
for nseg in (1, 2, 4, 8, 16):
    seglen = N // nseg
    def run():
        K.fft_x_bin_multi(specs, N, nlines, 0, 0, nseg, nlines * seglen if nseg > 1 else 0, psum, ns, count=True)
    for _ in range(2): run()
    K.timing(True)
    for _ in range(5): run()
    v = K.timing_list("fft_x"); K.timing(False)
    ms = float(np.mean(v))
    print("N=%d planes=%d nseg=%d  %.3f ms  %.0f GB/s" % (N, planes, nseg, ms, ncomp * 8.0 * nlines * N / ms / 1e6), flush=True)
