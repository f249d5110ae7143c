"""Pencil kernel (fused deposit + z pass) timing at C4: python tools/time_pencil.py [N Np]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device, synth
K = device.default_kernels()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Np = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
pos, vel, rho = synth.particles_device(K, 4, Np, 1.0)
for q, name in ((device.VELOCITY, "velocity"), (device.MOMENTUM, "momentum"), (device.ENERGY, "energy")):
    zimg = K.empty((1 if q == device.ENERGY else 3, K.zimage_elems(N, N)), torch.complex64)
    for _ in range(2):
        K.deposit_fft_z(pos, vel, rho, N, 1.0, 0, N, q, zimg=zimg)
    K.timing(True)
    for _ in range(3):
        K.deposit_fft_z(pos, vel, rho, N, 1.0, 0, N, q, zimg=zimg)
    tz, td = K.timing_list("fft_z"), K.timing_list("deposit")
    K.timing(False)
    print("N=%d Np=%d %-9s pencil %.2f ms   sort %.2f ms" % (N, Np, name, float(np.mean(tz)), float(np.sum(td)) / 3), flush=True)
    del zimg
