"""Timing experiments for the deposit stage (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device, synth
K = device.default_kernels()
N, L = 512, 1.0
for Np in (0, 1000, 10_000_000):
    pos, vel, mass, dens = synth.particles(7, max(Np, 1), L)
    pos, vel, dens = pos[:Np], vel[:Np], dens[:Np]
    dpos, dvel, drho = K.to_device(pos.reshape(-1, 3)), K.to_device(vel.reshape(-1, 3)), K.to_device(dens)
    out = K.empty((3, N, N, N), torch.float32)
    for q, name in ((device.VELOCITY, "velocity"), (device.ENERGY, "energy"), (device.VM, "vm")):
        o = K.empty(({0: 3, 2: 1, 3: 4}[q], N, N, N), torch.float32)
        for _ in range(3):
            K.deposit_field(dpos, dvel, drho, N, L, 0, N, q, out=o)
        K.timing(True)
        for _ in range(5):
            K.deposit_field(dpos, dvel, drho, N, L, 0, N, q, out=o)
        t = K.timing_get()
        K.timing(False)
        print("Np=%9d %-9s count+scan+scatter %.3f ms   brick %.3f ms" % (Np, name, t["deposit"][1] / 5, t["algebra"][1] / 5), flush=True)
