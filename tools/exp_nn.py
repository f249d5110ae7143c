"""NN resample timing (GPU box): python tools/exp_nn.py N Np"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch, time
from vpower import device, synth
K = device.default_kernels()
N = int(sys.argv[1]); Np = int(float(sys.argv[2]))
pos, vel, mass, dens = synth.particles(3, Np, 1.0)
dpos = K.to_device(pos); payload = K.density_velocity_vector(K.to_device(vel), K.to_device(dens))
ax = np.linspace(0.5 / N, 1 + 0.5 / N, N)
out = K.empty((4, N, N, N), torch.float32)
for _ in range(2): K.nn_resample(dpos, payload, (ax, ax, ax), 0, N, out=out)
K.timing(True)
for _ in range(3): K.nn_resample(dpos, payload, (ax, ax, ax), 0, N, out=out)
t = K.timing_get(); K.timing(False)
print("N=%d Np=%g: build %.3f ms, query %.3f ms  -> %.3g queries/s" % (N, Np, t["nn_build"][1] / 3, t["nn_query"][1] / 3, N ** 3 / (t["nn_query"][1] / 3 * 1e-3)), flush=True)
# search only (no payload gather / grid write): index output
import ctypes as C
from vpower import _ffi
idx = K.empty((N, N, N), torch.int32)
axs = [np.ascontiguousarray(ax, dtype=np.float64)] * 3
work = K.workspace("nn", K.lib.vps_nn_workspace_bytes(Np, 0))
def idx_only():
    K._stream()
    K._chk(K.lib.vps_nn_resample(K.ctx, K._ptr(dpos), 0, None, Np, 4, _ffi.as_dp(axs[0]), N, _ffi.as_dp(axs[1]), N,
                                 _ffi.as_dp(axs[2]), N, 0, N, None, K._ptr(idx), K._ptr(work)))
for _ in range(2): idx_only()
K.timing(True)
for _ in range(3): idx_only()
t = K.timing_get(); K.timing(False)
print("   index only: query %.3f ms" % (t["nn_query"][1] / 3), flush=True)
