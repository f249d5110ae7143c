"""Field-parallel decomposition at C4's full size on ONE GPU: the shares of ranks 0..W-1 are run one after the other, their shell
sums added and shell counts MAX-ed (what FieldComm's reductions do), and the result compared with the one-GPU step on the same
particles: python tools/check_fields_full_size.py [W]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
import bench
from vpower import device, synth
K = device.default_kernels()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, Np = 2048, 100_000_000
Q = ("velocity", "momentum", "energy")
pos, vel, rho = synth.particles_device(K, synth.BASE_SEED + 4, Np, 1.0)
class OneOf(device.FieldComm):
    def __init__(self, W, r):
        super().__init__(enabled=False)
        self.field_world, self.field_rank = W, r
    def all_reduce_sum(self, t):
        return t
wl = bench.Workload(K, device.SlabComm(enabled=False), N, 1.0, "ngp", Q, "library", pos, vel, rho)
ref = wl.step()
del wl; K._work.clear(); torch.cuda.empty_cache()
psum = {q: 0.0 for q in Q}; ns = {q: 0 for q in Q}
for r in range(W):
    wl = bench.Workload(K, OneOf(W, r), N, 1.0, "ngp", Q, "library", pos, vel, rho)
    t = wl.step()
    for q in Q:
        psum[q] = psum[q] + t[q][:, 2]
        ns[q] = np.maximum(ns[q], t[q][:, 3])
    print("rank %d of %d: %s" % (r, W, wl.my_units), flush=True)
    del wl; K._work.clear(); torch.cuda.empty_cache()
for q in Q:
    ok = ref[q][:, 3] > 0
    rel = float(np.max(np.abs(psum[q][ok] - ref[q][ok, 2]) / np.abs(ref[q][ok, 2])))
    print("%-9s counts equal: %s   max rel deviation of the shell sums: %.3g" % (q, bool(np.array_equal(ns[q], ref[q][:, 3])), rel), flush=True)
    assert np.array_equal(ns[q], ref[q][:, 3]) and rel < 2e-5
print("field-parallel shares of %d ranks reproduce the one-GPU tables at 2048^3 / 1e8 particles" % W)
