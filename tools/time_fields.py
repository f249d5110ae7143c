"""One rank's share of the field-parallel decomposition at C4, timed on one GPU: python tools/time_fields.py W r [W r ...]
(rank r of W: its contiguous block of [v0 v1 v2 p0 p1 p2 e]; the closing reductions are local)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
import bench
from vpower import device, synth
K = device.default_kernels()
N, Np = 2048, 100_000_000
pos, vel, rho = synth.particles_device(K, synth.BASE_SEED + 4, Np, 1.0)
class OneOf(device.FieldComm):
    def __init__(self, W, r):
        super().__init__(enabled=False)
        self.field_world, self.field_rank = W, r
    def all_reduce_sum(self, t):
        return t
args = [int(a) for a in sys.argv[1:]] or [2, 0]
for W, r in zip(args[0::2], args[1::2]):
    wl = bench.Workload(K, OneOf(W, r), N, 1.0, "ngp", ("velocity", "momentum", "energy"), "library", pos, vel, rho)
    for _ in range(2): wl.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): wl.step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print("rank %d of %d: fields %s  %.1f ms per step" % (r, W, wl.my_units, ms), flush=True)
    del wl; K._work.clear(); torch.cuda.empty_cache()
