"""End-to-end timing of the NN route through the library API (GPU box): python tools/exp_nnpath.py N Np"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import interp, synth
N = int(sys.argv[1]); Np = int(float(sys.argv[2]))
pos, vel, mass, dens = synth.particles(3, Np, 1.0)
gp = interp.GasParticles(pos, mass, dens, vel, 1.0)
def sync(): torch.cuda.synchronize()
for rep in range(2):
    sync(); t0 = time.perf_counter()
    box = gp.ann_interp_to_field(N)
    sync(); t1 = time.perf_counter()
    times = {}
    for q in ("velocity", "momentum", "energy"):
        sync(); a = time.perf_counter()
        sp = box.spctrm(q)
        sync(); times[q] = (time.perf_counter() - a) * 1e3
    print("N=%d Np=%g: ann_interp_to_field %.1f ms (incl. host->device copies); spctrm " % (N, Np, (t1 - t0) * 1e3)
          + ", ".join("%s %.1f ms" % kv for kv in times.items()), flush=True)
    del box
    torch.cuda.empty_cache()
