"""Does replaying the device part of a step as a HIP graph shrink the inter-kernel gaps?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "large-velocity-power-spectrum_amd"))
import numpy as np, torch
from vpower import device, synth
K = device.default_kernels()
N, Np, off = synth.CONFIGS["C2"]; L = 1.0
pipe = device.PowerPipeline(N, L, kernels=K, comm=device.SlabComm(enabled=False))
pos, vel, mass, dens = synth.particles(synth.BASE_SEED + off, Np, L)
dpos, dvel, drho = K.to_device(pos), K.to_device(vel), K.to_device(dens)
psum, nsample = pipe.new_accumulators(); acc = pipe._acc_buf
spec3 = K.empty((3, N // 2, N, N), torch.complex64); nyq3 = K.empty((3, N, N), torch.complex64)
def dev_part():
    acc.zero_()
    K.deposit_fft_zy(dpos, dvel, drho, N, L, 0, N, device.VELOCITY, spec=spec3, nyq=nyq3)
    pipe.accumulate_spectra(spec3, nyq3, psum, nsample)
def timeit(fn, n=20):
    for _ in range(3): fn(); tab = pipe.finish(psum, nsample)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); tab = pipe.finish(psum, nsample)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, tab
t_plain, tab0 = timeit(dev_part)
print("plain launches: %.3f ms/step" % t_plain, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): dev_part()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    dev_part()
t_graph, tab1 = timeit(g.replay)
print("graph replay:   %.3f ms/step" % t_graph, "tables equal:", np.allclose(tab0[:, 2], tab1[:, 2], rtol=1e-6), flush=True)
