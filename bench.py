#!/usr/bin/env python3
"""bench.py -- the hot path particles -> P(k) on MI355X, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--config C2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the path over one synthetic particle set already resident in
HBM: NGP deposit of [rho v, rho] into per-pencil buckets (rank, scan, scatter), one fused
kernel that accumulates each pencil in LDS, forms v = rho v / rho and z-transforms it, three y passes + all-to-all (N>1) + x pass with fused |F|^2 shell binning, shell all-reduce,
download of the (nbins,) sums, P(k) table.  Workload: BASELINE.json configs[1]
(512^3 grid, 1e7 particles, velocity P(k), nearest-grid-point deposition); with N>1 the SAME
grid is slab-decomposed over the N GPUs (strong scaling, one RCCL all-to-all per field).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(sample_N=256, density=10_000_000 / 512 ** 3):
    """The oracle (numpy restatement of the reference, kind 'port') timed on this host on a
    bounded sample of the same workload: same particle density, grid 256^3 instead of 512^3
    (1/8 of the cells and particles), one core (the reference pins FFTW to threads=1,
    vpower/interp.py:1382)."""
    from oracle import vps_oracle as orc
    from vpower import synth
    Np = int(round(density * sample_N ** 3))
    pos, vel, mass, dens = synth.particles(synth.BASE_SEED + 2, Np, 1.0)
    t0 = time.perf_counter()
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    grid = orc.deposit_to_grid_fast(vec, pos, sample_N, 1.0)
    v, m = orc.vm_from_vec_grid(grid, 1.0 / sample_N, zero_empty=True)
    t1 = time.perf_counter()
    P = orc.vector_power(v[..., 0], v[..., 1], v[..., 2], 1.0, sample_N)
    t2 = time.perf_counter()
    orc.spectrum_table(P, 1.0, sample_N, "library")
    t3 = time.perf_counter()
    total = t3 - t0
    if sample_N < 512 and total < 4.0:
        # fast host: the 1/8 sample is too short to time, run the whole workload instead
        return cpu_baseline(512, density)
    frac = "the whole workload" if sample_N == 512 else "1/8 of the workload at equal particle density"
    return {
        "value": sample_N ** 3 * 3 / total, "unit": "grid cells*components/s", "cores": 1, "kind": "port",
        "sample": "oracle/vps_oracle.py on %d^3 cells, %d particles (%s), float64, 1 thread: "
                  "deposit %.2fs, 3 FFTs+power %.2fs, pair+hist %.2fs"
                  % (sample_N, Np, frac, t1 - t0, t2 - t1, t3 - t2),
        "seconds": total,
    }


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "particles gridded/s + 3D FFT cells/s (Ngrid\u00b3) at 1/2/4/8 GPUs; HBM % of roofline"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", help="C1..C5 of vpower.synth.CONFIGS (grid, particles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="separate deposit and z-pass kernels (grid through HBM)")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra instrumented steps for the roofline")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="diagnostic: time ONE rank's share of a G-rank slab decomposition on one GPU "
                         "(x-slab N/G, segmented x pass, exchanges skipped; the spectrum is not meaningful)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # VPS_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, exchanges staged through host memory); the real runs use RCCL.
    backend = os.environ.get("VPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from vpower import device, synth
    K = device.default_kernels(local)
    N, Np, off = synth.CONFIGS[args.config]
    L = 1.0
    comm = device.SlabComm()
    if args.emulate_ranks > 1 and world == 1:
        class _OneOfG(device.SlabComm):
            """rank 0 of G without peers: the local z/y output stands in for the exchanged buffer
            (same sizes and segment layout), reductions are local"""
            def __init__(self, G):
                super().__init__(enabled=False)
                self.world, self.rank = G, 0
            def all_to_all_start(self, send):
                return send, None
            def all_reduce_sum(self, t):
                return t
        comm = _OneOfG(args.emulate_ranks)
    pipe = device.PowerPipeline(N, L, kernels=K, comm=comm, flavour="library")
    nx, x0 = pipe.nx, pipe.x0

    # ---- inputs: generated on the host once, resident in HBM before the timed region ----
    pos, vel, mass, dens = synth.particles(synth.BASE_SEED + off, Np, L)
    dpos, dvel, drho = K.to_device(pos), K.to_device(vel), K.to_device(dens)
    del pos, vel, mass, dens
    grid = K.empty((3, nx, N, N), torch.float32)
    psum, nsample = pipe.new_accumulators()      # two views of one buffer: one fill, one D2H copy per step
    acc_buf = pipe._acc_buf
    G = comm.world
    nkz, nky = N // 2 // G, N // G

    fused = K.fused_supported(N, device.VELOCITY) and not args.unfused
    if fused:
        spec3 = K.empty((3, N // 2, N, nx), torch.complex64)
        nyq3 = K.empty((3, N, nx), torch.complex64)

    def step():
        acc_buf.zero_()
        if fused:
            # deposit + v = rho v / rho + z pass in one kernel (pencil buckets), then the y passes
            K.deposit_fft_zy(dpos, dvel, drho, N, L, x0, nx, device.VELOCITY, spec=spec3, nyq=nyq3)
        else:
            K.deposit_field(dpos, dvel, drho, N, L, x0, nx, device.VELOCITY, out=grid)
        if fused:
            pipe.accumulate_spectra(spec3, nyq3, psum, nsample)     # 3 x (all-to-all, x pass + binning)
        else:
            pipe.accumulate([grid[0], grid[1], grid[2]], psum, nsample)   # 3 x (z/y passes, all-to-all, x pass)
        tab = pipe.finish(psum, nsample)     # all-reduce, D2H, table
        tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2
        return tab

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tab = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tab = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    cells = float(N) ** 3 * 3 * world / comm.world   # grid cells x components per step (whole job; emulation: one rank's share)

    # ---- per-kernel durations (HIP events on the library's stream), untimed extra steps ----
    K.timing(True)
    for _ in range(args.profile_steps):
        step()
    tim = K.timing_get()
    per = {k: K.timing_list(k) for k in ("fft_z", "fft_y", "fft_x", "deposit", "algebra")}
    K.timing(False)
    nst = args.profile_steps
    # y and x launches alternate main / Nyquist-plane; the main launch is the big one
    main_y, main_x = per["fft_y"][0::2], per["fft_x"][0::2]
    NH = N // 2
    mean = lambda v: float(np.mean(v)) if len(v) else 0.0
    if fused:
        # one pencil launch per step: the bucket records in (20 B per particle, once: the later rounds re-read
        # them from registers / L2), three z-transformed fields out
        z_bytes = 3 * 8.0 * nx * N * (NH + 1) + 20.0 * Np / G
        z_per_field = z_bytes / 3
    else:
        z_bytes = 4.0 * nx * N * N + 8.0 * nx * N * (NH + 1)
        z_per_field = z_bytes
    alg_bytes = {   # algorithmic HBM bytes per main launch (DESIGN.md "Kernels")
        "fft_z": z_bytes,
        "fft_y": 16.0 * nx * N * NH,
        "fft_x": 3 * 8.0 * nkz * N * N,   # one launch transforms and bins the three components
    }
    avg_ms = {"fft_z": mean(per["fft_z"]), "fft_y": mean(main_y), "fft_x": mean(main_x),
              # gridding stage: rank+scan+scatter ("deposit") and, unfused, brick accumulate+write ("algebra");
              # fused, the accumulation lives inside the z-pass launch
              "deposit": mean(per["deposit"]) + mean(per["algebra"])}
    step_kernel_ms = {k: v[1] / nst for k, v in tim.items() if v[0]}
    dom = max(("fft_z", "fft_y", "fft_x"), key=lambda k: step_kernel_ms.get(k, 0.0))
    ach = alg_bytes[dom] / (avg_ms[dom] * 1e-3) / 1e9
    fft_ms = sum(step_kernel_ms.get(k, 0.0) for k in ("fft_z", "fft_y", "fft_x"))
    fft_bytes = 3 * (z_per_field + 16.0 * nx * N * (NH + 1) + 8.0 * (nkz * N + nky) * N)
    traffic = None
    tr_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tr_path) and args.config == "C2" and world == 1 and comm.world == 1:
        try:
            traffic = json.load(open(tr_path)).get("fft_z_fused" if (dom == "fft_z" and fused) else dom)
        except Exception:
            traffic = None

    out = {
        "metric": baseline_metric(),
        "value_definition": "N^3 x components / step time: the whole path particles -> P(k) table; stage rates in "
                            "particles_per_s and fft_cells_per_s, HBM fraction in roofline",
        "value": cells * args.steps / dt,
        "unit": "grid cells*components/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d^3 grid, %d particles, velocity P(k), NGP deposit, library binning"
                               % (args.config, N, Np), "grid": N, "particles": Np,
                   "path": "fused deposit+z pass (pencil buckets)" if fused else "deposit -> grid -> z pass",
                   "parallelism": ("x-slab x%d, 1 all-to-all/field" % world) if comm.world == world else
                                  ("EMULATED rank 0 of %d on one GPU, exchanges skipped (diagnostic)" % comm.world)},
        "particles_per_s": Np / (avg_ms["deposit"] * 1e-3) if avg_ms["deposit"] > 0 else None,
        "fft_cells_per_s": cells / (fft_ms * 1e-3) / 1.0 if fft_ms > 0 else None,
        "fft_stage": {"ms_per_step": fft_ms, "algorithmic_GBs": fft_bytes / (fft_ms * 1e-3) / 1e9 if fft_ms else None,
                      "frac_of_hbm_peak": fft_bytes / (fft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fft_ms else None},
        "kernel_ms_per_step": step_kernel_ms,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": avg_ms[dom]},
    }
    if rank == 0 and world == 1 and comm.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        if not os.environ.get("VPS_BENCH_NOCHECK"):      # (timing-only kernel variants produce garbage)
            assert np.isfinite(tab[:, 2]).all() and tab[:, 3].sum() > 0
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
