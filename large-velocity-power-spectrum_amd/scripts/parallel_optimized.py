#!/usr/bin/env python3
"""GPU drop-in for the reference driver scripts/parallel_optimized.py.

    python parallel_optimized.py -i snapshot.hdf5 -o outdir -N 1024 -f
    python -m torch.distributed.run --nproc-per-node 8 parallel_optimized.py -i ... -N 2048 -f

Same flags (parallel_optimized.py:42-61), same preprocessing (:280-291), same output
`<out>/Pk.txt` = np.savetxt of the (nbins,4) float32 table [k, P, Psum, Nsample] (:473),
same module-level helper names (`planner`, `FFTW_power`, `FFTW_vector_power`, `pair_power`,
`hist_sample`).  Differences: the nearest-neighbour search is EXACT (Annoy's answers are
approximate and unpinned), and the O(N^3 m^3) fold across MPI ranks is replaced by one
slab-decomposed N^3 FFT (identical spectrum), so `-M` and `-b` are
accepted and ignored and any number of ranks dividing N/2 works (not only cubes).
"""
import argparse
import datetime
import os
import sys
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

SNAPSHOT = "snapshot_550.hdf5"
SAVEDIR = "../output/"
NBUFFER = 5000
NTOT = 1000
MAXNBOX = 500
LTOT = 1
remove_bulk_velocity = True


def build_parser():
    p = argparse.ArgumentParser(description="Compute the velocity power spectrum on MI355X GPUs.",
                                usage="python %(prog)s [options]  (torchrun for several GPUs)")
    p.add_argument("-i", "--input", nargs="?", type=str, default=SNAPSHOT, help="Path to the snapshot file.")
    p.add_argument("-o", "--output", nargs="?", type=str, default=SAVEDIR, help="Directory to save the power spectrum.")
    p.add_argument("-N", "--ntot", nargs="?", type=int, default=NTOT, help="Total resolution.")
    p.add_argument("-M", "--maxnbox", nargs="?", type=int, default=MAXNBOX, help="Accepted for compatibility; unused.")
    p.add_argument("-l", "--ltot", nargs="?", type=int, default=LTOT, help="Total length of the box.")
    p.add_argument("-b", "--nbuffer", nargs="?", type=int, default=NBUFFER, help="Accepted for compatibility; unused.")
    p.add_argument("-f", action="store_true", help="Skip confirmation and start the computation.")
    return p


def planner(n_total_res, l_total_length, n_box_affordable, n_total_threads):
    """The reference's fold planner (parallel_optimized.py:70-88), kept for callers that
    inspect it; the GPU path does not fold."""
    ntpa = round(n_total_threads ** (1 / 3))
    assert ntpa ** 3 == n_total_threads, \
        "Number of threads must be a cube of an integer. Support for any number is not yet implemented."
    ntpa = int(ntpa)
    loops_per_axis = 1
    n_full = n_total_res / ntpa
    assert n_full.is_integer(), "Divided Nbox must be an integer."
    n_box = n_full
    while n_box > n_box_affordable or not n_box.is_integer():
        loops_per_axis += 1
        n_box = n_full / loops_per_axis
    n_box = int(n_box)
    return loops_per_axis ** 3, ntpa, n_box, n_box / n_total_res * l_total_length


def FFTW_power(f, Lbox, Nsize):
    """0.5*|const*FFT3(f)|^2, float32 (parallel_optimized.py:124-141)."""
    from vpower import interp
    return interp._scalar_power(f, Lbox, Nsize).astype(np.float32)


def FFTW_vector_power(fx, fy, fz, Lbox, Nsize):
    """parallel_optimized.py:92-120."""
    from vpower import interp
    return interp._vector_power(fx, fy, fz, Lbox, Nsize).astype(np.float32)


def pair_power(Pk, Lbox, Nbox, shift=np.array([0, 0, 0])):
    """(n,2) [|k|, P]; each axis shifted by -shift[i] where shift[i] != 0
    (parallel_optimized.py:145-172)."""
    from vpower import device
    k = device.default_kernels()
    ks = device.k_axis(Lbox, Nbox)
    axes = [ks - shift[i] if shift[i] != 0 else ks for i in range(3)]
    return np.column_stack((k.pair_k(*axes).cpu().numpy(), np.ravel(Pk)))


def hist_sample(Pk_pair, kmin, kmax, spacing):
    """(nbins,4) [centre, P, Psum, Nsample] with np.linspace edges and
    n_bins=int((kmax-kmin)/spacing)+1; P is NaN in empty bins (parallel_optimized.py:176-190)."""
    from vpower import device
    k = device.default_kernels()
    centers, edges = device.bin_edges(kmin, kmax, spacing, "script")
    pair = np.asarray(Pk_pair, dtype=np.float64)
    psum, ns = k.hist_pairs(k.to_device(pair[:, 0]), k.to_device(pair[:, 1]), edges)
    psum, ns = psum.cpu().numpy(), ns.cpu().numpy().astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P = psum / ns
    return np.column_stack((centers, P, psum, ns))


def load_particles(path):
    """coords, mass, velocity of PartType0 (parallel_optimized.py:272-276); `.npz` accepted."""
    if path.endswith(".npz"):
        z = np.load(path)
        return z["Coordinates"], z["Masses"], z["Velocities"]
    import h5py
    with h5py.File(path, "r") as f:
        return (f["PartType0/Coordinates"][:], f["PartType0/Masses"][:], f["PartType0/Velocities"][:])


def velocity_spectrum(coords, mass, velocity, ntot, ltot, comm=None, kernels=None):
    """The body of main() after loading: preprocessing, exact NN at x=i*LCELL (float32
    lattice, :343-346), raw velocity gather (:351), P(k) (:409-463).  Returns the float32
    (nbins,4) table that rank 0 saves."""
    import torch
    from vpower import device
    k = kernels if kernels is not None else device.default_kernels()
    pipe = device.PowerPipeline(ntot, ltot, kernels=k, comm=comm, flavour="script")
    lcell = ltot / ntot
    ax = np.array([i * lcell for i in range(ntot)], dtype=np.float32).astype(np.float64)
    # preprocessing of :280-291 on the device, in the snapshot's coordinate dtype
    c = np.asarray(coords)
    pos = k.to_device(c if c.dtype == np.float32 else c.astype(np.float64))
    vel = k.to_device(np.asarray(velocity), torch.float32)
    k.preprocess(pos, vel, k.to_device(np.asarray(mass), torch.float32), True, remove_bulk_velocity)
    # Annoy holds float32 coordinates (add_item, :308): search them as float32
    pos = pos.to(torch.float32)
    grid, _ = k.nn_resample(pos, vel, (ax, ax, ax), pipe.x0, pipe.nx)
    psum, ns = pipe.accumulate([grid[0], grid[1], grid[2]])
    tab = pipe.finish(psum, ns)
    tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2
    tab = np.array(tab, dtype=np.float32)                       # :436
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tab[:, 1] = tab[:, 2] / tab[:, 3] * (4 * np.pi * tab[:, 0] ** 2)   # :463
    return tab


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    outputfile = os.path.join(args.output, "Pk.txt")
    assert os.path.isdir(args.output), "Output directory does not exist."
    assert os.path.isfile(args.input), "Snapshot file does not exist."
    if rank == 0:
        print(f"[{datetime.datetime.now()}] Plan: one {args.ntot}^3 FFT on {world} GPU(s), no folding.", flush=True)
        if not args.f:
            print("Accept plan? (y/n)", flush=True)
            if input() != "y":
                print("Plan rejected.", flush=True)
                ok = False
            else:
                ok = True
        else:
            ok = True
    else:
        ok = True
    if world > 1:
        flag = torch.tensor([1 if ok else 0])
        if torch.cuda.is_available():
            flag = flag.cuda()
        dist.broadcast(flag, 0)
        ok = bool(flag.item())
    if not ok:
        return 0
    print(f"[{datetime.datetime.now()}] Load snapshot: {args.input}", flush=True) if rank == 0 else None
    coords, mass, velocity = load_particles(args.input)
    tab = velocity_spectrum(coords, mass, velocity, args.ntot, args.ltot)
    if rank == 0:
        np.savetxt(outputfile, tab)
        print(f"[{datetime.datetime.now()}] Saved: {outputfile}", flush=True)
    if world > 1:
        dist.barrier()
    return 0


if __name__ == "__main__":
    assert main() == 0, "Program stopped before completion."
