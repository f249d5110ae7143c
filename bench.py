#!/usr/bin/env python3
"""bench.py -- the hot path particles -> P(k) on MI355X, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--config C4]

`--gpus N` with N > 1 starts its own N ranks (a `python -m torch.distributed.run` child, before
this process touches the GPU) unless it already runs under a launcher (WORLD_SIZE set):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the path over one synthetic particle set already resident in HBM,
for the quantities the config names (vpower.synth.WORKLOADS; BASELINE.json configs):

    C1  128^3,  1e5 particles: exact NN on the script lattice, raw velocities, script binning
    C2  512^3,  1e7 particles: NGP deposit, velocity P(k)
    C3  1024^3, 5e7 particles: exact-NN resampling (library lattice), momentum P(k)
    C4  2048^3, 1e8 particles: NGP deposit, velocity + momentum + kinetic-energy P(k)   <- default
    C5  4096^3, 1e9 particles: NGP deposit, kinetic-energy P(k) (needs 8 GPUs: 275 GB per real field)

NGP route per quantity: bucket sort of the particle records by z-pencil (first quantity of a step
only), one fused kernel that accumulates each pencil in LDS, forms v / p / E and z-transforms it,
y passes, all-to-all (N > 1), x pass with fused |F|^2 shell binning, shell all-reduce, download of
the (nbins,) sums, P(k) table.  With N > 1 the SAME grid is slab-decomposed over the N GPUs
(strong scaling, one RCCL all-to-all per scalar field).

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).  Besides the timed
region the default run also (a) runs the SAME step function on a small sample of the workload and
compares its tables with the CPU oracle (`parity`), (b) times the oracle on that sample on one
core and on all cores (`cpu_baseline`, `cpu_baseline_allcores`), (c) at N = 1 with the default
config, times short runs of C2 and C3 (`other_configs`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "large-velocity-power-spectrum_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
PSUM_RTOL = 2e-5        # Psum per non-empty bin vs the float64 oracle


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C4", help="C1..C5 (vpower.synth.CONFIGS / WORKLOADS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short C2 / C3 runs of the default line")
    ap.add_argument("--no-full-check", action="store_true",
                    help="skip the full-size checks of the timed step's tables (exact shell counts, Parseval over all modes)")
    ap.add_argument("--unfused", action="store_true", help="NGP route: separate deposit and z-pass kernels (grid through HBM)")
    ap.add_argument("--profile-steps", type=int, default=2, help="extra instrumented steps for the roofline")
    ap.add_argument("--decomposition", choices=("auto", "slab", "fields"), default="auto",
                    help="several GPUs: 'slab' = 1-D x-slabs with one all-to-all per field -- what BASELINE.json's configs and the "
                         "north star name, and the only choice beyond one GPU's memory (C5); 'fields' = every rank transforms whole "
                         "grids of its share of the step's scalar fields, only shell tables cross the node (grids that fit one GPU); "
                         "'auto' = slab.  Where both are possible the one not chosen is timed as well and reported under `alternative`")
    ap.add_argument("--no-alternative", action="store_true",
                    help="several GPUs: skip the extra timed leg with the other decomposition")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="diagnostic: time ONE rank's share of a G-rank slab decomposition on one GPU "
                         "(x-slab N/G, segmented x pass, exchanges skipped; the spectrum is not meaningful)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check: every rank joins a gloo group, all-reduces its rank and rank 0 prints "
                         "{'dry_run': true, 'n_gpus': N}; no GPU is touched")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="with --dry-run: this rank exits with code 3")
    return ap.parse_args(argv)


def dry_run(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        return 2
    if rank == args.dry_run_fail_rank:
        return 3
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child
    `torch.distributed.run` and relay its output and exit code.  This process has not imported
    torch yet, so it never touches the GPU; nothing is re-exec'ed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------
# the workload: device step and CPU oracle of the same quantities
# ------------------------------------------------------------------------------------------------
def baseline_json():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))
    except Exception:
        return {"metric": "particles gridded/s + 3D FFT cells/s (Ngrid³) at 1/2/4/8 GPUs; HBM % of roofline",
                "configs": []}


NCOMP = {"velocity": 3, "momentum": 3, "energy": 1}


def kernel_sources_sha16():
    """Hash of the HIP sources the profiled kernels are built from (csrc/*.hip, *.h)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "large-velocity-power-spectrum_amd", "csrc", "*.hip"))
                    + glob.glob(os.path.join(ROOT, "large-velocity-power-spectrum_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(cfg, kernel):
    """(bytes per launch or None, where the figure comes from) for the dominant kernel of `cfg`, from the newest tracked
    profiles/rNN_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 correction)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None, "no tracked PMC profile"
    path = files[-1]
    rel = os.path.relpath(path, ROOT)
    try:
        d = json.load(open(path))
    except Exception as e:
        return None, "%s unreadable: %s" % (rel, e)
    sha = d.get("kernel_sources_sha16")
    if sha is not None and sha != kernel_sources_sha16():
        return None, "%s is stale: the kernel sources changed after it was taken (not measured in this run)" % rel
    val = d.get(cfg, {}).get(kernel)
    return val, "%s (rocprofv3 --pmc passes of an earlier run of this command%s; not measured in this run)" % (
        rel, "" if sha is not None else ", taken before source hashes were recorded")


class Workload:
    """One config's step on the device: particles resident in HBM -> {quantity: (nbins,4) table}.
    The same class runs the timed region, the instrumented steps and the small oracle-checked
    sample, so what is checked is what is timed."""

    def __init__(self, K, comm, N, L, route, quantities, flavour, pos, vel, rho, unfused=False):
        import torch
        from vpower import device
        self.K, self.comm, self.N, self.L = K, comm, int(N), float(L)
        self.route, self.quantities, self.flavour = route, tuple(quantities), flavour
        self.pos, self.vel, self.rho = pos, vel, rho
        self.dev = device
        self.pipe = device.PowerPipeline(N, L, kernels=K, comm=comm, flavour=flavour)
        self.nx, self.x0 = self.pipe.nx, self.pipe.x0
        self.psum, self.nsample = self.pipe.new_accumulators()
        self.acc_buf = self.pipe._acc_buf
        nx = self.nx
        maxc = max(NCOMP[q] for q in self.quantities)
        self.slab_particles = None
        if isinstance(comm, device.FieldComm) and not (route == "ngp" and not unfused
                                                        and all(K.fused_supported(N, device.QUANTITY[q]) for q in self.quantities)):
            raise SystemExit("--decomposition fields: the field-parallel step is the fused NGP path (grid sizes the pencil kernel covers)")
        if route == "ngp":
            self.fused = (not unfused) and all(K.fused_supported(N, device.QUANTITY[q]) for q in self.quantities)
            if self.fused and self.pipe.chunked and nx < N:
                # a rank holds the whole (replicated) particle set but sorts only those inside its slab: the workspace is
                # sized for them (counted once, outside the timed region; 1 % head room for nothing -- the count is exact)
                self.slab_particles = K.count_in_slab(pos, N, L, self.x0, nx)
            # Several ranks: the kz chunks of ALL quantities of the step form one bounded pipeline (PowerPipeline.pipelined_quantities:
            # two chunks in flight) -- quantity q+1's deposit + z pass is issued while q's last chunks are still crossing the node.
            # One z image (every y pass of q is enqueued before q+1's deposit overwrites it) and two chunks' send / receive buffers,
            # whatever the rank count.  VPS_PIPELINE_QUANTITIES=0: one quantity after the other, tables in between.
            self.pipelined = (self.pipe.chunked and not isinstance(comm, device.LibraryComm) and self.fused
                              and os.environ.get("VPS_PIPELINE_QUANTITIES") != "0")
            if self.pipelined:
                self.zimg = K.empty((maxc, K.zimage_elems(N, nx)), torch.complex64)
                self.acc_q = [self.pipe.new_accumulators() for _ in self.quantities]
            elif self.fused and self.pipe.chunked:
                self.zimg = K.empty((maxc, K.zimage_elems(N, nx)), torch.complex64)
            elif self.fused and isinstance(comm, device.FieldComm):
                # field-parallel ranks: whole grids, one scalar field at a time (device.FieldComm)
                self.my_units = comm.mine(self.quantities)
                mc = max([sum(1 for u in self.my_units if u[0] == q) for q in self.quantities] + [1])
                self.spec = K.empty((mc, N // 2, N, nx), torch.complex64)
                self.nyq = K.empty((mc, N, nx), torch.complex64)
            elif self.fused:
                self.spec = K.empty((maxc, N // 2, N, nx), torch.complex64)
                self.nyq = K.empty((maxc, N, nx), torch.complex64)
            else:
                self.grid = K.empty((maxc, nx, N, N), torch.float32)
        elif route == "nn":
            self.fused = False
            Lcell = self.L / N
            ax = np.linspace(Lcell / 2, self.L + Lcell / 2, N)      # interp.py:1063
            self.axes = (ax, ax, ax)
            self.grid = K.empty((4, nx, N, N), torch.float32)
        elif route == "script":
            self.fused = False
            lcell = self.L / N
            ax = np.array([i * lcell for i in range(N)], dtype=np.float32).astype(np.float64)   # script:343-346
            self.axes = (ax, ax, ax)
            self.grid = K.empty((3, nx, N, N), torch.float32)
        else:
            raise SystemExit("unknown route %r" % route)

    def fields_per_step(self):
        return sum(NCOMP[q] for q in self.quantities)

    def describe_path(self):
        if self.route == "ngp":
            if getattr(self, "pipelined", False):
                return "fused deposit+z pass (pencil buckets); kz chunks of all quantities in one pipeline, two in flight"
            if isinstance(self.comm, self.dev.FieldComm):
                return "fused deposit+z pass (pencil buckets), one scalar field per launch; fields dealt out over the ranks"
            return "fused deposit+z pass (pencil buckets)" if self.fused else "deposit -> grid -> z pass"
        if self.route == "nn":
            if len(self.quantities) == 1:
                return "exact-NN resample (library lattice) with the %s field(s) formed in its epilogue -> z pass" % self.quantities[0]
            return "exact-NN resample (library lattice) with v, m formed in its epilogue -> z pass (p = v*m formed in the pass)"
        return "exact-NN resample (script lattice, raw velocities) -> grid -> z pass"

    def _quantity_accs(self):
        """One (buffer, shell sums, shell counts) triple per quantity for the CURRENT pipeline, zeroed."""
        import torch
        if getattr(self, "_qacc_pipe", None) is not self.pipe:
            nb = self.pipe.nbins
            self._qacc = []
            for _ in self.quantities:
                buf = self.K.zeros((2 * nb,), torch.float64)
                self._qacc.append((buf, buf[:nb], buf[nb:].view(torch.int64)))
            self._qacc_pipe = self.pipe
        for buf, _, _ in self._qacc:
            buf.zero_()
        return self._qacc

    def _quantity_tables(self, accs):
        out = {}
        for i, q in enumerate(self.quantities):
            tab = self.pipe.finish(accs[i][1], accs[i][2], buf=accs[i][0])     # all-reduce, one D2H copy, table
            tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2              # interp.py:590 / script:434
            out[q] = tab
        return out

    def _table(self):
        tab = self.pipe.finish(self.psum, self.nsample)     # all-reduce, D2H, table
        tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2              # interp.py:590 / script:434
        return tab

    def step(self):
        K, dev, N, L, nx, x0 = self.K, self.dev, self.N, self.L, self.nx, self.x0
        out = {}
        if self.route == "ngp" and getattr(self, "pipelined", False):
            state = {"token": None}

            def producer(i, q):
                def produce():
                    z = K.deposit_fft_z(self.pos, self.vel, self.rho, N, L, x0, nx, dev.QUANTITY[q], zimg=self.zimg[:NCOMP[q]],
                                        reuse_sort=state["token"], slab_particles=self.slab_particles)
                    state["token"] = K.fused_token()
                    return [z[c] for c in range(NCOMP[q])]
                return produce
            for ps, ns in self.acc_q:
                ps.zero_()
                ns.zero_()
            # one bounded pipeline of kz-chunk jobs over all quantities (PowerPipeline.pipelined_quantities)
            self.pipe.pipelined_quantities([producer(i, q) for i, q in enumerate(self.quantities)], self.acc_q)
            for i, q in enumerate(self.quantities):
                tab = self.pipe.finish(*self.acc_q[i])
                tab[:, 1] *= 4 * np.pi * tab[:, 0] ** 2
                out[q] = tab
            return out
        if self.route == "ngp" and self.fused and isinstance(self.comm, dev.FieldComm):
            # field-parallel: this rank's share of the step's scalar fields, whole grid each; the closing reduction of a
            # quantity adds the shell sums of all ranks (a vector quantity's |F|^2 are summed over components anyway)
            token = None
            self.pipe.prepare()
            accs = self._quantity_accs()
            for i, q in enumerate(self.quantities):
                comps = [c for (qq, c) in self.my_units if qq == q]
                if comps:       # this rank's components of q in ONE launch (None: the energy field)
                    nc = len(comps)
                    with K.binning_only():
                        spec, nyq = K.deposit_fft_zy(self.pos, self.vel, self.rho, N, L, 0, N, dev.QUANTITY[q],
                                                     spec=self.spec[:nc], nyq=self.nyq[:nc], reuse_sort=token,
                                                     component=None if comps[0] is None else comps)
                    token = K.fused_token()
                    self.pipe.accumulate_spectra(spec, nyq, accs[i][1], accs[i][2])
            return self._quantity_tables(accs)      # (reductions and copies after every launch of the step has been issued)
        if self.route == "ngp" and self.fused and not self.pipe.chunked:
            # one GPU: deposit + field algebra + z pass in one kernel (pencil buckets), then the y passes (their output goes
            # straight into the binning x pass: rows beyond the last shell edge are not stored); the second and third quantity
            # of a step reuse the first one's bucket sort.  Every quantity has its own accumulators, so the whole step is
            # issued before the first table is copied back (the host never waits between quantities).
            token = None
            self.pipe.prepare()
            accs = self._quantity_accs()
            # momentum and, later in the step, kinetic energy: the momentum launch leaves the energy field's z image behind as
            # well (its rounds accumulate the cell totals of rho v_c that field is made of; device.Kernels.deposit_fft_zy)
            # (a fourth z image: 4 N^3 bytes more -- 29 N^3 in all at C4 -- taken only where that leaves a tenth of the device free)
            share = ("momentum" in self.quantities and "energy" in self.quantities
                     and self.quantities.index("momentum") < self.quantities.index("energy")
                     and os.environ.get("VPS_SHARE_ENERGY") != "0" and K.share_energy_fits(N, nx))
            for i, q in enumerate(self.quantities):
                qi, nc = dev.QUANTITY[q], NCOMP[q]
                with K.binning_only():
                    spec, nyq = K.deposit_fft_zy(self.pos, self.vel, self.rho, N, L, x0, nx, qi,
                                                 spec=self.spec[:nc], nyq=self.nyq[:nc], reuse_sort=token, share_energy=share)
                token = K.fused_token()
                self.pipe.accumulate_spectra(spec, nyq, accs[i][1], accs[i][2])
            return self._quantity_tables(accs)
        if self.route == "ngp":
            token = None
            for q in self.quantities:
                qi, nc = dev.QUANTITY[q], NCOMP[q]
                self.acc_buf.zero_()
                if self.fused and self.pipe.chunked:
                    # several ranks: the fused kernel stops after the z pass; y pass, exchange and x pass run chunk by chunk
                    z = K.deposit_fft_z(self.pos, self.vel, self.rho, N, L, x0, nx, qi, zimg=self.zimg[:nc], reuse_sort=token,
                                        slab_particles=self.slab_particles)
                    token = K.fused_token()
                    self.pipe.accumulate_zimages([z[i] for i in range(nc)], self.psum, self.nsample)
                else:
                    g = K.deposit_field(self.pos, self.vel, self.rho, N, L, x0, nx, qi, out=self.grid[:nc])
                    self.pipe.accumulate([g[i] for i in range(nc)], self.psum, self.nsample)
                out[q] = self._table()
            return out
        if self.route == "nn" and len(self.quantities) == 1:
            # one quantity (C3): the search's epilogue writes the fields its spectrum transforms (vps_nn_resample_quantity) --
            # no mass channel, no weighted z pass; what `gp.ann_interp_to_field(N).spctrm(q)` runs (vpower/interp.py)
            q = self.quantities[0]
            payload = K.density_velocity_vector(self.vel, self.rho)              # interp.py:199-213
            f, _ = K.nn_resample_quantity(self.pos, payload, self.axes, x0, nx, L / N, dev.QUANTITY[q], out=self.grid[:NCOMP[q]])
            self.acc_buf.zero_()
            self.pipe.accumulate([f[i] for i in range(NCOMP[q])], self.psum, self.nsample)
            out[q] = self._table()
            return out
        if self.route == "nn":
            payload = K.density_velocity_vector(self.vel, self.rho)              # interp.py:199-213
            g, _ = K.nn_resample_field(self.pos, payload, self.axes, x0, nx, L / N, out=self.grid)   # + interp.py:272-273
            for q in self.quantities:
                self.acc_buf.zero_()
                if q == "momentum":
                    self.pipe.accumulate([g[0], g[1], g[2]], self.psum, self.nsample, weight=g[3])
                elif q == "velocity":
                    self.pipe.accumulate([g[0], g[1], g[2]], self.psum, self.nsample)
                else:
                    e = K.field_algebra_out(g, dev.ENERGY, dev.FLAG_INPUT_IS_VM, L / N)
                    self.pipe.accumulate([e[0]], self.psum, self.nsample)
                out[q] = self._table()
            return out
        # script route: raw velocity gather (script:351), complex64 powers (:409-411), float32 table (:436-463)
        g, _ = K.nn_resample(self.pos, self.vel, self.axes, x0, nx, out=self.grid)
        self.acc_buf.zero_()
        self.pipe.accumulate([g[0], g[1], g[2]], self.psum, self.nsample)
        tab = np.array(self._table(), dtype=np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            tab[:, 1] = tab[:, 2] / tab[:, 3] * (4 * np.pi * tab[:, 0] ** 2)
        out["velocity"] = tab
        return out


def fields_possible(K, N, route, quantities, unfused=False):
    """Can every rank hold WHOLE grids (the field-parallel decomposition, device.FieldComm)?  The fused NGP path at a size whose
    buffers (spectrum of up to three components, z images, sort workspace: ~26 N^3 bytes) fit three quarters of the GPU."""
    import torch
    from vpower import device
    if route != "ngp" or unfused or not all(K.fused_supported(N, device.QUANTITY[q]) for q in quantities):
        return False
    return 26.0 * float(N) ** 3 < 0.78 * torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory


def oracle_tables(route, quantities, flavour, N, L, pos, vel, dens):
    """The CPU oracle (oracle/vps_oracle.py: numpy restatement of the reference) on the same
    particles -> ({quantity: table}, {stage: seconds})."""
    from oracle import vps_oracle as orc
    t = {}
    t0 = time.perf_counter()
    if route == "script":
        tab, _ = orc.script_pipeline(pos, vel, N, L)
        t["total"] = time.perf_counter() - t0
        return {"velocity": tab}, t
    vec = orc.density_velocity_vector(vel.astype(np.float64), dens.astype(np.float64))
    if route == "ngp":
        grid = orc.deposit_to_grid_fast(vec, pos, N, L)
        v, m = orc.vm_from_vec_grid(grid, L / N, zero_empty=True)
    else:
        ax = orc.lattice_axes_library(L, N)
        grid, _ = orc.ann_interpolate(pos, (ax, ax, ax), vec, N)
        v, m = orc.vm_from_vec_grid(grid, L / N)
    t["gridding"] = time.perf_counter() - t0
    out = {}
    for q in quantities:
        t1 = time.perf_counter()
        out[q] = orc.box_spctrm(v[..., 0], v[..., 1], v[..., 2], m, L / N, q, flavour=flavour)
        t["spctrm_" + q] = time.perf_counter() - t1
    t["total"] = time.perf_counter() - t0
    return out, t


def compare_tables(dev_tabs, ora_tabs):
    """nsample bit-equal, max relative Psum deviation over the non-empty bins."""
    eq, worst = True, 0.0
    for q, ref in ora_tabs.items():
        got = dev_tabs[q]
        ref = np.asarray(ref, dtype=np.float64)
        eq = eq and bool(np.array_equal(got[:, 3], ref[:, 3]))
        ok = ref[:, 3] > 0
        if ok.any():
            worst = max(worst, float(np.max(np.abs(got[ok, 2] - ref[ok, 2]) / np.abs(ref[ok, 2]))))
    return eq, worst


def sample_size(route, N, Np, cpu=False):
    """Grid of the oracle-checked sample: same particle density, at most 256^3 (NGP) or 128^3 (the NN routes: the
    oracle's kd-tree search is the slow part).  cpu: the sample the CPU baseline is timed on -- the size of BASELINE config 2,
    512^3 (configs 3-5 are extrapolated from there), 256^3 for the NN routes."""
    if cpu:
        Ns = min(N, 512 if route == "ngp" else 256 if route == "nn" else 128)
    else:
        Ns = min(N, 256 if route == "ngp" else 128)
    Nps = max(1000, int(round(Np * (Ns / N) ** 3)))
    return Ns, Nps


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------
def run_config(args, cfg, K, comm, world, rank, backend, steps, warmup, profile_steps, want_parity, want_cpu):
    """Times one config; returns the result dict (rank 0 fills the CPU legs)."""
    import torch
    import torch.distributed as dist
    from vpower import device, synth
    N, Np, off = synth.CONFIGS[cfg]
    route, quantities, flavour = synth.WORKLOADS[cfg]
    rehearsal = None
    if os.environ.get("VPS_BENCH_GRID"):       # rehearsal of a config's code path at a size that fits the box (never a result)
        N = int(os.environ["VPS_BENCH_GRID"])
        Np = int(float(os.environ.get("VPS_BENCH_PARTICLES", Np)))
        rehearsal = "REHEARSAL at %d^3 / %d particles (VPS_BENCH_GRID): not the config's size" % (N, Np)
    L = 1.0
    G = comm.world
    lognormal = cfg != "C1"      # C1 follows the script, which ignores densities

    # ---- inputs resident in HBM before the timed region (every rank holds the whole particle set) ----
    if Np >= 50_000_000:
        dpos, dvel, drho = synth.particles_device(K, synth.BASE_SEED + off, Np, L, lognormal)
        data = "synthetic (generated on the device: torch Philox, 1e7-particle chunks; preprocessing by vps_preprocess)"
    else:
        pos, vel, mass, dens = synth.particles(synth.BASE_SEED + off, Np, L, lognormal)
        dpos, dvel, drho = K.to_device(pos), K.to_device(vel), K.to_device(dens)
        del pos, vel, mass, dens
        data = "synthetic"
    wl = Workload(K, comm, N, L, route, quantities, flavour, dpos, dvel, drho, unfused=args.unfused)
    nx = wl.nx
    nchunks = wl.pipe.nchunks
    exchange_rows = None
    if G > 1 and wl.pipe.chunked:
        # slot j of band c of a rank is plane c*G*nkc + j*G + rank (include/vps_hip.h: vps_fft_y): the rows this rank keeps,
        # and the share of a block's rows that crosses the node (the blocks leave the rows no shell can reach at home)
        nkc_ = N // 2 // G // nchunks
        wl_keep = float(np.mean([wl.pipe.kept_row_fraction(p_, p_ + 1)
                                 for c in range(nchunks) for p_ in range(c * G * nkc_ + comm.rank, (c + 1) * G * nkc_, G)]))
        wl.pipe.prepare()
        with wl.pipe._bin_scope():
            packed_ = K.y_packed(N)
        exchange_rows = (sum(K.chunk_block(N, nx, G, nchunks, c, packed_) for c in range(nchunks)) /
                         float(sum(K.chunk_block(N, nx, G, nchunks, c, False) for c in range(nchunks))))
    elif G > 1:
        wl_keep = wl.pipe.kept_row_fraction(comm.rank * (N // 2 // G), (comm.rank + 1) * (N // 2 // G))
    else:
        wl_keep = wl.pipe.kept_row_fraction()
    nkz, nky, NH = N // 2 // G, N // G, N // 2
    nfields = wl.fields_per_step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        tabs = wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tabs = wl.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / steps * 1e3
    fw = getattr(comm, "field_world", 1)          # field-parallel ranks (device.FieldComm): G = 1, the FIELDS are dealt out
    cells = float(N) ** 3 * nfields * world / (G * fw)   # grid cells x scalar fields per step (whole job; emulation: one rank's share)

    # ---- per-kernel durations (HIP events on the library's stream), untimed extra steps ----
    K.timing(True)
    for _ in range(profile_steps):
        wl.step()
    tim = K.timing_get()
    per = {k: np.asarray(K.timing_list(k)) for k in ("fft_z", "fft_y", "fft_x", "deposit", "algebra", "nn_build", "nn_query")}
    K.timing(False)
    nst = max(profile_steps, 1)
    step_kernel_ms = {k: v[1] / nst for k, v in tim.items() if v[0] and not k.startswith("exchange")}
    # ---- slab exchange: how long it takes and how much of it the step cannot hide (several ranks; max over ranks) ----
    #   kernel_ms            this rank's kernels per step (HIP events on the library's stream)
    #   exposed_exchange_ms  per step, the time the compute stream stood still waiting for blocks to arrive, with the step's own
    #                        overlap (events around every wait of a chunk's x pass on its all-to-all)
    #   exchange_ms          per step, the exchanges by themselves: torch transport -- one extra step with ONE chunk in flight
    #                        (nothing overlaps: exchange start -> arrival); library transport -- the grouped send / recv intervals
    #                        on the library's communication stream (VPS_K_EXCHANGE)
    exchange = None
    slab_run = wl.pipe.chunked and (G > 1 or bool(getattr(comm, "force", False)))
    if slab_run:
        tsum = lambda name: tim.get(name, (0, 0.0))[1] / nst
        if isinstance(comm, device.LibraryComm):
            exchange = {"kernel_ms": sum(step_kernel_ms.values()), "exchange_ms": tsum("exchange"),
                        "exposed_exchange_ms": tsum("exchange_wait"),
                        "how": "HIP events inside the library: communication stream around each chunk's ncclSend / ncclRecv group; "
                               "context stream around each wait of an x pass on its chunk"}
        else:
            wl.pipe.instr = []
            wl.step()
            exposed, _ = wl.pipe.exchange_times()
            prev_inflight = os.environ.get("VPS_A2A_INFLIGHT")
            os.environ["VPS_A2A_INFLIGHT"] = "1"
            try:
                wl.pipe.instr = []
                wl.step()
                _, span = wl.pipe.exchange_times()
            finally:
                wl.pipe.instr = None
                if prev_inflight is None:
                    os.environ.pop("VPS_A2A_INFLIGHT", None)
                else:
                    os.environ["VPS_A2A_INFLIGHT"] = prev_inflight
            exchange = {"kernel_ms": sum(step_kernel_ms.values()), "exchange_ms": span, "exposed_exchange_ms": exposed,
                        "how": "torch events on the compute stream: exposed = around every wait on a chunk's all-to-all in a normal "
                               "step; exchange = start -> arrival in one extra step with a single chunk in flight"}
        if world > 1:
            for key in ("kernel_ms", "exchange_ms", "exposed_exchange_ms"):
                t = torch.tensor([exchange[key]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                exchange[key] = float(t.item())
            exchange["over"] = "max over the %d ranks" % world
    # y and x launches come as main launches (whole fields, or kz chunks of them) and small Nyquist-plane launches
    # (one plane against hundreds): the main ones are those within a factor 8 of the longest
    def main_of(v, within=8.0):
        return v[v * within >= v.max()] if len(v) else v
    # (NN search: the search kernel proper; the exact fallback for the points it leaves open is its own, much shorter launch --
    #  both are in kernel_ms_per_step["nn_query"])
    main = {"fft_y": main_of(per["fft_y"]), "fft_x": main_of(per["fft_x"]), "fft_z": per["fft_z"], "nn_query": main_of(per["nn_query"], 2.0)}
    Nps = Np / G      # particles inside one rank's slab (uniform positions)
    # algorithmic HBM bytes of the main launches of ONE step, per kernel family (DESIGN.md "Kernels"): what the kernels
    # have to move -- rows (ky, kz) beyond the last shell edge are neither written by the y pass nor read by the x pass
    keep = wl_keep
    nfl = len(wl.my_units) if fw > 1 else nfields     # scalar fields THIS rank transforms per step (field-parallel: its share)
    step_bytes = {"fft_y": nfl * 8.0 * nx * N * NH * (1.0 + keep), "fft_x": nfl * 8.0 * nkz * N * N * keep}
    if route == "ngp" and wl.fused and fw > 1:
        step_bytes["fft_z"] = nfl * (8.0 * nx * N * (NH + 1) + 20.0 * Nps)
    elif route == "ngp" and wl.fused:
        step_bytes["fft_z"] = sum(NCOMP[q] * 8.0 * nx * N * (NH + 1) + 20.0 * Nps for q in quantities)
    else:
        wread = 4.0 * nx * N * N if (route == "nn" and "momentum" in quantities and len(quantities) > 1) else 0.0
        step_bytes["fft_z"] = nfields * (4.0 * nx * N * N + wread + 8.0 * nx * N * (NH + 1))
    if route != "ngp":
        C_ = 4 if route == "nn" else 3
        Cout = nfields if (route == "nn" and len(quantities) == 1) else C_     # one quantity: only its fields are written
        step_bytes["nn_query"] = Np * (12.0 + 4 * C_) + 4.0 * Cout * nx * N * N
    launches_per_step = {k: max(len(v) // nst, 1) for k, v in main.items() if len(v)}
    kms = {k: float(np.sum(v)) / nst for k, v in main.items() if len(v)}        # ms per step in main launches
    if not kms:      # (a field-parallel rank beyond the last field: it only takes part in the reductions; never rank 0)
        kms, launches_per_step, step_bytes = {"fft_y": 1e-9}, {"fft_y": 1}, dict(step_bytes, fft_y=0.0)
    dom = max(kms, key=lambda k: kms[k])
    ach = step_bytes[dom] / (kms[dom] * 1e-3) / 1e9
    fft_ms = sum(step_kernel_ms.get(k, 0.0) for k in ("fft_z", "fft_y", "fft_x"))
    fft_bytes = step_bytes["fft_z"] + nfl * (8.0 * nx * N * (NH + 1) * (1.0 + keep) + 8.0 * (nkz * N + nky) * N * keep)
    grid_ms = step_kernel_ms.get("deposit", 0.0) + step_kernel_ms.get("algebra", 0.0) \
        + step_kernel_ms.get("nn_build", 0.0) + step_kernel_ms.get("nn_query", 0.0)
    # HBM traffic of the dominant kernel from the PMC counters: NOT measured in this run (counters need rocprofv3 passes of
    # their own) but copied from the newest tracked profile -- named in `traffic_source`, and dropped when the kernel sources
    # have changed since that profile was taken (tools/pmc_summary.py records their hash)
    traffic, traffic_source = None, None
    if world == 1 and G == 1:
        traffic, traffic_source = pmc_traffic(cfg, dom)
    bj = baseline_json()
    idx = int(cfg[1]) - 1
    cfg_text = bj["configs"][idx] if idx < len(bj.get("configs", [])) else cfg
    res = {
        "ms_per_step": ms_per_step,
        "value": cells * steps / dt,
        "config": {"workload": "%s: %s" % (cfg, cfg_text),
                   "deviation": rehearsal or "; ".join(x for x in (
                       ("strong-scaled over %d GPU(s) of one node" % world) if cfg in ("C4", "C5") and world != 8 else None,
                       ("field-parallel decomposition (--decomposition fields) instead of the slab all-to-all the config names: every "
                        "rank transforms whole grids and nothing but shell tables crosses xGMI; the slab run of the same step is timed "
                        "under `alternative`") if fw > 1 else None) if x) or None,
                   "grid": N, "particles": Np, "route": route, "quantities": list(quantities),
                   "scalar_fields_per_step": nfields, "path": wl.describe_path(),
                   "decomposition": ("fields" if fw > 1 else "slab") if (world > 1 or G > 1 or slab_run) else "none (one GPU)",
                   **({"transport": ("libvps_hip.so: grouped ncclSend / ncclRecv on its own stream (RCCL)" if isinstance(comm, device.LibraryComm)
                                     else "torch.distributed.all_to_all_single (%s)" % (backend if world > 1 else "emulated, no peers")),
                       "chunks": nchunks, "chunks_in_flight": (2 if isinstance(comm, device.LibraryComm) else wl.pipe.inflight_max()),
                       "exchange_row_fraction": exchange_rows, "slab_exchange": exchange,
                       "ms_per_step_slab": ms_per_step} if slab_run else {}),
                   **({"ms_per_step_fields": ms_per_step} if fw > 1 else {}),
                   "parallelism": ("x-slab x%d, one message per field and pair of ranks in %d kz chunks (unbinned rows left out, Nyquist rows inside it)"
                                   % (world, nchunks)) if (G == world and world > 1)
                   else ("field-parallel: the step's %d scalar fields dealt out over %d ranks, whole %d^3 grid per GPU, only shell tables cross the node"
                         % (nfields, fw, N)) if fw > 1
                   else "one GPU, no exchange" if G == world else ("EMULATED rank 0 of %d on one GPU, exchanges skipped (diagnostic)" % G)},
        "data": data,
        "particles_per_s": Np / (grid_ms * 1e-3) if grid_ms > 0 else None,
        "gridding_note": ("bucket sort only: the LDS accumulation of the fused path lives in the fft_z launch"
                          if (route == "ngp" and wl.fused) else "cell list + NN search (+ algebra)" if route != "ngp"
                          else "bucket sort + brick accumulate"),
        "fft_cells_per_s": cells / (fft_ms * 1e-3) if fft_ms > 0 else None,
        "fft_stage": {"ms_per_step": fft_ms, "algorithmic_GBs": fft_bytes / (fft_ms * 1e-3) / 1e9 if fft_ms else None,
                      "frac_of_hbm_peak": fft_bytes / (fft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fft_ms else None},
        "kernel_ms_per_step": step_kernel_ms,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": step_bytes[dom] / launches_per_step[dom],
                     "avg_launch_ms": kms[dom] / launches_per_step[dom],
                     "launches_per_step": launches_per_step[dom]},
        "per_kernel_frac_of_hbm_peak": {k: step_bytes[k] / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS for k in kms},
        **({"nn_query_note": "the search writes only the %d field(s) of the one quantity (vps_nn_resample_quantity): %.1f GB of algorithmic "
                             "bytes; on SURVEY.md 8(d) A2's four BoxField channels (%.1f GB) the same launch would read %.3f of peak"
                             % (nfields, step_bytes["nn_query"] / 1e9, (Np * 28.0 + 16.0 * nx * N * N) / 1e9,
                                (Np * 28.0 + 16.0 * nx * N * N) / (kms["nn_query"] * 1e-3) / 1e9 / HBM_PEAK_GBS)}
           if (route == "nn" and len(quantities) == 1 and "nn_query" in kms) else {}),
        "kept_row_fraction": keep,
        "launch_ms": {k: [round(float(x), 3) for x in v[:12]] for k, v in per.items() if len(v) and k in ("nn_query", "nn_build")},
    }
    if exchange_rows is not None:
        res["exchange_row_fraction"] = exchange_rows   # rows per kz plane in the exchanged blocks / N
    if exchange is not None:
        res["slab_exchange"] = exchange
    finite = all(np.isfinite(t[:, 2]).all() and t[:, 3].sum() > 0 for t in tabs.values())
    if rank == 0 and not os.environ.get("VPS_BENCH_NOCHECK"):      # (timing-only kernel variants produce garbage)
        assert finite, "non-finite shell sums"
    # ---- the tables of the TIMED, full-size step itself: exact shell counts; Parseval over all modes (NGP routes) ----
    # (the oracle cannot follow at this size; these are the size-independent properties of SURVEY.md section 4, computed with
    # torch float64 as the calculator: oracle/gpu_checks.py.  What they exercise are the kernels that dominate the timed
    # region -- wide y pass, 512-thread pencils, 64-bit sort keys, row-cut packing -- which the small oracle sample cannot.)
    if not args.no_full_check and (G == world or (fw > 1 and fw == world)) and not os.environ.get("VPS_BENCH_NOCHECK"):
        from oracle import gpu_checks as chk
        counts = chk.shell_counts_exact(K.device, N, wl.pipe.k2, wl.pipe.thr)
        full = {"nsample_exact": bool(all(np.array_equal(t[:, 3], counts) for t in tabs.values())),
                "what": "timed step's tables: Nsample per shell vs the exact lattice count"}
        if route == "ngp" and world == 1:
            kmin_, kmax_, kres_ = chk.all_mode_k_range(N, L)
            pall = device.PowerPipeline(N, L, kernels=K, comm=comm, flavour="script", kmin=kmin_, kmax=kmax_, kres=kres_)
            keep_pipe, keep_acc = wl.pipe, (wl.psum, wl.nsample, wl.acc_buf)
            wl.pipe = pall                                      # the same step(), its shells widened to the corners of the k cube
            wl.psum, wl.nsample = pall.new_accumulators()
            wl.acc_buf = pall._acc_buf
            tall = wl.step()
            wl.pipe, (wl.psum, wl.nsample, wl.acc_buf) = keep_pipe, keep_acc
            # (x-slabs of 32 planes at 2048^3: ten float64 arrays of a slab are alive at once -- 11 GB instead of 43 next to the workload)
            want = chk.parseval_targets(chk.ngp_moments_float64(dpos, dvel, drho, N, L, quantities, rows=32 if N >= 2048 else 128), N)
            worst = 0.0
            for q in quantities:
                t_ = tall[q]
                got = float(np.sum(t_[:, 2])) * (2 * np.pi / L) ** 3
                worst = max(worst, abs(got - want[q]) / want[q])
            full["parseval_max_rel"] = worst
            full["what"] += "; sum over ALL modes of Psum (2 pi/L)^3 vs 0.5 (<f^2> - <f>^2) of the float64 NGP fields, every quantity"
            del pall, tall
        res["full_size_check"] = full
        if rank == 0:
            assert full["nsample_exact"], "shell counts of the timed step differ from the exact lattice counts"
            assert full.get("parseval_max_rel", 0.0) <= PSUM_RTOL, "Parseval of the timed step: %.3g" % full["parseval_max_rel"]
    # ---- gridding alone (BASELINE's "particles gridded/s"): the standalone NGP deposit -- bucket sort, LDS accumulation of
    # [rho v, rho], field algebra, the three velocity fields written to HBM -- timed by itself.  (In the fused path the
    # accumulation lives inside the z-pass launch and cannot be timed apart; the sort alone would flatter.)
    gridding = None
    own_tabs = tabs
    del wl, tabs
    K._work.clear()
    torch.cuda.empty_cache()
    if route == "ngp" and G == world and world == 1:
        try:
            g_ = K.deposit_field(dpos, dvel, drho, N, L, 0, N, device.VELOCITY)
            K.timing(True)
            for _ in range(2):
                K.deposit_field(dpos, dvel, drho, N, L, 0, N, device.VELOCITY, out=g_)
            tg = K.timing_get()
            K.timing(False)
            del g_
            sort_ms, acc_ms = tg["deposit"][1] / 2, tg["algebra"][1] / 2
            gridding = {"standalone_deposit_ms": sort_ms + acc_ms, "sort_ms": sort_ms, "accumulate_and_write_ms": acc_ms,
                        "particles_per_s": Np / ((sort_ms + acc_ms) * 1e-3),
                        "algorithmic_GBs": (28.0 * Np + 12.0 * float(N) ** 3) / ((sort_ms + acc_ms) * 1e-3) / 1e9,
                        "what": "vps_deposit_field(velocity): sort + LDS accumulation + algebra + 3 float32 fields of N^3 to HBM"}
            res["gridding"] = gridding
            res["particles_per_s_sort_only"] = res["particles_per_s"]
            res["particles_per_s"] = gridding["particles_per_s"]
            res["gridding_note"] = ("standalone NGP deposit (sort + accumulation + fields written); the fused step does the "
                                    "accumulation inside the z-pass launch: see gridding, particles_per_s_sort_only")
        except Exception as e:      # (out of memory on a box with less HBM: keep the sort-only figure, say so)
            res["gridding_note"] += "; standalone deposit not timed: %s" % str(e)[:80]
            K.timing(False)
    # ---- several GPUs, a grid that fits one of them: the same step under the OTHER decomposition, reported beside `value` ----
    # (xGMI is point-to-point: the slab all-to-all moves the whole half spectrum of every field through the links, at two ranks
    #  through ONE link; dealing the seven scalar fields out moves nbins numbers.  Whichever of the two is not the run's own
    #  decomposition is timed here on the same particles, and its tables are checked against the run's own, both at full size.)
    other = "slab" if fw > 1 else "fields"
    can_other = (world > 1 and (G == world or (fw > 1 and fw == world)) and not args.no_alternative
                 and fields_possible(K, N, route, quantities, args.unfused)
                 and (other == "fields" or (N % world == 0 and (N // 2) % world == 0)))
    if can_other:
        ocomm = device.FieldComm() if other == "fields" else device.SlabComm()
        owl, ok_local = None, 1
        try:        # (whole-grid buffers are ~150-220 GB at 2048^3: a rank that cannot allocate them says so, and every rank skips the leg)
            owl = Workload(K, ocomm, N, L, route, quantities, flavour, dpos, dvel, drho)
            if other == "fields":
                K.workspace("fused", K.lib.vps_deposit_fft_zy_workspace_bytes(dpos.shape[0], N, N))
            otabs = owl.step()      # (first warm-up step: the step's own buffers; sizes are the same on every rank)
        except Exception as e:      # (out of memory, or anything else the first step of the other decomposition raises: the
            ok_local = 0            #  run's own result must not be lost to the extra leg)
            why = "%s: %s" % (type(e).__name__, str(e)[:160])
        else:
            why = None
        okt = torch.tensor([ok_local], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            res["alternative"] = {"decomposition": other, "skipped": "a rank could not run its first step" + (" (%s)" % why if why else "")}
        else:
            for _ in range(max(warmup - 1, 0)):
                otabs = owl.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                otabs = owl.step()
            barrier()
            dto = time.perf_counter() - t0
            t = torch.tensor([dto], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dto = float(t.item())
            eq, worst = compare_tables(otabs, own_tabs)
            K.timing(True)              # one more, instrumented step: this rank's kernel time (the rest of a slab step is exchange)
            owl.step()
            otim = {k: v[1] for k, v in K.timing_get().items() if v[0]}
            K.timing(False)
            res["alternative"] = {
                "kernel_ms_per_step_rank0": otim,
                "decomposition": other, "ms_per_step": dto / steps * 1e3, "value": float(N) ** 3 * nfields * steps / dto,
                "unit": "grid cells*components/s",
                "parallelism": ("the step's %d scalar fields dealt out over %d ranks in contiguous blocks (a rank's components of a quantity "
                                "in one launch), whole %d^3 grid per GPU, particles replicated; only the shell tables cross the node"
                                % (nfields, world, N)) if other == "fields" else
                               ("x-slab x%d, one message per field and pair of ranks in %d kz chunks (unbinned rows left out), %s"
                                % (world, owl.pipe.nchunks, owl.describe_path())),
                "vs_own_tables": {"nsample_equal": eq, "psum_max_rel": worst},
                "note": "python bench.py --gpus N --decomposition %s makes this the reported decomposition" % other,
                **({"why": "xGMI is point-to-point: the slab transform sends every field's half spectrum through the links once (at two "
                           "ranks through ONE link), dealing the scalar fields out sends nbins numbers -- but it stops at 7 fields "
                           "(C4) and at grids that fit one GPU, and it is not the decomposition BASELINE.json names"}
                   if other == "fields" else {})}
            res["config"]["ms_per_step_" + other] = dto / steps * 1e3     # (both decompositions' step times inside `config`)
            if rank == 0 and not os.environ.get("VPS_BENCH_NOCHECK"):
                assert eq and worst <= PSUM_RTOL, "field-parallel and slab tables differ: %s %.3g" % (eq, worst)
            del otabs
        del owl
        K._work.clear()
        torch.cuda.empty_cache()
    # release the big buffers before the sample / the next config
    del dpos, dvel, drho
    K._work.clear()
    torch.cuda.empty_cache()

    # ---- the same step function on a small sample, against the oracle ----
    if want_parity and (G == world or (fw > 1 and fw == world)):
        Ns, Nps_ = sample_size(route, N, Np)
        while fw == 1 and Ns % (2 * world):      # (slabs: N/2 divisible by the ranks; field-parallel ranks hold whole grids)
            Ns *= 2
        pos, vel, mass, dens = synth.particles(synth.BASE_SEED + 100 + off, Nps_, L, lognormal)
        swl = Workload(K, comm, Ns, L, route, quantities, flavour, K.to_device(pos), K.to_device(vel),
                       K.to_device(dens), unfused=args.unfused)
        dev_tabs = swl.step()
        del swl
        if rank == 0:
            ora_tabs, _ = oracle_tables(route, quantities, flavour, Ns, L, pos, vel, dens)
            eq, worst = compare_tables(dev_tabs, ora_tabs)
            res["parity"] = {"sample": "%d^3 grid, %d particles (the config's particle density), same step() as the timed region, "
                                       "vs oracle/vps_oracle.py" % (Ns, Nps_),
                             "nsample_equal": eq, "psum_max_rel": worst, "psum_rtol": PSUM_RTOL}
            if not os.environ.get("VPS_BENCH_NOCHECK"):
                assert eq, "shell counts differ from the oracle"
                assert worst <= PSUM_RTOL, "shell sums differ from the oracle: %.3g" % worst
        K._work.clear()
        torch.cuda.empty_cache()
    return res


def cpu_baseline_legs(cfg):
    """The CPU side of the measurement (rank 0, one GPU run only; AFTER every device timing: the threaded leg leaves worker
    pools behind that slow the host's kernel launches).  One core: the oracle at the size of BASELINE config 2 (512^3 --
    SURVEY.md 8d makes it the base from which configs 3-5 are extrapolated; 256^3 for the NN route), at the config's particle
    density, stage by stage.  All cores (NGP): the same sample through oracle/allcores.py, every stage spread over the cores."""
    from vpower import synth
    from oracle import vps_oracle as orc
    N, Np, off = synth.CONFIGS[cfg]
    route, quantities, flavour = synth.WORKLOADS[cfg]
    L, lognormal = 1.0, cfg != "C1"
    nfields = sum(NCOMP[q] for q in quantities)
    out = {}
    Nc, Npc = sample_size(route, N, Np, cpu=True)
    pos, vel, _, dens = synth.particles(synth.BASE_SEED + 200 + off, Npc, L, lognormal)
    _, t1 = oracle_tables(route, quantities, flavour, Nc, L, pos, vel, dens)
    del pos, vel, dens
    scale = (float(N) / Nc) ** 3
    unit = "grid cells*components/s"
    out["cpu_baseline"] = {
        "value": Nc ** 3 * nfields / t1["total"], "unit": unit, "cores": 1, "kind": "port",
        "sample": "oracle/vps_oracle.py (numpy, float64, 1 thread -- the reference pins FFTW threads=1, interp.py:1382) on "
                  "%d^3 cells, %d particles = 1/%d of the workload at equal particle density: %s"
                  % (Nc, Npc, round(scale), ", ".join("%s %.2fs" % kv for kv in t1.items())),
        "seconds": t1["total"], "stage_seconds": dict(t1),
        "extrapolated_seconds_full_size": t1["total"] * scale,
        "extrapolation": "by the algorithmic-bytes ratio (N/%d)^3 (SURVEY.md 8d); not measured" % Nc,
        "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
    if route == "ngp":
        # all host cores, the WHOLE step spread over them the way the reference spreads its work over MPI ranks (every rank a part of
        # the volume, scripts/parallel_optimized.py:201-491 under `mpiexec -n R`): oracle/allcores.py -- one process per x-slab for
        # gridding, |F|^2 and both histograms, threaded transforms -- on the SAME sample as the one-core leg.  A child process of its
        # own: it forks its workers, and they must not inherit this process's GPU state.
        # (two worker counts: process start-up and shared-memory page faults grow with the count, the slabs shrink with it --
        #  which one wins depends on the host; the faster run is the baseline, both are listed)
        ncpu = os.cpu_count() or 1
        tries, runs = sorted({max(1, min(ncpu, Nc // 2) // 4), max(1, min(ncpu // 2, 128, Nc // 2))}), {}
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        for nthr in tries:
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "allcores.py"), "--grid", str(Nc), "--particles", str(Npc),
                   "--workers", str(nthr), "--quantities", ",".join(quantities), "--flavour", flavour,
                   "--seed", str(synth.BASE_SEED + 200 + off), "--lognormal", str(int(lognormal))]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
                runs[nthr] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:      # (the headline must not be lost to the extra CPU leg)
                runs[nthr] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        good = {k: v for k, v in runs.items() if "seconds" in v}
        if good:
            nthr = min(good, key=lambda k: good[k]["seconds"])
            t2 = good[nthr]
            out["cpu_baseline_allcores"] = {
                "value": Nc ** 3 * nfields / t2["seconds"], "unit": unit, "cores": nthr, "kind": "port",
                "sample": "oracle/allcores.py on the one-core leg's sample (%d^3 cells, %d particles): every stage spread over %d worker "
                          "processes by x-slab (gridding, field algebra, |F|^2, both histograms) + threaded 3-D transforms: %s"
                          % (Nc, Npc, nthr, ", ".join("%s %.2fs" % kv for kv in t2["stage_seconds"].items())),
                "seconds": t2["seconds"], "stage_seconds": t2["stage_seconds"],
                "seconds_by_workers": {str(k): v.get("seconds", v.get("error")) for k, v in runs.items()},
                "speedup_over_one_core": t1["total"] / t2["seconds"],
                "extrapolated_seconds_full_size": t2["seconds"] * scale,
                "cpu_model": cpu_model(), "host_cores": ncpu}
        else:
            out["cpu_baseline_allcores"] = {"skipped": str(runs)}
    elif route != "script":
        Ns, Nps_ = sample_size(route, N, Np)
        pos, vel, _, dens = synth.particles(synth.BASE_SEED + 100 + off, Nps_, L, lognormal)
        nthr = os.cpu_count() or 1
        orc.set_fft_workers(nthr)
        try:
            _, t2 = oracle_tables(route, quantities, flavour, Ns, L, pos, vel, dens)
        finally:
            orc.set_fft_workers(1)
        out["cpu_baseline_allcores"] = {
            "value": Ns ** 3 * nfields / t2["total"], "unit": unit, "cores": nthr, "kind": "port",
            "sample": "the %d^3 parity sample (%d particles); the 3-D transforms threaded over %d cores (scipy.fft workers), "
                      "the kd-tree search and histograms remain single-threaded numpy: %s"
                      % (Ns, Nps_, nthr, ", ".join("%s %.2fs" % kv for kv in t2.items())),
            "seconds": t2["total"], "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
    return out


def hbm_probe(torch, nbytes=4 << 30):
    """Plain streaming rates of THIS box's HBM through the runtime's own fill and copy kernels (after all timing, workloads
    freed): what a kernel that only writes (the pencil kernel's stores) or reads and writes (the y pass) a large buffer
    can hope for at best.  Information beside `roofline`, whose peak stays the guide's 8 TB/s."""
    try:
        n = nbytes // 4
        a = torch.empty(n, dtype=torch.float32, device="cuda")
        b = torch.empty(n, dtype=torch.float32, device="cuda")

        def timed(f, reps=10):
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3
        tf, tc = timed(lambda: a.zero_()), timed(lambda: b.copy_(a))
        del a, b
        return {"buffer_GB": round(nbytes / 1e9, 2), "fill_GBs": round(nbytes / tf / 1e9, 1),
                "copy_GBs_read_plus_written": round(2 * nbytes / tc / 1e9, 1),
                "note": "torch zero_() / copy_() on the bench box; reference points, not the roofline's peak"}
    except Exception as e:      # (never let a probe take the line down)
        return {"skipped": repr(e)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))

    if args.dry_run:
        raise SystemExit(dry_run(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # VPS_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, exchanges staged through host memory); the real runs use RCCL.
    backend = os.environ.get("VPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from vpower import device, synth
    K = device.default_kernels(local)
    comm = device.SlabComm()
    decomposition = args.decomposition
    if decomposition == "auto":
        # `value` is the decomposition the metric names (BASELINE.json config 4 / 5: "slab all-to-all"); the field-parallel
        # split -- the better fit for point-to-point xGMI while a grid fits one GPU, DESIGN.md section 4 -- is the `alternative`
        decomposition = "slab"
    if decomposition == "fields" and world > 1:
        if args.decomposition == "fields" and args.config in synth.CONFIGS and not fields_possible(
                K, int(os.environ.get("VPS_BENCH_GRID", synth.CONFIGS[args.config][0])), synth.WORKLOADS[args.config][0],
                synth.WORKLOADS[args.config][1], args.unfused):
            raise SystemExit("--decomposition fields: %s does not fit -- every rank would hold whole grids (fused NGP path, about "
                             "26 N^3 bytes of one GPU's HBM); use --decomposition slab" % args.config)
        comm = device.FieldComm()
    elif os.environ.get("VPS_BENCH_TRANSPORT") == "library" and ((world > 1 and backend == "nccl")
                                                                   or (world == 1 and os.environ.get("VPS_FORCE_COLLECTIVES") == "1")):
        # (one rank with VPS_FORCE_COLLECTIVES=1: the library's chunk pipeline and RCCL plumbing on a single GPU -- a rehearsal)
        # the exchange inside libvps_hip.so (vps_spectrum_zimages: RCCL send / recv groups on the library's own stream)
        # instead of torch.distributed.all_to_all_single; torch only moves the 128-byte id and times the run
        comm = device.LibraryComm(K)
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

    cfg = args.config
    if cfg not in synth.CONFIGS:
        raise SystemExit("unknown config %s" % cfg)
    single = world == 1 and comm.world == 1
    if single and not os.environ.get("VPS_BENCH_GRID") and (
            26.0 * float(synth.CONFIGS[cfg][0]) ** 3 >= 0.78 * torch.cuda.get_device_properties(local).total_memory):
        raise SystemExit("%s: a %d^3 grid does not fit one GPU -- run it on its ranks (--gpus 8) or time one rank's share "
                         "(--emulate-ranks 8)" % (cfg, synth.CONFIGS[cfg][0]))
    res = run_config(args, cfg, K, comm, world, rank, backend, args.steps, args.warmup, args.profile_steps,
                     want_parity=not args.no_parity and (comm.world == world or getattr(comm, "field_world", 1) == world),
                     want_cpu=single and not args.no_cpu_baseline)
    out = {
        "metric": baseline_json()["metric"],
        "value_definition": "N^3 x scalar fields / step time: the whole path particles -> P(k) tables of the config's "
                            "quantities; stage rates in particles_per_s and fft_cells_per_s, HBM fraction in roofline",
        "value": res.pop("value"), "unit": "grid cells*components/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": res.pop("data"),
    }
    out.update(res)
    if single and cfg == "C4" and not args.no_other_configs:
        # the two single-GPU configs of BASELINE.json, short runs, reported beside the headline
        other = {}
        for c in ("C2", "C3"):
            r = run_config(args, c, K, comm, world, rank, backend, steps=5, warmup=2, profile_steps=2,
                           want_parity=not args.no_parity, want_cpu=False)
            other[c] = {k: r[k] for k in ("ms_per_step", "value", "config", "particles_per_s", "fft_cells_per_s", "fft_stage",
                                          "kernel_ms_per_step", "roofline", "per_kernel_frac_of_hbm_peak", "parity", "full_size_check", "gridding",
                                          "particles_per_s_sort_only", "gridding_note", "launch_ms", "nn_query_note") if k in r}
            other[c]["steps"] = 5
        out["other_configs"] = other
    if rank == 0:
        # device memory: the library's buffers are torch allocations (device.Kernels.empty / workspace), so torch's own
        # high-water mark covers them; the driver's free / total as a cross-check
        free_b, total_b = torch.cuda.mem_get_info()
        out["device_memory"] = {"peak_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
                                "peak_reserved_GB": round(torch.cuda.max_memory_reserved() / 1e9, 2),
                                "total_GB": round(total_b / 1e9, 2)}
    if single and rank == 0:
        out["hbm_probe"] = hbm_probe(torch)
    if rank == 0 and single and not args.no_cpu_baseline:
        out.update(cpu_baseline_legs(cfg))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
