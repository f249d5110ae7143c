"""ctypes binding of libvps_hip.so (include/vps_hip.h).

This is the only place the shared library is loaded.  There is NO CPU fallback: if the
library is missing or a call fails the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("VPS_LIB_PATH") or os.path.join(_HERE, "libvps_hip.so")  # override: experiments only
SOURCES = ("api.hip", "deposit.hip", "nn.hip", "fft.hip", "hist.hip", "preprocess.hip", "comm.hip")
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics")

# every symbol include/vps_hip.h declares: (name, restype, argtypes)
_i64 = C.c_int64
_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
SYMBOLS = (
    ("vps_create", C.c_int, (C.POINTER(_vp), C.c_int)),
    ("vps_destroy", C.c_int, (_vp,)),
    ("vps_last_error", C.c_char_p, (_vp,)),
    ("vps_set_stream", C.c_int, (_vp, _vp)),
    ("vps_sync", C.c_int, (_vp,)),
    ("vps_version", C.c_int, ()),
    ("vps_set_option", C.c_int, (C.c_char_p, C.c_double)),
    ("vps_get_option", C.c_double, (C.c_char_p, C.c_double)),
    ("vps_device_info", C.c_int, (_vp, C.POINTER(_i64))),
    ("vps_binning_mode", C.c_int, (_vp,)),
    ("vps_malloc", C.c_int, (_vp, C.POINTER(_vp), C.c_size_t)),
    ("vps_free", C.c_int, (_vp, _vp)),
    ("vps_memset", C.c_int, (_vp, _vp, C.c_int, C.c_size_t)),
    ("vps_memcpy_h2d", C.c_int, (_vp, _vp, _vp, C.c_size_t)),
    ("vps_memcpy_d2h", C.c_int, (_vp, _vp, _vp, C.c_size_t)),
    ("vps_timing_enable", C.c_int, (_vp, C.c_int)),
    ("vps_timing_reset", C.c_int, (_vp,)),
    ("vps_timing_get", C.c_int, (_vp, C.c_int, C.POINTER(_i64), _dp)),
    ("vps_timing_list", C.c_int, (_vp, C.c_int, _dp, _i64, C.POINTER(_i64))),
    ("vps_preprocess", C.c_int, (_vp, _vp, C.c_int, _vp, _vp, _i64, C.c_int, C.c_int, _dp, _dp)),
    ("vps_totals", C.c_int, (_vp, _vp, _i64, _i64, _vp, _i64, _dp)),
    ("vps_cell_index", C.c_int, (_vp, _vp, C.c_int, _i64, C.c_int, C.c_double, _vp)),
    ("vps_deposit_workspace_bytes", C.c_size_t, (_i64, C.c_int, C.c_int, C.c_int)),
    ("vps_deposit_ngp", C.c_int, (_vp, _vp, C.c_int, _vp, _i64, C.c_int, C.c_int, C.c_double,
                                  C.c_int, C.c_int, _vp, _vp)),
    ("vps_deposit_field", C.c_int, (_vp, _vp, C.c_int, _vp, _vp, _i64, C.c_int, C.c_double, C.c_int,
                                    C.c_int, C.c_int, C.c_int, _vp, _vp)),
    ("vps_deposit_fft_zy_supported", C.c_int, (_vp, C.c_int, C.c_int)),
    ("vps_deposit_fft_zy_workspace_bytes", C.c_size_t, (_i64, C.c_int, C.c_int)),
    ("vps_deposit_fft_zy_workspace_bytes_shared", C.c_size_t, (_i64, C.c_int, C.c_int)),
    ("vps_deposit_fft_zy", C.c_int, (_vp, _vp, C.c_int, _vp, _vp, _i64, C.c_int, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_int, _vp, _vp, _vp)),
    ("vps_density_velocity_vector", C.c_int, (_vp, _vp, _vp, _i64, _vp)),
    ("vps_nn_workspace_bytes", C.c_size_t, (_i64, C.c_int, _i64)),
    ("vps_nn_resample_field", C.c_int, (_vp, _vp, C.c_int, _vp, _i64, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int,
                                        C.c_double, _vp, _vp, _vp)),
    ("vps_nn_resample_quantity", C.c_int, (_vp, _vp, C.c_int, _vp, _i64, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int,
                                           C.c_double, C.c_int, C.c_int, _vp, _vp, _vp)),
    ("vps_nn_resample", C.c_int, (_vp, _vp, C.c_int, _vp, _i64, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                  _dp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp)),
    ("vps_field_algebra", C.c_int, (_vp, C.c_int, C.c_int, C.c_double, _vp, _i64)),
    ("vps_field_algebra_out", C.c_int, (_vp, C.c_int, C.c_int, C.c_double, _vp, _i64, _vp)),
    ("vps_fft_supported", C.c_int, (C.c_int,)),
    ("vps_set_binning", C.c_int, (_vp, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_double)),
    ("vps_set_bin_only", C.c_int, (_vp, C.c_int)),
    ("vps_set_window", C.c_int, (_vp, C.c_int, _vp)),
    ("vps_assign_expand", C.c_int, (_vp, _vp, C.c_int, _vp, _i64, C.c_int, C.c_int, C.c_double, C.c_int, _vp, _vp)),
    ("vps_fft_workspace_bytes", C.c_size_t, (C.c_int, C.c_int)),
    ("vps_fft_zy", C.c_int, (_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp)),
    ("vps_fft_zy_weighted", C.c_int, (_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp)),
    ("vps_fft_zimage_bytes", C.c_size_t, (C.c_int, C.c_int)),
    ("vps_fft_z", C.c_int, (_vp, C.c_int, C.c_int, _vp, _vp, _vp)),
    ("vps_deposit_fft_z_workspace_bytes", C.c_size_t, (_i64, C.c_int, C.c_int)),
    ("vps_deposit_fft_z", C.c_int, (_vp, _vp, C.c_int, _vp, _vp, _i64, C.c_int, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.c_int, _vp, _vp)),
    ("vps_count_in_slab", _i64, (_vp, _vp, C.c_int, _i64, C.c_int, C.c_double, C.c_int, C.c_int)),
    ("vps_deposit_fft_z_workspace_bytes_slab", C.c_size_t, (_i64, _i64, C.c_int, C.c_int)),
    ("vps_deposit_fft_z_slab", C.c_int, (_vp, _vp, C.c_int, _vp, _vp, _i64, _i64, C.c_int, C.c_double, C.c_int, C.c_int,
                                         C.c_int, C.c_int, _vp, _vp)),
    ("vps_fft_y_chunk_elems", _i64, (C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)),
    ("vps_fft_y_packed", C.c_int, (_vp, C.c_int)),
    ("vps_fft_y_chunk_block", _i64, (_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)),
    ("vps_fft_y", C.c_int, (_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp)),
    ("vps_fft_x", C.c_int, (_vp, C.c_int, _i64, _i64, C.c_int, _vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp)),
    ("vps_fft_x_bin", C.c_int, (_vp, C.c_int, _i64, _i64, C.c_int, C.POINTER(_vp), C.c_int, C.c_int, _i64, C.c_int,
                               _vp, _vp)),
    ("vps_fft_x_bin_chunk", C.c_int, (_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp),
                                     C.c_int, C.c_int, _vp, _vp)),
    ("vps_comm_unique_id", C.c_int, (C.c_char_p,)),
    ("vps_comm_create", C.c_int, (_vp, C.c_int, C.c_int, C.c_char_p)),
    ("vps_comm_destroy", C.c_int, (_vp,)),
    ("vps_comm_info", C.c_int, (_vp, C.POINTER(C.c_int), C.POINTER(C.c_int))),
    ("vps_spectrum_zimages_workspace_bytes", C.c_size_t, (C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)),
    ("vps_spectrum_zimages", C.c_int, (_vp, C.c_int, C.c_int, C.POINTER(_vp), C.c_int, C.c_int, _vp, C.c_int, _vp, _vp)),
    ("vps_allreduce_shells", C.c_int, (_vp, _vp, _vp, C.c_int)),
    ("vps_power_workspace_bytes", C.c_size_t, (C.c_int,)),
    ("vps_power_bin", C.c_int, (_vp, C.c_int, _vp, _vp, _vp, _vp)),
    ("vps_rfft3", C.c_int, (_vp, C.c_int, _vp, _vp, _vp)),
    ("vps_power_grid", C.c_int, (_vp, C.c_int, _vp, _vp, _vp)),
    ("vps_pair_k", C.c_int, (_vp, C.c_int, _dp, _dp, _dp, _vp)),
    ("vps_hist_pairs", C.c_int, (_vp, _vp, _vp, _i64, _dp, C.c_int, _vp, _vp)),
)

K_DEPOSIT, K_ALGEBRA, K_FFT_Z, K_FFT_Y, K_FFT_X, K_NN_BUILD, K_NN_QUERY, K_MISC, K_EXCHANGE, K_EXCHANGE_WAIT = range(10)
KERNEL_KINDS = {"deposit": K_DEPOSIT, "algebra": K_ALGEBRA, "fft_z": K_FFT_Z, "fft_y": K_FFT_Y,
                "fft_x": K_FFT_X, "nn_build": K_NN_BUILD, "nn_query": K_NN_QUERY, "misc": K_MISC,
                "exchange": K_EXCHANGE, "exchange_wait": K_EXCHANGE_WAIT}


ABI_VERSION = 6   # include/vps_hip.h: VPS_ABI_VERSION
FFT_PARTS = 4   # fft.hip is compiled once per family of line lengths (-DVPS_FFT_PART=k)


class VpsError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into vpower/libvps_hip.so (in tree)."""
    import glob
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    # every header is a dependency of every unit (scan.h, vps_internal.h, the ABI header, ...)
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(REPO_ROOT, "include", "*.h")))
    deps = srcs + headers
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    # the flags are part of what an object was built from: a change rebuilds everything
    stamp_path = os.path.join(objdir, "flags.stamp")
    stamp = " ".join((hipcc,) + HIPCC_FLAGS + tuple(SOURCES)) + " parts=%d" % FFT_PARTS
    try:
        same_flags = open(stamp_path).read() == stamp
    except OSError:
        same_flags = False
    if not same_flags:
        force = True
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    units = []
    for s in srcs:
        stem = os.path.basename(s).replace(".hip", "")
        if stem == "fft":
            # one object per family of line lengths (fft.hip: "translation-unit split"), compiled side by side
            units += [(s, os.path.join(objdir, "fft_p%d.o" % k), ["-DVPS_FFT_PART=%d" % k]) for k in range(FFT_PARTS)]
        elif stem == "nn":
            # the device assembly is kept for the static check of the hand-made LDS pipeline (vpower/_asmcheck.py)
            units.append((s, os.path.join(objdir, stem + ".o"), ["-save-temps=obj"]))
        else:
            units.append((s, os.path.join(objdir, stem + ".o"), []))
    for s, o, extra in units:
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), newest_header):
            continue
        cmd = [hipcc, *HIPCC_FLAGS, *extra, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise VpsError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    _check_nn_pipeline(objdir, rebuilt=any(c[-1].endswith("nn.o") for c, _ in procs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise VpsError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    with open(stamp_path, "w") as f:
        f.write(stamp)
    return LIB_PATH


NN_CHECK_REPORT = os.path.join(CSRC, "build", "nn_pipeline_check.json")


def _source_sha(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def _check_nn_pipeline(objdir, rebuilt):
    """nn.hip's column kernel requests LDS reads in one inline-assembly statement and waits for them in another; between the
    two the destination registers must not be touched.  Checked on the device assembly of THIS build (kept by
    -save-temps); the verdict is written next to the objects, the temporaries are removed.  A violation stops the build."""
    import glob, json
    from . import _asmcheck
    asm = os.path.join(objdir, "nn-hip-amdgcn-amd-amdhsa-gfx950.s")
    if rebuilt:
        if not os.path.exists(asm):
            raise VpsError("nn.hip was compiled but its device assembly (%s) is missing: cannot check the LDS pipeline" % asm)
        n, bad = _asmcheck.check(asm)
        with open(NN_CHECK_REPORT, "w") as f:
            json.dump({"requests": n, "violations": len(bad), "source_sha16": _source_sha(os.path.join(CSRC, "nn.hip")),
                       "first": ["line %d: %s: %s" % b for b in bad[:5]]}, f)
        for tmp in glob.glob(os.path.join(objdir, "nn-hip-*")) + glob.glob(os.path.join(objdir, "nn-host-*")) + \
                glob.glob(os.path.join(objdir, "nn.hip-hip-*")):
            os.remove(tmp)
        if bad or n == 0:
            raise VpsError("nn.hip: the inline-assembly LDS pipeline is not intact in this build (%d requests, %d violations; %s)"
                           % (n, len(bad), NN_CHECK_REPORT))


_lib = None


def lib():
    """The loaded library with typed entry points; raises VpsError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VpsError(
            "libvps_hip.so is not built (%s). Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:
        raise VpsError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise VpsError("libvps_hip.so lacks symbol %s" % name) from e
        fn.restype = res
        fn.argtypes = list(args)
    _lib = handle
    _apply_env_options()
    return _lib


# Switches of the library (include/vps_hip.h: vps_set_option).  The library itself never reads the environment; the
# variables VPS_OPT_<NAME> (e.g. VPS_OPT_NN_KAPPA=1.3) are mapped ONCE, here, and recorded in OPTIONS so that a run can
# report them.
OPTION_NAMES = ("no_fast_binning", "no_pair_binning", "nn_query_centric", "nn_column", "nn_build_atomic", "nn_kappa", "nn_stats", "sort_groups",
                "sort_staged", "sort_atomic", "nn_ablate", "no_int_binning", "x_wg_per_cu")
OPTIONS = {}


def set_option(name, value):
    """value None restores the default."""
    v = float("nan") if value is None else float(value)
    if lib().vps_set_option(name.encode(), v) != 0:
        raise VpsError("unknown library option %r" % name)
    if value is None:
        OPTIONS.pop(name, None)
    else:
        OPTIONS[name] = float(value)


class option:
    """with _ffi.option("nn_query_centric", 1): ...  -- sets a switch for the duration of the block."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = OPTIONS.get(self.name)
        set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def _apply_env_options():
    for n in OPTION_NAMES:
        v = os.environ.get("VPS_OPT_" + n.upper())
        if v is not None:
            try:
                set_option(n, float(v))
            except ValueError:
                raise VpsError("VPS_OPT_%s=%r is not a number" % (n.upper(), v))


def as_dp(arr):
    return arr.ctypes.data_as(_dp)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
