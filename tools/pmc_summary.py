#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/gpu_profile_round.sh:
   python tools/pmc_summary.py gpurun_out/prof/r03 profiles/r03
   -> <dst>_pmc_summary.json   per kernel: the launch with the largest FETCH_SIZE / WRITE_SIZE (raw KB), corrected HBM
                               bytes, VALUBusy / LDSBankConflict / MemUnitStalled of its longest launch
      <dst>_pmc_traffic.json   {config: {kernel family: HBM bytes of its main launch}} -- what bench.py copies into
                               roofline.traffic
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of the bytes of coalesced streaming reads);
both counters are in KB."""
import csv, json, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.split(r"\((?![a-z]*\))", name)[0].strip()


def per_kernel(path, counters):
    mx = defaultdict(dict)
    if not os.path.exists(path):
        return mx
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c not in counters:
            continue
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        if c not in mx[k] or v > mx[k][c]:
            mx[k][c] = v
    return mx


f = per_kernel(src + "_pmc_fetch.csv", ("FETCH_SIZE",))
w = per_kernel(src + "_pmc_write.csv", ("WRITE_SIZE",))
u = per_kernel(src + "_pmc_valu.csv", ("VALUBusy", "LDSBankConflict", "MemUnitStalled"))
summary = {}
for k in sorted(set(f) | set(w) | set(u)):
    if k.startswith("at::") or "rocclr" in k:
        continue
    fe, wr = f.get(k, {}).get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
    summary[k] = {"FETCH_SIZE_KB_max": fe, "WRITE_SIZE_KB_max": wr,
                  "hbm_bytes_corrected": int(2 * 1024 * fe + 1024 * wr), **{c: v for c, v in u.get(k, {}).items()}}
json.dump(summary, open(dst + "_pmc_summary.json", "w"), indent=1)

# main launch of each hot kernel family per config (template arguments carry the line length: N = 2048 -> C4, ...)
fam = {
    "C4": {"fft_z": "pencil_fft_z_kernel<1024, 8, false>", "fft_y": "fft_transpose_pass_wide<2048, 8, 128, true>",
           "fft_x": "fft_x_pass<2048, 2, 0"},
    "C2": {"fft_z": "pencil_fft_z_kernel<256, 16, false>", "fft_y": "fft_transpose_pass<512, 16, false, true, false",
           "fft_x": "fft_x_pass<512, 8, 0"},
    "C3": {"fft_z": "fft_transpose_pass<512, 8, true", "fft_y": "fft_transpose_pass_wide<1024, 8, 64, true>",
           "fft_x": "fft_x_pass<1024, 4, 0", "nn_query": "nn_column_kernel<float, 4>"},
}
traffic = {"_note": "HBM bytes of the MAIN (largest) launch of each hot kernel, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in "
                    "separate passes over `python3 bench.py` (tools/gpu_profile_round.sh), FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md section HBM, KB*1024; raw per-kernel maxima in *_pmc_summary.json. For kernels whose "
                    "launches differ in size (x pass / pencil kernel of a vector field vs a scalar field) this is the "
                    "largest one."}
for cfg, m in fam.items():
    traffic[cfg] = {}
    for key, prefix in m.items():
        hits = [v["hbm_bytes_corrected"] for k, v in summary.items() if k.startswith(prefix)]
        if hits:
            traffic[cfg][key] = max(hits)
# the kernel sources the counters belong to: bench.py drops the figure when they have changed since
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
traffic["kernel_sources_sha16"] = bench.kernel_sources_sha16()
json.dump(traffic, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
