#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/gpu_profile_round.sh into the judged artefacts:
   python tools/pmc_summary.py gpurun_out/prof/r01 profiles/r01
   -> profiles/r01_final_pmc_summary.json (per-kernel max FETCH_SIZE / WRITE_SIZE of any launch, raw KB)
      profiles/r01_pmc_traffic.json       (HBM bytes of the MAIN launch of each hot kernel, corrected)
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of coalesced streaming reads); units KB."""
import csv, json, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def per_kernel(path, counter):
    mx = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.split(r"\((?![a-z]*\))", name)[0].strip()
        mx[name] = max(mx[name], float(r["Counter_Value"]))
    return mx


f = per_kernel(src + "_pmc_fetch.csv", "FETCH_SIZE")
w = per_kernel(src + "_pmc_write.csv", "WRITE_SIZE")
summary = {k: {"FETCH_SIZE_KB_max": f.get(k, 0.0), "WRITE_SIZE_KB_max": w.get(k, 0.0),
               "hbm_bytes_corrected": int(2 * 1024 * f.get(k, 0.0) + 1024 * w.get(k, 0.0))}
           for k in sorted(set(f) | set(w)) if not k.startswith("at::") and "rocclr" not in k}
json.dump(summary, open(dst + "_final_pmc_summary.json", "w"), indent=1)

keymap = {"fft_z_fused": "pencil_fft_z_kernel", "fft_z": "fft_transpose_pass<512, 16, true>",
          "fft_y": "fft_transpose_pass<512, 16, false>", "fft_x": "fft_x_pass<512, 8, 0",
          "brick_accumulate": "brick_accumulate_kernel", "brick_rank": "brick_rank_kernel",
          "brick_scatter": "brick_scatter_kernel", "sort_hist": "sort_hist_kernel",
          "sort_scatter": "sort_scatter_staged_kernel", "sort_fine": "sort_fine_kernel"}
traffic = {"_note": "HBM bytes per MAIN launch at C2 (512^3, 1e7 particles), rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                    "in separate passes (tools/gpu_profile_round.sh + tools/pmc_summary.py), FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md section HBM (gfx950 reports half of coalesced streaming reads), KB*1024; raw "
                    "per-kernel maxima in the *_final_pmc_summary.json next to this file"}
try:
    traffic.update({k: v for k, v in json.load(open(dst + "_pmc_traffic.json")).items() if k != "_note"})
except Exception:
    pass
for key, pat in keymap.items():
    hits = [v["hbm_bytes_corrected"] for k, v in summary.items() if k.startswith(pat)]
    if hits:
        traffic[key] = max(hits)
json.dump(traffic, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
