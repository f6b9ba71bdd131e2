#!/usr/bin/env python3
"""Condenses a tools/r04_profile.sh (or profile_r01.sh) run (gpurun_out/prof_<tag>/) into profiles/<round>/ + profiles/traffic.json.
bench.py pre-rolls a whole episode before its timed region, so the rocprofv3 stats average thousands of k_step launches; the
summary therefore also carries the average over the LAST <timed> launches of the kernel trace (= the timed region bench.py's own
HIP events bracket) -- the number that has to agree with the bench line's roofline.kernel_ms.
usage: tools/summarize_profile.py <tag> <key e.g. G_f64_65536> <round dir e.g. r03> [timed launches, default 50] [pmc timed, default 10]"""
import csv
import glob
import json
import os
import shutil
import sys

tag, key, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 50
timed_pmc = int(sys.argv[5]) if len(sys.argv) > 5 else 10
base = f"gpurun_out/prof_{tag}"
out_dir = f"profiles/{rnd}"
os.makedirs(out_dir, exist_ok=True)
ks = glob.glob(f"{base}/stats/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"{out_dir}/{key}_kernel_stats.csv")
step_row = [r for r in csv.DictReader(open(ks)) if "k_step" in r["Name"]][0]
out = {"kernel": step_row["Name"][:60] + "...", "calls": int(step_row["Calls"]), "avg_ns": float(step_row["AverageNs"]),
       "min_ns": float(step_row["MinNs"]), "max_ns": float(step_row["MaxNs"]), "pct_of_gpu_time": float(step_row["Percentage"])}
tr = glob.glob(f"{base}/stats/*/*_kernel_trace.csv")
if tr:
    rows_t = [r for r in csv.DictReader(open(tr[0])) if "k_step" in r["Kernel_Name"]]
    rows_t.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows_t[-timed:]]
    out["timed_region"] = {"launches": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d),
                           "note": f"the last {len(d)} k_step launches of the kernel trace = bench.py's timed region (the {len(rows_t) - len(d)} "
                                   "launches before them are the warm-up and the one-episode pre-roll)"}
    if os.path.exists(f"{base}/bench_stats.json"):
        try:
            out["bench_line_of_this_run"] = json.loads(open(f"{base}/bench_stats.json").read().strip().splitlines()[-1])
        except Exception:
            pass
n_env = int(key.split("_")[-1])
for kind in ("fetch", "write"):
    f = glob.glob(f"{base}/pmc_{kind}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows][-timed_pmc:]  # the timed region's launches (dispatch order)
    out[rows[0]["Counter_Name"]] = dict(per_launch_values_KB=vals, mean_KB=sum(vals) / len(vals))
    # (rocprofv3's dispatch record: its VGPR_Count is NOT the compiler's register count -- 124 for a kernel the compiler
    # allocates 246 VGPRs for; the compiler's own remark is attached below as `compiler_resources`)
    out["rocprofv3_dispatch_fields"] = dict(VGPR_Count=int(rows[0]["VGPR_Count"]), Accum_VGPR_Count=int(rows[0]["Accum_VGPR_Count"]),
                                            SGPR_Count=int(rows[0]["SGPR_Count"]), LDS_Block_Size=int(rows[0]["LDS_Block_Size"]),
                                            Scratch_Size=int(rows[0]["Scratch_Size"]), Workgroup_Size=int(rows[0]["Workgroup_Size"]),
                                            Grid_Size=int(rows[0]["Grid_Size"]))
kres = f"{out_dir}/kernel_resources.txt"  # `python tools/kernel_resources.py k_step > profiles/<round>/kernel_resources.txt` (no GPU needed)
if os.path.exists(kres):
    preset, dtype = key.split("_")[0], key.split("_")[1]
    shape = "1+0/1+0" if preset == "T" else "2+2/4+4"
    vw = "VW2" if preset == "T" else "VW8"
    real = "double" if dtype == "f64" else "float"
    rows_k = [ln.strip() for ln in open(kres) if shape in ln and f" {real} {vw} " in ln]
    out["compiler_resources"] = dict(source="hipcc -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py), same sources",
                                     k_step_instantiations=rows_k)
f_, w_ = out["FETCH_SIZE"]["mean_KB"] * 1024, out["WRITE_SIZE"]["mean_KB"] * 1024
out["hbm_bytes_per_launch"] = dict(
    fetch_raw=f_, fetch_corrected_x2=2 * f_, write=w_, total=2 * f_ + w_,
    note="gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); "
         "FETCH_SIZE and WRITE_SIZE come from separate --pmc passes; record loads are contiguous lane-strided runs")
out["hbm_bytes_per_env_step"] = (2 * f_ + w_) / n_env
json.dump(out, open(f"{out_dir}/{key}_summary.json", "w"), indent=1)
tf = "profiles/traffic.json"
t = json.load(open(tf)) if os.path.exists(tf) else {}
t[key] = {"bytes": 2 * f_ + w_, "source": f"{out_dir}/{key}_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, "
                                            f"tools/r04_profile.sh {tag})"}
json.dump(t, open(tf, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("avg_ns", "hbm_bytes_per_env_step")}))
