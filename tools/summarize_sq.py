#!/usr/bin/env python3
"""Condenses a tools/pmc_sq2.sh run (gpurun_out/pmc2_<tag>/) into profiles/<round>/<key>_sq.json: per-launch means of the SQ
counters of k_step and the ratios DESIGN.md quotes.  usage: tools/summarize_sq.py <tag> <key e.g. G_f64_65536> <round dir>"""
import collections
import csv
import glob
import json
import sys

tag, key, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for sub in "abc":
    for f in glob.glob(f"gpurun_out/pmc2_{tag}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_step" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
waves = c.get("SQ_WAVES", 1.0)
out = {"kernel": "k_step", "launches_averaged": len(next(iter(acc.values()))), "per_launch": c,
       "per_wavefront": {k: v / waves for k, v in c.items() if k != "SQ_WAVES"},
       "ratios": {
           "valu_active_over_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
           "wait_any_over_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
           "lds_active_over_wave_cycles": c["SQ_ACTIVE_INST_LDS"] / c["SQ_WAVE_CYCLES"],
           "lds_bank_conflict_over_lds_active": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"],
           "f64_arith_share_of_valu_insts": (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]) / c["SQ_INSTS_VALU"],
       },
       "note": "SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; three separate --pmc passes of bench.py (tools/pmc_sq2.sh)"}
json.dump(out, open(f"profiles/{rnd}/{key}_sq.json", "w"), indent=1)
print(json.dumps(out["ratios"], indent=1))
