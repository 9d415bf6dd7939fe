#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/ into profiles/ (tracked) and rebuilds
profiles/traffic.json from the PMC passes.   usage: python tools/stash_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    return hits[0] if hits else None


for p in glob.glob(os.path.join(src, tag + "_*.json")) + glob.glob(os.path.join(src, tag + "_role_cycles_*.txt")) + glob.glob(os.path.join(src, tag + "_bench_models.txt")) + glob.glob(os.path.join(src, tag + "_tick_trace.txt")):
    if os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, os.path.basename(p)))
for p in ("f32", "mixed", "f64"):
    for kind in ("stats", "stats256"):
        st = one("%s_%s_%s/*/*_kernel_stats.csv" % (tag, kind, p))
        if st:
            shutil.copy(st, os.path.join(dst, "%s_kernel_%s_%s.csv" % (tag, kind, p)))

traffic_path = os.path.join(dst, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
traffic["_comment"] = ("HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB); gfx950 "
                       "correction per MI355X_MICROARCH.md: FETCH_SIZE x2 for reads, WRITE_SIZE exact. valu_active = SQ_ACTIVE_INST_VALU x 4 / "
                       "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) from the instruction-mix pass. bench.py replays these for the matching workload.")


def rows_of(path, counter=None):
    rows = [r for r in csv.DictReader(open(path)) if "synth_kernel" in r["Kernel_Name"]]
    return [r for r in rows if counter is None or r["Counter_Name"] == counter]


for suffix, key_fmt in (("", "batch4096_frames7500_delay2_%s"), ("256", "batch256_frames500_delay1_%s")):
    for p in ("f32", "mixed", "f64"):
        vals, kernel, srcs = {}, None, []
        for counter, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cc = one("%s_%s%s_%s/*/*_counter_collection.csv" % (tag, short, suffix, p))
            if not cc:
                continue
            rows = rows_of(cc, counter)
            if not rows:
                continue
            outp = os.path.join(dst, "%s_pmc_%s_size%s_%s.csv" % (tag, short, suffix, p))
            with open(outp, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)
            vals[short] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
            kernel = rows[0]["Kernel_Name"]
            srcs.append("profiles/" + os.path.basename(outp))
        if len(vals) != 2:
            continue
        entry = {"fetch_size_kib": vals["fetch"], "write_size_kib": vals["write"],
                 "bytes": vals["fetch"] * 1024 * 2 + vals["write"] * 1024, "kernel": kernel, "source": srcs}
        inst = one("%s_inst_%s/*/*_counter_collection.csv" % (tag, p)) if suffix == "" else None
        if inst:
            acc = {}
            for r in rows_of(inst):
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            mean = {k: sum(v) / len(v) for k, v in acc.items()}
            if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
                entry["valu_active"] = mean["SQ_ACTIVE_INST_VALU"] * 4.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                entry["valu_active_source"] = "profiles/%s_pmc_instruction_mix.json (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles)" % tag
                mix = json.load(open(os.path.join(dst, "%s_pmc_instruction_mix.json" % tag))) if os.path.exists(os.path.join(dst, "%s_pmc_instruction_mix.json" % tag)) else {}
                mix[key_fmt % p] = {"kernel": kernel, "counters_mean_per_launch": mean, "valu_active_fraction_of_simd_cycles": entry["valu_active"],
                                    "wave_wait_any_fraction": mean.get("SQ_WAIT_ANY", 0.0) / mean["SQ_WAVE_CYCLES"] if mean.get("SQ_WAVE_CYCLES") else None}
                json.dump(mix, open(os.path.join(dst, "%s_pmc_instruction_mix.json" % tag), "w"), indent=1)
        traffic[key_fmt % p] = entry
json.dump(traffic, open(traffic_path, "w"), indent=1)
print("stashed", tag, sorted(k for k in traffic if not k.startswith("_")))
