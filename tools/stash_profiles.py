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


for name in ("bench.json", "bench_f64.json", "bench_batches.jsonl", "bench_config4.jsonl", "bench_aux.json", "bench_m5.json", "bench_model5.json",
             "config3_parity.json", "model5_parity.json", "role_cycles_f64_u1.txt", "role_cycles_f32_u1.txt",
             "role_cycles_f32_u4.txt", "role_cycles_m5.txt"):
    p = os.path.join(src, "%s_%s" % (tag, name))
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
for p in ("f32", "f64", "m5"):
    st = one("%s_stats_%s/*/*_kernel_stats.csv" % (tag, p))
    if st:
        shutil.copy(st, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, p)))
traffic = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB); gfx950 "
           "correction per MI355X_MICROARCH.md: FETCH_SIZE x2 for reads, WRITE_SIZE exact. bench.py reports these bytes as "
           "roofline.traffic for the matching workload."}
for p in ("f64", "f32", "m5"):
    vals, kernel, srcs = {}, None, []
    for counter, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        cc = one("%s_%s_%s/*/*_counter_collection.csv" % (tag, short, p))
        if not cc:
            continue
        rows = [r for r in csv.DictReader(open(cc)) if "synth_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        out = os.path.join(dst, "%s_pmc_%s_size_%s.csv" % (tag, short, p))
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        vals[short] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        kernel = rows[0]["Kernel_Name"]
        srcs.append("profiles/" + os.path.basename(out))
    if len(vals) == 2:
        traffic["batch256_frames500_delay1_" + ("f64_model5" if p == "m5" else p)] = {"fetch_size_kib": vals["fetch"], "write_size_kib": vals["write"],
                                                     "bytes": vals["fetch"] * 1024 * 2 + vals["write"] * 1024, "kernel": kernel, "source": srcs}
if len(traffic) > 1:
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("stashed", tag)
