#!/usr/bin/env python3
"""tools/collect_round.py <tag>: copy what tools/profile_round.sh and tools/final_round.sh left under gpurun_out/ into
profiles/ under the names profiles/README.md lists (run here, after the gpurun call has merged its output back)."""
import csv
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/%s/" % tag
for a, b in [("%s_counts.json" % tag, "%s_counts.json" % tag), ("%s_counts.txt" % tag, "%s_counts_summary.txt" % tag),
             ("%s_issue_rates.json" % tag, "%s_issue_rates.json" % tag), ("issue_rates.txt", "%s_issue_rates.txt" % tag),
             ("kernel_stats.csv", "%s_mega_kernel_stats.csv" % tag),
             ("wf_cornell_mis_kernel_stats.csv", "%s_wavefront_cornell_kernel_stats.csv" % tag),
             ("wf_final_rr_kernel_stats.csv", "%s_wavefront_scene09_kernel_stats.csv" % tag)]:
    shutil.copy(src + a, "profiles/" + b)
for f in ("regions", "top_regions", "scene_regions", "time_random", "bench_scenes"):
    shutil.copy("gpurun_out/%s_%s.txt" % (tag, f), "profiles/%s_%s.txt" % (tag, f))
rows = [r for r in csv.DictReader(open(src + "kernel_trace_once.csv")) if "k_mega" in r["Kernel_Name"]]
with open("profiles/%s_mega_once_per_workload.txt" % tag, "w") as f:
    f.write("one render per workload under rocprofv3 --kernel-trace (order: cornell_mis cornell_literal final_rr final_mis "
            "mis_spheres c5_shard)\n")
    for r in rows:
        f.write("%s  %.3f ms  VGPR %s SGPR %s scratch %s LDS %s\n" %
                (r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6,
                 r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("Scratch_Size", r.get("Private_Segment_Size", "?")),
                 r.get("LDS_Block_Size", "?")))
print(open("profiles/%s_mega_once_per_workload.txt" % tag).read())
print("counts measured on sources", json.load(open("profiles/%s_counts.json" % tag))["_source_hash"])
