#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs of tools/profile_round.sh into <tag>_counts.json: per workload and pipeline, the counter sums
of ONE render (what bench.py's roofline scales to its own sample count) + a readable summary.

    pmc_counts.py <dir> <tag> <workload> [<workload> ...]

Every pass rendered each workload exactly once, in the given order; a render is one k_mega dispatch (+ k_resolve) or,
for the wavefront pipeline, every wf_* dispatch between two wf_init launches."""
import collections
import csv
import json
import os
import sys

out, tag, workloads = sys.argv[1], sys.argv[2], sys.argv[3:]
once = json.load(open(os.path.join(out, "once_bench.json")))
lines = [once] + once.get("extra", [])
info = {}
for name, ln in zip(workloads, lines):
    W, H = [int(x) for x in ln["config"]["workload"].split()[1].split("x")]
    info[name] = {"pipeline": ln["config"]["pipeline"], "samples": None, "line": ln}


def renders(path):
    """dispatch rows grouped per render, in launch order: list of {counter: sum} (+ kernel name)."""
    rows = list(csv.DictReader(open(path)))
    by_dispatch = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        d = by_dispatch.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "c": collections.defaultdict(float)})
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    groups, cur = [], None
    for d in by_dispatch.values():
        k = d["kernel"]
        mega, wf_start = "k_mega" in k, "wf_init" in k
        if mega or wf_start:
            cur = {"kernel": k.split("(")[0], "c": collections.defaultdict(float), "n": 0}
            groups.append(cur)
        if cur is None or not (mega or "wf_" in k or "k_resolve" in k):
            continue
        for cn, v in d["c"].items():
            cur["c"][cn] += v
        cur["n"] += 1
    return groups


counts = {}
# ---- calibration of FETCH_SIZE / WRITE_SIZE on the 8-byte-per-lane stream (4 launches of 1 GiB in + 1 GiB out)
calib = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(out, "calib_%s.csv" % cn)
    if os.path.exists(f):
        tot = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_stream8" in r["Kernel_Name"] and r["Counter_Name"] == cn)
        calib[cn] = {"counter_bytes": tot * 1024, "true_bytes": 4 * (1 << 30), "true_over_counter": 4 * (1 << 30) / (tot * 1024) if tot else None}
passes = {}
# per-dispatch megakernel durations of the once-per-workload trace (kernel_trace_once.csv), in launch order
kernel_ns = {}
kt = os.path.join(out, "kernel_trace_once.csv")
if os.path.exists(kt):
    rows = sorted((r for r in csv.DictReader(open(kt)) if "k_mega" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    for i, r in enumerate(rows):
        kernel_ns[i] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for p in ("sq", "f64", "f32", "fetch", "write", "grbm"):
    f = os.path.join(out, "pmc_%s.csv" % p)
    if os.path.exists(f):
        passes[p] = renders(f)
        assert len(passes[p]) == len(workloads), (p, len(passes[p]), len(workloads))
for i, name in enumerate(workloads):
    ln = info[name]["line"]
    samples = ln["value"] * 1e6 * ln["ms_per_step"] * 1e-3  # Msamples/s x seconds of the single step
    c = {"samples": round(samples), "kernel": passes["sq"][i]["kernel"], "dispatches": passes["sq"][i]["n"]}
    sq = passes["sq"][i]["c"]
    c["insts_valu"] = sq["SQ_INSTS_VALU"]
    c["thread_cycles_valu"] = sq["SQ_THREAD_CYCLES_VALU"]
    wc = sq["SQ_WAVE_CYCLES"]
    c["sq"] = {"wait_frac": sq["SQ_WAIT_ANY"] / wc, "wait_inst_frac": sq["SQ_WAIT_INST_ANY"] / wc,
               "issue_frac": sq["SQ_ACTIVE_INST_ANY"] / wc, "valu_active_frac": sq["SQ_ACTIVE_INST_VALU"] / wc,
               "wave_cycles": wc}
    if "f64" in passes:
        f = passes["f64"][i]["c"]
        c["f64"] = {"add": f["SQ_INSTS_VALU_ADD_F64"], "mul": f["SQ_INSTS_VALU_MUL_F64"], "fma": f["SQ_INSTS_VALU_FMA_F64"],
                    "trans": f["SQ_INSTS_VALU_TRANS_F64"], "int32": f["SQ_INSTS_VALU_INT32"], "cvt": f["SQ_INSTS_VALU_CVT"],
                    "salu": f["SQ_INSTS_SALU"], "smem": f["SQ_INSTS_SMEM"]}
        # flop per wave-level instruction and lane: add / mul / transcendental 1, fma 2
        c["f64_flop"] = f["SQ_INSTS_VALU_ADD_F64"] + f["SQ_INSTS_VALU_MUL_F64"] + f["SQ_INSTS_VALU_TRANS_F64"] + \
            2 * f["SQ_INSTS_VALU_FMA_F64"]
    if "grbm" in passes and kernel_ns.get(i):
        # effective shader clock of the counter pass: GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        c["clock_ghz"] = round(passes["grbm"][i]["c"]["GRBM_GUI_ACTIVE"] / 8.0 / kernel_ns[i], 3)
    if "f32" in passes:
        f = passes["f32"][i]["c"]
        c["f32"] = {k.replace("SQ_INSTS_VALU_", "").replace("SQ_", "").lower(): v for k, v in f.items()}
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md,
    # HBM): the calibration stream of this run (8 bytes per lane, known size) gives the factor applied here
    if "fetch" in passes and "write" in passes:
        kf = (calib.get("FETCH_SIZE") or {}).get("true_over_counter") or 1.0
        kw = (calib.get("WRITE_SIZE") or {}).get("true_over_counter") or 1.0
        c["fetch_bytes"] = passes["fetch"][i]["c"]["FETCH_SIZE"] * 1024 * kf
        c["write_bytes"] = passes["write"][i]["c"]["WRITE_SIZE"] * 1024 * kw
        c["fetch_note"] = "counter x 1024 x the factor measured on an 8-byte-per-lane stream of known size (%.3f / %.3f)" % (kf, kw)
    counts.setdefault(name, {})[info[name]["pipeline"]] = c
# ---- the wavefront pipeline (one render per file)
for w in ("cornell_mis", "final_rr"):
    bj = os.path.join(out, "wf_%s_bench.json" % w)
    if not os.path.exists(bj):
        continue
    ln = json.load(open(bj))
    c = {"samples": round(ln["value"] * 1e6 * ln["ms_per_step"] * 1e-3), "kernel": "wf_extend_ls + wf_shade + wf_connect_ls + wf_compact"}
    for cn, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
        f = os.path.join(out, "wf_%s_%s.csv" % (w, cn))
        if os.path.exists(f):
            raw = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == cn and ("wf_" in r["Kernel_Name"] or "k_resolve" in r["Kernel_Name"])) * 1024
            k = (calib.get(cn) or {}).get("true_over_counter") or 1.0
            c[key + "_raw"] = raw
            c[key] = raw * k
    c["fetch_note"] = "counter x 1024 x the factor measured on an 8-byte-per-lane stream of known size (calibration below)"
    f = os.path.join(out, "wf_%s_sq.csv" % w)
    if os.path.exists(f):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "wf_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        c["stages"] = {k: {"insts_valu": v["SQ_INSTS_VALU"], "lane_util": v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_INSTS_VALU"], 1) / 64,
                           "wait_frac": v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)} for k, v in acc.items() if v["SQ_WAVE_CYCLES"] > 1e6}
        c["insts_valu"] = sum(v["SQ_INSTS_VALU"] for v in acc.values())
        c["thread_cycles_valu"] = sum(v["SQ_THREAD_CYCLES_VALU"] for v in acc.values())
    counts.setdefault(w, {})["wavefront"] = c
if calib:
    counts["_calibration_8B_per_lane"] = calib
# what the counters were measured on: bench.py compares this with the library it runs (roofline.stale)
import importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
counts["_source_hash"] = importlib.import_module("ray_tracing-rendering_amd.build").source_hash()
json.dump(counts, open(os.path.join(out, "%s_counts.json" % tag), "w"), indent=1, sort_keys=True)
with open(os.path.join(out, "%s_counts.txt" % tag), "w") as f:
    for cn, v in calib.items():
        f.write("calibration %s: counter %.4g bytes for %.4g true bytes (factor %.3f)\n" % (cn, v["counter_bytes"], v["true_bytes"], v["true_over_counter"] or 0))
    for name in workloads:
        for pipe, c in counts[name].items():
            if pipe == "wavefront":
                f.write("%s [wavefront] samples %.4g  HBM-side bytes per render: fetch %.4g write %.4g (%.1f B/sample)\n" % (
                    name, c["samples"], c.get("fetch_bytes", 0), c.get("write_bytes", 0),
                    (c.get("fetch_bytes", 0) + c.get("write_bytes", 0)) / c["samples"]))
                for k, v in c.get("stages", {}).items():
                    f.write("    %-40s VALU insts %.4g  lane_util %.1f%%  wait %.1f%%\n" % (k, v["insts_valu"], 100 * v["lane_util"], 100 * v["wait_frac"]))
                continue
            lu = c["thread_cycles_valu"] / c["insts_valu"] / 64
            f.write("%s [%s] %s\n  samples %.4g  VALU insts/sample %.1f  lane_util %.1f%%  wait %.1f%%  issue %.1f%%  valu-active %.1f%%\n" % (
                name, pipe, c["kernel"][:70], c["samples"], c["insts_valu"] / c["samples"], 100 * lu,
                100 * c["sq"]["wait_frac"], 100 * c["sq"]["issue_frac"], 100 * c["sq"]["valu_active_frac"]))
            if "f64" in c:
                tot = c["insts_valu"]
                f.write("  of VALU: f64 add %.1f%% mul %.1f%% fma %.1f%% trans %.1f%%  int32 %.1f%% cvt %.1f%%;  SALU/VALU %.2f  SMEM/VALU %.3f\n" % (
                    100 * c["f64"]["add"] / tot, 100 * c["f64"]["mul"] / tot, 100 * c["f64"]["fma"] / tot,
                    100 * c["f64"]["trans"] / tot, 100 * c["f64"]["int32"] / tot, 100 * c["f64"]["cvt"] / tot,
                    c["f64"]["salu"] / tot, c["f64"]["smem"] / tot))
            if "fetch_bytes" in c:
                f.write("  HBM-side bytes per render: fetch %.4g  write %.4g  (%.1f B/sample)\n" % (
                    c["fetch_bytes"], c["write_bytes"], (c["fetch_bytes"] + c["write_bytes"]) / c["samples"]))
print(open(os.path.join(out, "%s_counts.txt" % tag)).read())
