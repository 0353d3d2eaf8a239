#!/usr/bin/env python3
"""Benchmark of the integrator loop on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline = BASELINE.json's metric: Msamples/s of the Cornell box 800x800, spp 400, depth 50, MIS
integrator 4 (scene 21, the reference's Cornell box with NEE/MIS: SURVEY F1).  A step is one full
render.  Image tiles (16x16, the reference's own work unit, renderer/renderer.h:40-62) are dealt
round-robin to the ranks; there is no data-path collective: each rank writes its tiles into its own
device framebuffer and, after the timed region, sends the tiles it owns (nothing else) to rank 0
(SURVEY 8e).  Total work is fixed as N grows ("strong" scaling on the named config).

Rank 0 prints ONE JSON line.  `value` is whole-job Msamples/s = W*H*spp*K / max-over-ranks seconds,
with scene and framebuffer resident in HBM.  At N = 1 the same run also
  * times every other BASELINE configuration (`extra`: scene07/i4 literal, scene09/i1, scene22/i4,
    scene23/i4 1080p spp 1024, and ONE rank's eighth of C5 = scene21 4096x4096 spp 4096), each with
    its own roofline and cpu_baseline;
  * renders a 64x64 crop of every configuration at its FULL spp, with the chunking the timed render
    used, and compares it with the CPU oracle on the same seeds (`parity`);
  * times the reference's own tile-threaded renderer on the host cores (`cpu_baseline`).

`roofline`: the megakernel keeps path state in registers and LDS, so its binding limit is not HBM
(profiles/: 0.06 % of the algorithmic bytes reach memory) but the vector unit.  `bound` is "fp64_valu" and
`frac` = COUNTED FP64 flop / 78.6 TFLOP/s: the SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 class counters of the
committed rocprofv3 pass (profiles/rNN_counts.json, scaled to this run's sample count; fma = 2 flop) x 64 lanes x
lane utilisation / the HIP-event kernel time.  Beside it `issue` says how full the SIMDs' instruction issue was:
every counted instruction class priced with its MEASURED cost (profiles/rNN_issue_rates.json, tools/issue_rates.py:
cycles per SIMD and wave-instruction with four waves resident -- FP64 add / mul / fma / compare 2.8, v_rcp / v_rsq
f64 10.2, 32-bit VALU 1.7, scalar ALU 2.8) against SIMDs x clock x time.  The counter file carries a hash of the
sources it was measured on; `stale` is true when the library that ran is built from other sources.  The figure
SURVEY 8(d) prescribes -- algorithmic state bytes of a wavefront tracer (172 B/sample + 280 B/closest segment +
168 B/shadow segment) over the same time against 8 TB/s -- is kept as `hbm_equivalent`; it is the binding roofline
only when the wavefront pipeline ran, and then `bound` says "hbm".  `traffic` = HBM bytes of one render from
separate FETCH_SIZE / WRITE_SIZE passes (same file).
"""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this host driver needs dmabuf IPC (RCCL / sharing device memory across processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6    # FP64 vector: 256 CUs x 4 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz
B_SAMPLE, B_CLOSEST, B_SHADOW = 172, 280, 168  # SURVEY 8(d), FP64 state
HEADLINE = "cornell_mis"

WORKLOADS = {
    # name: scene id, integrator, W, H, spp, tile_stride (this process renders tiles index % stride == rank)
    "cornell_mis": dict(scene=21, integ=4, W=800, H=800, spp=400),       # C2 headline
    "cornell_literal": dict(scene=7, integ=4, W=800, H=800, spp=400),    # C2 literal: scene07 + integrator4 (near-black)
    "cornell_rr": dict(scene=7, integ=1, W=800, H=800, spp=400),         # lit scene07 (README numbers)
    "final_rr": dict(scene=9, integ=1, W=800, H=800, spp=500),           # C3
    "final_mis": dict(scene=22, integ=4, W=800, H=800, spp=500),         # C3 NEE twin
    "mis_spheres": dict(scene=23, integ=4, W=1920, H=1080, spp=1024),    # C4
    "c5_shard": dict(scene=21, integ=4, W=4096, H=4096, spp=4096, stride=8),  # C5: rank 0's share of 8
}
EXTRAS = ["cornell_literal", "final_rr", "final_mis", "mis_spheres", "c5_shard"]
# CPU-baseline sample per workload on a >= 128-thread host (about 5-25 s each); fewer cores: spp scaled down
CPU_SAMPLE_SPP = {"cornell_mis": 400, "cornell_literal": 200, "cornell_rr": 200, "final_rr": 64, "final_mis": 48,
                  "mis_spheres": 64, "c5_shard": 0}


def load_scene(pkg, scene_id):
    """Flattened scene.  Prefers the product's host builder; falls back to the fixture that was
    walked from the reference's object graph (same bytes, see tests/test_host_scenes.py)."""
    try:
        hs = pkg.hostscene
        return hs.build_scene(scene_id)
    except (ImportError, AttributeError, OSError):
        import gzip
        p = os.path.join(ROOT, "tests", "golden", "scene%02d.rtrs" % scene_id)
        if os.path.exists(p):
            return pkg.Scene.load(p)
        with gzip.open(p + ".gz", "rb") as f:
            return pkg.Scene.from_bytes(f.read())


ISSUE_COSTS = {}
LIB_HASH = None


def load_issue_costs():
    """Newest profiles/rNN_issue_rates.json (tools/issue_rates.py): cycles a wave spends per instruction with four waves
    on a SIMD, folded to cycles per SIMD."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_issue_rates.json")))
    if not files:
        return {}
    with open(files[-1]) as f:
        w = json.load(f)["cycles_per_wave_instruction_at_4_waves_per_simd"]
    return {"file": "profiles/" + os.path.basename(files[-1]),
            "cycles_per_simd": {"f64_arith": round((w["v_fma_f64"] + w["v_add_f64"] + w["v_mul_f64"]) / 12, 3),
                                "f64_trans": round((w["v_rcp_f64"] + w["v_rsq_f64"]) / 8, 3),
                                "valu_32bit": round((w["v_mov_b32"] + w["v_fma_f32"]) / 8, 3),
                                "salu": round(w["s_and_b64"] / 4, 3)}}


def committed_counts():
    """Newest profiles/rNN_counts.json: per workload, the rocprofv3 counter sums of ONE render."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counts.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def cpu_baseline(pkg, scene, name, wl):
    """Reference CPU path on this machine's host cores, bounded sample."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_time")
    spp = CPU_SAMPLE_SPP.get(name) or 64
    W = wl["W"]
    if name == "c5_shard":
        W, spp = 1024, 64  # same scene and integrator as the headline; the CPU rate does not depend on the image size
    if cores < 128:
        spp = max(4, spp * cores // 128)
    if os.path.exists(ref):
        try:
            t0 = time.time()
            out = subprocess.run([ref, "time", str(wl["scene"]), str(wl["integ"]), str(W), str(spp)], check=True,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900).stdout.decode()
            info = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
            return {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": info["threads"],
                    "kind": "reference",
                    "sample": "unmodified reference Renderer::render, scene%02d integrator%d %dx%d spp=%d "
                              "(%.1f s in render, %.1f s wall, own thread-hash RNG)" %
                              (wl["scene"], wl["integ"], W, info["height"], spp, info["seconds"], time.time() - t0)}
        except Exception:
            pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _golden as G  # oracle binding (test infrastructure), used here only as the CPU baseline
    H = wl["H"] * W // wl["W"]
    spp = max(1, spp // 4)
    p = pkg.make_params(W, H, spp, integrator=wl["integ"], seed=1)
    t0 = time.time()
    _, st = G.oracle_render(scene, p, threads=cores)
    sec = time.time() - t0
    return {"value": round(st["samples"] / sec * 1e-6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle port, tile-threaded, scene%02d integrator%d %dx%d spp=%d (%.1f s)" %
                      (wl["scene"], wl["integ"], W, H, spp, sec)}


def crop_parity(pkg, ctx, scene, wl, chunks_used, pipeline, timed_image=None, timed_mask=None):
    """64x64 crop of the full-size image at the FULL spp against the CPU oracle on the same seeds.  `timed_image`: the
    framebuffer the TIMED steps wrote (gathered on rank 0): its crop is compared too, so the check covers the very
    summation that was timed (its chunking differs from a crop render's: other tile count)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _golden as G  # the checker (test infrastructure)
    A = pkg._abi
    W, H, spp = wl["W"], wl["H"], wl["spp"]
    x0, y0 = (W // 2 - 32) // 16 * 16, (H // 2 - 32) // 16 * 16
    region = (x0, y0, x0 + 64, y0 + 64)
    t0 = time.time()

    def params(chunks):
        return A.make_params(W, H, spp, integrator=wl["integ"], seed=1, region=region, pipeline=pipeline,
                             spp_chunks=chunks)
    ref, _ = G.oracle_render(scene, params(1), threads=os.cpu_count() or 1)
    timed = ctx.render(params(0))             # spp_chunks = 0: the library's own partial sums, as in the timed render
    crop_chunks = ctx.stats()["spp_chunks"]
    seq = ctx.render(params(1))               # one running sum per pixel, like renderer.h:72-79
    out = {"crop": list(region), "spp": spp, "samples": 64 * 64 * spp,
           "rel_l2_vs_oracle": G.rel_l2(timed, ref), "spp_chunks": crop_chunks, "spp_chunks_of_the_timed_render": chunks_used,
           "rel_l2_chunks1_vs_oracle": G.rel_l2(seq, ref), "bit_exact_chunks1": bool(np.array_equal(seq, ref)),
           "rel_l2_chunked_vs_chunks1": G.rel_l2(timed, seq),
           "rmse_gamma_vs_oracle": float(np.sqrt(np.mean((np.clip(np.sqrt(timed), 0, 1) -
                                                          np.clip(np.sqrt(ref), 0, 1)) ** 2))),
           "seconds": round(time.time() - t0, 2)}
    if timed_image is not None:
        crop = timed_image[y0:y0 + 64, x0:x0 + 64]
        m = np.ones((64, 64), dtype=bool) if timed_mask is None else timed_mask[y0:y0 + 64, x0:x0 + 64]
        out["timed_framebuffer_pixels_compared"] = int(m.sum())
        out["rel_l2_timed_framebuffer_vs_oracle"] = G.rel_l2(crop[m], ref[m]) if m.any() else None
    no_libm = wl["scene"] in (7, 21)  # + - * / sqrt only on these paths: the device must match bit for bit
    out["bar"] = "bit-exact (chunks=1), rel-L2 <= 1e-13 chunked" if no_libm else "rel-L2 <= 1e-3"
    out["ok"] = bool(out["bit_exact_chunks1"] and out["rel_l2_vs_oracle"] <= 1e-13) if no_libm \
        else bool(out["rel_l2_vs_oracle"] <= 1e-3 and out["rel_l2_chunks1_vs_oracle"] <= 1e-3)
    timed = out.get("rel_l2_timed_framebuffer_vs_oracle")
    if timed is not None:
        out["ok"] = bool(out["ok"] and timed <= (1e-13 if no_libm else 1e-3))
    for k in ("rel_l2_vs_oracle", "rel_l2_chunks1_vs_oracle", "rel_l2_chunked_vs_chunks1", "rmse_gamma_vs_oracle",
              "rel_l2_timed_framebuffer_vs_oracle"):
        if out.get(k) is not None:
            out[k] = float("%.3e" % out[k])
    return out


def roofline(name, pipe_name, counts, samples, closest, shadow, kernel_ms, n_gpus):
    """Roofline object of one workload (per GPU over its kernels)."""
    t = kernel_ms * 1e-3
    alg_bytes = samples * B_SAMPLE + closest * B_CLOSEST + shadow * B_SHADOW
    hbm_eq = alg_bytes / n_gpus / t * 1e-9
    c = (counts.get(name) or {}).get(pipe_name) or {}
    per = samples / float(c["samples"]) if c.get("samples") else None
    traffic = None
    if per is not None and c.get("fetch_bytes") is not None and c.get("write_bytes") is not None:
        traffic = (c["fetch_bytes"] + c["write_bytes"]) * per / n_gpus
    eq = {"achieved": round(hbm_eq, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_eq / HBM_PEAK_GBS, 5),
          "algorithmic_bytes_per_sample": round(alg_bytes / samples, 1)}
    if pipe_name == "wavefront" or per is None or not c.get("insts_valu"):
        r = dict(eq)
        r["bound"] = "hbm"
        r["traffic"] = traffic
        r["kernel_ms"] = round(kernel_ms, 3)
        if traffic is not None:
            r["hbm_actual_GBs"] = round(traffic / t * 1e-9, 2)
            r["traffic_over_algorithmic"] = round(traffic / (alg_bytes / n_gpus), 3)
        if pipe_name != "wavefront":
            r["note"] = "no committed VALU count for this kernel: algorithmic-bytes figure only (the megakernel is not HBM-bound)"
        return r
    insts = c["insts_valu"] * per / n_gpus          # wave-level VALU instructions of this GPU's share
    lane_util = c["thread_cycles_valu"] / c["insts_valu"] / 64.0
    f64 = c.get("f64") or {}
    # the roofline proper: FP64 flop the class counters saw (add / mul / transcendental 1, fma 2 per active lane)
    flop = (c.get("f64_flop") or 0.0) * per / n_gpus * 64 * lane_util
    tflops = flop / t * 1e-12
    r = {"bound": "fp64_valu", "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": round(tflops / FP64_PEAK_TFLOPS, 5), "traffic": traffic, "kernel_ms": round(kernel_ms, 3),
         "achieved_is": "counted FP64 flop (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes x lane utilisation) / kernel time",
         "valu_insts_per_sample": round(c["insts_valu"] / c["samples"], 1), "lane_utilisation": round(lane_util, 4),
         "hbm_equivalent": eq}
    if traffic is not None:
        r["hbm_actual_GBs"] = round(traffic / t * 1e-9, 2)
    # how full the SIMDs' instruction issue was: each class at its measured cost (cycles per SIMD and wave-instruction)
    costs = ISSUE_COSTS.get("cycles_per_simd") or {}
    if f64 and costs:
        arith = (f64["add"] + f64["mul"] + f64["fma"]) * per / n_gpus
        trans = f64["trans"] * per / n_gpus
        other = max(insts - arith - trans, 0.0)            # compares, selects, moves, integer, conversions
        salu = (f64.get("salu", 0.0) + f64.get("smem", 0.0)) * per / n_gpus
        clock = c.get("clock_ghz") or 2.4
        avail = 1024 * clock * 1e9 * t                       # SIMD cycles of the kernel
        need_lo = arith * costs["f64_arith"] + trans * costs["f64_trans"] + other * costs["valu_32bit"] + salu * costs["salu"]
        need_hi = need_lo + other * (costs["f64_arith"] - costs["valu_32bit"])  # every "other" VALU an FP64 compare
        r["issue"] = {"valu_f64_arith": round(arith), "valu_f64_transcendental": round(trans), "valu_other": round(other),
                      "scalar": round(salu), "costs_cycles_per_simd": costs, "costs_file": ISSUE_COSTS.get("file"),
                      "clock_ghz": clock, "clock_is": "GRBM_GUI_ACTIVE / 8 / kernel time of the counter pass" if c.get("clock_ghz") else "assumed",
                      "utilisation": [round(need_lo / avail, 4), round(need_hi / avail, 4)],
                      "utilisation_is": "SIMD cycles the counted instructions need at their measured issue costs / SIMDs x clock "
                                        "x time; [other VALU priced as 32-bit moves, as FP64 compares]"}
    r["counts_source_hash"] = c.get("source_hash") or counts.get("_source_hash")
    r["library_source_hash"] = LIB_HASH
    r["stale"] = bool(r["counts_source_hash"] != LIB_HASH)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS), help="the line's main workload")
    ap.add_argument("--extras", default=None,
                    help="comma list of further workloads timed in the same run ('none'; default: every other "
                         "BASELINE config when the main workload is the headline and N = 1)")
    ap.add_argument("--extra-steps", type=int, default=2)
    ap.add_argument("--extra-warmup", type=int, default=1)
    ap.add_argument("--pipeline", default="auto", choices=["auto", "mega", "wavefront"])
    ap.add_argument("--sorted-shading", action="store_true",
                    help="experiment: RTR_FLAG_SORTED_SHADING on the timed renders (invalidates the headline)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-spp crop check against the oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (gloo + RTR_BENCH_ONE_GPU=1 rehearses the N>1 path on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("ray_tracing-rendering_amd")
    A = pkg._abi
    R = pkg.renderer
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if os.environ.get("RTR_BENCH_ONE_GPU"):
        local_rank = 0  # rehearsal: every rank drives GPU 0
    torch.cuda.set_device(local_rank)
    gloo = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            # RCCL carries the barrier and the max-over-ranks reduction; the framebuffer gather is host-side (gloo)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            gloo = dist.new_group(backend="gloo")
        else:
            dist.init_process_group("gloo")
            gloo = dist.group.WORLD

    pipeline = {"auto": A.PIPELINE_AUTO, "mega": A.PIPELINE_MEGAKERNEL, "wavefront": A.PIPELINE_WAVEFRONT}[args.pipeline]
    counts, counts_file = committed_counts()
    global ISSUE_COSTS, LIB_HASH
    ISSUE_COSTS = load_issue_costs()
    LIB_HASH = pkg.build.source_hash()
    ctx = pkg.Context(local_rank)
    stream = torch.cuda.Stream()  # a real (non-null) hipStream_t the library launches on; events use it too
    ctx.set_stream(stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(name, steps, warmup, with_cpu, with_parity):
        """Time one workload; returns its (sub-)line on rank 0."""
        wl = dict(WORKLOADS[name])
        if args.spp:
            wl["spp"] = args.spp
        W, H, spp = wl["W"], wl["H"], wl["spp"]
        stride = wl.get("stride", 1) * world
        first = rank  # with a workload-level stride this process plays rank `rank` of `stride`
        scene = load_scene(pkg, wl["scene"])
        ctx.upload(scene)
        fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
        params = A.make_params(W, H, spp, integrator=wl["integ"], seed=1, pipeline=pipeline, spp_chunks=0,
                               tile_first=first, tile_stride=stride,
                               flags=A.FLAG_SORTED_SHADING if args.sorted_shading else 0)

        def step():
            ctx.render_into(params, fb.data_ptr(), W, blocking=False)

        for _ in range(warmup):
            step()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            step()
            b.record(stream)
        barrier()
        sec = time.perf_counter() - t0
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
        st = ctx.stats()  # last step of this rank
        t = torch.tensor([sec, sum(kernel_ms) / len(kernel_ms), float(st["samples"]), float(st["closest_segments"]),
                          float(st["shadow_segments"])], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        if world > 1:
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            sec_max, kms_max = float(tmax[0]), float(tmax[1])
            samples, closest, shadow = float(tsum[2]), float(tsum[3]), float(tsum[4])
        else:
            sec_max, kms_max = float(t[0]), float(t[1])
            samples, closest, shadow = float(t[2]), float(t[3]), float(t[4])

        # host-side framebuffer gather (outside the timed region): every rank sends only the tiles it owns
        g0 = time.perf_counter()
        image, covered = R.gather_tiles(fb, W, H, rank, world, group=gloo, tile_first=first, tile_stride=stride)
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank != 0:
            return None

        # every tile of ranks 0..world-1 of `stride` exactly once, nothing else
        want = np.zeros_like(covered)
        for r in range(world):
            for tile in R.tiles_of_rank(W, H, r, stride):
                x0, y0, _, _ = R.tile_rect(W, H, tile)
                want[y0 // 16, x0 // 16] += 1
        assert np.array_equal(covered, want), "the gathered tiles are not the tiles the ranks own"
        px_mask = np.repeat(np.repeat(want, 16, axis=0), 16, axis=1)[:H, :W].astype(bool)
        total = float(px_mask.sum()) * spp
        assert abs(samples - total) < 0.5, "ranks rendered %d of %d samples" % (samples, total)
        mean = float(image[px_mask].mean())
        assert mean > 0 and bool(np.isfinite(image).all()), "framebuffer is empty or not finite"
        value = total * steps / sec_max * 1e-6
        pipe_name = {1: "megakernel", 2: "wavefront"}.get(st["pipeline"], "?")
        headline = name == HEADLINE and not args.spp and not args.sorted_shading
        share = " (tiles index %% %d == 0: one rank's share of C5)" % stride if wl.get("stride", 1) > 1 else ""
        line = {
            "metric": "Msamples/sec, Cornell Box 800x800 spp=400 MIS" if headline else "Msamples/sec, " + name,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(sec_max / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "scene%02d %dx%d spp=%d depth=50 integrator%d (%s)%s" %
                                   (wl["scene"], W, H, spp, wl["integ"], name, share),
                       "pipeline": pipe_name, "spp_chunks": st["spp_chunks"],
                       "parallelism": "tiles%%%d" % stride, "seed": 1, "flags_in_effect": st["flags_in_effect"]},
            "roofline": roofline(name, pipe_name, {} if (args.spp or st["flags_in_effect"]) else counts, samples, closest, shadow, kms_max, world),
            "segments_per_sample": {"closest": round(closest / samples, 4), "shadow": round(shadow / samples, 4)},
            "gather_ms": round(gather_ms, 2), "gather_bytes_per_rank": R.gather_tiles.bytes_sent,
            "image_mean": round(mean, 6),
        }
        if with_parity:
            line["parity"] = crop_parity(pkg, ctx, scene, wl, st["spp_chunks"], pipeline, image, px_mask)
        if with_cpu:
            line["cpu_baseline"] = cpu_baseline(pkg, scene, name, wl)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 2)
        return line

    single = world == 1
    extras = []
    if args.extras is None:
        extras = EXTRAS if (single and args.workload == HEADLINE and not args.spp) else []
    elif args.extras != "none":
        extras = [e for e in args.extras.split(",") if e]
    line = run(args.workload, args.steps, args.warmup, single and not args.no_cpu_baseline,
               single and not args.no_parity)
    if rank == 0 and counts_file:
        line["counts_file"] = "profiles/" + counts_file
    subs = []
    for e in extras:
        sub = run(e, args.extra_steps, args.extra_warmup, single and not args.no_cpu_baseline, single and not args.no_parity)
        if sub is not None:
            subs.append(sub)
    if rank == 0:
        if subs:
            line["extra"] = subs
        checks = [line.get("parity")] + [s.get("parity") for s in subs]
        if any(c is not None for c in checks):
            line["parity_ok"] = all(c["ok"] for c in checks if c is not None)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and line.get("parity_ok") is False:
        raise SystemExit("bench.py: a full-spp crop differs from the oracle beyond its bar (see `parity`)")


if __name__ == "__main__":
    main()
