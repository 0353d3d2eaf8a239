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

`roofline`: the megakernel keeps path state in registers and LDS, so its binding limit is FP64
vector issue, not HBM (profiles/: 0.06 % of the algorithmic bytes reach memory).  `bound` is
"fp64_valu": achieved = (wave-level VALU instructions of the render, from the committed rocprofv3
SQ pass in profiles/*_counts.json, scaled to this run's sample count) x 64 lanes x lane utilisation
x 2 flop / HIP-event kernel time, against the 78.6 TFLOP/s FP64 vector peak (= 16 lanes/clk on
1024 SIMDs at 2.4 GHz, FMA = 2 flop: every instruction is priced as an FP64 FMA slot).  The figure
SURVEY 8(d) prescribes -- algorithmic state bytes of a wavefront tracer (172 B/sample + 280
B/closest segment + 168 B/shadow segment) over the same time against 8 TB/s -- is kept as
`hbm_equivalent`; it is the binding roofline only when the wavefront pipeline ran, and then
`bound` says "hbm".  `traffic` = HBM bytes of one render from separate FETCH_SIZE / WRITE_SIZE
passes (same file).
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


def crop_parity(pkg, ctx, scene, wl, chunks_used, pipeline):
    """64x64 crop of the full-size image at the FULL spp against the CPU oracle on the same seeds."""
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
    no_libm = wl["scene"] in (7, 21)  # + - * / sqrt only on these paths: the device must match bit for bit
    out["bar"] = "bit-exact (chunks=1), rel-L2 <= 1e-13 chunked" if no_libm else "rel-L2 <= 1e-3"
    out["ok"] = bool(out["bit_exact_chunks1"] and out["rel_l2_vs_oracle"] <= 1e-13) if no_libm \
        else bool(out["rel_l2_vs_oracle"] <= 1e-3 and out["rel_l2_chunks1_vs_oracle"] <= 1e-3)
    for k in ("rel_l2_vs_oracle", "rel_l2_chunks1_vs_oracle", "rel_l2_chunked_vs_chunks1", "rmse_gamma_vs_oracle"):
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
    tflops = insts * 64 * lane_util * 2 / t * 1e-12  # every active lane-slot priced as one FP64 FMA
    r = {"bound": "fp64_valu", "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": round(tflops / FP64_PEAK_TFLOPS, 5), "traffic": traffic, "kernel_ms": round(kernel_ms, 3),
         "valu_insts_per_sample": round(c["insts_valu"] / c["samples"], 1), "lane_utilisation": round(lane_util, 4),
         # share of the chip's VALU issue slots (4 clk per wave64 FP64-rate instruction) that held an instruction
         "issue_frac": round(insts * 4 / (1024 * 2.4e9 * t), 4),
         "hbm_equivalent": eq}
    if traffic is not None:
        r["hbm_actual_GBs"] = round(traffic / t * 1e-9, 2)
    if c.get("f64_flop") is not None:  # class counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64), when the box has them
        r["fp64_flop_counted_TFLOPs"] = round(c["f64_flop"] * per / n_gpus * 64 * lane_util / t * 1e-12, 3)
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
                               tile_first=first, tile_stride=stride)

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
        headline = name == HEADLINE and not args.spp
        share = " (tiles index %% %d == 0: one rank's share of C5)" % stride if wl.get("stride", 1) > 1 else ""
        line = {
            "metric": "Msamples/sec, Cornell Box 800x800 spp=400 MIS" if headline else "Msamples/sec, " + name,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(sec_max / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "scene%02d %dx%d spp=%d depth=50 integrator%d (%s)%s" %
                                   (wl["scene"], W, H, spp, wl["integ"], name, share),
                       "pipeline": pipe_name, "spp_chunks": st["spp_chunks"],
                       "parallelism": "tiles%%%d" % stride, "seed": 1},
            "roofline": roofline(name, pipe_name, {} if args.spp else counts, samples, closest, shadow, kms_max, world),
            "segments_per_sample": {"closest": round(closest / samples, 4), "shadow": round(shadow / samples, 4)},
            "gather_ms": round(gather_ms, 2), "gather_bytes_per_rank": R.gather_tiles.bytes_sent,
            "image_mean": round(mean, 6),
        }
        if with_parity:
            line["parity"] = crop_parity(pkg, ctx, scene, wl, st["spp_chunks"], pipeline)
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
