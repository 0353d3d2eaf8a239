#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the integrator loop on BASELINE.json's configuration C2
(Cornell box 800x800, spp 400, depth 50, MIS integrator 4) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full render of the workload.  Image tiles (16x16, the reference's own work unit,
renderer/renderer.h:40-62) are dealt round-robin to the ranks; there is no data-path collective:
each rank writes its tiles into its own device framebuffer and the host gathers them after the
timed region.  Total work is fixed as N grows ("strong" scaling on the named config).

Rank 0 prints ONE JSON line.  `value` is whole-job Msamples/s = W*H*spp*K / max-over-ranks seconds,
with scene and framebuffer resident in HBM.  `roofline` prices the algorithmic state traffic of a
wavefront path tracer (172 B/sample + 280 B/closest-hit segment + 168 B/shadow
segment, FP64 state) against the 8 TB/s HBM peak, with the device time measured by HIP events on
the launch stream.  `cpu_baseline` times the reference's own tile-threaded renderer
(oracle/_ref/ref_time, kind "reference") or, where that binary is absent, the CPU oracle port
(kind "port") on the host cores of the same machine, on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
B_SAMPLE, B_CLOSEST, B_SHADOW = 172, 280, 168  # SURVEY 8(d), FP64 state

WORKLOADS = {
    # name: (scene id, integrator, W, H, spp)
    "cornell_mis": (21, 4, 800, 800, 400),      # C2 headline: Cornell box with NEE/MIS
    "cornell_literal": (7, 4, 800, 800, 400),   # C2 literal: scene07 + integrator4 (near-black)
    "cornell_rr": (7, 1, 800, 800, 400),        # lit scene07 (README numbers)
    "final_rr": (9, 1, 800, 800, 500),          # C3
    "final_mis": (22, 4, 800, 800, 500),        # C3 NEE twin
    "mis_spheres": (23, 4, 1920, 1080, 1024),   # C4
}


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


def cpu_baseline(pkg, scene, scene_id, integ, W, H):
    """Reference CPU path on this machine's host cores, bounded sample (about 10-30 s)."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_time")
    spp = 256 if cores >= 64 else (64 if cores >= 16 else 32)
    if os.path.exists(ref) and W == H:
        try:
            t0 = time.time()
            out = subprocess.run([ref, "time", str(scene_id), str(integ), str(W), str(spp)], check=True,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600).stdout.decode()
            info = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
            return {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": info["threads"],
                    "kind": "reference",
                    "sample": "unmodified reference Renderer::render, scene%02d integrator%d %dx%d spp=%d "
                              "(%.1f s, own thread-hash RNG)" % (scene_id, integ, W, info["height"], spp,
                                                                 time.time() - t0)}
        except Exception:
            pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _golden as G  # oracle binding (test infrastructure), used here only as the CPU baseline
    p = pkg.make_params(W, H, spp, integrator=integ, seed=1)
    t0 = time.time()
    _, st = G.oracle_render(scene, p, threads=cores)
    sec = time.time() - t0
    return {"value": round(st["samples"] / sec * 1e-6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle port, tile-threaded, scene%02d integrator%d %dx%d spp=%d (%.1f s)" %
                      (scene_id, integ, W, H, spp, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_mis", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="auto", choices=["auto", "mega", "wavefront"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (gloo + RTR_BENCH_ONE_GPU=1 rehearses the N>1 path on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("ray_tracing-rendering_amd")
    A = pkg._abi
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            gloo = dist.new_group(backend="gloo")
        else:
            dist.init_process_group("gloo")
            gloo = dist.group.WORLD

    scene_id, integ, W, H, spp = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    scene = load_scene(pkg, scene_id)
    pipeline = {"auto": A.PIPELINE_AUTO, "mega": A.PIPELINE_MEGAKERNEL, "wavefront": A.PIPELINE_WAVEFRONT}[args.pipeline]

    ctx = pkg.Context(local_rank)
    stream = torch.cuda.Stream()  # a real (non-null) hipStream_t the library launches on; events use it too
    ctx.set_stream(stream.cuda_stream)
    ctx.upload(scene)
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    params = A.make_params(W, H, spp, integrator=integ, seed=1, pipeline=pipeline, spp_chunks=0, tile_first=rank,
                           tile_stride=world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.render_into(params, fb.data_ptr(), W, blocking=False)

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
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

    # host-side framebuffer gather (outside the timed region), then a coverage check
    g0 = time.perf_counter()
    host = fb.cpu()
    if world > 1:
        parts = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, parts, dst=0, group=gloo)
        if rank == 0:
            host = torch.stack(parts).sum(0)  # ranks own disjoint tiles; the rest of each buffer is zero
    gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        total = float(W) * H * spp
        assert abs(samples - total) < 0.5, "ranks rendered %d of %d samples" % (samples, total)
        mean = float(host.mean())
        assert mean > 0 and bool(torch.isfinite(host).all()), "framebuffer is empty or not finite"
        value = total * args.steps / sec_max * 1e-6
        alg_bytes = samples * B_SAMPLE + closest * B_CLOSEST + shadow * B_SHADOW  # whole job, one step
        achieved = alg_bytes / world / (kms_max * 1e-3) * 1e-9  # GB/s per GPU over its kernels
        pipe_name = {1: "megakernel", 2: "wavefront"}.get(st["pipeline"], "?")
        traffic = None  # HBM bytes of one render from separate rocprofv3 --pmc passes (profiles/r01_traffic.json)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)
            if not args.spp and world == 1:
                traffic = tj.get(args.workload, {}).get(pipe_name)
        except (OSError, ValueError):
            pass
        line = {
            "metric": "Msamples/sec, Cornell Box 800x800 spp=400 MIS" if args.workload == "cornell_mis" and not args.spp
                      else "Msamples/sec, " + args.workload,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(sec_max / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "scene%02d %dx%d spp=%d depth=50 integrator%d (%s)" %
                                   (scene_id, W, H, spp, integ, args.workload),
                       "pipeline": pipe_name,
                       "parallelism": "tiles%%%d" % world, "seed": 1},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": round(kms_max, 3),
                         "algorithmic_bytes_per_sample": round(alg_bytes / samples, 1)},
            "segments_per_sample": {"closest": round(closest / samples, 4), "shadow": round(shadow / samples, 4)},
            "gather_ms": round(gather_ms, 2), "image_mean": round(mean, 6),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, scene, scene_id, integ, W, H)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 2)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
