#!/usr/bin/env python3
"""bench.py -- Msamples/s of the MI355X path-tracing integrator on BASELINE.json's headline workload.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full pass of the hot path over the workload: one frame of the synthetic Cornell-box-shaped scene
(BASELINE.json configs[1]: 512x512, 1024 spp, Lambertian-only) rendered through the C ABI (jp_render), scene
already resident in HBM, film download included.  At N > 1 the row bands of the film
(the reference's FRenderTask unit, integrator.cc:53: 20 rows; here the largest height <= 20 that deals evenly, 16 rows
for 512 rows on 2/4/8 ranks) are dealt round-robin to the ranks, each rank renders its bands into a device film that
is zero elsewhere, and ONE RCCL reduce(sum) over xGMI assembles the film on rank 0.
Weak scaling: the sample count grows with N (spp = 1024 * N), so every GPU traces the same number of paths as in
the 1-GPU run.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline     : dominant kernel class, algorithmic bytes per launch / HIP-event launch duration (events on the kernel
                 stream, taken on the last step of the timed region) vs 8 TB/s HBM
  cpu_baseline : the oracle restatement of the reference CPU path timed on this box's host cores on a bounded
                 sample (whole 20-row bands at the full spp), which is also the parity sample (l2_vs_cpu_ref)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# algorithmic bytes per unit, SURVEY.md section 8d records attributed to the kernel that moves them (DESIGN.md "Roofline")
B_EXTEND_PER_RAY = 32 + 8                     # ray read + hit record write
B_SHADE_PER_PATH_IN = 8 + 40                  # hit record read + path state read
B_SHADE_PER_SURVIVOR = 32 + 40                # next ray write + path state write
B_SHADE_PER_SHADOW_RAY = 48                   # shadow ray + contribution + pixel write
B_SHADOW_PER_RAY = 48                         # the same record read back
B_PER_SEGMENT, B_PER_SHADOW, B_FILM_PER_PIXEL = 160, 96, 12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU share (total spp = spp * gpus)")
    ap.add_argument("--full-materials", action="store_true", help="configs[2]: metal tall box instead of Lambertian-only")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity sample")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra single-lane step behind roofline.exclusive (profiling runs: keeps rocprofv3's per-kernel averages to the timed configuration)")
    ap.add_argument("--band-rows", type=int, default=0, help="band height for sharding / lanes (0: largest height <= 20 that deals evenly)")
    ap.add_argument("--cpu-bands", type=int, default=2, help="20-row bands rendered by the CPU oracle at the full spp (parity sample)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import jet_pbrt_amd as jp
    from jet_pbrt_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (n, n))
        n = world
    dist = None
    backend = os.environ.get("JETPBRT_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path on a 1-GPU box (ranks share the card)
    ndev = torch.cuda.device_count()
    dev = local_rank % max(1, ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, H = args.width, args.height
    spp_total = args.spp * n
    be = scenes.build_cornell(scenes.HostBackend("bench"), W, H, lambert_only=not args.full_materials)
    scene = be.flatten()
    ctx = jp.Context(dev)
    ctx.upload(scene)
    # bands that deal evenly over the ranks (16 rows for 512 rows, N <= 32); inside a rank the library splits the shard's rows
    # over its stream lanes by itself
    band_rows = args.band_rows if args.band_rows > 0 else jp.distributed.balanced_band_rows(H, n)
    params = jp.render_params(W, H, spp_total, 5, 1234, band_rows=band_rows, shard_index=rank, shard_count=n)
    film_dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if world > 1 else None

    def step():
        if world == 1:
            return ctx.render(params)                      # jp_render: kernels + film download, blocking
        ctx.render_device(params, film_dev.data_ptr(), sync=True)
        if backend == "nccl":
            dist.reduce(film_dev, dst=0, op=dist.ReduceOp.SUM)   # RCCL over xGMI onto rank 0's film
            return film_dev.cpu().numpy() if rank == 0 else None
        host = film_dev.cpu()                              # rehearsal backend: the same reduce on host tensors
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        return host.numpy() if rank == 0 else None

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    film = None
    for i in range(args.steps):
        if i == args.steps - 1:
            ctx.set_profiling(True)                        # per-launch HIP events (kernel stream) on the last timed step: roofline below
        film = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    samples_per_step = W * H * spp_total
    value = samples_per_step * args.steps / dt / 1e6

    # ---- roofline of the dominant kernel class: per-launch HIP events on the kernel stream, last timed step ----
    c = ctx.counters()                                     # counters + per-class event times of the last timed step
    lanes = int(ctx.build_info().lanes_last_render)
    lanes_note = ", %d stream lanes per GPU" % lanes
    # With several lanes a launch shares the GPU with the other lanes' kernels, so its duration (and the per-launch roofline
    # figure the contract asks for) reflects that sharing.  One extra, untimed step on ONE lane gives the same kernels' figures
    # when each has the GPU to itself -- reported next to the contract figure as roofline.exclusive.
    c1 = None
    if lanes > 1 and "JETPBRT_LANES" not in os.environ and not args.no_exclusive:
        os.environ["JETPBRT_LANES"] = "1"
        try:
            if world == 1:
                ctx.render(params)
            else:
                ctx.render_device(params, film_dev.data_ptr(), sync=True)
            c1 = ctx.counters()
        finally:
            del os.environ["JETPBRT_LANES"]
    ctx.set_profiling(False)
    roof = None
    if rank == 0:
        survivors = max(0, c.closest_rays - c.samples)     # rays written by k_shade (every ray but the camera rays)
        # (ms, launches, algorithmic bytes = SURVEY.md section 8d per-unit figure x units the class processes,
        #  bytes attributed to the kernel that actually moves each record -- DESIGN.md section 6)
        cls = {
            "k_extend": (c.extend_ms, c.extend_launches, B_PER_SEGMENT * c.closest_rays, B_EXTEND_PER_RAY * c.closest_rays),
            "k_shade": (c.shade_ms, c.shade_launches, B_PER_SEGMENT * c.closest_rays,
                        B_SHADE_PER_PATH_IN * c.closest_rays + B_SHADE_PER_SURVIVOR * survivors + B_SHADE_PER_SHADOW_RAY * c.shadow_rays),
            "k_shadow": (c.shadow_ms, c.shadow_launches, B_PER_SHADOW * c.shadow_rays, B_SHADOW_PER_RAY * c.shadow_rays),
        }
        dom = max(cls, key=lambda k: cls[k][0])
        ms, launches, nbytes, attributed = cls[dom]
        launches = max(1, launches)
        achieved = (nbytes / launches) / (ms / launches * 1e-3) / 1e9 if ms > 0 else 0.0
        bytes_per_sample = (B_PER_SEGMENT * c.closest_rays + B_PER_SHADOW * c.shadow_rays) / max(1, c.samples) + B_FILM_PER_PIXEL / spp_total
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel") == dom and tj.get("workload") == ("cornell_full" if args.full_materials else "cornell_lambert") and n == 1:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launch_ms_avg": round(ms / launches, 4), "launches": int(launches), "algorithmic_bytes_per_launch": int(nbytes / launches),
                "unit_bytes": B_PER_SHADOW if dom == "k_shadow" else B_PER_SEGMENT, "units_per_launch": int((c.shadow_rays if dom == "k_shadow" else c.closest_rays) / launches),
                "attributed_bytes_per_launch": int(attributed / launches),
                "class_ms": {k: round(v[0], 3) for k, v in cls.items()},
                # two stream lanes: the lanes' kernels overlap, so a launch's duration includes the time it shares the
                # GPU with the other lane's kernel; kernel_time_over_wall is the average number of kernels in flight
                "lanes": lanes, "kernel_time_over_wall": round((c.extend_ms + c.shade_ms + c.shadow_ms + c.other_ms) / max(1e-9, c.render_ms), 3),
                "exclusive": None,
                "whole_path": {"bytes_per_sample": round(bytes_per_sample, 1), "segments_per_sample": round(c.closest_rays / max(1, c.samples), 3),
                               "shadow_rays_per_sample": round(c.shadow_rays / max(1, c.samples), 3),
                               "achieved_GBps": round(value * 1e6 * bytes_per_sample / 1e9, 1),
                               "frac": round(value * 1e6 * bytes_per_sample / 1e9 / (HBM_PEAK_GBS * n), 4)}}

    if rank == 0 and roof is not None and c1 is not None:
        ms1 = {"k_extend": c1.extend_ms, "k_shade": c1.shade_ms, "k_shadow": c1.shadow_ms}
        n1 = {"k_extend": c1.extend_launches, "k_shade": c1.shade_launches, "k_shadow": c1.shadow_launches}
        units1 = {"k_extend": B_PER_SEGMENT * c1.closest_rays, "k_shade": B_PER_SEGMENT * c1.closest_rays, "k_shadow": B_PER_SHADOW * c1.shadow_rays}
        dom1 = max(ms1, key=lambda k: ms1[k])
        excl = {}
        for k in (roof["kernel"], dom1):
            a1 = units1[k] / max(1e-9, ms1[k] * 1e-3) / 1e9
            excl[k] = {"launch_ms_avg": round(ms1[k] / max(1, n1[k]), 4), "launches": int(n1[k]), "achieved": round(a1, 1), "frac": round(a1 / HBM_PEAK_GBS, 4)}
        roof["exclusive"] = {"lanes": 1, "dominant": dom1, "kernels": excl}

    # ---- parity sample + CPU baseline (rank 0, N = 1 only); oracle/ is the checker here, never the thing measured ----
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu and (n == 1 or os.environ.get("JETPBRT_BENCH_PARITY_ALL")):
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import harness as Hn
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = min(avail, 16)                           # the reference renders with 16 threads (main.cc:156)
        nbands = (H + 19) // 20
        # (a) parity at the FULL spp on whole 20-row bands (bands are independent under the counter sampler)
        k = max(1, nbands // max(1, args.cpu_bands))
        idx = (nbands // 2) % k
        p = jp.render_params(W, H, spp_total, 5, 1234, band_rows=20, shard_index=idx, shard_count=k)
        ref, _ = Hn.oracle_render(scene, p, threads)
        rows = np.zeros(H, bool)
        for y0, y1 in jp.distributed.bands_of(H, idx, k):
            rows[y0:y1] = True
        d = np.sqrt(((film[rows] - ref[rows]) ** 2).sum(-1))
        parity = {"mean_per_pixel_l2": float(d.mean()), "max_per_pixel_l2": float(d.max()), "pixels": int(d.size), "spp": spp_total,
                  "tolerance": 1e-4, "sample": "bands b %% %d == %d of %d" % (k, idx, nbands),
                  "bit_identical": bool(np.array_equal(film[rows].view(np.uint32), ref[rows].view(np.uint32))),
                  "libm_sincosf": int(ctx.build_info().libm_sincosf)}
        # (b) CPU baseline: the whole frame at a reduced spp (throughput does not depend on spp), 20-row tasks
        cpu_spp = 64
        pc = jp.render_params(W, H, cpu_spp, 5, 1234, sampler_mode=jp.JP_SAMPLER_STOCK_MT19937)
        t1 = time.perf_counter(); Hn.oracle_render(scene, pc, threads); t_port = time.perf_counter() - t1
        port_v = W * H * cpu_spp / t_port / 1e6
        cpu = {"value": round(port_v, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
               "sample": "whole %dx%d frame at %d spp = %d samples in %.1f s; oracle/pt_oracle.cc, stock mt19937_64 sampler, 20-row tasks on %d std::threads (of %d visible CPUs)" % (
                   W, H, cpu_spp, W * H * cpu_spp, t_port, threads, avail)}
        if Hn.have_ref():
            try:
                rb = scenes.build_cornell(Hn.RefBackend("bench"), W, H, lambert_only=not args.full_materials)
                t1 = time.perf_counter(); rb.render(W, H, cpu_spp, 5, 0, 1234, threads); t_ref = time.perf_counter() - t1
                cpu = {"value": round(W * H * cpu_spp / t_ref / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "reference",
                       "sample": "whole %dx%d frame at %d spp = %d samples in %.1f s; oracle/_ref (unmodified reference: DoRender per 20-row task on its FParallelSystem, FRandomSampler), %d threads (of %d visible CPUs)" % (
                           W, H, cpu_spp, W * H * cpu_spp, t_ref, threads, avail),
                       "port_value": round(port_v, 3)}
            except Exception as e:                         # the prebuilt reference library is optional on the GPU box
                cpu["reference_error"] = str(e)

    if rank == 0:
        out = {
            "metric": "Msamples/sec (whole node) + per-pixel L2 vs CPU ref, cornell_box 1024spp",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cornell_box %dx%d, %d spp (%d per GPU share), %s, maxDepth 5, counter sampler seed 1234" % (
                W, H, spp_total, args.spp, "full bsdf.cc + microfacet.cc materials" if args.full_materials else "Lambertian-only BSDF"),
                "parallelism": ("%d-row band shard x%d + RCCL film reduce" % (band_rows, n) if n > 1 else "single GPU") + lanes_note},
            "roofline": roof, "cpu_baseline": cpu, "l2_vs_cpu_ref": parity,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
