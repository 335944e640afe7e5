#!/usr/bin/env python3
"""bench.py -- Msamples/s of the MI355X path-tracing integrator on BASELINE.json's workloads.

  python bench.py [--gpus N] [--steps K] [--warmup W]                 (driver contract; N = 1 by default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus 8 --scene bunny --width 1920 --height 1080 --spp 4096 --scaling strong      (configs[4] as written)

A "step" is one full pass of the hot path over the workload: one frame rendered through the C ABI (jp_render), scene
already resident in HBM, film download included.  The HEADLINE workload is BASELINE.json
configs[2] -- the reference's own Cornell scene (main.cc:27-33: metal tall box) at 512x512, 1024 spp -- timed over EXACTLY
--steps frames after --warmup untimed ones.  At N = 1 the line also carries a `configs` object with one full sub-record per
single-GPU configuration of BASELINE.json (configs[1] Lambertian-only Cornell, configs[2] full materials, configs[3] the
bunny scene of main.cc:64-111 at 800x600, 2048 spp) and -- round 4 -- configs[4]'s frame (1920x1080, 4096 spp) rendered WHOLE on the
one GPU: the N = 1 figure an 8-GPU run of that frame divides by; each is timed over its own region of >= --min-seconds (default 10 s);
their figures are summarised in `config.sub` of the one stdout line, the full sub-records (roofline, cpu_baseline,
l2_vs_cpu_ref) are written to stderr as one `configs_detail {...}` line.

At N > 1 the row bands of the film (the reference's FRenderTask unit, integrator.cc:53: 20 rows; here the largest height
<= 20 that deals evenly) are dealt round-robin to the ranks, each rank renders its bands into a device film that is zero
elsewhere, and ONE RCCL gather of the ranks' packed rows over xGMI assembles the film on rank 0.  --scaling weak (default): the sample count
grows with N (spp = --spp * N), every GPU traces as many paths as in the 1-GPU run; --scaling strong: the frame is fixed.
An N > 1 run checks itself (`shard_check` in the line; the run FAILS otherwise): the process group has exactly --gpus ranks, every
rank reports the samples of its own bands (one all_gather of the per-rank counters) and they add up to the frame, and every rank's bands arrived
in the assembled film (no rank's rows are all zero).

Rank 0 prints one JSON line.  Besides the contract fields:
  roofline     : dominant kernel class, algorithmic bytes per launch / HIP-event launch duration (events on the kernel
                 stream, taken on the last step of the timed region) vs 8 TB/s HBM; `traffic` = rocprofv3 PMC bytes per unit
                 (profiles/traffic_r04.json, named once in `notes`) x this run's units per launch; `attributed` = the same launch by
                 the bytes that kernel class itself moves; `alone` = every class with the GPU to itself: contract / attributed HBM
                 fraction and the vector-instruction issue fraction (instructions per unit from the SQ counter passes under profiles/)
  config.sub   : one compact summary per single-GPU BASELINE.json config (value, ms, whole-path fraction, parity, CPU reference);
                 the full sub-records follow under `configs`
  cpu_baseline : the unmodified reference (oracle/_ref) -- or the oracle port where that library is absent -- timed on this
                 box's host cores on a bounded sample, at 16 threads (main.cc:156) and at min(cores, bands)
  l2_vs_cpu_ref: parity sample.  Cornell: whole 20-row bands at the full spp against the oracle.  Bunny: the whole film at the
                 full spp against the device's reference-tree mode (bit-identical to the oracle, checked in the same run on
                 an oracle band at reduced spp)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
VALU_CLOCK_GHZ = 2.2             # shader clock this workload sustains (profiles/r02g_clocks_power.txt: 2.17-2.24 GHz at 1.15-1.28 kW; 2.4 GHz idle)
VALU_PEAK_GINST = 256 * 4 * VALU_CLOCK_GHZ / 2.0    # wave64 instructions/ns: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles
# algorithmic bytes per unit, SURVEY.md section 8d records attributed to the kernel that moves them (DESIGN.md "Roofline")
B_EXTEND_PER_RAY = 32 + 8                     # ray read + hit record write
B_SHADE_PER_PATH_IN = 8 + 40                  # hit record read + path state read
B_SHADE_PER_SURVIVOR = 32 + 40                # next ray write + path state write
B_SHADE_PER_SHADOW_RAY = 48                   # shadow ray + contribution + pixel write
B_SHADOW_PER_RAY = 48                         # the same record read back
B_PER_SEGMENT, B_PER_SHADOW, B_FILM_PER_PIXEL = 160, 96, 12

# BASELINE.json configs -> (scene key, width, height, spp)
CONFIGS = {
    1: ("cornell_lambert", 512, 512, 1024),
    2: ("cornell", 512, 512, 1024),
    3: ("bunny", 800, 600, 2048),
    4: ("bunny", 1920, 1080, 4096),
}
SCENE_LABEL = {
    "cornell_lambert": "cornell_box, Lambertian-only BSDF",
    "cornell": "cornell_box, full bsdf.cc + microfacet.cc materials (main.cc:27-33)",
    "bunny": "bunny scene of main.cc:64-111 (4 x 69,938-triangle procedural stand-in + 2 rectangles = 279,754 primitives)",
}


def build_scene(scenes, backend, key, W, H):
    if key == "bunny":
        return scenes.build_bunny(backend, W, H)
    return scenes.build_cornell(backend, W, H, lambert_only=(key == "cornell_lambert"))


def load_profile_constants():
    """rocprofv3 figures measured in an earlier profiling run and committed under profiles/ (never measured inside this run)"""
    for name in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            d["_file"] = "profiles/" + name
            return d
        except Exception:
            continue
    return {}


def r4(x):
    return float("%.4g" % x)


def roofline_record(c, c1, value, spp_total, n, lanes, scene_key, prof):
    """roofline of the dominant kernel class from the per-launch HIP events of one profiled step (counters c), plus the same
    kernels with the GPU to themselves (c1: one extra step on ONE stream lane) and the whole-path figure.  Every class carries two
    byte counts: `contract` = SURVEY.md section 8d's per-unit figure (160 B per segment / 96 B per shadow ray: the whole record set,
    whichever kernel moves it) x units, and `attributed` = the bytes that class itself reads and writes."""
    survivors = max(0, c.closest_rays - c.samples)     # rays written by k_shade (every ray but the camera rays)

    def classes(x):
        sv = max(0, x.closest_rays - x.samples)
        d = {
            "k_extend": (x.extend_ms, x.extend_launches, B_PER_SEGMENT * x.closest_rays, x.closest_rays, B_EXTEND_PER_RAY * x.closest_rays),
            "k_shade": (x.shade_ms, x.shade_launches, B_PER_SEGMENT * x.closest_rays, x.closest_rays,
                        B_SHADE_PER_PATH_IN * x.closest_rays + B_SHADE_PER_SURVIVOR * sv + B_SHADE_PER_SHADOW_RAY * x.shadow_rays),
            "k_shadow": (x.shadow_ms, x.shadow_launches, B_PER_SHADOW * x.shadow_rays, x.shadow_rays, B_SHADOW_PER_RAY * x.shadow_rays),
        }
        whole = B_PER_SEGMENT * x.closest_rays + B_PER_SHADOW * x.shadow_rays
        if x.path_launches:                              # fused schedule: ONE kernel moves the whole record set
            d = {"k_path": (x.path_ms, x.path_launches, whole, x.samples, whole)}
        return d

    cls = classes(c)
    dom = max(cls, key=lambda k: cls[k][0])
    ms, launches, nbytes, units, attributed = cls[dom]
    launches = max(1, launches)
    achieved = (nbytes / launches) / (ms / launches * 1e-3) / 1e9 if ms > 0 else 0.0
    att_achieved = (attributed / launches) / (ms / launches * 1e-3) / 1e9 if ms > 0 else 0.0
    bytes_per_sample = (B_PER_SEGMENT * c.closest_rays + B_PER_SHADOW * c.shadow_rays) / max(1, c.samples) + B_FILM_PER_PIXEL / spp_total
    ps = prof.get(scene_key) or {}
    pk = ps.get(dom) or {}
    traffic = int(pk["hbm_bytes_per_unit"] * units / launches) if "hbm_bytes_per_unit" in pk else None
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launch_ms_avg": r4(ms / launches), "launches": int(launches), "algorithmic_bytes_per_launch": int(nbytes / launches),
            "unit_bytes": (B_PER_SHADOW if dom == "k_shadow" else (round(bytes_per_sample, 1) if dom == "k_path" else B_PER_SEGMENT)), "units_per_launch": int(units / launches),
            "attributed": {"bytes_per_launch": int(attributed / launches), "achieved": round(att_achieved, 1), "frac": round(att_achieved / HBM_PEAK_GBS, 4)},
            "class_ms": {k: round(v[0], 2) for k, v in cls.items()},
            # several stream lanes: the lanes' kernels overlap, so a launch's duration includes the time it shares the
            # GPU with the other lanes' kernels; kernel_time_over_wall is the average number of kernels in flight
            "lanes": lanes, "kernel_time_over_wall": round((c.extend_ms + c.shade_ms + c.shadow_ms + c.path_ms + c.other_ms) / max(1e-9, c.render_ms), 2),
            "whole_path": {"bytes_per_sample": round(bytes_per_sample, 1), "segments_per_sample": round(c.closest_rays / max(1, c.samples), 3),
                           "shadow_rays_per_sample": round(c.shadow_rays / max(1, c.samples), 3),
                           "achieved_GBps": round(value * 1e6 * bytes_per_sample / 1e9, 1),
                           "frac": round(value * 1e6 * bytes_per_sample / 1e9 / (HBM_PEAK_GBS * n), 4)}}
    # per class with the GPU to itself (one stream lane): contract and attributed HBM fractions, PMC bytes per unit, and the
    # vector-instruction issue roofline (wave64 VALU instructions per unit from the SQ counter pass x units / time)
    src = c1 if c1 is not None else c
    alone = {}
    for k, (ms1, n1, nb1, u1, at1) in classes(src).items():
        if ms1 <= 0:
            continue
        e = {"ms": r4(ms1 / max(1, n1)), "frac": round(nb1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 3), "attributed_frac": round(at1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
        pv = ps.get(k) or {}
        if "hbm_bytes_per_unit" in pv:
            e["pmc_bytes_per_unit"] = pv["hbm_bytes_per_unit"]
        if "valu_insts_per_unit" in pv:
            g = pv["valu_insts_per_unit"] * u1 / (ms1 * 1e-3) / 1e9
            e["valu_frac"] = round(g / VALU_PEAK_GINST, 3); e["lane_util"] = pv.get("lane_utilisation")
            e["simd_cycles_per_valu_inst"] = round(256 * 4 * VALU_CLOCK_GHZ / g, 2)   # 2.5-2.8: a pure v_add / v_mul stream; 3.9-4.4: fma / min / max / cmp / cvt / cndmask; the kernels' mix is ~85 % the latter (profiles/r04a_valu_issue_cost.txt)
        if "valu_busy" in pv:                                # [round 4] VALUBusy by rocprofv3's derived-metric formula, from the PMC pass (every dispatch profiled on its own)
            e["valu_busy_pmc"] = pv["valu_busy"]
        alone[k] = e
    roof["alone" if c1 is not None else "per_class"] = alone
    roof["notes"] = {"alone": "one extra untimed step on ONE stream lane: every kernel has the GPU to itself" if c1 is not None else "from the timed configuration",
                     "valu_peak": "256 CUs x 4 SIMDs x %.1f GHz / 2 cycles = %.0f G wave64 instructions/s -- reachable only by v_add / v_mul / v_and; this mix (fma, min / max, compare, convert: ~4 cycles; IEEE division 42) saturates at about half of it: valu_busy (profiles/r04a_valu_issue_cost.txt)" % (VALU_CLOCK_GHZ, VALU_PEAK_GINST),
                     "pmc": "%s: rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE and SQ_INSTS_VALU per unit, measured in an earlier profiling run of this scene" % prof.get("_file", "-")}
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="timed frames of the headline workload (80 frames ~ 10 s on one GPU)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", choices=sorted(SCENE_LABEL), default=None, help="headline scene (default: configs[2], cornell)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config index of the headline workload")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (weak scaling: per GPU share, total spp = spp * gpus)")
    ap.add_argument("--full-materials", action="store_true", help="kept for round-1 command lines: same as --scene cornell")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--configs", default="1,2,3,4", help="sub-records at N = 1: comma list of BASELINE.json config indices, '' for none")
    ap.add_argument("--min-seconds", type=float, default=10.0, help="length of each sub-record's timed region")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity samples")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra single-lane step behind roofline.alone (profiling runs: keeps rocprofv3's per-kernel averages to the timed configuration)")
    ap.add_argument("--band-rows", type=int, default=0, help="band height for sharding / lanes (0: largest height <= 20 that deals evenly)")
    ap.add_argument("--cpu-bands", type=int, default=2, help="20-row bands rendered by the CPU oracle at the full spp (Cornell parity sample)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the pool's driver supports dmabuf IPC only: RCCL across processes fails without it (already exported on the GPU boxes)
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

    prof = load_profile_constants()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ctx = jp.Context(dev)
    Hn = None
    if rank == 0 and not args.no_cpu:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import harness as Hn                                # oracle/ bindings: the checker and the CPU baseline, never the thing measured

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(scene_key, W, H, spp, steps, warmup, min_seconds=None, with_cpu=True, tag="", parity_fn=None):
        """upload, warm up, time `steps` frames (or as many as fill min_seconds), profile the last one; returns the record"""
        spp_total = spp * n if args.scaling == "weak" else spp
        be = build_scene(scenes, scenes.HostBackend("bench"), scene_key, W, H)
        scene = be.flatten()
        ctx.upload(scene)
        band_rows = args.band_rows if args.band_rows > 0 else jp.distributed.balanced_band_rows(H, n)
        params = jp.render_params(W, H, spp_total, 5, 1234, band_rows=band_rows, shard_index=rank, shard_count=n)
        film_dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if world > 1 else None

        def step():
            if world == 1:
                return ctx.render(params)                      # jp_render: kernels + film download, blocking
            ctx.render_device(params, film_dev.data_ptr(), sync=True)
            if backend == "nccl":                              # RCCL over xGMI: every rank sends its own rows, packed (1/N of the film)
                full = jp.distributed.assemble_bands(film_dev, H, band_rows, dist)
                return full.cpu().numpy() if rank == 0 else None
            full = jp.distributed.assemble_bands(film_dev.cpu(), H, band_rows, dist)   # rehearsal backend: the same on host tensors
            return full.numpy() if rank == 0 else None

        ctx.set_profiling(False)
        for _ in range(warmup):
            step()
        fence()
        if min_seconds is not None:                           # sub-records: as many frames as fill the region (>= 3)
            tw = time.perf_counter()
            step()                                            # one more untimed frame, after the allocations of the first
            fence()
            tw = time.perf_counter() - tw
            steps = max(3, int(min_seconds / max(1e-3, tw) + 0.999))
        t0 = time.perf_counter()
        film = None
        for i in range(steps):
            if i == steps - 1:
                ctx.set_profiling(True)                        # per-launch HIP events (kernel stream) on the last timed step: roofline below
            film = step()
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        samples_per_step = W * H * spp_total
        value = samples_per_step * steps / dt / 1e6
        c = ctx.counters()                                     # counters + per-class event times of the last timed step
        shard_check = None
        if dist is not None:
            # the N > 1 run checks itself: exactly N ranks took part, each traced the samples of its own bands, together the whole
            # frame, and the assembled film has no empty row (a rank whose rows never arrived would leave its bands zero)
            mine = torch.tensor([int(c.samples)], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            per = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(per, mine)
            per = [int(t.item()) for t in per]
            expect = [sum(y1 - y0 for y0, y1 in jp.distributed.bands_of(H, r, world, band_rows)) * W * spp_total for r in range(world)]
            shard_check = {"world_size": dist.get_world_size(), "ranks_expected": n, "samples_per_rank": per, "samples_expected_per_rank": expect,
                           "samples_total": sum(per), "samples_frame": samples_per_step}
            ok = dist.get_world_size() == n and per == expect and sum(per) == samples_per_step
            if rank == 0:
                # (a black row is legitimate -- the Cornell camera looks past the box; a rank whose rows never arrived leaves ALL its bands zero)
                rows_filled = np.abs(film.reshape(H, -1)).max(axis=1) > 0
                empty = [r for r in range(world) if not any(rows_filled[y0:y1].any() for y0, y1 in jp.distributed.bands_of(H, r, world, band_rows))]
                shard_check["ranks_with_empty_bands"] = empty; shard_check["empty_rows"] = int((~rows_filled).sum()); shard_check["finite"] = bool(np.isfinite(film).all())
                ok = ok and not empty and shard_check["finite"]
            # every rank learns the verdict (rank 0 alone sees the film): a failed check ends ALL ranks together instead of leaving the others in a collective
            okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = bool(int(okt.item()))
            shard_check["ok"] = ok
            if not ok:
                if rank == 0:
                    sys.stderr.write("bench.py: the %d-rank frame is incomplete: %s\n" % (n, json.dumps(shard_check)))
                raise SystemExit(3)
        bi = ctx.build_info()
        lanes = int(bi.lanes_last_render)
        # With several lanes a launch shares the GPU with the other lanes' kernels, so its duration (and the per-launch roofline
        # figure the contract asks for) reflects that sharing.  One extra, untimed step on ONE lane gives the same kernels' figures
        # when each has the GPU to itself -- reported next to the contract figure as roofline.alone.
        c1 = None
        if lanes > 1 and ctx.get_options().lanes == 0 and not args.no_exclusive:
            ctx.set_options(lanes=1)                           # JpOptions (ABI 7): schedule fields apply to the next render
            try:
                step()
                c1 = ctx.counters()
            finally:
                ctx.set_options()
        ctx.set_profiling(False)
        rec = None
        if rank == 0:
            rec = {"value": round(value, 2), "unit": "Msamples/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
                   "timed_region_s": round(dt, 3),
                   "workload": "%s, %dx%d, %d spp%s, maxDepth 5, counter sampler seed 1234" % (
                       SCENE_LABEL[scene_key], W, H, spp_total, (" (%d per GPU share)" % spp) if (n > 1 and args.scaling == "weak") else ""),
                   "traversal_mode": int(bi.traversal_mode), "lanes": lanes,
                   "roofline": roofline_record(c, c1, value, spp_total, n, lanes, scene_key, prof)}
            if shard_check is not None:
                rec["shard_check"] = shard_check
            if parity_fn is not None and n == 1 and not args.no_cpu:
                rec["cpu_baseline"], rec["l2_vs_cpu_ref"] = parity_fn(scene_key, scene, film, W, H, spp_total)
            elif with_cpu and Hn is not None and (n == 1 or os.environ.get("JETPBRT_BENCH_PARITY_ALL")):
                rec["cpu_baseline"], rec["l2_vs_cpu_ref"] = cpu_and_parity(scene_key, scene, film, W, H, spp_total)
            else:
                rec["cpu_baseline"], rec["l2_vs_cpu_ref"] = None, None
        return rec, band_rows, lanes

    def cpu_and_parity(scene_key, scene, film, W, H, spp_total):
        """oracle/ is the checker here, never the thing measured (rank 0, N = 1)"""
        nbands = (H + 19) // 20
        t16 = min(avail, 16)                               # the reference renders with 16 threads (main.cc:156)
        tb = min(avail, nbands)                            # one thread per 20-row task: all the parallelism the reference's decomposition has
        bi = ctx.build_info()
        if scene_key != "bunny":
            # parity at the FULL spp on whole 20-row bands (bands are independent under the counter sampler)
            k = max(1, nbands // max(1, args.cpu_bands))
            idx = (nbands // 2) % k
            p = jp.render_params(W, H, spp_total, 5, 1234, band_rows=20, shard_index=idx, shard_count=k)
            ref, _ = Hn.oracle_render(scene, p, tb)
            rows = np.zeros(H, bool)
            for y0, y1 in jp.distributed.bands_of(H, idx, k):
                rows[y0:y1] = True
            d = np.sqrt(((film[rows] - ref[rows]) ** 2).sum(-1))
            parity = {"mean_per_pixel_l2": float(d.mean()), "max_per_pixel_l2": float(d.max()), "pixels": int(d.size), "spp": spp_total,
                      "tolerance": 1e-4, "reference": "oracle/pt_oracle.cc (pinned bit-exact to the compiled reference), counter sampler",
                      "sample": "bands b %% %d == %d of %d" % (k, idx, nbands),
                      "bit_identical": bool(np.array_equal(film[rows].view(np.uint32), ref[rows].view(np.uint32))),
                      "libm_sincosf": int(bi.libm_sincosf)}
            cpu_spp = 32
        else:
            # The CPU oracle needs ~40 min for this frame.  The device's reference-tree mode walks the
            # reference's own tree with the reference's semantics and is bit-identical to the oracle (asserted here on one
            # oracle band at reduced spp, and by tests/test_gpu_parity.py), so it serves as the full-size reference.
            rb = scenes.HostBackend("bench_ref")
            rb.set_reference_tree(True)
            build_scene(scenes, rb, scene_key, W, H)
            rscene = rb.flatten()
            rctx = jp.Context(dev)
            try:
                rctx.upload(rscene)
                t1 = time.perf_counter()
                ref = rctx.render(jp.render_params(W, H, spp_total, 5, 1234))
                t_ref_gpu = time.perf_counter() - t1
                t1 = time.perf_counter(); rctx.render(jp.render_params(W, H, spp_total, 5, 1234)); t_ref_warm = time.perf_counter() - t1   # (a second frame: without the allocations)
                b = 17 if nbands > 17 else nbands // 2
                # the link to the CPU oracle: rows spread over the whole image (2-row bands j with j % 30 == 8: ten tasks for ten
                # oracle threads) at 192 spp, strictly compared
                low = min(192, spp_total)
                pb = jp.render_params(W, H, low, 5, 1234, band_rows=2, shard_index=8, shard_count=30)
                gband = rctx.render(pb)
                Hn.libc_srand(1)                            # the reference process' rand() state when it builds its tree
                t1 = time.perf_counter()
                oband, _ = Hn.oracle_render(rscene, pb, tb)
                t_link = time.perf_counter() - t1
                lrows = np.zeros(H, bool)
                for y0, y1 in jp.distributed.bands_of(H, 8, 30, 2):
                    lrows[y0:y1] = True
                link = bool(np.array_equal(gband[lrows].view(np.uint32), oband[lrows].view(np.uint32)))
                link_l2 = float(np.sqrt(((gband[lrows] - oband[lrows]) ** 2).sum(-1)).mean())
                # ... and at the FULL sample count (round 4): twelve single rows across the image (rows j % 50 == 37: twelve oracle tasks) of the
                # full-size reference-tree film itself against the CPU oracle -- every sample index of the frame, strictly compared
                pf_rows = jp.render_params(W, H, spp_total, 5, 1234, band_rows=1, shard_index=37, shard_count=50)
                Hn.libc_srand(1)
                t1 = time.perf_counter()
                ofull, _ = Hn.oracle_render(rscene, pf_rows, tb)
                t_full = time.perf_counter() - t1
                frows = np.zeros(H, bool)
                for y0, y1 in jp.distributed.bands_of(H, 37, 50, 1):
                    frows[y0:y1] = True
                link_full = bool(np.array_equal(ref[frows].view(np.uint32), ofull[frows].view(np.uint32)))
                link_full_l2 = float(np.sqrt(((ref[frows] - ofull[frows]) ** 2).sum(-1)).mean())
            finally:
                rctx.close()
            # the certified walk over the same (reference) tree -- FScene::certifiedWalk, DESIGN.md "Certified walk": speed, and how far its film is from the verbatim one
            certified = None
            cb = scenes.HostBackend("bench_cert")
            cb.set_reference_tree(True, certified=True)
            build_scene(scenes, cb, scene_key, W, H)
            cctx = jp.Context(dev)
            try:
                cctx.upload(cb.flatten())
                pf = jp.render_params(W, H, spp_total, 5, 1234)
                cctx.render(pf)                               # allocations, lanes
                t1 = time.perf_counter(); cfilm = cctx.render(pf); t_c = time.perf_counter() - t1
                cc = cctx.counters(); cbi = cctx.build_info()
                dc = np.sqrt(((cfilm - ref) ** 2).sum(-1))
                certified = {"Msamples_s": round(W * H * spp_total / t_c / 1e6, 1), "verbatim_Msamples_s": round(W * H * spp_total / t_ref_warm / 1e6, 1),
                             "mean_per_pixel_l2_vs_verbatim": float(dc.mean()), "fraction_pixels_identical": float((cfilm == ref).all(-1).mean()),
                             "rays_walked_again": int(cc.certified_fallback_rays), "rays": int(cc.closest_rays + cc.shadow_rays), "nodes": int(cbi.certified_nodes), "on": bool(cbi.certified_walk)}
            finally:
                cctx.close()
            d = np.sqrt(((film - ref) ** 2).sum(-1))
            band = d[b * 20:b * 20 + 20]
            parity = {"mean_per_pixel_l2": float(d.mean()), "max_per_pixel_l2": float(d.max()), "pixels": int(d.size), "spp": spp_total,
                      "tolerance": 1e-4,
                      "reference": "device reference-tree mode (traversal mode 5: the reference's rand()-driven tree, its box test and order) at the full size, %.1f s" % t_ref_gpu,
                      "reference_vs_oracle": {"sample": "%d rows spread over the image (2-row bands j %% 30 == 8) at %d spp, oracle/pt_oracle.cc on the CPU, %.1f s" % (int(lrows.sum()), low, t_link), "bit_identical": link, "mean_per_pixel_l2": link_l2,
                                              "full_spp": {"sample": "%d single rows (j %% 50 == 37) at the full %d spp = %d samples, %.1f s on %d threads" % (int(frows.sum()), spp_total, int(frows.sum()) * W * spp_total, t_full, tb),
                                                           "bit_identical": link_full, "mean_per_pixel_l2": link_full_l2}},
                      "fraction_pixels_identical": float((film == ref).all(-1).mean()), "fraction_pixels_gt_1e-3": float((d > 1e-3).mean()),
                      "band_through_meshes": {"band": b, "mean_per_pixel_l2": float(band.mean()), "fraction_pixels_identical": float((film[b * 20:b * 20 + 20] == ref[b * 20:b * 20 + 20]).all(-1).mean())},
                      "note": "the default path walks its own SAH tree: ~3e-4 of the samples through the meshes find a different first hit than the reference's rand()-driven tree does (fringe hits, DESIGN.md Numerics)",
                      "certified_walk": certified,
                      "libm_sincosf": int(bi.libm_sincosf)}
            cpu_spp = 4
        # CPU baseline: the whole frame at a reduced spp (throughput does not depend on spp), 20-row tasks, stock sampler
        pc = jp.render_params(W, H, cpu_spp, 5, 1234, sampler_mode=jp.JP_SAMPLER_STOCK_MT19937)
        nsamp = W * H * cpu_spp
        t1 = time.perf_counter(); Hn.oracle_render(scene, pc, t16); t_port = time.perf_counter() - t1
        cpu = {"value": round(nsamp / t_port / 1e6, 3), "unit": "Msamples/s", "cores": t16, "kind": "port",
               "sample": "whole %dx%d frame at %d spp = %d samples in %.1f s; oracle/pt_oracle.cc, stock mt19937_64 sampler, 20-row tasks on %d std::threads (of %d visible CPUs)" % (
                   W, H, cpu_spp, nsamp, t_port, t16, avail)}
        if Hn.have_ref():
            try:
                rb = build_scene(scenes, Hn.RefBackend("bench"), scene_key, W, H)
                t1 = time.perf_counter(); rb.render(W, H, cpu_spp, 5, 0, 1234, t16); t_ref = time.perf_counter() - t1
                cpu = {"value": round(nsamp / t_ref / 1e6, 3), "unit": "Msamples/s", "cores": t16, "kind": "reference",
                       "sample": "whole %dx%d frame at %d spp = %d samples in %.1f s; oracle/_ref (unmodified reference: DoRender per 20-row task on its FParallelSystem, FRandomSampler), %d threads (of %d visible CPUs)" % (
                           W, H, cpu_spp, nsamp, t_ref, t16, avail),
                       "port_value": round(nsamp / t_port / 1e6, 3)}
                if tb != t16:
                    t1 = time.perf_counter(); rb.render(W, H, cpu_spp, 5, 0, 1234, tb); t_refb = time.perf_counter() - t1
                    cpu["at_min_cores_bands"] = {"value": round(nsamp / t_refb / 1e6, 3), "cores": tb,
                                                 "note": "threads = min(visible CPUs, %d bands): every 20-row task of the reference on its own thread" % nbands}
            except Exception as e:                         # the prebuilt reference library is optional on the GPU box
                cpu["reference_error"] = str(e)
        if tb != t16 and "at_min_cores_bands" not in cpu:
            t1 = time.perf_counter(); Hn.oracle_render(scene, pc, tb); t_pb = time.perf_counter() - t1
            cpu["at_min_cores_bands"] = {"value": round(nsamp / t_pb / 1e6, 3), "cores": tb, "kind": "port"}
        return cpu, parity

    def parity_config4(scene_key, scene, film, W, H, spp_total):
        """configs[4] on ONE GPU: the rows rank 0 of 8 would own (15-row bands b % 8 == 0), default path vs the device's reference-tree
        mode at the full 4096 spp (the pairing of tests/test_gpu_parity.py::test_config4_...; the CPU reference of this scene is timed in
        the configs[3] record -- its rate does not depend on the frame size)"""
        worldc = 8
        band = jp.distributed.balanced_band_rows(H, worldc)
        rows = np.zeros(H, bool)
        for y0, y1 in jp.distributed.bands_of(H, 0, worldc, band):
            rows[y0:y1] = True
        rb = scenes.HostBackend("bench_ref4")
        rb.set_reference_tree(True)
        build_scene(scenes, rb, scene_key, W, H)
        rctx = jp.Context(dev)
        try:
            rctx.upload(rb.flatten())
            t1 = time.perf_counter()
            ref = rctx.render(jp.render_params(W, H, spp_total, 5, 1234, band_rows=band, shard_index=0, shard_count=worldc))
            t_ref = time.perf_counter() - t1
        finally:
            rctx.close()
        d = np.sqrt(((film[rows] - ref[rows]) ** 2).sum(-1))
        parity = {"mean_per_pixel_l2": float(d.mean()), "max_per_pixel_l2": float(d.max()), "pixels": int(d.size), "spp": spp_total, "tolerance": 1e-4,
                  "reference": "device reference-tree mode (traversal mode 5) on the rows of rank 0 of 8 (%d-row bands b %% 8 == 0: %d rows), %.1f s; linked to the CPU oracle in the configs[3] record" % (band, int(rows.sum()), t_ref),
                  "fraction_pixels_identical": float((film[rows] == ref[rows]).all(-1).mean()), "fraction_pixels_gt_1e-3": float((d > 1e-3).mean()),
                  "outside_shard_zero": bool((ref[~rows] == 0).all())}
        return None, parity

    # ---- headline ----------------------------------------------------------------------------------------------------------
    hk, hw, hh, hspp = CONFIGS[args.config]
    if args.full_materials:
        hk = "cornell"
    if args.scene:
        hk = args.scene
    hw = args.width or hw; hh = args.height or hh; hspp = args.spp or hspp
    sub_ids = [int(x) for x in args.configs.split(",") if x.strip()] if (n == 1 and args.configs) else []
    sub_ids = [i for i in sub_ids if i in (1, 2, 3, 4)]
    custom = bool(args.scene or args.width or args.height or args.spp or args.full_materials or args.config != 2)
    if custom:
        sub_ids = [] if args.configs == "1,2,3,4" else sub_ids          # a custom headline runs alone unless sub-records are asked for
    head_in_sub = (not custom) and 2 in sub_ids
    head, band_rows, lanes = run_workload(hk, hw, hh, hspp, args.steps, args.warmup, with_cpu=not head_in_sub)
    subs = {}
    for i in sub_ids:
        k, w, h, s = CONFIGS[i]
        rec, _, _ = run_workload(k, w, h, s, 0, 1, min_seconds=args.min_seconds, parity_fn=(parity_config4 if i == 4 else None))
        if rank == 0:
            subs["configs[%d]" % i] = rec
    if rank == 0:
        if head_in_sub:                                       # the headline's CPU / parity legs are those of its long sub-record
            head["cpu_baseline"] = subs["configs[2]"]["cpu_baseline"]; head["l2_vs_cpu_ref"] = subs["configs[2]"]["l2_vs_cpu_ref"]
        out = {
            "metric": "Msamples/sec (whole node) + per-pixel L2 vs CPU ref, cornell_box 1024spp",
            "value": head["value"], "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": args.scaling if n > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["workload"],
                       "parallelism": ("%d-row band shard x%d + RCCL gather of the ranks' rows" % (band_rows, n) if n > 1 else "single GPU") + ", %d stream lanes per GPU" % lanes},
            "roofline": head["roofline"], "cpu_baseline": head["cpu_baseline"], "l2_vs_cpu_ref": head["l2_vs_cpu_ref"],
            "timed_region_s": head["timed_region_s"],
        }
        if head.get("shard_check") is not None:
            out["shard_check"] = head["shard_check"]
        if subs:
            out["config"]["sub"] = {k.replace("configs[", "c").replace("]", ""): {
                "Msamples_s": v["value"], "ms": v["ms_per_step"], "frames": v["steps"], "s": v["timed_region_s"], "whole_path_frac": v["roofline"]["whole_path"]["frac"],
                "dominant": v["roofline"]["kernel"], "frac": v["roofline"]["frac"],
                "cyc_per_valu": {k2.replace("k_", ""): e2.get("simd_cycles_per_valu_inst") for k2, e2 in (v["roofline"].get("alone") or {}).items() if isinstance(e2, dict) and "simd_cycles_per_valu_inst" in e2} or None,
                "l2": (None if not v.get("l2_vs_cpu_ref") else r4(v["l2_vs_cpu_ref"]["mean_per_pixel_l2"])),
                "identical": (None if not v.get("l2_vs_cpu_ref") else (v["l2_vs_cpu_ref"].get("bit_identical") if "bit_identical" in v["l2_vs_cpu_ref"] else round(v["l2_vs_cpu_ref"].get("fraction_pixels_identical", 0), 4))),
                "cpu_ref_Msamples_s": (None if not v.get("cpu_baseline") else v["cpu_baseline"]["value"]),
                **({"certified": {"Msamples_s": v["l2_vs_cpu_ref"]["certified_walk"]["Msamples_s"], "l2_vs_verbatim": r4(v["l2_vs_cpu_ref"]["certified_walk"]["mean_per_pixel_l2_vs_verbatim"]),
                                  "identical": round(v["l2_vs_cpu_ref"]["certified_walk"]["fraction_pixels_identical"], 6)}} if (v.get("l2_vs_cpu_ref") or {}).get("certified_walk") else {})} for k, v in subs.items()}
            # the full sub-records (roofline, cpu_baseline, l2_vs_cpu_ref of every config) go to stderr: the ONE line on stdout stays
            # small enough for a log tail to keep whole, with every sub-record's figures in config.sub
            sys.stderr.write("configs_detail " + json.dumps(subs, separators=(",", ":")) + "\n"); sys.stderr.flush()
        print(json.dumps(out, separators=(",", ":")), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
