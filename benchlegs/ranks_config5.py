"""bench.py --route ranks --workload config5: BASELINE configs[4] with one process per GPU over torch.distributed."""
import ctypes as C
import json
import os
import time

import numpy as np

from .common import ROOT, HBM_PEAK_GBS, INFINITY_CACHE_BYTES, PG_TIMEOUT, Leg, emit, flush_c_stdio


def run_config5(args):
    """BASELINE.json configs[4]: 16 synthetic 1920x1080 streams, 16/N per GPU, wavefront invalid-depth compaction and a
    voxel-grid downsample of the stitched cloud on rank 0. A step = one frame-set through
      rank r : rasters -> voxel partials of its cameras (pcs_process_frames_voxel_partials_device; the points themselves
               are never written: the voxel sums are integers, so the grid of the union IS the grid of the stitched cloud)
      all    : all_gather of the partial counts, ONE grouped exchange of keys + partials to rank 0 (N > 1)
      rank 0 : sort + segmented mean over everybody's partials (pcs_voxel_grid_from_partials_device).
    The layout it replaces: src/pcs-multicamera-client.cpp:373-409 (concatenate on the centre) +
    src/pcs-multicamera-optimized.cpp:226-248 (downsample there)."""
    import hashlib
    import torch
    import torch.distributed as dist
    from pointcloud_stitching_amd import synthetic as Syn
    from pointcloud_stitching_amd.api import PcsContext
    from pointcloud_stitching_amd.stitch import ShardedVoxelGrid
    from pointcloud_stitching_amd.types import POINT_SHORTS, FLAG_DROP_INVALID

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libpcs_hip has no CPU fallback")
    debug_gloo = args.debug_backend == "gloo"
    if debug_gloo:
        local_rank = 0
    local_rank %= max(torch.cuda.device_count(), 1)      # (a launcher that shows every rank only its own GPU: index 0)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if debug_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=PG_TIMEOUT)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=PG_TIMEOUT)

    full = (args.streams, args.width, args.height) == (8, 1280, 720)        # the stitch workload's defaults: not given
    total_streams, W, H = (16, 1920, 1080) if full else (args.streams, args.width, args.height)
    if total_streams % world:
        raise SystemExit(f"config5 shards {total_streams} streams over {world} GPUs: not divisible")
    S, LEAF = total_streams // world, args.leaf
    npts = W * H
    cfgs = [Syn.synth_stream_config(W, H, rank * S + s) for s in range(S)]
    ctx = PcsContext(cfgs, device=local_rank, flags=FLAG_DROP_INVALID)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    in_bytes = S * npts * 5
    R = max(args.ring, 2) if args.ring else max(3, -(-2 * INFINITY_CACHE_BYTES // in_bytes) + 2)
    dep0 = [torch.from_numpy(Syn.synth_depth(W, H, rank * S + s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
    col0 = [torch.from_numpy(Syn.synth_color(W, H, rank * S + s)).to(dev) for s in range(S)]
    sets = [(dep0, col0)] + [([d.clone() for d in dep0], [c.clone() for c in col0]) for _ in range(R - 1)]
    VP = C.c_void_p
    ptrs = [([t.data_ptr() for t in d], [t.data_ptr() for t in c]) for d, c in sets]
    svg = ShardedVoxelGrid(ctx, LEAF, dev)
    out_shorts = svg.total_cap * POINT_SHORTS
    vox = torch.empty(out_shorts if rank == 0 else 8, dtype=torch.int16, device=dev)
    k = [0]

    def pre():
        d, c = ptrs[k[0] % R]; k[0] += 1
        svg.pre_aggregate(d, c)

    def reduce_():
        if rank != 0:
            return
        if world == 1:      # nothing to size on the host: the partial count is read from device memory
            ctx.voxel_grid_from_partials_device(svg.keys.data_ptr(), svg.parts.data_ptr(), svg.cap, LEAF, vox.data_ptr(), out_shorts,
                                                svg.n_vox.data_ptr(), d_n_partials=svg.n_local.data_ptr())
        else:
            svg.reduce(vox.data_ptr(), out_shorts)

    def step():
        pre()
        if world > 1:
            svg.exchange()
        reduce_()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- correctness before timing: the root's voxel cloud against the committed oracle digests where they apply --------------
    step(); torch.cuda.synchronize(dev)
    check = {}
    if rank == 0:
        nv = svg.voxels(vox.data_ptr(), out_shorts) if world > 1 else int(svg.n_vox[0].item())
        digest = hashlib.sha256(vox[:nv * POINT_SHORTS].cpu().numpy().tobytes()).hexdigest()
        check = {"voxels": nv, "voxel_sha256": digest, "golden": None}
        gpath = os.path.join(ROOT, "tests", "golden", "config5_digests.json")
        if (total_streams, W, H) == (16, 1920, 1080) and os.path.exists(gpath):
            gold = json.load(open(gpath))["voxel"].get(str(LEAF))
            if gold:
                check["golden"] = bool(gold["voxels"] == nv and gold["sha256"] == digest)
                if not check["golden"]:
                    raise SystemExit(f"bench aborted: config5 voxel cloud differs from the oracle digest (leaf {LEAF} mm)")
    for _ in range(max(args.warmup, 3)):
        step()
    barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.timer_end()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        red = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if debug_gloo else dev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed = float(red.item())

    # ---- the phases on their own (synchronised between them: a diagnostic, not the timed region) ---------------------------------
    ph = {"kernel": 0.0, "exchange": 0.0, "root_voxel": 0.0}
    n_ph = 10
    for _ in range(n_ph):
        barrier(); a = time.perf_counter()
        pre(); torch.cuda.synchronize(dev); b = time.perf_counter()
        if world > 1:
            svg.exchange(); torch.cuda.synchronize(dev)
        c = time.perf_counter()
        reduce_(); torch.cuda.synchronize(dev); d = time.perf_counter()
        ph["kernel"] += b - a; ph["exchange"] += c - b; ph["root_voxel"] += d - c
    # the dominant kernel, by HIP events on the launch stream
    ctx.timer_begin()
    for _ in range(20):
        pre()
    ctx.timer_end()
    kern_ms = ctx.timer_elapsed_ms() / 20
    m_local = int(svg.n_local[0].item())
    counts = svg.counts if world > 1 else [m_local]
    if world > 1:
        mx = torch.tensor([ph["kernel"], ph["exchange"], ph["root_voxel"]], dtype=torch.float64,
                          device=torch.device("cpu") if debug_gloo else dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        ph = dict(zip(("kernel", "exchange", "root_voxel"), [float(x) for x in mx.tolist()]))

    if world > 1:
        flush_c_stdio()           # (as in main: nothing of any rank may follow rank 0's line on the shared stdout)
        dist.barrier()
    if rank == 0:
        pts_step = total_streams * npts
        ms_per_step = elapsed * 1e3 / args.steps
        algo = S * npts * 5 + m_local * 40            # this rank's launch: 2 B Z16 + 3 B RGB8 per pixel in, 40 B per partial out
        ach = algo / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpoints/s in (16x1920x1080 streams: deproject+transform+RGB+pack, invalid-depth compaction, voxel grid of the stitched cloud)",
            "value": round(pts_step * args.steps / elapsed / 1e6, 1), "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4]: {total_streams} synthetic {W}x{H} Z16+RGB8 streams, {S} per GPU x {world} GPU(s), "
                                   f"PCS_FLAG_DROP_INVALID (wavefront invalid-depth compaction), voxel-grid downsample (leaf {LEAF} mm) of the "
                                   f"stitched cloud on rank 0: per-rank voxel partials, one exchange of the partials, one sort + segmented mean",
                       "streams_total": total_streams, "streams_per_gpu": S, "width": W, "height": H, "leaf_mm": LEAF,
                       "ring_frame_sets": R, "ring_inputs_between_rereads_mbytes": round((R - 1) * in_bytes / 1e6, 1),
                       "parallelism": f"streams sharded {S}/GPU x {world}", "pipeline": "synchronous per step (the exchange is sized by "
                       "data-dependent counts: one host round trip per step at N > 1; none at N = 1)"},
            "check": check,
            "phases_ms": {"kernel": round(ph["kernel"] * 1e3 / n_ph, 4), "exchange": round(ph["exchange"] * 1e3 / n_ph, 4),
                          "root_voxel": round(ph["root_voxel"] * 1e3 / n_ph, 4),
                          "note": "host clock with a device synchronisation after each phase, max over ranks; the timed region has none at N = 1"},
            "partials_per_rank": counts, "partials_total": int(sum(counts)),
            "exchange_bytes_per_step": int(sum(counts[1:]) * 40),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": None, "kernel": "pcs_fused_voxel_partials_kernel", "avg_launch_ms": round(kern_ms, 5),
                         "algorithmic_bytes_per_launch": int(algo),
                         "note": "rank 0's pre-aggregation launch: 5 B per pixel in + 40 B per partial out; the kernel is VALU / LDS bound "
                                 "(deprojection + pack + voxel key + LDS hash table per pixel), not HBM bound",
                         "timing": "hipEvent pair on the launch stream around 20 back-to-back launches"},
        }
        if debug_gloo:
            out["debug"] = "gloo control-flow test: all ranks on one GPU, host-staged exchange; numbers are meaningless"
        if world == 1 and not args.no_cpu_baseline:
            with Leg(out, "cpu_baseline"):
                from oracle import pcs_oracle as O
                ns = min(8, S)
                hd = [Syn.synth_depth(W, H, s) for s in range(ns)]
                hc = [Syn.synth_color(W, H, s) for s in range(ns)]
                best, passes, t_end = float("inf"), 0, time.perf_counter() + args.cpu_seconds
                while passes < 1 or time.perf_counter() < t_end:
                    tc = time.perf_counter()
                    st_, _ = O.process_frames(cfgs[:ns], hd, hc, FLAG_DROP_INVALID, 1)
                    O.voxel_grid(st_, LEAF)
                    best = min(best, time.perf_counter() - tc); passes += 1
                out["cpu_baseline"] = {"value": round(ns * npts / best / 1e6, 2), "unit": "Mpoints/s", "cores": 1, "kind": "port",
                                       "sample": f"{ns} of {total_streams} streams: deprojection + pack + compaction + stitch + voxel grid by the "
                                                 f"scalar CPU oracle, best of {passes} passes ({best:.2f} s each); the reference itself has no "
                                                 f"voxel grid (src/pcs-multicamera-optimized.cpp:17 only includes the header)"}
        emit(out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()

