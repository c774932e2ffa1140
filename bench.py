#!/usr/bin/env python3
"""bench.py — Mpoints/s stitched for 8 x 1280x720 synthetic streams (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                       (any N; N > 1 = the one-process node route)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W         (the same: rank 0 drives the node, the others exit)
    ... --route ranks                                                    (one process per GPU over torch.distributed / RCCL)

A "step" = one pass of the fused deproject -> transform -> RGB attach -> pack kernel over one frame-set
already resident in HBM. The frame-sets live in a ring whose INPUT rasters alone are more than twice the
256 MiB Infinity Cache, and ONE launch counter runs through pre-heat, warm-up and the timed region, so a
slot is never re-read before at least 2 x 256 MiB of other inputs went by: every read comes from HBM.

Workload (the default, `--scaling strong`): 8 streams IN TOTAL, 8/N per GPU.
  N = 1  BASELINE.json configs[2]: 8 streams batched on one GPU (the metric's configuration).
  N = 8  BASELINE.json configs[3]: one stream per GPU, the packed payloads gathered to rank 0 over RCCL/xGMI in
         camera order (what /root/reference's src/pcs-multicamera-client.cpp:373-409 does over TCP).
`--scaling weak` keeps 8 streams PER GPU (64 streams at N = 8) — not a BASELINE configuration, kept as an option.
Routes at N > 1. `node` (the default): ONE process drives the N GPUs through libpcs_node — the C++ host over
the C ABI that `north_star` asks for: per-GPU contexts of libpcs_hip, ncclCommInitAll, one grouped ncclSend/ncclRecv to
GPU 0 per frame-set, pipelined as submit(k+1); wait(k). It works however the script is launched: plain, or under
torch.distributed.run (rank 0 does the work, the other ranks exit at once). `ranks`: one process per GPU, the exchange
through torch.distributed (pointcloud_stitching_amd/stitch.py); needs torch.distributed.run — launched plain it re-executes
itself under it.
Rank 0 prints ONE JSON line. At N = 1 it also carries the other kernels' legs (ordered compaction, K frame-sets per
launch, the batched a2 twin), each with its own algorithmic byte model, and the CPU baseline.

Layout: this file parses the command line, picks the route, measures the headline and assembles the ONE line; the rig (ring of
frame-sets, launch forms) is benchlegs/rig.py, every other leg is a function in benchlegs/legs_*.py that returns its object and
runs under a guard that names a failed leg under `leg_errors` instead of costing the line (benchlegs/common.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlegs.common import (ALGO_BYTES_PER_POINT, PACK_BYTES_PER_POINT, HBM_PEAK_GBS, INFINITY_CACHE_BYTES, PG_TIMEOUT, POLICY,      # noqa: E402,F401
                              Leg, cpu_baseline, emit, flush_c_stdio, run_leg)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): --streams cameras IN TOTAL, sharded streams/N per GPU (BASELINE configs[2] at "
                         "N=1, configs[3] at N=8); weak: --streams cameras PER GPU")
    ap.add_argument("--streams", type=int, default=8, help="camera streams (total for strong scaling, per GPU for weak)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--ring", type=int, default=0,
                    help="frame-sets resident in HBM (ring); default: enough that a slot is re-read only after > 2x the "
                         "256 MiB Infinity Cache of other input rasters (16 for 8 x 1280x720)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the headline leg: skip compaction / batched / pack / cache-resident / two-stream / rotation legs")
    ap.add_argument("--no-cache-leg", action="store_true",
                    help="skip the informational legs (Infinity-Cache-resident ring, two HIP streams)")
    ap.add_argument("--no-config5", action="store_true", help="skip the 16 x 1920x1080 compaction + voxel-grid leg")
    ap.add_argument("--no-gather", action="store_true", help="N>1: shard only, skip the gather to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive host-pointer measurement")
    ap.add_argument("--no-general-rotation", action="store_true",
                    help="skip the extra leg that times the same workload with a non-identity depth->colour rotation")
    ap.add_argument("--debug-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: exercise the N>1 control flow on ONE GPU (all ranks on device 0, gathers staged through "
                         "host memory). For testing the script only — the numbers mean nothing.")
    ap.add_argument("--payload-skew", type=int, default=0,
                    help="diagnostic: offset the payload pointer by this many bytes (4 = the reference's buffer+2 shorts) "
                         "to force the generic (unaligned) store path")
    ap.add_argument("--mode", choices=["dense", "drop_invalid", "cutoff", "pack", "pack_batch", "batch", "batch_drop_invalid"], default="dense",
                    help="diagnostic: make another kernel the headline of the line (for profiling one kernel at a time): "
                         "the compaction path, the a2 twin per stream / batched, or K frame-sets per launch")
    ap.add_argument("--workload", choices=["stitch", "config5"], default="stitch",
                    help="stitch (default): the metric's workload (BASELINE configs[2] at N=1, configs[3] at N=8). config5: BASELINE "
                         "configs[4] — 16 x 1920x1080 streams sharded 16/N per GPU, invalid-depth compaction, voxel grid of the "
                         "stitched cloud on rank 0 (per-rank voxel partials, one exchange, one sort + segmented mean)")
    ap.add_argument("--leaf", type=int, default=50, help="config5: voxel leaf in millimetres")
    ap.add_argument("--route", choices=["auto", "node", "ranks"], default="auto",
                    help="how N GPUs are driven. node: ONE process, libpcs_node (C++ host, ncclCommInitAll, one grouped "
                         "ncclSend/ncclRecv to GPU 0 per frame-set, pipelined submit/wait). ranks: one process per GPU over "
                         "torch.distributed (needs torch.distributed.run; launched plain it re-executes itself under it). "
                         "auto (default): node for N > 1, the single-GPU legs for N = 1")
    ap.add_argument("--node-direct-child", action="store_true", help=argparse.SUPPRESS)      # (the direct-store leg's own process)
    ap.add_argument("--node-devices", type=str, default="",
                    help="node route: explicit device ids, one per peer (default 0..N-1). A repeated id makes virtual peers of "
                         "one GPU whose transfers become RCCL self send/recv pairs: `--gpus 2 --node-devices 0,0` runs the N = 2 "
                         "flow on a one-GPU box. For testing the flow — the numbers then say nothing about scaling")
    ap.add_argument("--batch-sets", type=int, default=4, help="frame-sets per launch of the batched-dense leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--preheat-ms", type=float, default=400.0,
                    help="untimed launches before the warm-up steps so clocks/power state settle (the first "
                         "~10 ms after idle run ~15 %% slower on MI355X)")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: profiles/traffic.json")
    return ap.parse_args()



def _reexec_under_torchrun(args):
    """--route ranks launched plain with N > 1: run the same command line under torch.distributed.run (one rank per GPU)
    and hand its output and exit code through. Never a SystemExit for a launcher reason."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def choose_route(route, gpus, node_devices, debug_backend, world, env, visible_devices):
    """How `--gpus N` is driven. `visible_devices` is a callable (only asked when it matters).
    node:  ONE process, libpcs_node — any N > 1, launched plain or one rank per GPU (rank 0 drives the node, the others exit).
    ranks: one process per GPU over torch.distributed — N = 1's single-GPU legs; the gloo control-flow test; and a launcher
           that hides all but one GPU from every rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank: rank 0 cannot
           drive GPUs it does not see, and every rank of such a launcher reaches the same verdict on its own). A box that
           simply has fewer GPUs than asked for stays on the node route, which folds the peers onto the visible ones."""
    if route != "auto":
        return route
    route = "node" if ((gpus > 1 or node_devices) and debug_backend != "gloo") else "ranks"
    if route == "node" and world > 1 and not node_devices:
        hidden = any(env.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        if hidden and 0 < visible_devices() < gpus:
            route = "ranks"
    return route




def headline(args, g, world, elapsed, gpu_ms, shard_gpu_ms, long_sample, traffic, traffic_src, policy, gather, checked_slots):
    """The contract's keys + `roofline` from the timed region. Two clocks bracket that ONE region and the line says which is which:
    `value` / `ms_per_step` are the HOST wall clock between the barriers (the contract's), `roofline.avg_launch_ms` / `achieved` /
    `frac` the hipEvent pair on the launch stream inside it (the kernel's own average launch duration, what rocprofv3 reports);
    `roofline.frac_wall` prices the same bytes with ms_per_step, so the headline and a fraction follow from one number."""
    W, H, S, R, strong, total_streams = g.W, g.H, g.S, g.R, g.strong, g.total_streams
    set_points, sets_per_launch, KB, kept_frac = g.set_points, g.sets_per_launch, g.KB, g.kept_frac
    total_points = set_points * sets_per_launch * world * args.steps
    ms_per_step = elapsed * 1e3 / args.steps
    kern_ms = gpu_ms / args.steps          # HIP-event bracket on the launch stream / launches
    roofline_timing = "hipEvent pair on the launch stream around the timed region / steps"
    if shard_gpu_ms is not None:
        # N > 1 with the gather: in the timed region the launch stream also waits for the exchange, so the bracket
        # there is a link figure. The kernel's own launch duration comes from the same K launches without it.
        kern_ms = shard_gpu_ms / args.steps
        roofline_timing = ("hipEvent pair on the launch stream around the same K launches WITHOUT the gather (rank 0); "
                           "the timed region's bracket includes waits for the exchange")
    bytes_pp = {"pack": PACK_BYTES_PER_POINT, "pack_batch": PACK_BYTES_PER_POINT,
                "drop_invalid": 5 + 10 * kept_frac, "batch_drop_invalid": 5 + 10 * kept_frac}.get(args.mode, ALGO_BYTES_PER_POINT)
    algo_bytes = set_points * sets_per_launch * bytes_pp
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    where = ("one GPU" if world == 1 else f"{world} GPUs, {S} per GPU") if strong else f"per GPU x {world} GPUs"
    cfg_name = ("BASELINE.json configs[2]" if world == 1 else
                "BASELINE.json configs[3]" if (strong and S == 1 and total_streams == 8) else
                f"{total_streams} streams sharded {S}/GPU" if strong else "weak scaling (not a BASELINE configuration)")
    out = {
        "metric": "Mpoints/s stitched (8x1280x720 streams: deproject+transform+RGB+pack)",
        "value": round(total_points / elapsed / 1e6, 1),
        "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{total_streams} synthetic {W}x{H} Z16+RGB8 streams on {where}, batched fused kernel, "
                               f"one extrinsic per stream ({cfg_name})"
                               + (", payloads gathered to rank 0 in camera order" if gather else ""),
                   "arithmetic": "f32 deprojection + affine (bit-exact vs the -m path), u16 depth in, u8 colour in, int16 records out",
                   "streams_total": total_streams, "streams_per_gpu": S, "width": W, "height": H,
                   "points_per_step_per_gpu": set_points * sets_per_launch,
                   "ring_frame_sets": R, "ring_mbytes": round(g.ring_bytes / 1e6, 1),
                   "ring_inputs_between_rereads_mbytes": round((R - 1) * g.in_bytes_per_set / 1e6, 1),
                   "ring_cold": bool(g.ring_cold),
                   "gather_to_rank0": bool(gather), "parallelism": f"streams sharded {S}/GPU x {world}"},
        "check": {"oracle_compared": {"slots": checked_slots, "streams": S, "records": "all"}},
        "parity": "bit-exact against this build's own restatement of the -m path (oracle/); the reference cannot be compiled here "
                  "(no librealsense): unpinned",
        "per_stream_fps": round(args.steps * sets_per_launch / elapsed, 1),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_wall": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "clocks": "achieved / frac / avg_launch_ms: hipEvent pair on the launch stream (the kernel's average launch duration); "
                               "frac_wall: the same bytes over ms_per_step, the host wall clock that `value` is computed from",
                     "traffic": traffic,
                     "traffic_source": traffic_src if traffic is not None else None,
                     "kernel": {"dense": "pcs_fused_dense_kernel", "batch": "pcs_fused_dense_batch_kernel",
                                "pack": "pcs_pack_dense_kernel", "pack_batch": "pcs_pack_batch_kernel",
                                "drop_invalid": "pcs_fused_count_kernel + pcs_scan_kernel + pcs_fused_emit_kernel (PCS_COMPACT_PATH=single: pcs_fused_compact_kernel)",
                                "batch_drop_invalid": "pcs_fused_count_batch_kernel + pcs_scan_batch_kernel + pcs_fused_emit_batch_kernel",
                                "cutoff": "pcs_fused_count_kernel + pcs_scan_kernel + pcs_fused_emit_kernel (PCS_COMPACT_PATH=single: pcs_fused_compact_kernel)"}[args.mode],
                     "arithmetic": policy, "avg_launch_ms": round(kern_ms, 5), "avg_launch_ms_unrounded": kern_ms,
                     "long_sample_ms": round(long_sample[0], 5) if long_sample else None,
                     "long_sample_launches": long_sample[1] if long_sample else None,
                     "long_sample_frac": round(algo_bytes / (long_sample[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if long_sample else None,
                     "algorithmic_bytes_per_launch": round(algo_bytes),
                     "algorithmic_bytes_per_point": round(bytes_pp, 3),
                     "timing": roofline_timing},
    }
    if args.mode != "dense":
        out["config"]["mode"] = {
            "pack": "diagnostic: a2 twin from resident vertices/texcoords, one launch per stream, 33 B/point",
            "pack_batch": "diagnostic: a2 twin from resident vertices/texcoords, all streams in one launch, 33 B/point",
            "batch": f"diagnostic: {KB} frame-sets per launch (pcs_process_frames_device_batch)",
            "drop_invalid": f"diagnostic: ordered invalid-depth compaction, kept fraction {kept_frac:.4f}, (5 + 10 rho) B/point",
            "batch_drop_invalid": f"diagnostic: ordered invalid-depth compaction of {KB} frame-sets per call (three launches for all of them), "
                                  f"kept fraction {kept_frac:.4f}, (5 + 10 rho) B/point",
            "cutoff": "diagnostic: ordered -c cutoff compaction (bytes priced as the dense kernel's 15 B/point: upper bound)",
        }[args.mode]
        out["roofline"]["traffic"] = None
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def visible():
        try:
            from pointcloud_stitching_amd import lib as _L
            return int(_L.load().pcs_device_count())
        except Exception:       # noqa: BLE001
            return 0
    route = choose_route(args.route, args.gpus, args.node_devices, args.debug_backend, world, os.environ, visible)
    if route == "node":
        if int(os.environ.get("RANK", "0")) != 0:
            return 0                     # under torch.distributed.run: rank 0's process drives every GPU of the node
        from benchlegs.node_route import run_node
        return run_node(args)
    if args.gpus > 1 and world == 1:
        return _reexec_under_torchrun(args)
    if args.workload == "config5":
        from benchlegs.ranks_config5 import run_config5
        return run_config5(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from pointcloud_stitching_amd.types import POINT_SHORTS

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libpcs_hip has no CPU fallback")
    # The CPU samples run FIRST, in child processes, while this process has not yet created a HIP context or a stream:
    # nothing of the GPU legs (runtime helper threads, pinned-memory traffic, thermal state of the host) can move them.
    cpu_first, cpu_first_error, cpu_single = None, None, None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu_first = cpu_baseline(args.width, args.height, args.streams, args.cpu_seconds)
        except Exception as e:       # noqa: BLE001 — a leg must never cost the line
            cpu_first_error = f"{type(e).__name__}: {e}"[:300]
        if args.mode == "dense" and not args.no_host_api:
            try:      # ONE frame per pass: what the reference's one-camera process does (host_api_single prints it beside the GPU's host forms)
                cpu_single = cpu_baseline(args.width, args.height, 1, max(args.cpu_seconds / 3.0, 0.5))
            except Exception:       # noqa: BLE001
                cpu_single = None
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

    from benchlegs.rig import Rig
    g = Rig(args, rank, world, local_rank)
    W, H, S, R, npts, ctx = g.W, g.H, g.S, g.R, g.npts, g.ctx
    strong, total_streams, set_points, payload_shorts = g.strong, g.total_streams, g.set_points, g.payload_shorts
    sets_per_launch, KB, launch = g.sets_per_launch, g.KB, g.launch

    gather = world > 1 and not args.no_gather
    variable = gather and g.mode_flags != 0        # compaction: per-rank counts differ -> counts all-gathered, payloads sent point to point
    g.variable = variable
    last_counts = [None]
    stitched = None
    if gather:
        from pointcloud_stitching_amd.stitch import RankStitcher
        st = RankStitcher()
        if rank == 0:
            stitched = [torch.empty(payload_shorts * world, dtype=torch.int16, device=dev) for _ in range(2)]

    pending = [None, None]
    red_dev = torch.device("cpu") if debug_gloo else dev      # where the tiny control reductions live

    class _Done:
        def wait(self):
            return True

    def gather_async(src, dst):
        if not debug_gloo:
            return st.gather_fixed(src, dst, async_op=True)
        torch.cuda.synchronize(dev)                       # script test only: stage through host memory
        host = src.view(torch.uint8).cpu()
        outs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, outs, dst=0)
        if rank == 0:
            db = dst.view(torch.uint8)
            for r, o in enumerate(outs):
                db[r * host.numel():(r + 1) * host.numel()].copy_(o)
        return _Done()

    step_no = [0]

    def step():
        if variable:
            # the kept counts are data dependent: launch, all_gather the totals (read from the device word the kernel wrote),
            # one grouped send/recv into the root buffer at the camera-order offsets. Synchronous per step.
            slot = g.counter % R
            launch()
            last_counts[0] = st.gather_variable(g.d_out[slot], g.d_cnt[S], stitched[0] if rank == 0 else None)
        elif gather:
            k = step_no[0]; step_no[0] = k + 1
            if pending[k & 1] is not None:       # the buffer pair (slot's out, stitched[k&1]) is free again
                pending[k & 1].wait()
            slot = g.counter % R
            launch()
            pending[k & 1] = gather_async(g.d_out[slot], stitched[k & 1] if rank == 0 else None)
        else:
            launch()

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait(); pending[i] = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # parity check against the oracle before timing (bench must not time a wrong kernel): EVERY stream of ring slots 0 and 1
    # (distinct frames), on the very buffers the timed region uses
    g.counter = 0
    launch(); torch.cuda.synchronize(dev)
    if sets_per_launch == 1 and R > 1:
        launch(); torch.cuda.synchronize(dev)          # slot 1 (the batched modes covered it with the first launch)
    checked_slots = [0, 1] if (R > 1 and (sets_per_launch == 1 or KB >= 2)) else [0]
    from oracle import pcs_oracle as O
    for slot in checked_slots:
        hd, hc = (g.host0, g.host1)[slot]
        if args.mode in ("pack", "pack_batch"):
            hd = g.host0[0]                            # the a2 twin's ring holds slot 0's vertices beside every slot's colour raster
        want, _ = O.process_frames(g.cfgs, hd, hc, g.mode_flags, 1)
        got = g.d_out[slot][:want.size].cpu().numpy().reshape(-1, 5)
        if got.shape != want.shape or (got != want).any():
            raise SystemExit(f"bench aborted: HIP output of ring slot {slot} differs from the oracle "
                             f"({int((got != want).any(axis=1).sum())} of {want.shape[0]} records)")

    g.preheat(launch, args.preheat_ms)
    gather_error = None
    if gather:
        # The exchange cannot be exercised on the single-GPU development boxes; if RCCL refuses it here the run
        # degrades to shard-only (and says so) instead of producing no line at all.
        try:
            first_slot = g.counter % R
            step(); step(); drain(); torch.cuda.synchronize(dev)
            ok = torch.tensor([1], dtype=torch.int32, device=red_dev)
        except Exception as e:          # noqa: BLE001
            gather_error = f"{type(e).__name__}: {e}"[:300]
            ok = torch.tensor([0], dtype=torch.int32, device=red_dev)
            pending[0] = pending[1] = None
        try:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                gather = False
        except Exception as e:          # noqa: BLE001
            gather = False
            gather_error = (gather_error or "") + f" | all_reduce: {e}"[:200]
        if rank == 0 and gather:
            # a7: rank r's payload must sit at [r*n, (r+1)*n) of the stitched buffer; rank 0's own slice is checkable here
            own_n = last_counts[0][0] * POINT_SHORTS if variable else payload_shorts
            own = stitched[0][:own_n]
            if variable:
                first_slot = (g.counter - 1) % R         # the slot of the most recent step
            if not torch.equal(own, g.d_out[first_slot][:own_n]):
                raise SystemExit("bench aborted: gathered slice of rank 0 differs from its payload")
    for _ in range(args.warmup):
        step()
    drain(); barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.timer_end()
    drain(); barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ctx.timer_elapsed_ms()

    # The driver's --steps 20 makes the contract bracket 0.5 ms long. The SAME launch loop for >= 50 ms, printed beside it
    # (roofline.long_sample_ms), says whether a low contract figure is the box or the run length.
    long_sample = None
    if world == 1:
        n_long = max(int(50.0 / max(gpu_ms / args.steps, 1e-3)) + 1, args.steps)
        long_ms = g.timed(launch, n_long)
        torch.cuda.synchronize(dev)
        long_sample = (long_ms, n_long)

    shard_only = None
    shard_gpu_ms = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if gather:
            # the same K steps without the exchange: what the kernels alone sustain when streams are only sharded
            barrier()
            t1 = time.perf_counter()
            ctx.timer_begin()
            for _ in range(args.steps):
                launch()
            ctx.timer_end()
            barrier()
            shard_gpu_ms = ctx.timer_elapsed_ms()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            shard_only = float(t.item())

    traffic, traffic_src = args.traffic, "--traffic"
    if traffic is None:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("workload") == f"{S}x{W}x{H}" and args.mode == "dense":
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/traffic.json (" + tj.get("tag", "?") + ")"
        except (OSError, ValueError, KeyError):
            traffic = None
    policy = POLICY[min(ctx.stream_math(s) for s in range(S))]

    if world > 1:
        # every rank's C stdio (RCCL's banner) is out before rank 0 prints the line: the ranks share one stdout under torchrun
        flush_c_stdio()
        dist.barrier()
    if rank == 0:
        out = headline(args, g, world, elapsed, gpu_ms, shard_gpu_ms, long_sample, traffic, traffic_src, policy, gather, checked_slots)
        kern_ms = out["roofline"]["avg_launch_ms_unrounded"]
        del out["roofline"]["avg_launch_ms_unrounded"]
        if debug_gloo:
            out["debug"] = "gloo control-flow test: all ranks on one GPU, host-staged gathers; numbers are meaningless"
        if gather_error:
            out["gather_error"] = gather_error
            out["config"]["gather_to_rank0"] = False
        if shard_only is not None:
            total_points = set_points * sets_per_launch * world * args.steps
            peer_bytes = (sum(last_counts[0][1:]) * 10 / max(world - 1, 1)) if variable else payload_shorts * 2
            gb = (world - 1) * peer_bytes * args.steps / elapsed / 1e9
            out["gather"] = {"root_ingest_GBps": round(gb, 1), "bytes_per_peer_per_step": int(peer_bytes),
                             "form": ("variable: all_gather of the kept counts (device) + grouped isend/irecv at camera-order offsets, "
                                      "synchronous per step") if variable else "fixed: dist.gather into views of the root buffer, double-buffered",
                             "counts_per_rank": last_counts[0] if variable else None,
                             "note": "value includes one RCCL gather of every rank's payload to rank 0 per step "
                                     "(double-buffered against the next kernel); it is bound by the peers' xGMI links "
                                     "into the root, not by the kernel",
                             "shard_only_value": round(total_points / shard_only / 1e6, 1),
                             "shard_only_ms_per_step": round(shard_only * 1e3 / args.steps, 5)}

        # ---- the other legs: each a function that returns its object, run under the guard (benchlegs/common.py: run_leg) ----------
        from benchlegs import legs_dense as LD
        extra = world == 1 and args.mode == "dense" and not args.no_extra_legs
        if extra:
            from benchlegs.legs_single import single_stream
            run_leg(out, "single_stream", single_stream, g)
            run_leg(out, "compaction", LD.compaction, g)
            run_leg(out, "batched_dense", LD.batched_dense, g)
            run_leg(out, "pack_twin", LD.pack_twin, g)
            run_leg(out, "centre_transform", LD.centre_transform, g)
            if not args.no_config5:
                from benchlegs.legs_config5 import config5_one_gpu
                run_leg(out, "config5_one_gpu", config5_one_gpu, g)
            if not args.no_cache_leg:
                run_leg(out, "infinity_cache_resident_inputs", LD.infinity_cache_resident_inputs, g)
                run_leg(out, "two_stream_overlap", LD.two_stream_overlap, g)
            if not args.no_general_rotation:
                run_leg(out, "general_rotation", LD.general_rotation, g)
                run_leg(out, "color_1080p", LD.color_1080p, g)
        if world == 1 and args.mode in ("dense", "drop_invalid", "cutoff"):
            run_leg(out, "per_launch_ms", LD.per_launch_ms, g, into=out["roofline"])
        if world == 1 and args.mode == "dense" and not args.no_host_api:
            from benchlegs import legs_host as LH
            run_leg(out, "host_api", LH.host_api, g)
            run_leg(out, "host_api_single", LH.host_api_single, g, cpu_single)
        if extra:
            run_leg(out, "per_kernel_unprofiled_us", per_kernel_unprofiled_us, out, kern_ms, S)
        if world == 1 and not args.no_cpu_baseline:
            def cpu_leg():
                if cpu_first is None:
                    raise RuntimeError(cpu_first_error or "cpu baseline did not run")
                return cpu_first
            if run_leg(out, "cpu_baseline", cpu_leg) is not None and args.mode == "dense":
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        emit(out)

    g.close()
    if world > 1:
        dist.destroy_process_group()


def per_kernel_unprofiled_us(out, kern_ms, S):
    """What one launch of each short kernel takes in a back-to-back loop bracketed by ONE hipEvent pair — no per-launch events, no
    profiler (rocprofv3 --kernel-trace adds ~1.5 us to every short kernel: profiles/README.md). Assembled from the legs' figures."""
    pk = {"pcs_fused_dense_kernel (8 x 720p)": round(kern_ms * 1e3, 2)}
    if "single_stream" in out:
        pk["pcs_fused_dense_kernel (1 x 720p)"] = round(out["single_stream"]["ms_per_frame"] * 1e3, 2)
    if "pack_twin" in out:
        pk["pcs_pack_dense_kernel (1 x 720p, per-camera launches of a frame-set)"] = round(out["pack_twin"]["per_stream_launches_ms_per_frame_set"] / S * 1e3, 2)
        if "single" in out["pack_twin"]:
            pk["pcs_pack_dense_kernel (1 x 720p)"] = round(out["pack_twin"]["single"]["ms_per_cloud"] * 1e3, 2)
        pk["pcs_pack_batch_kernel (8 x 720p)"] = round(out["pack_twin"]["batched_ms_per_frame_set"] * 1e3, 2)
    if "compaction" in out:
        pk["count + scan + emit (8 x 720p, drop-invalid)"] = round(out["compaction"]["ms_per_step"] * 1e3, 2)
        if "caller_counts" in out["compaction"]:
            pk["scan + emit (caller counts)"] = round(out["compaction"]["caller_counts"]["ms_per_step"] * 1e3, 2)
    return pk


if __name__ == "__main__":
    sys.exit(main() or 0)
