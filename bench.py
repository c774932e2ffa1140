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
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_POINT = 15            # 2 B Z16 + 3 B RGB8 + 10 B packed record
PACK_BYTES_PER_POINT = 33            # a2 twin: 12 B vertex + 8 B texcoord + 3 B RGB8 + 10 B record
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20     # MI355X memory-side cache: a ring whose inputs fit it is not an HBM measurement
# A collective that never completes (the RCCL paths have not met a multi-GPU box yet) must end the run, not hang it: the process
# group's watchdog gives up after this long.
import datetime
PG_TIMEOUT = datetime.timedelta(seconds=300)
POLICY = {0: "ieee", 1: "certified", 2: "certified+identityR", 3: "certified+noOverflow",
          4: "certified+identityR+noOverflow"}


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


def cpu_baseline(width, height, streams, budget_s):
    """The reference's `-m -t<N>` path restated (oracle/pcs_oracle_simd.c), timed on this host by a CHILD process
    (oracle/cpu_baseline.py) BEFORE any GPU leg: the OpenMP team is bound (OMP_PROC_BIND=close, OMP_PLACES=cores are in the
    child's environment when libgomp initialises), its buffers are first-touched by the team, no torch / HIP runtime
    threads run beside it, and `value` is the median over >= 30 passes at the best thread count (best / p10 / p90 beside
    it). Bracket A = the reference's own timed region (memset + pack, deprojection excluded, :291-293); bracket B adds the
    CPU deprojection, i.e. what the fused GPU kernel does."""
    import subprocess
    env = dict(os.environ)
    env["OMP_PROC_BIND"] = "close"
    env["OMP_PLACES"] = "cores"
    env.pop("OMP_NUM_THREADS", None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--width", str(width), "--height", str(height),
                        "--streams", str(streams), "--seconds", str(budget_s)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=max(600.0, 20 * budget_s))
    if r.returncode != 0:
        raise RuntimeError("cpu baseline child failed: " + r.stderr[-600:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


class Leg:
    """A leg of the line must never cost the line: an exception inside is recorded under `leg_errors` and swallowed."""

    def __init__(self, out, name):
        self.out, self.name = out, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            self.out.setdefault("leg_errors", {})[self.name] = f"{et.__name__}: {ev}"[:300]
            try:
                import torch
                torch.cuda.synchronize()
            except Exception:       # noqa: BLE001
                pass
            return True
        return False


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:          # noqa: BLE001
        pass
    sys.stdout.flush()


def emit(out):
    """The contract's ONE JSON line — and the LAST line on stdout: what C libraries left in the C stdio buffer (RCCL prints a
    version banner with printf when a communicator is created; into a pipe it would otherwise be flushed at exit, after this line)
    goes out first."""
    flush_c_stdio()
    print(json.dumps(out), flush=True)


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
        nv = int(svg.n_vox[0].item())
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


def run_node(args):
    """ONE process, N GPUs, libpcs_node (include/pcs_node.h) — the route `north_star` words: a C++ host over the C ABI,
    cameras sharded over the GPUs in camera order, one grouped RCCL exchange to GPU 0 per frame-set
    (src/pcs-multicamera-client.cpp:373-409's concatenation over xGMI instead of TCP). The loop is the pipelined one,
        submit(k+1); wait(k)
    so the kernels of frame-set k+1 overlap the exchange (and, for config5, the root's sort) of frame-set k.
      --workload stitch   8 x 1280x720 in total, 8/N per GPU (BASELINE configs[2] at N = 1, configs[3] at N = 8)
      --workload config5  16 x 1920x1080 in total, 16/N per GPU, invalid-depth compaction, voxel grid of the stitched cloud on
                          GPU 0 through voxel partials (BASELINE configs[4] at N = 8)
    Strong scaling: the work is fixed, the GPUs share it. Input rings are cold (per GPU, a slot is re-read after more than
    2 x 256 MiB of other rasters). Before timing, the root's result for ring slots 0 and 1 is compared with the CPU oracle,
    every stream of it (config5: the committed oracle digest)."""
    import hashlib
    import torch
    from pointcloud_stitching_amd import synthetic as Syn
    from pointcloud_stitching_amd import node as N
    from pointcloud_stitching_amd.api import PcsError
    from pointcloud_stitching_amd.types import POINT_SHORTS, FLAG_DROP_INVALID, FLAG_CUTOFF

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libpcs_hip has no CPU fallback")
    config5 = args.workload == "config5"
    devices = [int(x) for x in args.node_devices.split(",")] if args.node_devices else list(range(args.gpus))
    P = len(devices)
    if args.node_devices and P != args.gpus:
        raise SystemExit(f"--node-devices names {P} peers but --gpus is {args.gpus}")
    avail = torch.cuda.device_count()
    note = None
    if max(devices) >= avail:
        # fewer GPUs than asked for: never a reason to print no line — fold the peers onto the GPUs that exist, and say so
        note = f"{args.gpus} GPUs requested, {avail} visible: peers folded onto the visible GPUs (virtual peers)"
        devices = [d % avail for d in devices]
    defaults = (args.streams, args.width, args.height) == (8, 1280, 720)
    total_streams, W, H = (16, 1920, 1080) if (config5 and defaults) else (args.streams, args.width, args.height)
    if total_streams % P:
        # never an exit without a line: use the largest peer count <= P that divides the streams, on the first GPUs, and say so
        P2 = max(q for q in range(1, P + 1) if total_streams % q == 0)
        note = ((note + "; ") if note else "") + (f"{total_streams} streams do not divide over {P} peers: folded to {P2} peers "
                                                   f"({total_streams // P2} cameras each) on the first {P2} device entries")
        devices, P = devices[:P2], P2
    S, npts = total_streams // P, W * H
    virtual = len(set(devices)) < P
    LEAF = args.leaf
    flags = FLAG_DROP_INVALID if config5 else {"drop_invalid": FLAG_DROP_INVALID, "cutoff": FLAG_CUTOFF}.get(args.mode, 0)
    cfgs = [Syn.synth_stream_config(W, H, g) for g in range(total_streams)]

    out = {}
    node, node_error = None, None
    try:
        node = N.PcsNode(cfgs, devices=devices, flags=flags, node_flags=N.DIRECT_STORE if args.node_direct_child else 0)
    except PcsError as e:
        if args.node_direct_child:
            raise
        # RCCL would not come up: measure what the kernels alone sustain, say so, and still print a line
        node_error = f"{type(e).__name__}: {e}"[:300]
        node = N.PcsNode(cfgs, devices=devices, flags=flags, node_flags=N.NO_EXCHANGE)
    lib = node._lib
    cur = [node]                     # the node the loops below drive (the direct-store leg swaps in a second one)
    VP = C.c_void_p

    def check(rc):
        if rc:
            raise RuntimeError((lib.pcs_node_last_error(cur[0]._h) or b"").decode())

    # ---- rings of input rasters, each on its owning GPU ------------------------------------------------------------------------
    in_bytes_gpu = S * npts * 5
    R = max(args.ring, 2) if args.ring else max(4, -(-2 * INFINITY_CACHE_BYTES // in_bytes_gpu) + 2)
    DISTINCT = 2
    host = [([Syn.synth_depth(W, H, g, seed=Syn.SEED + 7919 * k) for g in range(total_streams)],
             [Syn.synth_color(W, H, g, seed=Syn.SEED + 7919 * k) for g in range(total_streams)]) for k in range(DISTINCT)]
    if config5:
        host[1] = host[0]            # the digest is of frame 0: every slot holds it (distinct ADDRESSES are what keeps the ring cold)
    ring = []                        # ring[slot] = (ctypes depth pointers, ctypes colour pointers), keeps: the tensors
    keep = []
    first = [None] * DISTINCT
    for slot in range(R):
        dps, cps = [], []
        src = slot % DISTINCT
        tens = []
        for g in range(total_streams):
            dev = torch.device("cuda", devices[g // S])
            if slot < DISTINCT:
                d = torch.from_numpy(host[src][0][g].reshape(-1).view(np.uint8)).to(dev)
                c = torch.from_numpy(host[src][1][g]).to(dev)
            else:
                d, c = first[src][g][0].clone(), first[src][g][1].clone()
            tens.append((d, c)); dps.append(d.data_ptr()); cps.append(c.data_ptr())
        if slot < DISTINCT:
            first[slot] = tens
        keep.append(tens)
        ring.append(((VP * total_streams)(*dps), (VP * total_streams)(*cps)))
    root_dev = torch.device("cuda", devices[0])
    cap = node.max_payload_shorts
    outs = [torch.empty(cap + 32, dtype=torch.int16, device=root_dev) for _ in range(2)]
    counter = [0]
    tick = C.c_int(-1)
    cnt_arr = (C.c_int * total_streams)()
    tot = C.c_int(0)

    def submit():
        k = counter[0]; counter[0] = k + 1
        dp, cp = ring[k % R]
        if config5:
            check(lib.pcs_node_submit_voxel_device(cur[0]._h, dp, cp, LEAF, VP(outs[k & 1].data_ptr()), cap, C.byref(tick)))
        else:
            check(lib.pcs_node_submit_device(cur[0]._h, dp, cp, VP(outs[k & 1].data_ptr()), cap, C.byref(tick)))
        return tick.value

    def wait(t):
        if config5:
            check(lib.pcs_node_wait_voxel(cur[0]._h, t, C.byref(tot)))
        else:
            check(lib.pcs_node_wait(cur[0]._h, t, cnt_arr, C.byref(tot)))
        return tot.value

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    # ---- correctness before timing: slots 0 and 1 through the pipelined pair, every stream, against the oracle -------------------
    from oracle import pcs_oracle as O
    t_a = submit(); t_b = submit()
    n_a = wait(t_a); got_a = outs[0][:n_a * POINT_SHORTS].cpu().numpy()
    n_b = wait(t_b); got_b = outs[1][:n_b * POINT_SHORTS].cpu().numpy()
    checked = {"slots": [0, 1], "streams": total_streams, "exchange": node_error is None}
    if config5:
        dig = [hashlib.sha256(g.tobytes()).hexdigest() for g in (got_a, got_b)]
        checked.update({"voxels": n_a, "voxel_sha256": dig[0], "golden": None})
        gpath = os.path.join(ROOT, "tests", "golden", "config5_digests.json")
        gold = json.load(open(gpath))["voxel"].get(str(LEAF)) if ((total_streams, W, H) == (16, 1920, 1080) and os.path.exists(gpath)) else None
        if gold and node_error is None:
            checked["golden"] = bool(gold["voxels"] == n_a == n_b and gold["sha256"] == dig[0] == dig[1])
            if not checked["golden"]:
                raise SystemExit(f"bench aborted: the node's voxel cloud differs from the oracle digest (leaf {LEAF} mm)")
        elif node_error is None:
            want, _ = O.process_frames(cfgs, host[0][0], host[0][1], flags, 1)
            wv = O.voxel_grid(want, LEAF)
            if n_a != wv.shape[0] or (got_a.reshape(-1, 5) != wv).any():
                raise SystemExit("bench aborted: the node's voxel cloud differs from the oracle")
    else:
        for slot, (n_got, got) in enumerate(((n_a, got_a), (n_b, got_b))):
            want, _ = O.process_frames(cfgs, host[slot][0], host[slot][1], flags, 1)
            if node_error is not None:                         # nothing was gathered: only the root's own slice is there
                own = sum(cnt_arr[:S]); want, got, n_got = want[:own], got[:own * POINT_SHORTS], own
            if n_got != want.shape[0] or (got.reshape(-1, 5) != want).any():
                raise SystemExit(f"bench aborted: the stitched cloud of ring slot {slot} differs from the oracle")

    # ---- settle clocks, warm up, time EXACTLY `steps` frame-sets: submit(k+1); wait(k) ----------------------------------------------
    def run(k_steps):
        t = submit()
        for _ in range(k_steps - 1):
            t2 = submit(); wait(t); t = t2
        wait(t)

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        run(20)
    if args.warmup:
        run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0

    # ---- where a frame-set's time goes (HIP events on the root GPU; a separate loop: the events cost host time) ----------------------
    node.set_timing(True)
    ph = {"kernel": [], "exchange": [], "root": [], "submit_host": [], "exchange_host": []}
    xbytes = reduced = 0
    n_ph = 30
    t = submit()
    for _ in range(n_ph):
        t2 = submit(); wait(t); t = t2
        st = node.last_stats()
        ph["kernel"].append(st["kernels_ms"]); ph["exchange"].append(st["exchange_ms"]); ph["root"].append(st["root_ms"])
        ph["submit_host"].append(st["submit_host_ms"]); ph["exchange_host"].append(st["exchange_host_ms"])
        xbytes, reduced = st["exchanged_bytes"] + st["direct_bytes"], st["reduced"]
        counts = [int(x) for x in cnt_arr] if not config5 else None       # of the same frame-set as xbytes
    wait(t)
    node.set_timing(False)
    kern_ms = float(np.median(ph["kernel"]))
    # ---- what answered and what connects the GPUs: the first multi-GPU record must explain itself --------------------------------
    rccl = {"runtime_version": node.rccl_version, "header_version": node.rccl_header_version, "library": node.rccl_library}
    links, link_error = [], None
    try:
        probe_bytes = 0 if not config5 else 8 << 20
        per_peer_ms = node.probe_links(probe_bytes, 5) if (P > 1 and node_error is None) else [0.0] * P
        pb = (S * npts * 10) if not config5 else probe_bytes
        for r in range(P):
            ln = node.link_info(r)
            ln["probe_ms"] = round(per_peer_ms[r], 5)
            ln["probe_GBps"] = round(pb / (per_peer_ms[r] * 1e-3) / 1e9, 1) if per_peer_ms[r] > 0 else None
            links.append(ln)
    except Exception as e:      # noqa: BLE001
        link_error = f"{type(e).__name__}: {e}"[:300]

    pts_step = total_streams * npts
    ms_per_step = elapsed * 1e3 / args.steps
    kept_root = float(np.mean([(d != 0).mean() for d in host[0][0][:S]]))
    if config5:
        bytes_root = S * npts * 5             # + 40 B per partial (not known per peer here): a lower bound, stated
        kern_name = "pcs_fused_voxel_partials_kernel"
        bpp_note = "root GPU's pre-aggregation launch: 5 B per pixel in (+ 40 B per partial out, not counted): VALU / LDS bound, not HBM bound"
    elif flags:
        bytes_root = S * npts * (5 + 10 * kept_root)
        kern_name = "pcs_fused_count_kernel + pcs_scan_kernel + pcs_fused_emit_kernel"
        bpp_note = "root GPU's launches for its own cameras: (5 + 10 rho) B per pixel"
    else:
        bytes_root = S * npts * ALGO_BYTES_PER_POINT
        kern_name = "pcs_fused_dense_kernel"
        bpp_note = "root GPU's launch for its own cameras: 15 B per point"
    ach = bytes_root / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    where = "one GPU" if P == 1 else f"{P} GPUs, {S} per GPU"
    if config5:
        cfg_name = "BASELINE.json configs[4]" if (total_streams, W, H, S) == (16, 1920, 1080, 2) else f"configs[4]'s pipeline, {S} cameras per GPU"
        metric = "Mpoints/s in (16x1920x1080 streams: deproject+transform+RGB+pack, invalid-depth compaction, voxel grid of the stitched cloud)"
        workload = (f"{cfg_name}: {total_streams} synthetic {W}x{H} Z16+RGB8 streams on {where}, PCS_FLAG_DROP_INVALID, voxel-grid downsample "
                    f"(leaf {LEAF} mm) of the stitched cloud on GPU 0: per-GPU voxel partials, one grouped exchange, one sort + segmented mean")
    else:
        cfg_name = ("BASELINE.json configs[2]" if P == 1 and total_streams == 8 else
                    "BASELINE.json configs[3]" if (S == 1 and total_streams == 8) else f"{total_streams} streams sharded {S}/GPU")
        metric = "Mpoints/s stitched (8x1280x720 streams: deproject+transform+RGB+pack)"
        workload = (f"{total_streams} synthetic {W}x{H} Z16+RGB8 streams on {where}, batched fused kernel, one extrinsic per stream ({cfg_name})"
                    + (", payloads gathered to GPU 0 in camera order" if P > 1 and node_error is None else ""))
    out.update({
        "metric": metric, "value": round(pts_step * args.steps / elapsed / 1e6, 1), "unit": "Mpoints/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "route": "node: one process, libpcs_node (C++ host over the C ABI), pcs_node_submit*/pcs_node_wait* pipelined",
                   "arithmetic": "f32 deprojection + affine (bit-exact vs the -m path), u16 depth in, u8 colour in, int16 records out",
                   "streams_total": total_streams, "streams_per_gpu": S, "width": W, "height": H,
                   "points_per_step": pts_step, "ring_frame_sets": R,
                   "ring_inputs_between_rereads_mbytes_per_gpu": round((R - 1) * in_bytes_gpu / 1e6, 1),
                   "ring_cold": bool((R - 1) * in_bytes_gpu >= 2 * INFINITY_CACHE_BYTES),
                   "gather_to_rank0": bool(P > 1 and node_error is None), "devices": devices,
                   "parallelism": f"streams sharded {S}/GPU x {P}"},
        "rccl_ranks": node.rccl_ranks, "direct_store_gather": bool(args.node_direct_child),
        "rccl": rccl,
        "links": {"per_peer": links, "probe": "pcs_node_probe_links: one ncclSend/ncclRecv pair at a time of one peer's payload "
                                               "(config5: 8 MiB) into GPU 0, event pair on GPU 0's communication stream, mean of 5; link_type "
                                               "per hipExtGetLinkTypeAndHopCount (4 = xGMI)", "error": link_error},
        "host_enqueue_ms": {"submit": round(float(np.median(ph["submit_host"])), 5), "exchange": round(float(np.median(ph["exchange_host"])), 5),
                            "note": "host time of ONE thread per frame-set: submit = every peer's kernels enqueued; exchange = the grouped "
                                    "ncclSend/ncclRecv (config5: + the root's sort + mean) enqueued"},
        "check": checked,
        "phases_ms": {"kernel": round(kern_ms, 5), "exchange": round(float(np.median(ph["exchange"])), 5),
                      "root": round(float(np.median(ph["root"])), 5),
                      "note": "medians of HIP-event brackets on GPU 0 over a separate loop of the same pipelined steps: kernel = the root's own "
                              "launch(es); exchange = group enqueued (every peer's kernels done) -> every payload landed; root = config5's sort + "
                              "segmented mean. They overlap across frame-sets: their sum is not ms_per_step"},
        "bytes_into_root_per_step": int(xbytes),
        "root_ingest_GBps": round(xbytes / (ms_per_step * 1e-3) / 1e9, 1),
        "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": kern_name, "avg_launch_ms": round(kern_ms, 5),
                     "algorithmic_bytes_per_launch": int(bytes_root), "note": bpp_note,
                     "timing": "hipEvent pair on GPU 0's kernel stream around its own launch(es), median over pipelined steps"},
    })
    if config5:
        out["partials_reduced_per_step"] = int(reduced)
        out["config"]["leaf_mm"] = LEAF
        if P == 1 and os.environ.get("PCS_NODE_ONE_CALL", "1") != "0":
            # one peer: submit enqueued the rasters -> voxels call (no partials leave the library). Beside it, the same loop with the
            # partials pipeline a node of several peers runs (pre-aggregation of k+1 beside the root's sort + mean of k)
            node.set_timing(False)
            node.set_one_call(False)
            try:
                run(max(args.warmup, 4)); sync_all()
                t1 = time.perf_counter(); run(args.steps); sync_all()
                out["one_peer"] = {"route": "pcs_process_frames_voxel_device enqueued at submit (warm bucket tail: 2 launches per frame-set)",
                                   "partials_pipeline_ms_per_step": round((time.perf_counter() - t1) * 1e3 / args.steps, 5),
                                   "note": "partials_pipeline = PCS_NODE_ONE_CALL=0: partials to caller-held arrays, sort + mean on a second "
                                           "context beside the next frame-set's pre-aggregation (what a node of several peers runs on its root)"}
            finally:
                node.set_one_call(True)
    else:
        out["per_stream_fps"] = round(args.steps / elapsed, 1)
        out["points_per_stream"] = counts
    if P > 1:
        out["scaling_note"] = ("strong scaling with a gather: every peer's packed cloud crosses ONE xGMI link into GPU 0 each step, so the step "
                               "is bound by bytes_into_root_per_step over the links (and by one host thread enqueueing for N GPUs), not by the "
                               "kernels; see DESIGN.md §9")
    if P > 1 and not config5 and flags == 0 and node_error is None and not args.node_direct_child:
        # the same frame loop with the gather done by the pack kernels' own stores into GPU 0's stitched buffer over xGMI
        # (PCS_NODE_DIRECT_STORE: no exchange step, no RCCL kernel) — reported beside the RCCL figure, never instead of it. In a
        # process of its own: peer-to-peer stores have never met a multi-GPU box, and a fault there must not cost the line.
        import subprocess
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--route", "node", "--node-direct-child", "--gpus", str(args.gpus),
                   "--steps", str(args.steps), "--warmup", str(args.warmup), "--preheat-ms", str(min(args.preheat_ms, 200.0)),
                   "--streams", str(args.streams), "--width", str(W), "--height", str(H)]
            if args.node_devices:
                cmd += ["--node-devices", args.node_devices]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"child exited {r.returncode}: {(r.stderr or r.stdout)[-200:]}")
            dch = json.loads(lines[-1])
            out["direct_store"] = {"ms_per_step": dch["ms_per_step"], "value": dch["value"], "checked_against_oracle": dch["check"],
                                   "phases_ms": {k: dch["phases_ms"][k] for k in ("kernel", "exchange")},
                                   "note": "PCS_NODE_DIRECT_STORE (its own process): every peer's pack kernel writes its records straight into "
                                           "its camera-order slice of GPU 0's stitched buffer (peer access over xGMI); no exchange step, no "
                                           "RCCL kernel"}
        except Exception as e:          # noqa: BLE001
            out["direct_store"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if virtual:
        out["debug"] = ("virtual peers: device ids repeat, the peers of one GPU share it and their transfers are RCCL self send/recv "
                        "pairs. Exercises the N > 1 flow; says nothing about scaling")
    if note:
        out["note"] = note
    if node_error:
        out["node_error"] = node_error
        out["config"]["gather_to_rank0"] = False
    try:
        node.close()          # (RCCL teardown before the line, so that nothing follows it)
    except Exception:          # noqa: BLE001
        pass
    emit(out)
    return 0


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
        return run_node(args)
    if args.gpus > 1 and world == 1:
        return _reexec_under_torchrun(args)
    if args.workload == "config5":
        return run_config5(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from pointcloud_stitching_amd import synthetic as Syn
    from pointcloud_stitching_amd.api import PcsContext
    from pointcloud_stitching_amd.types import POINT_SHORTS, FLAG_CUTOFF, FLAG_DROP_INVALID

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libpcs_hip has no CPU fallback")
    # The CPU sample runs FIRST, in a child process, while this process has not yet created a HIP context or a stream:
    # nothing of the GPU legs (runtime helper threads, pinned-memory traffic, thermal state of the host) can move it.
    cpu_first, cpu_first_error = None, None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu_first = cpu_baseline(args.width, args.height, args.streams, args.cpu_seconds)
        except Exception as e:       # noqa: BLE001 — a leg must never cost the line
            cpu_first_error = f"{type(e).__name__}: {e}"[:300]
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

    W, H = args.width, args.height
    strong = args.scaling == "strong"
    if strong:
        if args.streams % world:
            raise SystemExit(f"--scaling strong shards {args.streams} streams over {world} GPUs: not divisible")
        S = args.streams // world            # cameras on this GPU
        total_streams = args.streams
    else:
        S = args.streams
        total_streams = args.streams * world
    in_bytes_per_set = S * (W * H * 2 + W * H * 3)
    # a slot is re-read after R-1 other sets: (R-1) * inputs > 2 x Infinity Cache
    R = max(args.ring, 2) if args.ring else max(4, -(-2 * INFINITY_CACHE_BYTES // in_bytes_per_set) + 2)
    ring_cold = (R - 1) * in_bytes_per_set >= 2 * INFINITY_CACHE_BYTES
    if not args.ring:
        assert ring_cold, "default ring must keep every re-read >= 2 x 256 MiB of input traffic apart"
    npts = W * H
    set_points = S * npts
    # global camera index = rank*S + s  -> extrinsic transform[(rank*S+s) % 8], distinct seeds per camera
    cfgs = [Syn.synth_stream_config(W, H, rank * S + s) for s in range(S)]
    mode_flags = {"drop_invalid": FLAG_DROP_INVALID, "batch_drop_invalid": FLAG_DROP_INVALID, "cutoff": FLAG_CUTOFF}.get(args.mode, 0)
    ctx = PcsContext(cfgs, device=local_rank, flags=mode_flags)
    # One explicit HIP stream for everything this rank enqueues: the library's kernels (pcs_set_stream) and torch's own
    # work — the RCCL gather orders itself against torch's CURRENT stream. (torch's default stream has the handle 0, which
    # pcs_set_stream reads as "use the context's own stream": the kernels and the gather would then be unordered.)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)

    # Ring of frame-sets resident in HBM, carved from ONE slab at 256-byte granularity (power-of-two aligned
    # per-raster allocations alias in the Infinity Cache when the inputs are resident there).
    def up(nbytes):
        return (nbytes + 16 + 255) & ~255
    payload_shorts = set_points * POINT_SHORTS
    depth_b, color_b, out_b = up(npts * 2), up(cfgs[0].color_bytes), up(payload_shorts * 2 + 256)
    slab = torch.empty(R * (S * (depth_b + color_b) + out_b) + 256, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    off = (-base) % 256
    d_depth, d_color, d_out, host0, host1 = [], [], [], None, None
    DISTINCT = 4          # frame-sets generated on the host; further ring slots are device copies of these (distinct
    for slot in range(R):  # ADDRESSES are what defeats the caches; generating 16 sets in numpy would only cost start-up time)
        if slot < DISTINCT:
            dep = [Syn.synth_depth(W, H, rank * S + s, seed=Syn.SEED + 7919 * slot) for s in range(S)]
            col = [Syn.synth_color(W, H, rank * S + s, seed=Syn.SEED + 7919 * slot) for s in range(S)]
        if slot == 0:
            host0 = (dep, col)
        if slot == 1:
            host1 = (dep, col)
        dd, dc = [], []
        for s in range(S):
            v = slab[off:off + npts * 2]
            v.copy_(torch.from_numpy(dep[s].reshape(-1).view(np.uint8)) if slot < DISTINCT else d_depth[slot % DISTINCT][s])
            dd.append(v); off += depth_b
            v = slab[off:off + cfgs[0].color_bytes]
            v.copy_(torch.from_numpy(col[s]) if slot < DISTINCT else d_color[slot % DISTINCT][s])
            dc.append(v); off += color_b
        d_depth.append(dd); d_color.append(dc)
        sk = args.payload_skew & ~1
        d_out.append(slab[off + sk:off + sk + payload_shorts * 2].view(torch.int16)); off += out_b
    ring_bytes = R * (set_points * ALGO_BYTES_PER_POINT)
    kept_frac = float(np.mean([(d != 0).mean() for d in host0[0]]))      # rho of the invalid-drop compaction

    gather = world > 1 and not args.no_gather
    variable = gather and mode_flags != 0        # compaction: per-rank counts differ -> counts all-gathered, payloads sent point to point
    d_cnt = torch.zeros(S + 1, dtype=torch.int32, device=dev)
    last_counts = [None]
    stitched = None
    if gather:
        from pointcloud_stitching_amd.stitch import RankStitcher
        st = RankStitcher()
        if rank == 0:
            stitched = [torch.empty(payload_shorts * world, dtype=torch.int16, device=dev) for _ in range(2)]

    lib = ctx._lib
    h = ctx._h
    VP = C.c_void_p

    def check(rc, handle=None):
        if rc:
            raise RuntimeError(lib.pcs_last_error(handle or h).decode())

    call_args = []
    for slot in range(R):
        dp = (VP * S)(*[t.data_ptr() for t in d_depth[slot]])
        cp = (VP * S)(*[t.data_ptr() for t in d_color[slot]])
        call_args.append((dp, cp, VP(d_out[slot].data_ptr())))

    # ---- the a2 twin's inputs (vertices 12 B + texcoords 8 B per point), one copy per ring slot -------------------
    pack_ring = None

    def build_pack_ring():
        nonlocal pack_ring
        if pack_ring is not None:
            return pack_ring
        from pointcloud_stitching_amd.types import CloudDesc
        set_b = S * (up(npts * 12) + up(npts * 8))
        # re-read distance as above, over everything the kernel reads (vertices + texcoords + colour)
        Rp = max(3, -(-2 * INFINITY_CACHE_BYTES // (set_b + S * cfgs[0].color_bytes)) + 2)
        Rp = min(Rp, R)
        vt_slab = torch.empty(Rp * set_b + 256, dtype=torch.uint8, device=dev)
        vo = (-vt_slab.data_ptr()) % 256
        per_slot = [[] for _ in range(Rp)]
        for s in range(S):
            v, t = ctx0.deproject(s, host0[0][s])
            hv = torch.from_numpy(v.reshape(-1).view(np.uint8)); ht = torch.from_numpy(t.reshape(-1).view(np.uint8))
            for slot in range(Rp):
                o = vo + slot * set_b
                dv = vt_slab[o:o + npts * 12]; dv.copy_(hv if slot == 0 else vt_slab[vo:vo + npts * 12])
                dt = vt_slab[o + up(npts * 12):o + up(npts * 12) + npts * 8]
                dt.copy_(ht if slot == 0 else vt_slab[vo + up(npts * 12):vo + up(npts * 12) + npts * 8])
                per_slot[slot].append((dv.data_ptr(), dt.data_ptr()))
            vo += up(npts * 12) + up(npts * 8)
        descs = []
        for slot in range(Rp):
            arr = (CloudDesc * S)()
            for s in range(S):
                arr[s].stream, arr[s].n_points = s, npts
                arr[s].vertices, arr[s].texcoords = per_slot[slot][s]
                arr[s].color = d_color[slot][s].data_ptr()
                arr[s].pc_buffer = d_out[slot].data_ptr() + s * npts * 10
            descs.append(arr)
        pack_ring = {"R": Rp, "slab": vt_slab, "per_slot": per_slot, "descs": descs}
        return pack_ring

    # a context without predicate flags for the legs that need the plain configuration (and for deproject)
    ctx0 = ctx if mode_flags == 0 else PcsContext(cfgs, device=local_rank)
    if ctx0 is not ctx:
        ctx0.set_stream(stream.cuda_stream)

    # ---- launch forms; ALL of them take the slot from one monotonically increasing counter ---------------------------
    counter = [0]

    def next_slot(ring=R):
        k = counter[0]
        counter[0] = k + 1
        return k % ring

    def launch_dense(handle=None):
        dp, cp, out = call_args[next_slot()]
        check(lib.pcs_process_frames_device(handle or h, dp, cp, out, payload_shorts,
                                            VP(d_cnt.data_ptr()) if (variable and handle is None) else None), handle)

    def launch_pack_single():
        pr = build_pack_ring()
        slot = next_slot(pr["R"])
        for s in range(S):
            check(lib.pcs_copy_pointcloud_xyzrgb_to_buffer_device(
                ctx0._h, s, VP(pr["per_slot"][slot][s][0]), VP(pr["per_slot"][slot][s][1]), npts,
                VP(d_color[slot][s].data_ptr()), VP(d_out[slot].data_ptr() + s * npts * 10), None), ctx0._h)

    def launch_pack_batch():
        pr = build_pack_ring()
        slot = next_slot(pr["R"])
        check(lib.pcs_copy_pointclouds_xyzrgb_to_buffer_device(ctx0._h, S, pr["descs"][slot], None), ctx0._h)

    KB = max(1, min(args.batch_sets, R // 2))
    batch_args = []
    for g in range(R // KB):
        slots = [g * KB + k for k in range(KB)]
        dp = (VP * (KB * S))(*[t.data_ptr() for sl in slots for t in d_depth[sl]])
        cp = (VP * (KB * S))(*[t.data_ptr() for sl in slots for t in d_color[sl]])
        pp = (VP * KB)(*[d_out[sl].data_ptr() for sl in slots])
        batch_args.append((dp, cp, pp))

    def launch_batch(c=None):
        c = c or (ctx if args.mode == "batch_drop_invalid" else ctx0)
        dp, cp, pp = batch_args[next_slot(len(batch_args))]
        check(lib.pcs_process_frames_device_batch(c._h, KB, dp, cp, pp, payload_shorts, None), c._h)

    launch = {"dense": launch_dense, "drop_invalid": launch_dense, "cutoff": launch_dense, "pack": launch_pack_single,
              "pack_batch": launch_pack_batch, "batch": launch_batch, "batch_drop_invalid": launch_batch}[args.mode]
    sets_per_launch = KB if args.mode in ("batch", "batch_drop_invalid") else 1

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
            slot = counter[0] % R
            launch()
            last_counts[0] = st.gather_variable(d_out[slot], d_cnt[S], stitched[0] if rank == 0 else None)
        elif gather:
            k = step_no[0]; step_no[0] = k + 1
            if pending[k & 1] is not None:       # the buffer pair (slot's out, stitched[k&1]) is free again
                pending[k & 1].wait()
            slot = counter[0] % R
            launch()
            pending[k & 1] = gather_async(d_out[slot], stitched[k & 1] if rank == 0 else None)
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

    def preheat(fn, ms):
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < ms:      # untimed: settle clocks
            for _ in range(50):
                fn()
            torch.cuda.synchronize(dev)

    def timed(fn, n, c=ctx):
        """n launches of fn bracketed by a hipEvent pair on the launch stream -> ms per launch."""
        c.timer_begin()
        for _ in range(n):
            fn()
        c.timer_end()
        return c.timer_elapsed_ms() / n

    # parity check against the oracle before timing (bench must not time a wrong kernel): EVERY stream of ring slots 0 and 1
    # (distinct frames), on the very buffers the timed region uses
    counter[0] = 0
    launch(); torch.cuda.synchronize(dev)
    if sets_per_launch == 1 and R > 1:
        launch(); torch.cuda.synchronize(dev)          # slot 1 (the batched modes covered it with the first launch)
    checked_slots = [0, 1] if (R > 1 and (sets_per_launch == 1 or KB >= 2)) else [0]
    from oracle import pcs_oracle as O
    for slot in checked_slots:
        hd, hc = (host0, host1)[slot]
        if args.mode in ("pack", "pack_batch"):
            hd = host0[0]                              # the a2 twin's ring holds slot 0's vertices beside every slot's colour raster
        want, _ = O.process_frames(cfgs, hd, hc, mode_flags, 1)
        got = d_out[slot][:want.size].cpu().numpy().reshape(-1, 5)
        if got.shape != want.shape or (got != want).any():
            raise SystemExit(f"bench aborted: HIP output of ring slot {slot} differs from the oracle "
                             f"({int((got != want).any(axis=1).sum())} of {want.shape[0]} records)")

    preheat(launch, args.preheat_ms)
    gather_error = None
    if gather:
        # The exchange cannot be exercised on the single-GPU development boxes; if RCCL refuses it here the run
        # degrades to shard-only (and says so) instead of producing no line at all.
        try:
            first_slot = counter[0] % R
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
                first_slot = (counter[0] - 1) % R        # the slot of the most recent step
            if not torch.equal(own, d_out[first_slot][:own_n]):
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
        long_ms = timed(launch, n_long)
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
                       "ring_frame_sets": R, "ring_mbytes": round(ring_bytes / 1e6, 1),
                       "ring_inputs_between_rereads_mbytes": round((R - 1) * in_bytes_per_set / 1e6, 1),
                       "ring_cold": bool(ring_cold),
                       "gather_to_rank0": bool(gather), "parallelism": f"streams sharded {S}/GPU x {world}"},
            "check": {"oracle_compared": {"slots": checked_slots, "streams": S, "records": "all"}},
            "per_stream_fps": round(args.steps * sets_per_launch / elapsed, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src if traffic is not None else None,
                         "kernel": {"dense": "pcs_fused_dense_kernel", "batch": "pcs_fused_dense_batch_kernel",
                                    "pack": "pcs_pack_dense_kernel", "pack_batch": "pcs_pack_batch_kernel",
                                    "drop_invalid": "pcs_fused_count_kernel + pcs_scan_kernel + pcs_fused_emit_kernel (PCS_COMPACT_PATH=single: pcs_fused_compact_kernel)",
                                    "batch_drop_invalid": "pcs_fused_count_batch_kernel + pcs_scan_batch_kernel + pcs_fused_emit_batch_kernel",
                                    "cutoff": "pcs_fused_count_kernel + pcs_scan_kernel + pcs_fused_emit_kernel (PCS_COMPACT_PATH=single: pcs_fused_compact_kernel)"}[args.mode],
                         "arithmetic": policy, "avg_launch_ms": round(kern_ms, 5),
                         "long_sample_ms": round(long_sample[0], 5) if long_sample else None,
                         "long_sample_launches": long_sample[1] if long_sample else None,
                         "long_sample_frac": round(algo_bytes / (long_sample[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if long_sample else None,
                         "algorithmic_bytes_per_launch": round(algo_bytes),
                         "algorithmic_bytes_per_point": round(bytes_pp, 3),
                         "timing": roofline_timing},
        }
        if debug_gloo:
            out["debug"] = "gloo control-flow test: all ranks on one GPU, host-staged gathers; numbers are meaningless"
        if gather_error:
            out["gather_error"] = gather_error
            out["config"]["gather_to_rank0"] = False
        if shard_only is not None:
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

        extra = world == 1 and args.mode == "dense" and not args.no_extra_legs
        n_leg = max(200, min(args.steps, 600))
        if extra:
            with Leg(out, "compaction"):
                # ---- ordered compaction (invalid-depth drop, ~10 % of the synthetic pixels), cold ring --------------------
                ctx_c = PcsContext(cfgs, device=local_rank, flags=FLAG_DROP_INVALID)
                ctx_c.set_stream(stream.cuda_stream)
                d_cnt = torch.zeros(S + 1, dtype=torch.int32, device=dev)

                def launch_c(cnt=None):
                    dp, cp, outp = call_args[next_slot()]
                    check(lib.pcs_process_frames_device(ctx_c._h, dp, cp, outp, payload_shorts, cnt), ctx_c._h)
                launch_c(VP(d_cnt.data_ptr())); ctx_c.synchronize()
                kept = int(d_cnt[S].item())
                for _ in range(100):
                    launch_c()
                torch.cuda.synchronize(dev)
                ms_c = timed(launch_c, n_leg, ctx_c)
                rho = kept / set_points
                ach_c = set_points * (5 + 10 * rho) / (ms_c * 1e-3) / 1e9
                out["compaction"] = {"ms_per_step": round(ms_c, 5), "value": round(set_points / ms_c / 1e3, 1), "unit": "Mpoints/s in",
                                     "kept_fraction": round(rho, 4), "algorithmic_bytes_per_point": round(5 + 10 * rho, 3),
                                     "achieved": round(ach_c, 1), "frac": round(ach_c / HBM_PEAK_GBS, 4),
                                     "path": os.environ.get("PCS_COMPACT_PATH", "three (default: count + scan + emit)"),
                                     "note": "PCS_FLAG_DROP_INVALID, order-preserving (= the reference's -c -m -t1 order), same cold ring; "
                                             "bytes = 2 (Z16) + 3 (RGB8) + 10*rho (records)"}
                with Leg(out["compaction"], "caller_counts"):
                    # a producer that counts as it writes the depth image hands the per-tile kept counts over
                    # (pcs_process_frames_device_counted): scan + emit only, the Z16 rasters are read once
                    tcs = []
                    for slot in range(R):
                        tcs.append(torch.cat([d.view(torch.int16).ne(0).view(-1, 2048).sum(1, dtype=torch.int32) for d in d_depth[slot]]))

                    def launch_cc():
                        slot = next_slot()
                        dp, cp, outp = call_args[slot]
                        check(lib.pcs_process_frames_device_counted(ctx_c._h, dp, cp, VP(tcs[slot].data_ptr()), outp, payload_shorts, None), ctx_c._h)
                    if npts % 2048 == 0:
                        for _ in range(50):
                            launch_cc()
                        torch.cuda.synchronize(dev)
                        ms_cc = timed(launch_cc, n_leg, ctx_c)
                        ach_cc = set_points * (5 + 10 * rho) / (ms_cc * 1e-3) / 1e9
                        out["compaction"]["caller_counts"] = {
                            "ms_per_step": round(ms_cc, 5), "achieved": round(ach_cc, 1), "frac": round(ach_cc / HBM_PEAK_GBS, 4),
                            "note": "pcs_process_frames_device_counted: per-tile kept counts handed in by the producer of the depth "
                                    "image (here: computed beforehand, outside the timed region), scan + emit only"}
                if KB >= 2:
                    # K frame-sets per call: three launches (count, scan, emit) for all K sets, nothing order-dependent
                    def launch_cb():
                        dp, cp, pp = batch_args[next_slot(len(batch_args))]
                        check(lib.pcs_process_frames_device_batch(ctx_c._h, KB, dp, cp, pp, payload_shorts, None), ctx_c._h)
                    for _ in range(30):
                        launch_cb()
                    torch.cuda.synchronize(dev)
                    ms_cb = timed(launch_cb, max(50, n_leg // KB), ctx_c) / KB
                    ach_cb = set_points * (5 + 10 * rho) / (ms_cb * 1e-3) / 1e9
                    out["compaction"]["batched"] = {"frame_sets_per_call": KB, "ms_per_frame_set": round(ms_cb, 5),
                                                    "achieved": round(ach_cb, 1), "frac": round(ach_cb / HBM_PEAK_GBS, 4),
                                                    "note": "pcs_process_frames_device_batch with the predicate: count, scan and emit "
                                                            "launches shared by K frame-sets (throughput form)"}
                ctx_c.close()
                if "PCS_COMPACT_PATH" not in os.environ:
                    # the opt-in one-launch kernel (its forward progress assumes in-order workgroup dispatch; bounded waits and a
                    # three-pass re-run catch a violation — which is why it is not the default)
                    os.environ["PCS_COMPACT_PATH"] = "single"
                    try:
                        ctx_s = PcsContext(cfgs, device=local_rank, flags=FLAG_DROP_INVALID)
                    finally:
                        del os.environ["PCS_COMPACT_PATH"]
                    ctx_s.set_stream(stream.cuda_stream)

                    def launch_s():
                        dp, cp, outp = call_args[next_slot()]
                        check(lib.pcs_process_frames_device(ctx_s._h, dp, cp, outp, payload_shorts, None), ctx_s._h)
                    for _ in range(100):
                        launch_s()
                    torch.cuda.synchronize(dev)
                    ms_s = timed(launch_s, n_leg, ctx_s)
                    ctx_s.synchronize()          # raises if a placement wait ever expired
                    ach_s = set_points * (5 + 10 * rho) / (ms_s * 1e-3) / 1e9
                    out["compaction"]["single_pass_opt_in"] = {"ms_per_step": round(ms_s, 5), "achieved": round(ach_s, 1),
                                                               "frac": round(ach_s / HBM_PEAK_GBS, 4),
                                                               "note": "PCS_COMPACT_PATH=single: one launch, Z16 read once"}
                    ctx_s.close()
            with Leg(out, "batched_dense"):
                # ---- K frame-sets per launch (throughput form of the dense path) --------------------------------------------
                if KB >= 2:
                    for _ in range(30):
                        launch_batch()
                    torch.cuda.synchronize(dev)
                    ms_b = timed(launch_batch, max(50, n_leg // KB), ctx0) / KB
                    ach_b = set_points * ALGO_BYTES_PER_POINT / (ms_b * 1e-3) / 1e9
                    out["batched_dense"] = {"frame_sets_per_launch": KB, "ms_per_frame_set": round(ms_b, 5),
                                            "value": round(set_points / ms_b / 1e3, 1), "achieved": round(ach_b, 1),
                                            "frac": round(ach_b / HBM_PEAK_GBS, 4),
                                            "note": "pcs_process_frames_device_batch: the same tiles, K frame-sets share one launch's "
                                                    "fill and drain; a throughput figure (latency of a frame-set = the whole launch), "
                                                    "NOT the headline value"}
            with Leg(out, "pack_twin"):
                # ---- the a2 twin, one launch per camera vs all cameras in one launch -----------------------------------------
                for _ in range(20):
                    launch_pack_batch()
                torch.cuda.synchronize(dev)
                ms_pb = timed(launch_pack_batch, max(50, n_leg // 2), ctx0)
                for _ in range(10):
                    launch_pack_single()
                torch.cuda.synchronize(dev)
                ms_ps = timed(launch_pack_single, max(30, n_leg // 4), ctx0)
                out["pack_twin"] = {"batched_ms_per_frame_set": round(ms_pb, 5),
                                    "batched_achieved": round(set_points * PACK_BYTES_PER_POINT / (ms_pb * 1e-3) / 1e9, 1),
                                    "batched_frac": round(set_points * PACK_BYTES_PER_POINT / (ms_pb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "per_stream_launches_ms_per_frame_set": round(ms_ps, 5),
                                    "per_stream_launches_frac": round(set_points * PACK_BYTES_PER_POINT / (ms_ps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_point": PACK_BYTES_PER_POINT, "ring_frame_sets": pack_ring["R"],
                                    "note": "copyPointCloudXYZRGBToBufferSIMD's twin on device-resident rs2::points arrays "
                                            "(12 B vertex + 8 B texcoord + 3 B RGB in, 10 B out): pcs_copy_pointclouds_xyzrgb_to_buffer_device "
                                            "(one launch for all cameras) vs pcs_copy_pointcloud_xyzrgb_to_buffer_device per camera"}
        if extra:
            with Leg(out, "centre_transform"):
                # ---- what pcs-multicamera-optimized does to packed payloads on the centre (src/pcs-multicamera-optimized.cpp:226-265,
                # 289): decode, transform[i], re-encode, concatenate — one launch for all cameras, 10 B in + 10 B out per record.
                # Inputs: the payload slices of the ring's frame-sets (device-resident, cold), output: a ring of stitched buffers.
                from pointcloud_stitching_amd.types import TRANSFORMS as _TR
                xo = [torch.empty(payload_shorts + 64, dtype=torch.int16, device=dev) for _ in range(min(R, 8))]
                mats = [_TR[s % 8] for s in range(S)]
                xk = [0]

                def launch_xform():
                    k = xk[0]; xk[0] = k + 1
                    src = d_out[k % R].data_ptr()
                    ctx0.transform_payloads_device([src + s * npts * 10 for s in range(S)], [npts] * S, mats, 1,
                                                   xo[k % len(xo)].data_ptr(), payload_shorts)
                launch_xform(); torch.cuda.synchronize(dev)
                from oracle import pcs_oracle as _O
                got_x = xo[0][:payload_shorts].cpu().numpy().reshape(-1, 5)
                src0 = d_out[0][:payload_shorts].cpu().numpy().reshape(-1, 5)
                want_x = _O.transform_payload(src0[:npts], mats[0], 1)
                if (got_x[:npts] != want_x).any():
                    raise RuntimeError("centre transform differs from the oracle")
                for _ in range(20):
                    launch_xform()
                torch.cuda.synchronize(dev)
                ms_x = timed(launch_xform, max(50, n_leg // 2), ctx0)
                ach_x = set_points * 20 / (ms_x * 1e-3) / 1e9
                out["centre_transform"] = {"ms_per_frame_set": round(ms_x, 5), "achieved": round(ach_x, 1), "frac": round(ach_x / HBM_PEAK_GBS, 4),
                                           "algorithmic_bytes_per_point": 20, "kernel": "pcs_transform_payload_kernel",
                                           "note": "pcs_transform_payloads_device: the centre-side decode / pcl::transformPointCloud / re-encode of "
                                                   "pcs-multicamera-optimized over 8 packed 1280x720 payloads in one launch, camera-order "
                                                   "concatenation fused (CLI: -c ... -T <file>); camera 0 compared with the oracle before timing"}
        if extra and not args.no_config5:
            with Leg(out, "config5_one_gpu"):
                # ---- BASELINE configs[4] on ONE GPU: 16 x 1920x1080 -> invalid-depth compaction -> camera-order stitch -> voxel
                # grid of the stitched cloud, device-resident and asynchronous (the voxel grid reads the kept total from the
                # device); 4 input sets (664 MB) so that the rasters come from HBM
                W5, H5, S5, LEAF = 1920, 1080, 16, 50
                cfg5 = [Syn.synth_stream_config(W5, H5, s) for s in range(S5)]
                ctx5 = PcsContext(cfg5, device=local_rank, flags=FLAG_DROP_INVALID)
                ctx5.set_stream(stream.cuda_stream)
                n5 = W5 * H5
                dep5 = [torch.from_numpy(Syn.synth_depth(W5, H5, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S5)]
                col5 = [torch.from_numpy(Syn.synth_color(W5, H5, s)).to(dev) for s in range(S5)]
                sets5 = [(dep5, col5)] + [([d.clone() for d in dep5], [c.clone() for c in col5]) for _ in range(3)]
                pay5 = torch.empty(S5 * n5 * POINT_SHORTS, dtype=torch.int16, device=dev)
                vox5 = torch.empty(S5 * n5 * POINT_SHORTS, dtype=torch.int16, device=dev)
                cnt5 = torch.zeros(S5 + 1, dtype=torch.int32, device=dev)
                nv5 = torch.zeros(1, dtype=torch.int32, device=dev)
                args5 = [((VP * S5)(*[t.data_ptr() for t in d]), (VP * S5)(*[t.data_ptr() for t in c])) for d, c in sets5]
                k5 = [0]

                def compact5():
                    dp, cp = args5[k5[0] % 4]; k5[0] += 1
                    check(lib.pcs_process_frames_device(ctx5._h, dp, cp, VP(pay5.data_ptr()), pay5.numel(), VP(cnt5.data_ptr())), ctx5._h)

                def voxel5():
                    check(lib.pcs_voxel_grid_device_counted(ctx5._h, VP(pay5.data_ptr()), VP(cnt5.data_ptr() + 4 * S5), S5 * n5, LEAF,
                                                            VP(vox5.data_ptr()), vox5.numel(), VP(nv5.data_ptr())), ctx5._h)

                def both5():
                    compact5(); voxel5()
                for _ in range(3):
                    both5()
                torch.cuda.synchronize(dev)
                ms_c5 = timed(compact5, 30, ctx5)
                ms_v5 = timed(voxel5, 30, ctx5)
                ms_b5 = timed(both5, 30, ctx5)
                kept5, nvox5 = int(cnt5[S5].item()), int(nv5.item())

                def onecall5():       # rasters -> voxels, the stitched cloud never written
                    dp, cp = args5[k5[0] % 4]; k5[0] += 1
                    check(lib.pcs_process_frames_voxel_device(ctx5._h, dp, cp, LEAF, VP(vox5.data_ptr()), vox5.numel(), VP(nv5.data_ptr())), ctx5._h)
                for _ in range(3):
                    onecall5()
                torch.cuda.synchronize(dev)
                ms_o5 = timed(onecall5, 30, ctx5)
                nvox5_one = int(nv5.item())
                # the voxel cloud of the timed loop against the committed oracle digest (this IS the digest's workload)
                import hashlib
                dig5 = hashlib.sha256(vox5[:nvox5_one * POINT_SHORTS].cpu().numpy().tobytes()).hexdigest()
                gold5 = json.load(open(os.path.join(ROOT, "tests", "golden", "config5_digests.json")))["voxel"].get(str(LEAF))
                if gold5 and not (gold5["voxels"] == nvox5_one and gold5["sha256"] == dig5):
                    raise RuntimeError("config5 one-call voxel cloud differs from the oracle digest")
                # the same call with the LSD radix sort + segmented mean instead of the bucket tail (PCS_VOXEL_TAIL is read per call)
                tail_default = os.environ.get("PCS_VOXEL_TAIL")
                os.environ["PCS_VOXEL_TAIL"] = "lsd"
                try:
                    for _ in range(3):
                        onecall5()
                    torch.cuda.synchronize(dev)
                    ms_o5_lsd = timed(onecall5, 30, ctx5)
                finally:
                    if tail_default is None:
                        del os.environ["PCS_VOXEL_TAIL"]
                    else:
                        os.environ["PCS_VOXEL_TAIL"] = tail_default
                # ... and with the bucket tail held to its cold chain (every call partitions: histogram, column scan, scatter, reduce)
                regions_default = os.environ.get("PCS_VOXEL_REGIONS")
                os.environ["PCS_VOXEL_REGIONS"] = "0"
                try:
                    for _ in range(3):
                        onecall5()
                    torch.cuda.synchronize(dev)
                    ms_o5_cold = timed(onecall5, 30, ctx5)
                finally:
                    if regions_default is None:
                        del os.environ["PCS_VOXEL_REGIONS"]
                    else:
                        os.environ["PCS_VOXEL_REGIONS"] = regions_default
                bucket_default = tail_default in (None, "bucket")
                out["config5_one_gpu"] = {"workload": f"{S5} x {W5}x{H5} synthetic streams, PCS_FLAG_DROP_INVALID, voxel leaf {LEAF} mm",
                                          "points_in": S5 * n5, "points_kept": kept5, "voxels": nvox5,
                                          "compaction_ms": round(ms_c5, 4), "voxel_grid_ms": round(ms_v5, 4),
                                          "pipeline_ms_per_frame_set": round(ms_b5, 4),
                                          "value": round(S5 * n5 / ms_b5 / 1e3, 1), "unit": "Mpoints/s in",
                                          "one_call": {"ms_per_frame_set": round(ms_o5, 4), "value": round(S5 * n5 / ms_o5 / 1e3, 1),
                                                       "voxels": nvox5_one, "oracle_digest_ok": bool(gold5 is not None),
                                                       "kernels_per_call": (2 if regions_default != "0" else 5) if bucket_default else 13,
                                                       "cold_chain_ms_per_frame_set": round(ms_o5_cold, 4),
                                                       "lsd_tail_ms_per_frame_set": round(ms_o5_lsd, 4),
                                                       "algorithmic_bytes": int(5 * S5 * n5 + 10 * nvox5_one),
                                                       "frac": round((5 * S5 * n5 + 10 * nvox5_one) / (ms_o5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                       "note": "pcs_process_frames_voxel_device: the same voxel cloud straight from the "
                                                               "rasters; the stitched cloud is never written to HBM. Warm bucket tail: the "
                                                               "pre-aggregation puts every partial into its bucket's region (the previous "
                                                               "call's splitters), one reduce launch follows: 2 kernels per call. "
                                                               "cold_chain_*: PCS_VOXEL_REGIONS=0, every call partitions (histogram, column "
                                                               "scan, scatter, reduce: 5 kernels); lsd_tail_*: the round-4 tail (13 kernels) "
                                                               "forced for the same call; the timed loop's cloud is hashed against the "
                                                               "committed oracle digest"},
                                          "compaction_frac_of_hbm_peak": round(S5 * n5 * (5 + 10 * kept5 / (S5 * n5)) / (ms_c5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "note": "BASELINE.json configs[4] without the 2-per-GPU sharding: compaction + stitch + voxel grid "
                                                  "as two asynchronous device calls (pcs_process_frames_device, pcs_voxel_grid_device_counted)"}
                ctx5.close()
                del dep5, col5, sets5, pay5, vox5
        if extra and not args.no_cache_leg and R > 6:
            with Leg(out, "infinity_cache_resident_inputs"):
                # Informational: the same launches on a ring of 6 frame-sets, whose input rasters (221 MB for 8 x 720p) fit the
                # 256 MiB Infinity Cache — what the kernel reads when its inputs were produced or touched on the GPU just
                # before (and what an under-sized ring silently measures). NOT an HBM figure, NOT `value`.
                def launch6():
                    dp, cp, outp = call_args[next_slot(6)]
                    check(lib.pcs_process_frames_device(h, dp, cp, outp, payload_shorts, None))
                for _ in range(600):
                    launch6()
                torch.cuda.synchronize(dev)
                ms_c6 = timed(launch6, max(400, args.steps))
                out["infinity_cache_resident_inputs"] = {
                    "ms_per_step": round(ms_c6, 5), "value": round(set_points / ms_c6 / 1e3, 1),
                    "algorithmic_GBps": round(set_points * ALGO_BYTES_PER_POINT / (ms_c6 * 1e-3) / 1e9, 1),
                    "ring_frame_sets": 6, "input_mbytes": round(6 * in_bytes_per_set / 1e6, 1),
                    "note": "inputs served by the 256 MiB Infinity Cache, payload written to HBM; informational, not a roofline fraction"}
        if extra and not args.no_cache_leg:
            with Leg(out, "two_stream_overlap"):
                # Informational: the same cold launches alternated over two HIP streams (two contexts), so the drain of
                # launch k overlaps the fill of launch k+1 — what a throughput-oriented frame loop can sustain. It is NOT
                # `value` and not what `roofline` prices (each individual kernel gets longer when two overlap).
                ctx2 = PcsContext(cfgs, device=local_rank)           # its own non-blocking stream
                flip = [0]

                def launch2():
                    flip[0] ^= 1
                    launch_dense(ctx2._h if flip[0] else None)
                for _ in range(400):
                    launch2()
                torch.cuda.synchronize(dev); ctx2.synchronize()
                k2 = max(800, args.steps)
                t0o = time.perf_counter()
                for _ in range(k2):
                    launch2()
                torch.cuda.synchronize(dev); ctx2.synchronize()
                ms_o = (time.perf_counter() - t0o) * 1e3 / k2
                ctx2.close()
                out["two_stream_overlap"] = {"ms_per_step": round(ms_o, 5), "value": round(set_points / ms_o / 1e3, 1),
                                             "aggregate_GBps": round(set_points * ALGO_BYTES_PER_POINT / (ms_o * 1e-3) / 1e9, 1),
                                             "aggregate_frac_of_peak": round(set_points * ALGO_BYTES_PER_POINT / (ms_o * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                             "note": "host clock; consecutive cold launches alternate over two HIP streams and overlap; "
                                                     "informational (not the contract's value, not a per-kernel figure)"}
        if extra and not args.no_general_rotation:
            with Leg(out, "general_rotation"):
                # The synthetic configuration of SURVEY.md 8(d) has depth->colour R = I, which lets the kernel skip 15
                # individually-rounded flops per pixel; real D400 units report a small rotation. Same rasters, same
                # launch, R = 1 degree about a skewed axis:
                import math
                ang = math.radians(1.0)
                ax = np.array([0.3, 0.9, 0.3]); ax /= np.linalg.norm(ax)
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                Rm = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
                cfgs_r = [Syn.synth_stream_config(W, H, rank * S + s) for s in range(S)]
                for cfg_r in cfgs_r:
                    for k, v in enumerate(Rm.T.reshape(-1)):
                        cfg_r.depth_to_color.rotation[k] = float(v)
                ctx_r = PcsContext(cfgs_r, device=local_rank)
                ctx_r.set_stream(stream.cuda_stream)

                def launch_r():
                    launch_dense(ctx_r._h)
                preheat(launch_r, args.preheat_ms / 2)       # same clock settling as the headline leg
                ms_r = timed(launch_r, max(400, args.steps), ctx_r)
                ach_r = set_points * ALGO_BYTES_PER_POINT / (ms_r * 1e-3) / 1e9
                out["general_rotation"] = {"ms_per_step": round(ms_r, 5), "value": round(set_points / ms_r / 1e3, 1),
                                           "achieved": round(ach_r, 1), "frac": round(ach_r / HBM_PEAK_GBS, 4),
                                           "arithmetic": POLICY[min(ctx_r.stream_math(s) for s in range(S))],
                                           "note": "same rasters and launch with a 1-degree depth->colour rotation (what real cameras "
                                                   "report); the headline configuration has R = I per SURVEY.md 8(d)"}
                ctx_r.close()
        if extra and not args.no_general_rotation:
            with Leg(out, "color_1080p"):
                # The stream shapes a real D400 rig records (/root/reference's src/pcs-camera-grab-frames.cpp:69-70): depth
                # 1280x720 with COLOUR 1920x1080, a 1-degree depth->colour rotation and non-zero colour distortion
                # coefficients (inverse Brown-Conrady, the model D400 colour streams report). Every depth pixel gathers its
                # own texel from a raster 2.25 x its size (every third colour row and column is never touched), so the
                # algorithmic bytes stay 2 + 3 + 10 per point while the cache-line traffic of the gather grows.
                import math
                from pointcloud_stitching_amd.types import DISTORTION_INVERSE_BROWN_CONRADY
                CW, CH = 1920, 1080
                ang = math.radians(1.0)
                ax = np.array([0.3, 0.9, 0.3]); ax /= np.linalg.norm(ax)
                Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                Rm = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx @ Kx
                cfgs_c = [Syn.synth_stream_config(W, H, rank * S + s, color_size=(CW, CH)) for s in range(S)]
                for cfg_c in cfgs_c:
                    for k, v in enumerate(Rm.T.reshape(-1)):
                        cfg_c.depth_to_color.rotation[k] = float(v)
                    cfg_c.color.model = DISTORTION_INVERSE_BROWN_CONRADY
                    for k, v in enumerate((0.12, -0.28, 0.0008, -0.0005, 0.09)):
                        cfg_c.color.coeffs[k] = v
                ctx_k = PcsContext(cfgs_c, device=local_rank)
                ctx_k.set_stream(stream.cuda_stream)
                cb = cfgs_c[0].color_bytes
                in_set = S * (npts * 2 + cb)
                Rk = max(4, -(-2 * INFINITY_CACHE_BYTES // in_set) + 2)
                slab_k = torch.empty(Rk * S * (up(npts * 2) + up(cb)) + 256, dtype=torch.uint8, device=dev)
                ok_ = (-slab_k.data_ptr()) % 256
                hostc = [Syn.synth_color(CW, CH, rank * S + s) for s in range(S)]
                args_k, first_c = [], []
                for slot in range(Rk):
                    dps, cps = [], []
                    for s in range(S):
                        v = slab_k[ok_:ok_ + npts * 2]; v.copy_(d_depth[0][s]); dps.append(v.data_ptr()); ok_ += up(npts * 2)
                        v = slab_k[ok_:ok_ + cb]
                        if slot == 0:
                            v.copy_(torch.from_numpy(hostc[s])); first_c.append(v)
                        else:
                            v.copy_(first_c[s])
                        cps.append(v.data_ptr()); ok_ += up(cb)
                    args_k.append(((VP * S)(*dps), (VP * S)(*cps)))
                kk = [0]

                def launch_k():
                    dp, cp = args_k[kk[0] % Rk]; kk[0] += 1
                    check(lib.pcs_process_frames_device(ctx_k._h, dp, cp, VP(d_out[kk[0] % R].data_ptr()), payload_shorts, None), ctx_k._h)
                # parity spot check of camera 0 against the oracle before timing
                kk[0] = 0
                launch_k(); torch.cuda.synchronize(dev)
                from oracle import pcs_oracle as O
                want_k, _ = O.process_frames(cfgs_c[:1], host0[0][:1], hostc[:1], 0, 1)
                got_k = d_out[1 % R][:want_k.size].cpu().numpy().reshape(-1, 5)
                if (got_k != want_k).any():
                    raise RuntimeError("colour-1080p leg: HIP output differs from the oracle")
                preheat(launch_k, args.preheat_ms / 2)
                ms_k = timed(launch_k, max(300, args.steps), ctx_k)
                ach_k = set_points * ALGO_BYTES_PER_POINT / (ms_k * 1e-3) / 1e9
                # the same launch priced by the colour bytes it must TOUCH: every 128-byte line of the larger raster that holds
                # some pixel's texel (camera 0's map, from the oracle's texture coordinates), instead of 3 B per point
                _, tex = O.deproject(cfgs_c[0], host0[0][0])
                tx = np.clip((tex[:, 0] * np.float32(CW) + np.float32(0.5)).astype(np.int64), 0, CW - 1)
                ty = np.clip((tex[:, 1] * np.float32(CH) + np.float32(0.5)).astype(np.int64), 0, CH - 1)
                ok_px = host0[0][0].reshape(-1) != 0
                off_b = (ty * cfgs_c[0].color_stride + tx * 3)[ok_px]
                lines = np.union1d(off_b // 128, (off_b + 2) // 128).size
                touched_pp = lines * 128.0 / npts
                ach_t = set_points * (2 + 10 + touched_pp) / (ms_k * 1e-3) / 1e9
                out["color_1080p"] = {"ms_per_step": round(ms_k, 5), "value": round(set_points / ms_k / 1e3, 1),
                                      "achieved": round(ach_k, 1), "frac": round(ach_k / HBM_PEAK_GBS, 4),
                                      "touched_colour_bytes_per_point": round(touched_pp, 3),
                                      "frac_touched_bytes": round(ach_t / HBM_PEAK_GBS, 4),
                                      "arithmetic": POLICY[min(ctx_k.stream_math(s) for s in range(S))],
                                      "workload": f"{S} x (Z16 {W}x{H} + RGB8 {CW}x{CH}), 1-degree depth->colour rotation, inverse "
                                                  f"Brown-Conrady colour coefficients (0.12, -0.28, 0.0008, -0.0005, 0.09)",
                                      "algorithmic_bytes_per_point": ALGO_BYTES_PER_POINT, "ring_frame_sets": Rk,
                                      "pmc_traffic_bytes_per_launch": 135_950_000,
                                      "note": "the geometry a D400 rig records; bytes priced as 2 (Z16) + 3 (the point's own texel) + 10 "
                                              "(record). PMC (profiles/README.md, r03): 2 x FETCH_SIZE + WRITE_SIZE = 62.2 + 73.7 MB = 1.23 x "
                                              "algorithmic — the gather pulls in 95 % of the 2.25 x larger colour raster's lines. Of the gap to "
                                              "the same-size, undistorted launch (tools/color_probe.py) the distortion polynomial's ~25 "
                                              "individually rounded flops per pixel cost 2.7 us (VALU), the larger raster's gather 1.5 us"}
                ctx_k.close()
                del slab_k
        if world == 1 and args.mode in ("dense", "drop_invalid", "cutoff"):
            with Leg(out, "per_launch_ms"):
                # per-launch distribution (SURVEY.md 8d asks for median + min): a separate leg with a hipEvent pair
                # around every launch, so the event records stay out of the timed region above
                ctx.kernel_timing(True)
                for _ in range(300):
                    launch()
                per = np.sort(ctx.kernel_times_ms())       # synchronises the stream
                ctx.kernel_timing(False)
                if per.size:
                    out["roofline"]["per_launch_ms"] = {"n": int(per.size), "median": round(float(np.median(per)), 5),
                                                        "min": round(float(per[0]), 5), "p95": round(float(per[int(per.size * 0.95)]), 5),
                                                        "note": "one hipEvent pair per launch (includes event overhead); "
                                                                "avg_launch_ms above is the contract figure"}
        if world == 1 and args.mode == "dense" and not args.no_host_api:
            with Leg(out, "host_api"):
                # PCIe-inclusive: host pointers in, host buffer out (36.9 MB up + 73.7 MB down per frame-set),
                # pageable numpy memory like a caller of the reference's function would have. Never `value`.
                def time_host(dep, col, outbuf, reps=5):
                    ctx.process_frames(dep, col, out=outbuf)
                    t0h = time.perf_counter()
                    for _ in range(reps):
                        ctx.process_frames(dep, col, out=outbuf)
                    return (time.perf_counter() - t0h) / reps
                pg_out = np.zeros(2 + payload_shorts, np.int16)      # allocated and touched once, like the reference's buffer (:157)
                th = time_host(host0[0], host0[1], pg_out)
                pd = [ctx.host_array(d.shape, np.uint16) for d in host0[0]]
                pc = [ctx.host_array(c.shape, np.uint8) for c in host0[1]]
                for a, b in zip(pd + pc, host0[0] + host0[1]):
                    a[...] = b
                po = ctx.host_array((2 + payload_shorts,), np.int16)
                tp = time_host(pd, pc, po)
                # software-pipelined loop (pcs_submit_frames / pcs_collect_frames): upload of k+1 overlaps download of k
                po2 = ctx.host_array((2 + payload_shorts,), np.int16)

                sub_host = []

                def time_pipe(reps=8):
                    ta, tb = ctx.submit_frames(pd, pc), ctx.submit_frames(pd, pc)     # warm both slots
                    ctx.collect_frames(ta, po); ctx.collect_frames(tb, po2)
                    t0p = time.perf_counter()
                    t_prev = ctx.submit_frames(pd, pc)
                    for k in range(1, reps + 1):
                        ts = time.perf_counter()
                        t_next = ctx.submit_frames(pd, pc) if k < reps else None
                        if t_next is not None:
                            sub_host.append(time.perf_counter() - ts)
                        ctx.collect_frames(t_prev, po if k & 1 else po2)
                        t_prev = t_next
                    return (time.perf_counter() - t0p) / reps
                tpipe = time_pipe()
                # the two directions on their own (page-locked buffers), SURVEY.md 8d: "H2D/D2H reported separately"
                d_tmp = ctx.device_malloc(payload_shorts * 2)

                def time_copy(fn, reps=5):
                    fn()
                    t0c = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    return (time.perf_counter() - t0c) / reps
                up_bytes = sum(a.nbytes for a in pd + pc)

                def all_up():
                    o = 0
                    for a in pd + pc:
                        ctx.memcpy_h2d(d_tmp + o, a); o += (a.nbytes + 255) & ~255
                t_up = time_copy(all_up)
                pay = po[2:]
                t_dn = time_copy(lambda: ctx.memcpy_d2h(pay, d_tmp))
                ctx.device_free(d_tmp)
                out["host_api"] = {"ms_per_step": round(th * 1e3, 3), "value": round(set_points / th / 1e6, 1),
                                   "pinned_ms_per_step": round(tp * 1e3, 3), "pinned_value": round(set_points / tp / 1e6, 1),
                                   "pipelined_ms_per_step": round(tpipe * 1e3, 3), "pipelined_value": round(set_points / tpipe / 1e6, 1),
                                   "breakdown": {"d2h_alone_ms": round(t_dn * 1e3, 3), "h2d_alone_ms": round(t_up * 1e3, 3),
                                                 "submit_host_enqueue_ms": round(float(np.median(sub_host)) * 1e3, 3) if sub_host else None,
                                                 "rest_ms": round((tpipe - t_dn - (float(np.median(sub_host)) if sub_host else 0.0)) * 1e3, 3),
                                                 "note": "a pipelined step = the download of frame-set k (the longer direction; the upload of k+1 "
                                                         "runs beside it) + the host time of submit(k+1) — 2 x S hipMemcpyAsync + the launch — which "
                                                         "passes before collect(k) can enqueue that download + rest (the link's duplex penalty, "
                                                         "measured 1.42 vs 1.30 ms in tools/lab, event and synchronisation latency). Enqueueing the "
                                                         "download at SUBMIT time (destination named early) was built and measured in round 5: "
                                                         "2.02 instead of 1.63 ms — copies issued in that order run one after the other"},
                                   "h2d_ms": round(t_up * 1e3, 3), "h2d_GBps": round(up_bytes / t_up / 1e9, 1),
                                   "d2h_ms": round(t_dn * 1e3, 3), "d2h_GBps": round(pay.nbytes / t_dn / 1e9, 1),
                                   "unit": "Mpoints/s", "note": "pcs_process_frames, synchronous, per frame-set. ms_per_step: long-lived pageable "
                                   "(numpy) buffers = staged, H2D (36.9 MB) + kernel + D2H (73.7 MB). pinned_*: every buffer from pcs_host_malloc = "
                                   "ZERO COPY, the kernels read the rasters and write the payload over PCIe themselves, both directions at once. "
                                   "pipelined_*: pcs_submit_frames / pcs_collect_frames (staged, upload of k+1 overlaps download of k). All bounded "
                                   "by the host link, not by the kernel"}
        if extra:
            with Leg(out, "per_kernel_unprofiled_us"):
                # what one launch of each short kernel takes in a back-to-back loop bracketed by ONE hipEvent pair — no
                # per-launch events, no profiler (rocprofv3 --kernel-trace adds ~1.5 us to every short kernel: profiles/README.md)
                pk = {"pcs_fused_dense_kernel (8 x 720p)": round(kern_ms * 1e3, 2)}
                if "pack_twin" in out:
                    pk["pcs_pack_dense_kernel (1 x 720p)"] = round(out["pack_twin"]["per_stream_launches_ms_per_frame_set"] / S * 1e3, 2)
                    pk["pcs_pack_batch_kernel (8 x 720p)"] = round(out["pack_twin"]["batched_ms_per_frame_set"] * 1e3, 2)
                if "compaction" in out:
                    pk["count + scan + emit (8 x 720p, drop-invalid)"] = round(out["compaction"]["ms_per_step"] * 1e3, 2)
                    if "caller_counts" in out["compaction"]:
                        pk["scan + emit (caller counts)"] = round(out["compaction"]["caller_counts"]["ms_per_step"] * 1e3, 2)
                out["per_kernel_unprofiled_us"] = pk
        if world == 1 and not args.no_cpu_baseline:
            with Leg(out, "cpu_baseline"):
                if cpu_first is None:
                    raise RuntimeError(cpu_first_error or "cpu baseline did not run")
                out["cpu_baseline"] = cpu_first
                if args.mode == "dense":
                    out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        emit(out)

    if ctx0 is not ctx:
        ctx0.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
