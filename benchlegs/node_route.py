"""bench.py --route node: ONE process drives the N GPUs through libpcs_node (include/pcs_node.h)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

from .common import (ROOT, ALGO_BYTES_PER_POINT, HBM_PEAK_GBS, INFINITY_CACHE_BYTES, POLICY, Leg, emit, flush_c_stdio)


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
        how = ("every peer pre-aggregates into a voxel sink of the one GPU they share (nothing to exchange), one tail per frame-set"
               if node.voxel_sink else "per-GPU voxel partials, one grouped exchange, one sort + segmented mean")
        workload = (f"{cfg_name}: {total_streams} synthetic {W}x{H} Z16+RGB8 streams on {where}, PCS_FLAG_DROP_INVALID, voxel-grid downsample "
                    f"(leaf {LEAF} mm) of the stitched cloud on GPU 0: {how}")
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
        out["voxel_sink"] = bool(node.voxel_sink)
        if node.voxel_sink:
            # every peer shares GPU 0: the line above went through the sinks (no exchange). Beside it the route such a node exists to
            # exercise — partials to caller-held arrays, ONE grouped RCCL exchange (self send/recv), place + tail on the root's second context
            node.set_timing(False)
            node.set_voxel_sink(False)
            try:
                run(max(args.warmup, 4)); sync_all()
                t1 = time.perf_counter(); run(args.steps); sync_all()
                x_ms = (time.perf_counter() - t1) * 1e3 / args.steps
                node.set_timing(True)
                xph = {"kernel": [], "exchange": [], "root": []}
                t = submit()
                for _ in range(10):
                    t2 = submit(); wait(t); t = t2
                    st = node.last_stats()
                    xph["kernel"].append(st["kernels_ms"]); xph["exchange"].append(st["exchange_ms"]); xph["root"].append(st["root_ms"])
                wait(t)
                node.set_timing(False)
                out["same_gpu_peers"] = {
                    "route": "voxel sink (pcs_voxel_sink_*): the peers' pre-aggregations write into the workspace of a sink context of the GPU "
                             "they share, two sinks in turn; no exchange, no placement",
                    "partials_exchange": {"ms_per_step": round(x_ms, 5), "bytes_into_root_per_step": int(st["exchanged_bytes"]),
                                          "partials_reduced_per_step": int(st["reduced"]),
                                          "phases_ms": {k: round(float(np.median(v)), 5) for k, v in xph.items()},
                                          "note": "PCS_NODE_VOXEL_SINK=0: the route of peers on different GPUs, here on RCCL self send/recv"}}
            finally:
                node.set_timing(False)
                node.set_voxel_sink(True)
        if P == 1 and os.environ.get("PCS_NODE_ONE_CALL", "2") == "2":
            # one peer: submit enqueued the rasters -> voxels call (no partials leave the library), the two slots on two contexts in
            # turn. Beside it, the same loop on ONE context (frame-sets queue behind each other) and with the partials pipeline a node
            # of several peers runs (pre-aggregation of k+1 beside the root's sort + mean of k)
            node.set_timing(False)
            other = {}
            for mode, key in ((1, "one_context_ms_per_step"), (0, "partials_pipeline_ms_per_step")):
                node.set_one_call(mode)
                try:
                    run(max(args.warmup, 4)); sync_all()
                    t1 = time.perf_counter(); run(args.steps); sync_all()
                    other[key] = round((time.perf_counter() - t1) * 1e3 / args.steps, 5)
                finally:
                    node.set_one_call(2)
            out["one_peer"] = {"route": "pcs_process_frames_voxel_device enqueued at submit (warm bucket tail: 2 launches per frame-set), the two "
                                        "slots on two contexts of the peer used in turn: the tail of k runs beside the pre-aggregation of k+1",
                               **other,
                               "note": "one_context = PCS_NODE_ONE_CALL=1 (round 5's route); partials_pipeline = PCS_NODE_ONE_CALL=0: partials to "
                                       "caller-held arrays, sort + mean on a second context beside the next frame-set's pre-aggregation (what a "
                                       "node of several peers runs on its root)"}
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
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--route", "node", "--node-direct-child", "--gpus", str(args.gpus),
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
                        "pairs" + (" (config5: none by default — they pre-aggregate into a sink of that GPU; same_gpu_peers.partials_exchange "
                                   "is the exchange route)" if config5 and out.get("voxel_sink") else "")
                        + ". Exercises the N > 1 flow; says nothing about scaling")
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

