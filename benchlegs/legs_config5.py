"""BASELINE configs[4] on ONE GPU, riding on the default line."""
import hashlib
import json
import os

import numpy as np

from .common import ROOT, HBM_PEAK_GBS
from .rig import VP


def config5_one_gpu(g):
    """16 x 1920x1080 -> invalid-depth compaction -> camera-order stitch -> voxel grid of the stitched cloud, device-resident and
    asynchronous; and the same voxel cloud in ONE call from the rasters (pcs_process_frames_voxel_device)."""
    from pointcloud_stitching_amd.types import POINT_SHORTS, FLAG_DROP_INVALID
    torch, dev, lib, Syn, check, timed = g.torch, g.dev, g.lib, g.Syn, g.check, g.timed
    # ---- BASELINE configs[4] on ONE GPU: 16 x 1920x1080 -> invalid-depth compaction -> camera-order stitch -> voxel
    # grid of the stitched cloud, device-resident and asynchronous (the voxel grid reads the kept total from the
    # device); 4 input sets (664 MB) so that the rasters come from HBM
    W5, H5, S5, LEAF = 1920, 1080, 16, 50
    cfg5 = [Syn.synth_stream_config(W5, H5, s) for s in range(S5)]
    ctx5 = g.new_context(cfg5, flags=FLAG_DROP_INVALID)
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
    # ... and as a FRAME LOOP over two contexts used in turn (each its own stream, workspace, splitters, regions): the bucket tail of
    # frame-set k — a latency chain that leaves the chip almost empty — runs beside the pre-aggregation of k+1. What libpcs_node does
    # for a one-peer node (pcs_node_submit_voxel_device / pcs_node_wait_voxel) and what a caller's own loop can do with two pcs_ctx.
    ctx5b = g.new_context(cfg5, flags=FLAG_DROP_INVALID, own_stream=True)
    ctx5.set_stream(0)                                   # its own stream again: the two contexts must not share one ...
    beside = ctx5b.use_stream_beside(ctx5)               # ... nor one hardware queue (streams are dealt onto a few queues round robin)
    vox5b = torch.empty(S5 * n5 * POINT_SHORTS, dtype=torch.int16, device=dev)
    nv5b = torch.zeros(1, dtype=torch.int32, device=dev)
    pair = ((ctx5, vox5, nv5), (ctx5b, vox5b, nv5b))
    torch.cuda.synchronize(dev)

    def turn():
        c, v, nvx = pair[k5[0] & 1]
        dp, cp = args5[k5[0] % 4]; k5[0] += 1
        check(lib.pcs_process_frames_voxel_device(c._h, dp, cp, LEAF, VP(v.data_ptr()), v.numel(), VP(nvx.data_ptr())), c._h)
    import time
    for _ in range(8):
        turn()
    ctx5.synchronize(); ctx5b.synchronize()
    ms_pair = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(60):
            turn()
        ctx5.synchronize(); ctx5b.synchronize()
        ms_pair = min(ms_pair, (time.perf_counter() - t0) * 1e3 / 60)
    pair_ok = True
    for c, v, nvx in pair:
        m = int(nvx.item())
        dg = hashlib.sha256(v[:m * POINT_SHORTS].cpu().numpy().tobytes()).hexdigest()
        pair_ok = pair_ok and bool(gold5) and gold5["voxels"] == m and gold5["sha256"] == dg
    if gold5 and not pair_ok:
        raise RuntimeError("config5 two-context frame loop: a voxel cloud differs from the oracle digest")
    row_const = all(ctx5.stream_color_row_const(s) for s in range(S5))
    ctx5b.close()
    del vox5b
    ctx5.set_stream(g.stream.cuda_stream)
    res = {"workload": f"{S5} x {W5}x{H5} synthetic streams, PCS_FLAG_DROP_INVALID, voxel leaf {LEAF} mm",
           "points_in": S5 * n5, "points_kept": kept5, "voxels": nvox5,
           "compaction_ms": round(ms_c5, 4), "voxel_grid_ms": round(ms_v5, 4),
           "pipeline_ms_per_frame_set": round(ms_b5, 4),
           "value": round(S5 * n5 / ms_b5 / 1e3, 1), "unit": "Mpoints/s in",
           "one_call": {"ms_per_frame_set": round(ms_o5, 4), "value": round(S5 * n5 / ms_o5 / 1e3, 1),
                        "voxels": nvox5_one, "oracle_digest_ok": bool(gold5 is not None),
                        "kernels_per_call": (2 if regions_default != "0" else 5) if bucket_default else 13,
                        "cold_chain_ms_per_frame_set": round(ms_o5_cold, 4),
                        "lsd_tail_ms_per_frame_set": round(ms_o5_lsd, 4),
                        "frame_loop_two_contexts_ms_per_frame_set": round(ms_pair, 4),
                        "frame_loop_two_contexts_digest_ok": bool(pair_ok),
                        "frame_loop_frac": round((5 * S5 * n5 + 10 * nvox5_one) / (ms_pair * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "frame_loop_streams_seen_to_overlap": bool(beside),
                        "colour_row_from_table": bool(row_const),
                        "algorithmic_bytes": int(5 * S5 * n5 + 10 * nvox5_one),
                        "frac": round((5 * S5 * n5 + 10 * nvox5_one) / (ms_o5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "pcs_process_frames_voxel_device: the same voxel cloud straight from the "
                                "rasters; the stitched cloud is never written to HBM. Warm bucket tail: the "
                                "pre-aggregation puts every partial into its bucket's region (the previous "
                                "call's splitters), one reduce launch follows: 2 kernels per call. "
                                "frame_loop_two_contexts_*: the same call on two contexts used in turn (host clock over 3 x 60 calls, best): the bucket tail of k "
                                "runs beside the pre-aggregation of k+1; colour_row_from_table: CertRowConst certified for every stream. "
                                "cold_chain_*: PCS_VOXEL_REGIONS=0, every call partitions (histogram, column "
                                "scan, scatter, reduce: 5 kernels); lsd_tail_*: the round-4 tail (13 kernels) "
                                "forced for the same call; the timed loop's cloud is hashed against the "
                                "committed oracle digest"},
           "compaction_frac_of_hbm_peak": round(S5 * n5 * (5 + 10 * kept5 / (S5 * n5)) / (ms_c5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "note": "BASELINE.json configs[4] without the 2-per-GPU sharding: compaction + stitch + voxel grid "
                   "as two asynchronous device calls (pcs_process_frames_device, pcs_voxel_grid_device_counted)"}
    ctx5.close()
    del dep5, col5, sets5, pay5, vox5
    return res
