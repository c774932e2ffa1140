"""BASELINE.json configs[4] as a SHARDED pipeline: 16 x 1920x1080 streams, cameras split over GPUs / ranks, invalid-depth
compaction, voxel grid of the stitched cloud on the root.

The shape it replaces in the reference: src/pcs-multicamera-client.cpp:373-409 (camera-order concatenation on the centre)
+ src/pcs-multicamera-optimized.cpp:226-248 (downsample there). Here every GPU pre-aggregates its own cameras into voxel
partials, the partials are exchanged once, and the root runs one sort + segmented mean; because the voxel sums are integers
the result must be BYTE-identical to the voxel grid of the stitched cloud — which is what these tests hold it to: against the
CPU oracle at small sizes and against the committed full-size oracle digests (tests/golden/config5_digests.json)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext, PcsError
from pointcloud_stitching_amd.types import FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
CENTRAL = os.path.join(ROOT, "pointcloud_stitching_amd", "bin", "pcs-multicamera-optimized")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5_digests.json")))


def _upload(ctx, depth, color):
    dd = [ctx.device_malloc(max(d.nbytes, 16)) for d in depth]
    dc = [ctx.device_malloc(max(c.nbytes, 16)) for c in color]
    for ptr, a in zip(dd + dc, list(depth) + list(color)):
        ctx.memcpy_h2d(ptr, a)
    return dd, dc


def _sharded_voxels(cfgs, depth, color, shards, leaf, flags, downsample=1):
    """The partials route with `shards` contexts on this one GPU standing in for as many GPUs: shard r holds the cameras
    [r*per, (r+1)*per); the root is shard 0's context. Returns (voxel records, partials per shard)."""
    per = len(cfgs) // shards
    ctxs = [PcsContext(cfgs[r * per:(r + 1) * per], flags=flags, downsample=downsample) for r in range(shards)]
    try:
        root = ctxs[0]
        caps = [c.max_payload_shorts // 5 for c in ctxs]
        total = sum(caps)
        d_keys = root.device_malloc(total * 8 + 64)
        d_parts = root.device_malloc(total * 32 + 64)
        counts, off = [], 0
        for r, c in enumerate(ctxs):
            dd, dc = _upload(c, depth[r * per:(r + 1) * per], color[r * per:(r + 1) * per])
            # every shard writes into its own arrays ...
            k_r = c.device_malloc(caps[r] * 8 + 64); p_r = c.device_malloc(caps[r] * 32 + 64); n_r = c.device_malloc(64)
            c.process_frames_voxel_partials_device(dd, dc, leaf, k_r, p_r, caps[r], n_r)
            c.synchronize()
            m = np.empty(1, np.int32); c.memcpy_d2h(m, n_r)
            m = int(m[0])
            assert 0 <= m <= caps[r]
            # ... and the "exchange" lands them behind the earlier shards' on the root (host hop = the wire here)
            if m:
                hk = np.empty(m, np.uint64); hp = np.empty(m * 8, np.uint32)
                c.memcpy_d2h(hk, k_r); c.memcpy_d2h(hp, p_r)
                root.memcpy_h2d(d_keys + off * 8, hk); root.memcpy_h2d(d_parts + off * 32, hp)
            counts.append(m); off += m
            for ptr in dd + dc + [k_r, p_r, n_r]:
                c.device_free(ptr)
        d_out = root.device_malloc(max(off, 1) * 10 + 64)
        d_nv = root.device_malloc(64)
        root.voxel_grid_from_partials_device(d_keys, d_parts, off, leaf, d_out, max(off, 1) * 5, d_nv)
        root.synchronize()
        nv = np.empty(1, np.int32); root.memcpy_d2h(nv, d_nv)
        got = np.empty(max(int(nv[0]), 1) * 5, np.int16); root.memcpy_d2h(got, d_out)
        return got[:int(nv[0]) * 5].reshape(-1, 5), counts
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, FLAG_DROP_INVALID, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT | FLAG_DROP_INVALID])
@pytest.mark.parametrize("shape,shards", [((4, 320, 240), 2), ((6, 200, 96), 3), ((4, 203, 57), 4)])
def test_partials_of_sharded_cameras_reduce_to_the_voxel_grid_of_the_stitched_cloud(oracle, flags, shape, shards):
    """Shards' partials concatenated on the root == oracle voxel grid of the stitched cloud, at several leaves, with and
    without predicates, incl. a raster width that is not a multiple of 8 (the consecutive-pixel reader and, below 36 mm,
    the internal stitched-cloud route)."""
    n, w, h = shape
    cfgs, depth, color = S.synth_frame_set(n, w, h)
    stitched, _ = oracle.process_frames(cfgs, depth, color, flags, 1)
    for leaf in (20, 50, 200):
        want = oracle.voxel_grid(stitched, leaf)
        got, counts = _sharded_voxels(cfgs, depth, color, shards, leaf, flags)
        assert got.shape == want.shape and (got == want).all(), (leaf, counts)


@pytest.mark.gpu
def test_partials_with_a_stride_and_with_nothing_kept(oracle):
    cfgs, depth, color = S.synth_frame_set(4, 160, 120)
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, 3)
    want = oracle.voxel_grid(stitched, 64)
    got, _ = _sharded_voxels(cfgs, depth, color, 2, 64, FLAG_DROP_INVALID, downsample=3)
    assert got.shape == want.shape and (got == want).all()
    empty = [np.zeros_like(d) for d in depth]                                       # every pixel invalid: no partials at all
    got, counts = _sharded_voxels(cfgs, empty, color, 2, 50, FLAG_DROP_INVALID)
    assert got.shape == (0, 5) and counts == [0, 0]


@pytest.mark.gpu
def test_partials_entry_points_check_their_arguments():
    cfgs, depth, color = S.synth_frame_set(2, 64, 48)
    with PcsContext(cfgs) as ctx:
        dd, dc = _upload(ctx, depth, color)
        cap = ctx.max_payload_shorts // 5
        k = ctx.device_malloc(cap * 8 + 64); p = ctx.device_malloc(cap * 32 + 64); n = ctx.device_malloc(64)
        with pytest.raises(PcsError) as e:
            ctx.process_frames_voxel_partials_device(dd, dc, 50, k, p, cap - 1, n)          # worst case does not fit
        assert e.value.status == -5
        with pytest.raises(PcsError) as e:
            ctx.process_frames_voxel_partials_device(dd, dc, 0, k, p, cap, n)
        assert e.value.status == -1
        with pytest.raises(PcsError) as e:
            ctx.process_frames_voxel_partials_device(dd, dc, 50, k + 4, p, cap, n)          # misaligned key array
        assert e.value.status == -1
        with pytest.raises(PcsError) as e:
            ctx.voxel_grid_from_partials_device(k, p, 100, 50, k, 10, n)                     # output too small
        assert e.value.status == -5
        ctx.voxel_grid_from_partials_device(k, p, 0, 50, k, 0, n)                            # nothing in, nothing out
        ctx.synchronize()
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, n)
        assert int(nv[0]) == 0


def _sink_voxels(cfgs, sets, shards, leaf, flags, downsample=1, tails=None):
    """The sink route (pcs_voxel_sink_*) with `shards` contexts of this GPU pre-aggregating into ONE more context's workspace: call after
    call on the same sink (the second and later ones are warm: partials straight into the buckets' regions), the streams ordered by host
    synchronisation. Returns the voxel records of every call."""
    per = len(cfgs) // shards
    ctxs = [PcsContext(cfgs[r * per:(r + 1) * per], flags=flags, downsample=downsample) for r in range(shards)]
    sink_ctx = PcsContext(cfgs[:1])
    try:
        total = sum(c.max_payload_shorts // 5 for c in ctxs)
        d_out = sink_ctx.device_malloc(total * 10 + 64)
        d_nv = sink_ctx.device_malloc(64)
        outs = []
        for k, (depth, color) in enumerate(sets):
            if tails:
                sink_ctx.set_voxel_tail(tails[k % len(tails)])
            sink = sink_ctx.voxel_sink_begin(total, leaf)
            sink_ctx.synchronize()
            ptrs = []
            for r, c in enumerate(ctxs):
                dd, dc = _upload(c, depth[r * per:(r + 1) * per], color[r * per:(r + 1) * per])
                c.process_frames_voxel_into_sink_device(dd, dc, sink)
                ptrs.append((c, dd + dc))
            for c, pp in ptrs:
                c.synchronize()
                for q in pp:
                    c.device_free(q)
            sink_ctx.voxel_sink_finish(sink, d_out, total * 5, d_nv)
            sink_ctx.synchronize()
            nv = np.empty(1, np.int32); sink_ctx.memcpy_d2h(nv, d_nv)
            assert 0 <= int(nv[0]) <= total
            got = np.empty(max(int(nv[0]), 1) * 5, np.int16); sink_ctx.memcpy_d2h(got, d_out)
            outs.append(got[:int(nv[0]) * 5].reshape(-1, 5).copy())
        return outs
    finally:
        sink_ctx.close()
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, FLAG_DROP_INVALID, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT | FLAG_DROP_INVALID])
@pytest.mark.parametrize("shape,shards", [((4, 320, 240), 2), ((6, 200, 96), 3), ((4, 203, 57), 4), ((8, 320, 240), 8)])
def test_sink_of_several_contexts_is_the_voxel_grid_of_the_stitched_cloud(oracle, flags, shape, shards):
    """pcs_voxel_sink_begin / pcs_process_frames_voxel_into_sink_device / pcs_voxel_sink_finish: several contexts of one GPU write their
    partials into one context's workspace; the tail over the union == oracle voxel grid of the stitched cloud. Four calls per leaf on the
    same sink (cold, then warm with regions sized by the call before), a scene that CHANGES between the calls (other seeds: regions
    overflow into the general list), leaves on both sides of the bucket / LSD switch, a raster width that is not a multiple of 8."""
    n, w, h = shape
    cfgs = [S.synth_stream_config(w, h, s) for s in range(n)]
    sets = [([S.synth_depth(w, h, s, seed=S.SEED + 31 * f) for s in range(n)], [S.synth_color(w, h, s, seed=S.SEED + 31 * f) for s in range(n)])
            for f in (0, 0, 1, 2)]
    stitched = [oracle.process_frames(cfgs, d, c, flags, 1)[0] for d, c in sets]
    for leaf in (20, 50, 200):
        got = _sink_voxels(cfgs, sets, shards, leaf, flags)
        for k, g in enumerate(got):
            want = oracle.voxel_grid(stitched[k], leaf)
            assert g.shape == want.shape and (g == want).all(), (leaf, k)


@pytest.mark.gpu
def test_sink_with_a_stride_with_nothing_kept_and_across_tail_changes(oracle):
    VOXEL_TAIL_BUCKET, VOXEL_TAIL_LSD = 1, 2          # include/pcs_hip.h: PCS_VOXEL_TAIL_*
    cfgs, depth, color = S.synth_frame_set(4, 160, 120)
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, 3)
    want = oracle.voxel_grid(stitched, 64)
    for g in _sink_voxels(cfgs, [(depth, color)] * 3, 2, 64, FLAG_DROP_INVALID, downsample=3):      # the stride: through each context's own cloud
        assert g.shape == want.shape and (g == want).all()
    empty = [np.zeros_like(d) for d in depth]                                       # every pixel invalid: no partials at all
    for g in _sink_voxels(cfgs, [(empty, color), (depth, color), (empty, color)], 2, 50, FLAG_DROP_INVALID)[::2]:
        assert g.shape == (0, 5)
    # bucket, bucket (warm), LSD, bucket (its splitters are still this leaf's), LSD: the sink's workspace serves whichever tail a call takes
    stitched, _ = oracle.process_frames(cfgs, depth, color, 0, 1)
    want = oracle.voxel_grid(stitched, 50)
    for g in _sink_voxels(cfgs, [(depth, color)] * 5, 4, 50, 0, tails=[VOXEL_TAIL_BUCKET, VOXEL_TAIL_BUCKET, VOXEL_TAIL_LSD, VOXEL_TAIL_BUCKET, VOXEL_TAIL_LSD]):
        assert g.shape == want.shape and (g == want).all()


@pytest.mark.gpu
def test_sink_entry_points_check_their_arguments():
    cfgs, depth, color = S.synth_frame_set(2, 64, 48)
    with PcsContext(cfgs) as ctx, PcsContext(cfgs[:1]) as sink_ctx:
        dd, dc = _upload(ctx, depth, color)
        cap = ctx.max_payload_shorts // 5
        out = sink_ctx.device_malloc(cap * 10 + 64); n = sink_ctx.device_malloc(64)
        with pytest.raises(PcsError) as e:
            sink_ctx.voxel_sink_begin(cap, 0)
        assert e.value.status == -1
        with pytest.raises(PcsError) as e:
            sink_ctx.voxel_sink_begin(0, 50)
        assert e.value.status == -1
        import ctypes as C
        junk = (C.c_uint64 * 24)()
        with pytest.raises(PcsError) as e:
            ctx.process_frames_voxel_into_sink_device(dd, dc, junk)                          # not a sink
        assert e.value.status == -1
        small = sink_ctx.voxel_sink_begin(cap - 1, 50)
        with pytest.raises(PcsError) as e:
            ctx.process_frames_voxel_into_sink_device(dd, dc, small)                         # this context alone can exceed the sink
        assert e.value.status == -5
        sink = sink_ctx.voxel_sink_begin(cap, 50)                                            # (abandons the small one)
        sink_ctx.synchronize()
        ctx.process_frames_voxel_into_sink_device(dd, dc, sink)
        ctx.synchronize()
        with pytest.raises(PcsError) as e:
            sink_ctx.voxel_sink_finish(sink, out, cap * 5 - 1, n)                            # output too small
        assert e.value.status == -5
        with pytest.raises(PcsError) as e:
            ctx.voxel_sink_finish(sink, out, cap * 5, n)                                     # not the context that opened it
        assert e.value.status == -1
        sink_ctx.voxel_sink_finish(sink, out, cap * 5, n)
        sink_ctx.synchronize()
        with pytest.raises(PcsError) as e:
            sink_ctx.voxel_sink_finish(sink, out, cap * 5, n)                                # already finished
        assert e.value.status == -1
        # the sink context still serves its own calls afterwards
        d1, c1 = _upload(sink_ctx, depth[:1], color[:1])
        sink_ctx.process_frames_voxel_device(d1, c1, 50, out, cap * 5, n)
        sink_ctx.synchronize()


@pytest.mark.gpu
def test_config5_full_size_sharded_eight_plus_eight_against_oracle_digests():
    """16 x 1920x1080 split 8 + 8 over two contexts (two GPUs' worth of cameras), DROP_INVALID, 50 and 200 mm: the root's
    voxel cloud must hash to the oracle's digests of the voxel grid of the full stitched cloud."""
    cfgs, depth, color = S.synth_frame_set(16, 1920, 1080)
    for leaf, want in sorted(GOLD["voxel"].items()):
        got, counts = _sharded_voxels(cfgs, depth, color, 2, int(leaf), FLAG_DROP_INVALID)
        assert got.shape[0] == want["voxels"], (leaf, counts)
        assert hashlib.sha256(got.tobytes()).hexdigest() == want["sha256"], leaf
        assert sum(counts) * 40 < GOLD["points"] * 10 / 4          # the point of the route: the exchange is a fraction of the payloads


def _bench_line(*extra, launcher=()):
    r = subprocess.run([sys.executable, *launcher, BENCH, *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("leaf,port", [(50, 29581), (200, 29583)])
def test_two_ranks_run_config5_end_to_end_against_the_digests(leaf, port):
    """Two ranks (gloo, both on this one GPU, host-staged exchange) run bench.py's config-5 workload: rank r pre-aggregates
    cameras 8r..8r+7, the partial counts are all-gathered, keys + partials travel to rank 0, rank 0 reduces. bench.py itself
    aborts if the root's voxel cloud differs from the oracle digest; the line must say so and name configs[4]."""
    d = _bench_line("--workload", "config5", "--leaf", str(leaf), "--gpus", "2", "--steps", "2", "--warmup", "1",
                    "--debug-backend", "gloo", "--ring", "2",
                    launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)))
    want = GOLD["voxel"][str(leaf)]
    assert d["n_gpus"] == 2 and "configs[4]" in d["config"]["workload"] and d["config"]["streams_per_gpu"] == 8
    assert d["check"]["golden"] is True and d["check"]["voxels"] == want["voxels"] and d["check"]["voxel_sha256"] == want["sha256"]
    assert len(d["partials_per_rank"]) == 2 and all(m > 0 for m in d["partials_per_rank"])
    assert d["exchange_bytes_per_step"] == d["partials_per_rank"][1] * 40
    ph = d["phases_ms"]
    assert ph["kernel"] > 0 and ph["exchange"] > 0 and ph["root_voxel"] > 0
    assert abs(d["value"] - 16 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01


@pytest.mark.gpu
def test_config5_workload_on_one_gpu_has_the_contract_keys():
    d = _bench_line("--workload", "config5", "--steps", "5", "--warmup", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and "configs[4]" in d["config"]["workload"] and d["check"]["golden"] is True
    assert d["roofline"]["kernel"] == "pcs_fused_voxel_partials_kernel" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["exchange_bytes_per_step"] == 0


@pytest.mark.gpu
def test_two_ranks_gather_compacted_payloads_with_variable_counts():
    """--mode drop_invalid at N > 1: the kept counts differ per rank, so the gather is the variable form (counts all-gathered
    from the device word the kernel wrote, grouped isend / irecv at camera-order offsets) — not full-capacity buffers."""
    d = _bench_line("--mode", "drop_invalid", "--gpus", "2", "--steps", "4", "--warmup", "2", "--preheat-ms", "20",
                    "--debug-backend", "gloo",
                    launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", "29585"))
    g = d["gather"]
    assert d["config"]["gather_to_rank0"] is True and g["form"].startswith("variable")
    kept = g["counts_per_rank"]
    assert len(kept) == 2 and all(0.85 * 4 * 1280 * 720 < c < 0.95 * 4 * 1280 * 720 for c in kept)
    assert g["bytes_per_peer_per_step"] == kept[1] * 10


# ---- libpcs_node (one process, several GPUs): the one-GPU paths on hardware --------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("one_call", [True, False])
@pytest.mark.parametrize("flags", [0, FLAG_DROP_INVALID])
def test_node_voxel_routes_agree_with_the_oracle(oracle, flags, one_call, monkeypatch):
    """A node of one peer: route PARTIALS is the rasters -> voxels call enqueued at submit (no partials leave the library, the
    stats say 0), or — PCS_NODE_ONE_CALL=0 — the partials pipeline a node of several peers runs."""
    from pointcloud_stitching_amd.node import PcsNode, VOXEL_PARTIALS, VOXEL_PAYLOADS
    if not one_call:
        monkeypatch.setenv("PCS_NODE_ONE_CALL", "0")
    cfgs, depth, color = S.synth_frame_set(4, 320, 240)
    stitched, _ = oracle.process_frames(cfgs, depth, color, flags, 1)
    with PcsNode(cfgs, devices=[0], flags=flags) as node:
        for leaf in (25, 50, 200):
            want = oracle.voxel_grid(stitched, leaf)
            for route in (VOXEL_PARTIALS, VOXEL_PAYLOADS):
                got, stats = node.process_voxel(depth, color, leaf, route)
                assert got.shape == want.shape and (got == want).all(), (leaf, route)
                assert stats["voxels"] == want.shape[0] and stats["exchanged_bytes"] == 0
                if one_call and route == VOXEL_PARTIALS:
                    assert stats["partials"] == 0 and stats["kernels_ms"] > 0
                else:
                    assert stats["partials"] > 0 and stats["kernels_ms"] >= 0 and stats["root_voxel_ms"] > 0
        # the plain stitch still works on the same node between voxel calls
        buf, counts, size = node.process(depth, color)
        assert size == stitched.nbytes and (buf[2:2 + stitched.size].reshape(-1, 5) == stitched).all()


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [50, 200])
def test_central_cli_shards_config5_over_one_gpu_and_matches_the_digests(leaf):
    """`pcs-multicamera-optimized -i synth:1920x1080 -N 16 -Z -G 1 -V <leaf>`: the node library's voxel route through the CLI
    (frame 0 of the CLI's synthetic generator is the frame the digests were made from)."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        dump = os.path.join(tmp, "voxels.bin")
        r = subprocess.run([CENTRAL, "-i", "synth:1920x1080", "-N", "16", "-Z", "-G", "1", "-V", str(leaf), "-q", "-r", "1", "-t",
                            "-o", dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Voxel grid over 1 GPU(s)" in r.stdout and "partials ->" in r.stdout
        raw = open(dump, "rb").read()
    size = int(np.frombuffer(raw[:4], np.int32)[0])
    want = GOLD["voxel"][str(leaf)]
    assert size == want["voxels"] * 10 and len(raw) == size + 4
    assert hashlib.sha256(raw[4:]).hexdigest() == want["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-Z"], ["-Z", "-V", "50"], ["-d", "3"]])
def test_central_cli_over_virtual_peers_writes_the_same_bytes_as_one_gpu(extra):
    """`pcs-multicamera-optimized -G 0,0,0,0`: four peers on GPU 0, the gather (or the voxel-partials exchange) on real RCCL as
    self send/recv pairs — byte-identical to the same cameras on one context (no -G) and on a one-peer node (-G 1)."""
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, g in (("plain", []), ("one", ["-G", "1"]), ("four", ["-G", "0,0,0,0"])):
            dump = os.path.join(tmp, tag + ".bin")
            r = subprocess.run([CENTRAL, "-i", "synth:320x240", "-N", "8", "-q", "-r", "2", "-t", "-o", dump, *g, *extra],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert r.returncode == 0, (tag, r.stderr[-2000:])
            if tag == "four":
                assert "communicator of 1 rank(s)" in r.stdout
            outs.append(open(dump, "rb").read())
    assert len(outs[0]) > 4 and outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-Z"], ["-Z", "-V", "50"]])
def test_central_cli_pipelined_frame_loop_ends_on_the_same_bytes(extra):
    """`-P`: the C++ host's own frame loop over device-resident rasters — pcs_node_submit_device(k+1); pcs_node_wait(k) (with -V
    the voxel tickets), two frame-sets in flight over three peers. Its last frame-set (loop index 8 = ring slot 2) must equal
    what the synchronous host-form loop writes for its third frame-set."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, "pipe.bin"), os.path.join(tmp, "sync.bin")
        common = [CENTRAL, "-i", "synth:320x240", "-N", "6", "-q", "-t", "-G", "0,0,0", *extra]
        r = subprocess.run([*common, "-P", "-r", "4", "-o", a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Pipelined" in r.stdout and "ms per frame-set" in r.stdout and "per frame-set: kernels" in r.stdout
        r2 = subprocess.run([*common, "-r", "3", "-o", b], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r2.returncode == 0, r2.stderr[-2000:]
        pa, pb = open(a, "rb").read(), open(b, "rb").read()
    assert len(pa) > 4 and pa == pb


# ---- no GPU needed ------------------------------------------------------------------------------------------------------------
def test_aggregate_point_count_is_refused_before_any_device_is_touched():
    """64 streams of 4096 x 4096 would stitch to 1.07 G points: the int32 byte-count header (and the 32-bit per-stream
    bases) cannot hold that. Refused as an invalid argument, by the context and by the node, without needing a device."""
    from pointcloud_stitching_amd.node import PcsNode
    cfgs = [S.synth_stream_config(4096, 4096, s) for s in range(64)]
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs)
    assert e.value.status == -1 and "int32" in str(e.value)
    with pytest.raises(PcsError) as e:
        PcsNode(cfgs, devices=[0, 1, 2, 3], flags=0)
    assert e.value.status == -1 and "int32" in str(e.value)
    ok = [S.synth_stream_config(1920, 1080, s) for s in range(64)]               # 132.7 M points: fits; fails later, on the device
    try:
        PcsContext(ok).close()
    except PcsError as ex:
        assert ex.status == -2                                                    # no device here: that, and only that


_RCCL_WORLD1 = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
from pointcloud_stitching_amd.stitch import RankStitcher, ShardedVoxelGrid
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID
from oracle import pcs_oracle as O
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfgs, depth, color = S.synth_frame_set(3, 320, 240)
stitched, counts = O.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
with PcsContext(cfgs, device=0, flags=FLAG_DROP_INVALID) as ctx:
    ctx.set_stream(stream.cuda_stream)
    dd = [torch.from_numpy(d.reshape(-1).view(np.uint8)).to(dev) for d in depth]
    dc = [torch.from_numpy(c).to(dev) for c in color]
    n_max = sum(c.n_points for c in cfgs)
    pay = torch.zeros(n_max * 5, dtype=torch.int16, device=dev)
    cnt = torch.zeros(len(cfgs) + 1, dtype=torch.int32, device=dev)
    ctx.process_frames_device([t.data_ptr() for t in dd], [t.data_ptr() for t in dc], pay.data_ptr(), pay.numel(), cnt.data_ptr())
    st = RankStitcher()
    assert (st.rank, st.world, st.host_staged) == (0, 1, False)
    # fixed-size gather through RCCL (asynchronous form, as bench.py drives it), variable gather, counts from a device word
    out = torch.zeros_like(pay)
    st.gather_fixed(pay, out, async_op=True).wait()
    out2 = torch.zeros_like(pay)
    got_counts = st.gather_variable(pay, cnt[len(cfgs)], out2)
    torch.cuda.synchronize()
    total = int(cnt[len(cfgs)].item())
    assert got_counts == [total] and total * 5 == stitched.size
    assert (out[:stitched.size].cpu().numpy() == stitched.reshape(-1)).all()
    assert (out2[:stitched.size].cpu().numpy() == stitched.reshape(-1)).all()
    # the config-5 pipeline object on the RCCL group
    vox = torch.zeros(n_max * 5, dtype=torch.int16, device=dev)
    sv = ShardedVoxelGrid(ctx, 50, dev)
    sv.run([t.data_ptr() for t in dd], [t.data_ptr() for t in dc], vox.data_ptr(), vox.numel())
    torch.cuda.synchronize()
    nv = int(sv.n_vox[0].item())
    want = O.voxel_grid(stitched, 50)
    assert nv == want.shape[0] and (vox[:nv * 5].cpu().numpy().reshape(-1, 5) == want).all()
    assert sv.voxels(vox.data_ptr(), vox.numel()) == nv
    # a flagged bucket tail (fault injection): the device word says -1, voxels() runs the reduce again on the LSD tail
    import os
    os.environ["PCS_VOXEL_TAIL"] = "bucket"
    ctx.inject_voxel_stall(1)
    sv.run([t.data_ptr() for t in dd], [t.data_ptr() for t in dc], vox.data_ptr(), vox.numel())
    torch.cuda.synchronize()
    assert int(sv.n_vox[0].item()) == -1
    assert sv.voxels(vox.data_ptr(), vox.numel()) == nv and ctx.voxel_tail_reruns() == 1
    assert (vox[:nv * 5].cpu().numpy().reshape(-1, 5) == want).all()
    ctx.inject_voxel_stall(0)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


@pytest.mark.gpu
def test_exchange_classes_on_a_real_rccl_group_of_one():
    """Everything at N > 1 is tested with gloo (host-staged). This is the part of the RCCL route one GPU can exercise: a real
    "nccl" process group (world 1) under RankStitcher / ShardedVoxelGrid — the group comes up in this environment, the
    collectives take the dtypes and views they are handed, and stream ordering between the library's kernels and torch's
    collectives holds."""
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1, ROOT, "29611"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
