"""Voxel-grid downsample (BASELINE config 5). The op is defined by this build (the reference only #includes
PCL's header); the oracle here is the CPU statement of that definition, cross-checked against a numpy
formulation, and the HIP path must match it bit for bit."""
import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext, PcsError
from pointcloud_stitching_amd.types import FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID, FLAG_FORCE_IEEE


@pytest.fixture(autouse=True, params=["bucket", "bucket-cold", "lsd"])
def voxel_tail(request, monkeypatch):
    """Every test of this file runs with each tail of the voxel pipeline forced (PCS_VOXEL_TAIL / PCS_VOXEL_REGIONS are read at
    every call): the bucket tail as it runs by default — its first call on a context partitions the partials by key splitters
    (histogram, column scan, scatter: 5 launches), every later one has the pre-aggregation fill the buckets' regions itself (2
    launches) —, the bucket tail held to the cold chain, and the LSD radix sort + segmented mean (12 launches). Same bytes,
    whatever the leaf and the input."""
    monkeypatch.setenv("PCS_VOXEL_TAIL", "bucket" if request.param.startswith("bucket") else request.param)
    monkeypatch.setenv("PCS_VOXEL_REGIONS", "0" if request.param == "bucket-cold" else "1")
    return request.param


def numpy_voxel_grid(p, leaf):
    p = np.asarray(p, np.int16).reshape(-1, 5)
    if p.shape[0] == 0:
        return p.copy()
    xyz = p[:, :3].astype(np.int64)
    vox = np.floor_divide(xyz, leaf) + 32768
    key = (vox[:, 2] << 34) | (vox[:, 1] << 17) | vox[:, 0]
    order = np.argsort(key, kind="stable")
    key = key[order]; q = p[order]
    starts = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    cnt = np.diff(np.r_[starts, key.size])
    def seg(a):
        return np.add.reduceat(a.astype(np.int64), starts)
    c = q[:, 3].view(np.uint16).astype(np.int64)
    out = np.zeros((starts.size, 5), np.int16)
    for k in range(3):
        s = seg(q[:, k])
        out[:, k] = (np.sign(s) * (np.abs(s) // cnt)).astype(np.int16)          # C division truncates toward zero
    r, g, b = seg(c & 0xFF) // cnt, seg(c >> 8) // cnt, seg(q[:, 4].view(np.uint16).astype(np.int64) & 0xFF) // cnt
    out[:, 3] = (r | (g << 8)).astype(np.uint16).view(np.int16)
    out[:, 4] = b.astype(np.int16)
    return out


def random_payload(n, seed, span=3000):
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 5), np.int16)
    p[:, :3] = rng.integers(-span, span, (n, 3))
    p[:, 3] = rng.integers(0, 65536, n, dtype=np.uint16).view(np.int16)
    p[:, 4] = rng.integers(0, 256, n)
    return p


@pytest.mark.parametrize("n,leaf,span", [(0, 10, 100), (1, 10, 100), (1000, 1, 50), (5000, 37, 3000), (20000, 250, 32767),
                                         (4096, 32767, 32767)])
def test_oracle_matches_numpy_formulation(oracle, n, leaf, span):
    p = random_payload(n, 3 + n, span)
    if n > 8:
        p[:4, :3] = [(-32768, -32768, -32768), (32767, 32767, 32767), (-1, 0, 1), (-leaf, leaf - 1, -leaf - 1)]
    got = oracle.voxel_grid(p, leaf)
    assert (got == numpy_voxel_grid(p, leaf)).all()
    assert got.shape[0] <= max(n, 0)


def test_oracle_properties(oracle):
    p = random_payload(30000, 9, 2000)
    v = oracle.voxel_grid(p, 100)
    assert 0 < v.shape[0] < p.shape[0]
    # idempotence at the same leaf: every voxel already holds exactly one point, which is its own mean
    assert (oracle.voxel_grid(v, 100) == v).all()
    # leaf 1: every distinct (x,y,z) is its own voxel; coordinates survive unchanged
    d = oracle.voxel_grid(p, 1)
    assert d.shape[0] == np.unique(p[:, :3], axis=0).shape[0]
    # permutation invariance (integer sums)
    assert (oracle.voxel_grid(p[::-1].copy(), 100) == v).all()


def test_float_voxel_index_is_exact_for_every_coordinate():
    """The kernels' voxel index (pcs_voxel_agg.h: VoxelDiv) is (unsigned)fmaf((float)v, fl(1/leaf), fl((bias*leaf + 0.5)/leaf)).
    Restated here in numpy: v * inv + c is exact in float64 (40 + 24 significant bits, close exponents), so one rounding to
    float32 is the fused multiply-add's single rounding. Every int16 v, every leaf up to 2048, every 97th above and the ends
    (the full 32 767 x 65 536 sweep, run once in C with fmaf: 0 mismatches)."""
    v = np.arange(-32768, 32768, dtype=np.int64)
    leaves = list(range(1, 2049)) + list(range(2049, 32768, 97)) + [32766, 32767]
    for leaf in leaves:
        bias_leaf = (32768 + leaf - 1) // leaf * leaf
        inv = np.float32(1.0 / leaf); c = np.float32((bias_leaf + 0.5) / leaf)
        q = (v.astype(np.float64) * np.float64(inv) + np.float64(c)).astype(np.float32)
        got = q.astype(np.int64)                       # truncation; q > 0
        want = (v + bias_leaf) // leaf
        assert (got == want).all(), leaf


@pytest.mark.gpu
@pytest.mark.parametrize("n,leaf,span", [(0, 10, 100), (1, 10, 100), (1000, 1, 50), (5000, 37, 3000), (200000, 250, 32767),
                                         (100000, 32767, 32767), (300001, 20, 400)])
def test_gpu_voxel_grid_matches_oracle(oracle, n, leaf, span):
    p = random_payload(n, 11 + n, span)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        got = ctx.voxel_grid(p, leaf)
        again = ctx.voxel_grid(p[::-1].copy(), leaf)
    want = oracle.voxel_grid(p, leaf)
    assert got.shape == want.shape and (got == want).all()
    assert (again == want).all()


@pytest.mark.gpu
def test_gpu_voxel_grid_preaggregation_paths(oracle):
    """The pre-aggregation kernel's special paths: wavefront-uniform runs (reduced across the wavefront, lane 0
    adds), the same with a FULL hash table (the whole wavefront must pass its points through), ragged tails."""
    rng = np.random.default_rng(5)
    n = 3 * 8192 + 777
    p = np.zeros((n, 5), np.int16)
    p[:, 3] = rng.integers(0, 65536, n).astype(np.uint16).view(np.int16)
    p[:, 4] = rng.integers(0, 256, n)
    # workgroup 0: 4096 points in distinct voxels fill the 2048-slot table, then 4096 points inside ONE new voxel
    p[:4096, 0] = (np.arange(4096) % 64) * 100 - 3200
    p[:4096, 1] = (np.arange(4096) // 64) * 100 - 3200
    p[:4096, 2] = 0
    p[4096:8192, :3] = rng.integers(20000, 20090, (4096, 3))
    # workgroup 1: long runs, a few voxels in total (large-leaf case), with a run boundary inside a wavefront
    p[8192:16384, 0] = np.repeat(np.arange(8192 // 500 + 1) * 100, 500)[:8192] - 900
    p[8192:16384, 1:3] = rng.integers(0, 50, (8192, 2))
    # workgroup 2 + ragged tail: image-like smooth surface with noise
    m = n - 16384
    p[16384:, 0] = np.linspace(-3000, 3000, m).astype(np.int16)
    p[16384:, 1] = 5
    p[16384:, 2] = 2000 + rng.integers(-8, 9, m)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for leaf in (100, 25, 3000):
            got = ctx.voxel_grid(p, leaf)
            want = oracle.voxel_grid(p, leaf)
            assert got.shape == want.shape and (got == want).all(), leaf


@pytest.mark.gpu
def test_config5_pipeline_compaction_then_voxel_grid(oracle):
    """BASELINE configs[4] in miniature: 1080p-shaped streams, invalid-depth compaction, voxel grid on the
    stitched cloud — all on the device, compared end to end with the oracle."""
    cfgs, depth, color = S.synth_frame_set(4, 480, 270)
    stitched, counts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    want = oracle.voxel_grid(stitched, 25)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd = [ctx.device_malloc(d.nbytes) for d in depth]
        dc = [ctx.device_malloc(c.nbytes) for c in color]
        for ptr, a in zip(dd + dc, depth + color):
            ctx.memcpy_h2d(ptr, a)
        n_max = sum(c.n_points for c in cfgs)
        d_pay = ctx.device_malloc(n_max * 10 + 64)
        d_cnt = ctx.device_malloc(4 * 5)
        d_vox = ctx.device_malloc(n_max * 10 + 64)
        d_nv = ctx.device_malloc(4)
        ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
        ctx.synchronize()
        cnt = np.empty(5, np.int32); ctx.memcpy_d2h(cnt, d_cnt)
        assert list(cnt[:4]) == counts
        ctx.voxel_grid_device(d_pay, int(cnt[4]), 25, d_vox, n_max * 5, d_nv)
        ctx.synchronize()
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
        got = np.empty((int(nv[0]), 5), np.int16); ctx.memcpy_d2h(got, d_vox)
        assert got.shape == want.shape and (got == want).all()
        # and the counted form: the point count is taken from the compaction's device-side total
        ctx.memcpy_h2d(d_nv, np.zeros(1, np.int32))
        ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
        ctx.voxel_grid_device_counted(d_pay, d_cnt + 4 * 4, n_max, 25, d_vox, n_max * 5, d_nv)
        ctx.synchronize()
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
        got = np.empty((int(nv[0]), 5), np.int16); ctx.memcpy_d2h(got, d_vox)
    assert got.shape == want.shape and (got == want).all()
    assert want.shape[0] < stitched.shape[0]


@pytest.mark.gpu
def test_gpu_voxel_grid_argument_errors():
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    p = random_payload(10, 1)
    with PcsContext(cfgs) as ctx:
        with pytest.raises(PcsError) as e:
            ctx.voxel_grid(p, 0)
        assert e.value.status == -1
        with pytest.raises(PcsError) as e:
            ctx.voxel_grid(p, 40000)
        assert e.value.status == -1


@pytest.mark.gpu
def test_config5_full_size_against_oracle_digests():
    """BASELINE configs[4] at FULL size on one GPU: 16 x 1920x1080 synthetic streams -> invalid-depth compaction ->
    camera-order stitch -> voxel grid (50 mm and 200 mm) of the 29.8 M-point cloud, all device-resident. The CPU oracle
    needs about a minute for this, so its outputs are pinned as SHA-256 digests by
    tests/golden/make_config5_golden.py (committed, re-runnable) and compared here."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5_digests.json")))
    cfgs, depth, color = S.synth_frame_set(16, 1920, 1080)
    n_max = sum(c.n_points for c in cfgs)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd = [ctx.device_malloc(d.nbytes) for d in depth]
        dc = [ctx.device_malloc(c.nbytes) for c in color]
        for ptr, a in zip(dd + dc, depth + color):
            ctx.memcpy_h2d(ptr, a)
        d_pay = ctx.device_malloc(n_max * 10 + 64)
        d_cnt = ctx.device_malloc(4 * 17)
        ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
        ctx.synchronize()
        cnt = np.empty(17, np.int32); ctx.memcpy_d2h(cnt, d_cnt)
        assert list(cnt[:16]) == gold["counts"] and int(cnt[16]) == gold["points"]
        stitched = np.empty(int(cnt[16]) * 5, np.int16); ctx.memcpy_d2h(stitched, d_pay)
        assert hashlib.sha256(stitched.tobytes()).hexdigest() == gold["stitched_sha256"]
        d_vox = ctx.device_malloc(n_max * 10 + 64)
        d_nv = ctx.device_malloc(4)
        for leaf, want in sorted(gold["voxel"].items()):
            # the whole of config 5 as two asynchronous calls: the voxel grid reads the kept total from the device
            ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
            ctx.voxel_grid_device_counted(d_pay, d_cnt + 4 * 16, n_max, int(leaf), d_vox, n_max * 5, d_nv)
            ctx.synchronize()
            nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
            assert int(nv[0]) == want["voxels"], leaf
            got = np.empty(int(nv[0]) * 5, np.int16); ctx.memcpy_d2h(got, d_vox)
            assert hashlib.sha256(got.tobytes()).hexdigest() == want["sha256"], leaf


def _upload_rasters(ctx, depth, color):
    dd = [ctx.device_malloc(max(d.nbytes, 16)) for d in depth]
    dc = [ctx.device_malloc(max(c.nbytes, 16)) for c in color]
    for ptr, a in zip(dd + dc, list(depth) + list(color)):
        ctx.memcpy_h2d(ptr, a)
    return dd, dc


def _rasters_to_voxels(ctx, dd, dc, leaf, n_max):
    d_vox = ctx.device_malloc(n_max * 10 + 64)
    d_nv = ctx.device_malloc(4)
    ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
    ctx.synchronize()
    nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
    got = np.empty(max(int(nv[0]), 1) * 5, np.int16); ctx.memcpy_d2h(got, d_vox)
    ctx.device_free(d_vox); ctx.device_free(d_nv)
    return got[:int(nv[0]) * 5].reshape(-1, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT | FLAG_DROP_INVALID,
                                   FLAG_DROP_INVALID | FLAG_FORCE_IEEE])
@pytest.mark.parametrize("shapes", [[(640, 480)] * 2, [(1280, 720), (321, 243), (64, 48)], [(8, 1)], [(1920, 1080)]])
def test_rasters_to_voxel_grid_equals_stitch_then_voxel_grid(oracle, flags, shapes):
    """pcs_process_frames_voxel_device never writes the stitched cloud; its voxels must be exactly those of the oracle's
    voxel grid over the oracle's stitched cloud (same flags), for leaves from 'every point its own voxel' to 'one voxel'.
    Widths that are multiples of 8 are read in 64 x 64 squares (640 x 480: ragged last patch row; 1920 x 1080: many
    patches; 8 x 1: smaller than one lane's 8 pixels); the mixed set holds a 321-wide raster, so the whole frame-set is
    read in runs of consecutive pixels and, below 36 mm, through the internal stitched cloud."""
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    n_max = sum(c.n_points for c in cfgs)
    stitched, _ = oracle.process_frames(cfgs, depth, color, flags)
    with PcsContext(cfgs, flags=flags) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        for leaf in (1, 7, 36, 50, 200, 32767):
            got = _rasters_to_voxels(ctx, dd, dc, leaf, n_max)
            want = oracle.voxel_grid(stitched, leaf)
            assert got.shape == want.shape and (got == want).all(), (leaf, got.shape, want.shape)


@pytest.mark.gpu
def test_rasters_to_voxel_grid_nothing_kept_and_stride(oracle):
    cfgs, depth, color = S.synth_frame_set(3, 320, 240)
    n_max = sum(c.n_points for c in cfgs)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd, dc = _upload_rasters(ctx, [np.zeros_like(d) for d in depth], color)
        assert _rasters_to_voxels(ctx, dd, dc, 50, n_max).shape[0] == 0
        with pytest.raises(PcsError):
            ctx.process_frames_voxel_device(dd, dc, 0, dd[0], n_max * 5)
        with pytest.raises(PcsError):
            ctx.process_frames_voxel_device(dd, dc, 50, dd[0], n_max * 5 - 1)
    # with a stride the stitched cloud is built internally (the stride is defined on the order of the kept points)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID, downsample=3) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, downsample=3)
        got = _rasters_to_voxels(ctx, dd, dc, 100, n_max)
        want = oracle.voxel_grid(stitched, 100)
        assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_rasters_to_voxel_grid_full_table_passes_points_through(oracle):
    """Uniformly random depth (0 .. 65 m): neighbouring pixels are metres apart, so practically every pixel is its own
    voxel and the 2048-slot LDS table of a workgroup overflows: most runs take the pass-through route (square-patch
    reader: the raster's width is a multiple of 8)."""
    cfgs = [S.synth_stream_config(640, 480, 0)]
    rng = np.random.default_rng(5)
    depth = [rng.integers(0, 65536, 640 * 480, dtype=np.uint16)]
    color = [S.synth_color(640, 480, 0)]
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        for leaf in (1, 3, 40, 64):
            got = _rasters_to_voxels(ctx, dd, dc, leaf, cfgs[0].n_points)
            want = oracle.voxel_grid(stitched, leaf)
            assert got.shape == want.shape and (got == want).all(), leaf


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [FLAG_DROP_INVALID, FLAG_DROP_INVALID | FLAG_FORCE_IEEE])
def test_rasters_to_voxel_grid_coordinates_that_leave_int16(oracle, flags):
    """A smooth wall whose depth ramps across 32.768 m along every row: inside one lane's 8 pixels some world coordinates (in mm) still fit
    int16 and the next ones do not — the record keeps their low 16 bits, so they wrap to -32768 and land in voxels at the other end of
    the grid. The raster reader works on widened coordinates and must send exactly those lanes through its exact path."""
    w, h = 640, 136
    cfgs = [S.synth_stream_config(w, h, 0)]
    ramp = (32600 + (np.arange(w) * 400) // w).astype(np.uint16)                 # 32 600 .. 32 999 mm at depth_scale 0.001
    depth = [np.ascontiguousarray(np.tile(ramp, h))]
    depth[0][::97] = 0
    color = [S.synth_color(w, h, 0)]
    stitched, _ = oracle.process_frames(cfgs, depth, color, flags)
    xyz = stitched.reshape(-1, 5)[:, :3]
    assert ((xyz < -30000).any(0) & (xyz > 30000).any(0)).any()                  # the scene does straddle the wrap
    with PcsContext(cfgs, flags=flags) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        for leaf in (5, 50, 300):
            got = _rasters_to_voxels(ctx, dd, dc, leaf, cfgs[0].n_points)
            want = oracle.voxel_grid(stitched, leaf)
            assert got.shape == want.shape and (got == want).all(), leaf


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [12, 60])
def test_rasters_to_voxel_grid_more_streams_than_one_launch(oracle, leaf):
    """20 cameras = two launches of the raster reader (16 + 4) appending to the same partial arrays."""
    shapes = [(64, 48)] * 19 + [(128, 96)]
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    n_max = sum(c.n_points for c in cfgs)
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    want = oracle.voxel_grid(stitched, leaf)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        got = _rasters_to_voxels(ctx, dd, dc, leaf, n_max)
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_rasters_to_voxel_grid_random_configurations(oracle):
    """Random raster sizes (multiples of 8 or not, down to a single row / column), leaves, flags and scenes."""
    rng = np.random.default_rng(20260928)
    for trial in range(40):
        n = int(rng.integers(1, 4))
        shapes = []
        for s in range(n):
            w = int(rng.choice([8, 16, 64, 72, 200, 320, 333, 640, 1]) if rng.random() < 0.7 else rng.integers(1, 700))
            h = int(rng.choice([1, 2, 63, 64, 65, 129, 240]) if rng.random() < 0.7 else rng.integers(1, 300))
            shapes.append((max(w, 2) if h == 1 else w, h))                       # pcs_create refuses a colour raster of < 4 bytes
        if rng.random() < 0.5:
            shapes = [(w - w % 8 if w >= 8 else 8, h) for w, h in shapes]        # all multiples of 8: the square-patch reader
        flags = int(rng.choice([0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID]))
        leaf = int(rng.choice([1, 2, 5, 13, 29, 30, 36, 50, 77, 200, 1000, 32767]))
        cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
        depth = []
        for s, (w, h) in enumerate(shapes):
            kind = rng.random()
            if kind < 0.6:
                d = S.synth_depth(w, h, s, seed=int(rng.integers(1, 1 << 30)))
            elif kind < 0.8:
                d = rng.integers(0, 65536, w * h, dtype=np.uint16)
            else:
                d = np.full(w * h, int(rng.integers(0, 3000)), np.uint16)              # a wall (or nothing at all)
            depth.append(np.ascontiguousarray(d, np.uint16).reshape(-1))
        color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
        n_max = sum(c.n_points for c in cfgs)
        stitched, _ = oracle.process_frames(cfgs, depth, color, flags)
        want = oracle.voxel_grid(stitched, leaf)
        with PcsContext(cfgs, flags=flags) as ctx:
            dd, dc = _upload_rasters(ctx, depth, color)
            got = _rasters_to_voxels(ctx, dd, dc, leaf, n_max)
        assert got.shape == want.shape and (got == want).all(), (trial, shapes, flags, leaf, got.shape, want.shape)


@pytest.mark.gpu
def test_config5_full_size_rasters_to_voxels_digests():
    """BASELINE configs[4] at full size through the one-call form: 16 x 1920x1080 rasters -> voxel grid, no stitched
    cloud in between; same pinned oracle digests as the two-call form above."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5_digests.json")))
    cfgs, depth, color = S.synth_frame_set(16, 1920, 1080)
    n_max = sum(c.n_points for c in cfgs)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        for leaf, want in sorted(gold["voxel"].items()):
            got = _rasters_to_voxels(ctx, dd, dc, int(leaf), n_max)
            assert got.shape[0] == want["voxels"], leaf
            assert hashlib.sha256(got.tobytes()).hexdigest() == want["sha256"], leaf


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 4095, 4096, 4097, 8191, 8193, 70001])
def test_gpu_voxel_grid_sizes_around_the_kernels_block_borders(oracle, n):
    """n = 1 takes the narrow-load kernel; an even LAST record (odd n) is passed through as a single-point partial by the
    wide-load kernel; 256 / 4096 / 8192 are the block sizes of the segmented mean, the sort chunks and the pre-aggregation."""
    p = random_payload(n, 700 + n, 600)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for leaf in (1, 64, 5000):
            got = ctx.voxel_grid(p, leaf)
            want = oracle.voxel_grid(p, leaf)
            assert got.shape == want.shape and (got == want).all(), (n, leaf)


@pytest.mark.gpu
def test_gpu_voxel_grid_long_runs_cross_many_blocks(oracle):
    """Runs of equal keys far longer than a 256-element block of the segmented mean: every pre-aggregation workgroup
    contributes a partial to the SAME few voxels, so after the sort a run spans dozens of blocks (lead / trail pieces
    and the fix-up kernel), and the sums exceed 32 bits (64-bit accumulation across blocks)."""
    rng = np.random.default_rng(17)
    n = 120 * 8192 + 5
    p = np.zeros((n, 5), np.int16)
    p[:, 0] = rng.integers(30000, 32767, n)                # large coordinates: coordinate sums pass 2^31 quickly
    p[:, 1] = rng.integers(-32768, -30000, n)
    p[:, 2] = rng.integers(0, 3, n) * 1000                 # three voxels along z at leaf 5000
    p[:, 3] = rng.integers(0, 65536, n).astype(np.uint16).view(np.int16)
    p[:, 4] = rng.integers(0, 256, n)
    # plus a sprinkle of isolated points so that short and long runs are mixed in the sorted order
    p[::977, :3] = rng.integers(-20000, 20000, (p[::977].shape[0], 3))
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for leaf in (1000, 5000, 32767):
            got = ctx.voxel_grid(p, leaf)
            want = oracle.voxel_grid(p, leaf)
            assert got.shape == want.shape and (got == want).all(), leaf


@pytest.mark.gpu
def test_gpu_voxel_grid_device_unaligned_payload_and_async(oracle):
    """Device API: a payload that starts 2 bytes off a dword (the reference's buffer + 2 shorts convention shifted once
    more) takes the narrow-load pre-aggregation; two calls queued back to back without synchronising in between (the call
    no longer waits for the host anywhere) must both be right."""
    p = random_payload(50001, 23, 1500)
    want = {leaf: oracle.voxel_grid(p, leaf) for leaf in (10, 300)}
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        d_in = ctx.device_malloc(p.nbytes + 64)
        outs = {leaf: ctx.device_malloc(p.nbytes + 64) for leaf in want}
        cnts = {leaf: ctx.device_malloc(4) for leaf in want}
        for skew in (0, 2):
            ctx.memcpy_h2d(d_in + skew, p)
            for leaf in want:                                  # queued back to back
                ctx.voxel_grid_device(d_in + skew, p.shape[0], leaf, outs[leaf], p.size, cnts[leaf])
            ctx.synchronize()
            for leaf in want:
                nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, cnts[leaf])
                assert int(nv[0]) == want[leaf].shape[0], (skew, leaf)
                got = np.empty((int(nv[0]), 5), np.int16); ctx.memcpy_d2h(got, outs[leaf])
                assert (got == want[leaf]).all(), (skew, leaf)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 2, 777, 8192])
def test_gpu_voxel_grid_counted_reads_the_count_from_the_device(oracle, n):
    """pcs_voxel_grid_device_counted with a capacity far above the live count (incl. 0 and 1 live points)."""
    cap = 20000
    p = random_payload(cap, 41, 900)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        d_in = ctx.device_malloc(p.nbytes + 64); ctx.memcpy_h2d(d_in, p)
        d_out = ctx.device_malloc(p.nbytes + 64)
        d_n = ctx.device_malloc(4); ctx.memcpy_h2d(d_n, np.array([n], np.int32))
        d_nv = ctx.device_malloc(4); ctx.memcpy_h2d(d_nv, np.array([-7], np.int32))
        ctx.voxel_grid_device_counted(d_in, d_n, cap, 120, d_out, cap * 5, d_nv)
        ctx.synchronize()
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
        want = oracle.voxel_grid(p[:n], 120)
        assert int(nv[0]) == want.shape[0]
        if want.shape[0]:
            got = np.empty((int(nv[0]), 5), np.int16); ctx.memcpy_d2h(got, d_out)
            assert (got == want).all()


@pytest.mark.gpu
def test_voxel_entry_forms_interleaved_on_one_context_without_synchronising(oracle):
    """Every voxel call hands the next one its control block (two blocks at the start of the workspace, cleared by the
    previous call's block-scan kernel instead of a memset launch). One context, one stream, no host synchronisation between
    calls: payloads of growing and shrinking sizes (the workspace is re-allocated in between), the raster form, the partials
    exchange format, calls that fail their argument checks in the middle, and the host-pointer form — each against the oracle."""
    rng = np.random.default_rng(77)
    cfgs, depth, color = S.synth_frame_set(3, 320, 240)
    n_max = sum(c.n_points for c in cfgs)
    flags = FLAG_DROP_INVALID
    stitched, _ = oracle.process_frames(cfgs, depth, color, flags)
    with PcsContext(cfgs, flags=flags) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        sizes = [500, 70001, 3, 8192, 200000, 1, 4097, 150000]
        clouds = [random_payload(n, 100 + i, span=int(rng.choice([60, 3000, 32767]))) for i, n in enumerate(sizes)]
        big = max(max(sizes), n_max)
        d_in = [ctx.device_malloc(c.nbytes + 64) for c in clouds]
        for p, c in zip(d_in, clouds):
            ctx.memcpy_h2d(p, c)
        d_k = ctx.device_malloc(n_max * 8 + 64); d_p = ctx.device_malloc(n_max * 32 + 64); d_np = ctx.device_malloc(64)
        jobs = []                                    # (what, leaf, device output, device count, expected)
        def out_pair(n):
            return ctx.device_malloc(max(n, 1) * 10 + 64), ctx.device_malloc(64)
        for rep in range(3):
            for i, (n, c) in enumerate(zip(sizes, clouds)):
                leaf = int(rng.choice([1, 9, 37, 50, 250, 4000]))
                d_o, d_n = out_pair(n)
                ctx.voxel_grid_device(d_in[i], n, leaf, d_o, max(n, 1) * 5, d_n)
                jobs.append((f"payload {n}", leaf, d_o, d_n, oracle.voxel_grid(c, leaf)))
                if i % 3 == 0:                       # rasters -> voxels on the same workspace
                    leaf = int(rng.choice([12, 36, 60, 500]))
                    d_o, d_n = out_pair(n_max)
                    ctx.process_frames_voxel_device(dd, dc, leaf, d_o, n_max * 5, d_n)
                    jobs.append(("rasters", leaf, d_o, d_n, oracle.voxel_grid(stitched, leaf)))
                if i % 3 == 1:                       # the exchange format: partials, then the grid from them (count on the device)
                    leaf = int(rng.choice([7, 45, 300]))
                    d_o, d_n = out_pair(n_max)
                    ctx.process_frames_voxel_partials_device(dd, dc, leaf, d_k, d_p, n_max, d_np)
                    ctx.voxel_grid_from_partials_device(d_k, d_p, n_max, leaf, d_o, n_max * 5, d_n, d_n_partials=d_np)
                    jobs.append(("partials", leaf, d_o, d_n, oracle.voxel_grid(stitched, leaf)))
                if i % 4 == 2:                       # refused before anything is enqueued: must not disturb the hand-over
                    with pytest.raises(PcsError):
                        ctx.voxel_grid_device(d_in[i], n, 0, d_o, max(n, 1) * 5, d_n)
                    with pytest.raises(PcsError):
                        ctx.voxel_grid_device(d_in[i], n, 50, d_o, 1, d_n)
        ctx.synchronize()
        for what, leaf, d_o, d_n, want in jobs:
            nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_n)
            got = np.empty(max(int(nv[0]), 1) * 5, np.int16); ctx.memcpy_d2h(got, d_o)
            got = got[:int(nv[0]) * 5].reshape(-1, 5)
            assert got.shape == want.shape and (got == want).all(), (what, leaf, got.shape, want.shape)
        # the host-pointer form shares the workspace too
        got = ctx.voxel_grid(clouds[1], 50)
        want = oracle.voxel_grid(clouds[1], 50)
        assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_bucket_tail_stale_splitters_and_crowded_buckets(oracle, voxel_tail):
    """The bucket tail's splitters come from the PREVIOUS call on the context (they only decide balance). Worst cases for that:
    (1) cloud A = two far-apart clusters, then cloud B = 400 k distinct voxels that all lie BETWEEN A's clusters, i.e. inside one
    of A's key ranges — that bucket must be worked off in many passes over key sub-ranges; (2) then A again (B's splitters are
    useless for it); (3) a cloud with ~2 M distinct voxels, twice what 1024 tables of 1024 slots hold, whatever the splitters;
    (4) one voxel that collects 300 k points beside isolated ones (no skew: only distinct keys take slots). Every result against
    the oracle, on ONE context without synchronising in between."""
    rng = np.random.default_rng(91)
    def cloud(n, lo, hi):
        p = np.zeros((n, 5), np.int16)
        p[:, :3] = rng.integers(lo, hi, (n, 3))
        p[:, 3] = rng.integers(0, 65536, n).astype(np.uint16).view(np.int16)
        p[:, 4] = rng.integers(0, 256, n)
        return p
    a = np.concatenate([cloud(60000, -30000, -29000), cloud(60000, 29000, 30000)])
    bmid = cloud(400000, -4000, 4000)
    hot = np.concatenate([cloud(300000, 1000, 1040), cloud(5000, -32768, 32767)])
    many = cloud(2000000, -6000, 6000)
    leaf = 40
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for name, p, lf in (("A", a, leaf), ("B", bmid, leaf), ("A", a, leaf), ("hot", hot, leaf), ("B", bmid, leaf),
                            ("many", many, 32), ("A", a, 32), ("hot", hot, 32767)):
            got = ctx.voxel_grid(p, lf)
            want = oracle.voxel_grid(p, lf)
            assert got.shape == want.shape and (got == want).all(), (name, lf, voxel_tail)


@pytest.mark.gpu
def test_bucket_regions_that_overflow_or_do_not_fit(oracle, voxel_tail):
    """Warm bucket calls: the pre-aggregation puts every partial into its bucket's region — sized, like the splitters, by the
    PREVIOUS call on the context. What can go wrong with that, on one context, every result against the oracle: (1) the regions
    the previous (larger) cloud asked for do not fit this call's smaller workspace carve: nothing goes to a region, every partial
    takes the general list and every bucket gathers its own; (2) part of the cloud grows more than twofold — a dense slab of
    distinct voxels appears inside the old key range: those buckets' regions fill up, the rest overflows to the list, is
    gathered, and the buckets (beyond 8192 partials) are split into key ranges over both stretches; (3) the slab disappears again
    (regions mostly empty); (4) the same cloud twice (steady state)."""
    rng = np.random.default_rng(4242)
    def cloud(n, lo, hi):
        p = np.zeros((n, 5), np.int16)
        p[:, :3] = rng.integers(lo, hi, (n, 3))
        p[:, 3] = rng.integers(0, 65536, n).astype(np.uint16).view(np.int16)
        p[:, 4] = rng.integers(0, 256, n)
        return p
    base = cloud(300000, -2000, 2000)
    slab = np.concatenate([base, cloud(300000, -600, 600)])
    leaf = 16
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for name, p in (("slab", slab), ("base", base), ("base", base), ("slab", slab), ("slab", slab), ("base", base)):
            got = ctx.voxel_grid(p, leaf)
            want = oracle.voxel_grid(p, leaf)
            assert got.shape == want.shape and (got == want).all(), (name, voxel_tail)


@pytest.mark.gpu
def test_bucket_tail_is_the_default_at_config5_leaf_and_counts_its_launches(oracle, monkeypatch):
    """Without PCS_VOXEL_TAIL the 50 mm call takes the bucket tail: pcs_kernel_times_ms... is not what says so — the result is
    the same either way — so this checks the one observable: both forced forms and the default agree with the oracle on the
    config-5 shape at reduced size, interleaved on one context (the tails share the workspace and its control blocks)."""
    cfgs, depth, color = S.synth_frame_set(4, 640, 480)
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    want = oracle.voxel_grid(stitched, 50)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        dd, dc = _upload_rasters(ctx, depth, color)
        n_max = sum(c.n_points for c in cfgs)
        for tail in (None, "lsd", "bucket", None, "bucket", "lsd"):
            if tail is None:
                monkeypatch.delenv("PCS_VOXEL_TAIL", raising=False)
            else:
                monkeypatch.setenv("PCS_VOXEL_TAIL", tail)
            got = _rasters_to_voxels(ctx, dd, dc, 50, n_max)
            assert got.shape == want.shape and (got == want).all(), tail


@pytest.mark.gpu
def test_voxel_tail_preference_per_context(oracle, monkeypatch):
    """pcs_set_voxel_tail: a context's standing choice of tail (libpcs_node sets LSD for a root that holds every camera);
    PCS_VOXEL_TAIL still overrides. Same bytes whatever is chosen; an unknown value is refused."""
    monkeypatch.delenv("PCS_VOXEL_TAIL", raising=False)
    p = random_payload(120000, 5, 2500)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        for tail, leaf in ((1, 20), (2, 60), (0, 60), (1, 60), (2, 20)):
            ctx.set_voxel_tail(tail)
            got = ctx.voxel_grid(p, leaf)
            want = oracle.voxel_grid(p, leaf)
            assert got.shape == want.shape and (got == want).all(), (tail, leaf)
        with pytest.raises(PcsError) as e:
            ctx.set_voxel_tail(7)
        assert e.value.status == -1


def _points_as_partials(p, leaf):
    """Every point its own partial, in the exchange format (include/pcs_hip.h: pcs_voxel_partial + raw key): key = voxel indices
    (floor(v / leaf) + ceil(32768 / leaf)) packed z | y | x with axis_bits(leaf) bits each; sums = the point, n = 1."""
    max_index = 32767 // leaf + (32768 + leaf - 1) // leaf
    bits = 1
    while (1 << bits) <= max_index:
        bits += 1
    bias = (32768 + leaf - 1) // leaf
    v = np.floor_divide(p[:, :3].astype(np.int64), leaf) + bias
    keys = ((v[:, 2] << (2 * bits)) | (v[:, 1] << bits) | v[:, 0]).astype(np.uint64)
    parts = np.zeros((p.shape[0], 8), np.uint32)
    parts[:, 0:3] = p[:, :3].astype(np.int32).view(np.uint32)
    c = p[:, 3].view(np.uint16).astype(np.uint32)
    parts[:, 3], parts[:, 4], parts[:, 5] = c & 0xFF, c >> 8, p[:, 4].view(np.uint16).astype(np.uint32) & 0xFF
    parts[:, 6] = 1
    return keys, parts


@pytest.mark.gpu
def test_caller_held_partials_take_the_warm_path_from_the_second_call(oracle, voxel_tail):
    """pcs_voxel_grid_from_partials_device on ONE context, call after call (what the root of a multi-GPU voxel grid does): the first
    bucket call partitions the caller's list itself (histogram, column scan, scatter), every later one PLACES it into the regions the
    previous call sized (pcs_vox_bkt_place_kernel) and runs the bucket reduce alone. Sequence: a cloud, the same again (steady
    state), a cloud 30 % larger (regions sized for less: some overflow to the list), a cloud that moved (stale splitters), a dense
    slab inside the old key range (regions overflow massively), nothing at all, one partial, the first cloud again with its count
    read from DEVICE memory beside a larger capacity, and a different leaf (no splitters for it: cold again). The caller's arrays
    must come back untouched; every result against the oracle."""
    rng = np.random.default_rng(777)

    def cloud(n, lo, hi):
        p = np.zeros((n, 5), np.int16)
        p[:, :3] = rng.integers(lo, hi, (n, 3))
        p[:, 3] = rng.integers(0, 65536, n).astype(np.uint16).view(np.int16)
        p[:, 4] = rng.integers(0, 256, n)
        return p
    a = cloud(200000, -2500, 2500)
    bigger = np.concatenate([a, cloud(60000, -2500, 2500)])
    moved = cloud(200000, 4000, 9000)
    slab = np.concatenate([a, cloud(400000, -300, 300)])
    leaf = 40
    cap = 700000
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        d_keys = ctx.device_malloc(cap * 8 + 64); d_parts = ctx.device_malloc(cap * 32 + 64)
        d_out = ctx.device_malloc(cap * 10 + 64); d_nv = ctx.device_malloc(64); d_m = ctx.device_malloc(64)
        seq = [("a", a, leaf, False), ("a", a, leaf, False), ("bigger", bigger, leaf, False), ("moved", moved, leaf, False),
               ("slab", slab, leaf, False), ("empty", a[:0], leaf, False), ("one", a[:1], leaf, False), ("a, device count", a, leaf, True),
               ("a, other leaf", a, 55, False), ("a", a, leaf, False)]
        for name, p, lf, dev_count in seq:
            keys, parts = _points_as_partials(p, lf)
            if p.shape[0]:
                ctx.memcpy_h2d(d_keys, keys); ctx.memcpy_h2d(d_parts, parts)
            n = p.shape[0]
            if dev_count:
                ctx.memcpy_h2d(d_m, np.array([n], np.int32))
                ctx.voxel_grid_from_partials_device(d_keys, d_parts, cap, lf, d_out, cap * 5, d_nv, d_n_partials=d_m)
            else:
                ctx.voxel_grid_from_partials_device(d_keys, d_parts, n, lf, d_out, max(n, 1) * 5, d_nv)
            ctx.synchronize()
            nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
            want = oracle.voxel_grid(p, lf)
            assert int(nv[0]) == want.shape[0], (name, voxel_tail)
            if want.shape[0]:
                got = np.empty(want.size, np.int16); ctx.memcpy_d2h(got, d_out)
                assert (got.reshape(-1, 5) == want).all(), (name, voxel_tail)
            if n:
                back_k = np.empty_like(keys); back_p = np.empty_like(parts)
                ctx.memcpy_d2h(back_k, d_keys); ctx.memcpy_d2h(back_p, d_parts)
                assert (back_k == keys).all() and (back_p == parts).all(), name


@pytest.mark.gpu
def test_colour_row_certificate_and_the_reader_that_uses_it(oracle, monkeypatch):
    """CertRowConst: with depth->colour R = I, t_y = t_z = 0 and no distortion a pixel's colour row is a function of its raster row —
    if, and only if, a device sweep over every row x every Z16 value at pcs_create finds no exception. The voxel reader then reads the row
    from a table. (1) the synthetic configuration is certified, and PCS_ROW_CONST=0 turns that off; (2) t_y != 0, t_z != 0, a rotation, a
    distortion model or the half-pixel texture convention are never certified; (3) a colour principal point half a pixel off the depth
    one puts every row's py on a rounding boundary: the sweep must refuse (or the table must still be exact); (4) whatever was decided,
    the voxel cloud equals the oracle's — worst-case depth (uniform random Z16, so every row meets thousands of depth values)."""
    from pointcloud_stitching_amd.types import FLAG_TEXCOORD_HALF_PIXEL, DISTORTION_INVERSE_BROWN_CONRADY
    w, h, n = 640, 480, 2
    rng = np.random.default_rng(5)
    depth = [rng.integers(0, 65536, (h, w), dtype=np.uint16) for _ in range(n)]
    depth[0][:8, :] = 0; depth[1][:, :16] = 1                      # invalid rows, and the smallest valid depth
    color = [S.synth_color(w, h, s) for s in range(n)]

    def check(cfgs, flags, expect, env=None):
        if env is None:
            monkeypatch.delenv("PCS_ROW_CONST", raising=False)
        else:
            monkeypatch.setenv("PCS_ROW_CONST", env)
        stitched, _ = oracle.process_frames(cfgs, depth, color, flags, 1)
        with PcsContext(cfgs, flags=flags) as ctx:
            got_c = [ctx.stream_color_row_const(s) for s in range(n)]
            if expect is not None:
                assert got_c == [expect] * n, (got_c, expect)
            dd, dc = _upload_rasters(ctx, depth, color)
            for leaf in (40, 200):
                got = _rasters_to_voxels(ctx, dd, dc, leaf, n * w * h)
                want = oracle.voxel_grid(stitched, leaf)
                assert got.shape == want.shape and (got == want).all(), (leaf, got_c)
        return got_c

    base = lambda: [S.synth_stream_config(w, h, s) for s in range(n)]      # noqa: E731
    check(base(), FLAG_DROP_INVALID, True)
    check(base(), FLAG_DROP_INVALID, False, env="0")
    check(base(), 0, True)
    for mutate in ("ty", "tz", "rot", "dist"):
        cfgs = base()
        for c in cfgs:
            if mutate == "ty":
                c.depth_to_color.translation[1] = 0.0002
            elif mutate == "tz":
                c.depth_to_color.translation[2] = -0.0003
            elif mutate == "rot":
                a = np.radians(0.5)
                R = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]], np.float32)
                for k, v in enumerate(R.T.reshape(-1)):
                    c.depth_to_color.rotation[k] = float(v)
            else:
                c.color.model = DISTORTION_INVERSE_BROWN_CONRADY
                c.color.coeffs[0] = 0.05
        check(cfgs, FLAG_DROP_INVALID, False)
    check(base(), FLAG_DROP_INVALID | FLAG_TEXCOORD_HALF_PIXEL, False)
    cfgs = base()
    for c in cfgs:
        c.color.ppy = c.depth.ppy + 0.5          # py = r + 0.5 -> fma(v, H, 0.5) lands on integers: the row flips with the rounding of (z * my) / z
    decided = check(cfgs, FLAG_DROP_INVALID, None)
    assert decided == [False] * n, "a boundary configuration was certified row-constant"


@pytest.mark.gpu
def test_two_contexts_in_turn_on_streams_seen_to_overlap(oracle):
    """pcs_use_stream_beside: the second context of a pair used in turn takes a stream that is SEEN to run beside the first one's (streams
    are dealt onto a few hardware queues round robin; two on one queue do not overlap). Bytes are untouched — every call of the loop, on
    either context, equals the oracle; the refusals: the same context twice, a context on an adopted stream."""
    cfgs, depth, color = S.synth_frame_set(4, 640, 480)
    stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    want = oracle.voxel_grid(stitched, 50)
    n_max = sum(c.n_points for c in cfgs)
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as a, PcsContext(cfgs, flags=FLAG_DROP_INVALID) as b:
        found = b.use_stream_beside(a)
        assert found in (True, False)                         # (True on every box seen so far; False keeps the context's stream and is not an error)
        assert b.get_stream() != a.get_stream()
        dd, dc = _upload_rasters(a, depth, color)
        outs = [(c, c.device_malloc(n_max * 10 + 64), c.device_malloc(64)) for c in (a, b)]
        for k in range(8):                                    # calls in turn, nothing synchronised in between
            c, d_vox, d_nv = outs[k & 1]
            c.process_frames_voxel_device(dd, dc, 50, d_vox, n_max * 5, d_nv)
        for c, d_vox, d_nv in outs:
            c.synchronize()
            nv = np.empty(1, np.int32); c.memcpy_d2h(nv, d_nv)
            got = np.empty(max(int(nv[0]), 1) * 5, np.int16); c.memcpy_d2h(got, d_vox)
            assert int(nv[0]) == want.shape[0] and (got[:want.size].reshape(-1, 5) == want).all()
        with pytest.raises(PcsError):
            a.use_stream_beside(a)
        a.set_stream(b.get_stream())                          # adopted: its owner picks the stream
        with pytest.raises(PcsError):
            a.use_stream_beside(b)
        a.set_stream(0)
