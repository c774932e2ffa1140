// pcs_capi_voxel.cpp — the voxel-grid part of the C ABI of libpcs_hip.so (include/pcs_hip.h): the grid of a payload, of the rasters, of
// partials (the exchange format of a multi-GPU grid) and the sink several contexts of one device pre-aggregate into. The kernels and
// what bounds them: pcs_voxel.hip, pcs_kernels.hip (the raster / payload readers), DESIGN.md section 10. Not in the reference (it includes
// pcl/filters/voxel_grid.h and never instantiates it, src/pcs-multicamera-optimized.cpp:17): BASELINE configs[4] asks for it.

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>

#include "pcs_host.h"

using namespace pcs_host;

extern "C" {

// ---- voxel-grid downsample (not in the reference; defined in pcs_voxel.hip / DESIGN.md) ----------
static int voxel_grid_device_impl(pcs_ctx* c, const int16_t* d_payload, int n_points, const int32_t* d_n_points, int leaf_mm,
                                  int16_t* d_out, size_t out_shorts, int32_t* d_out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_points %d < 0", n_points);
    if (leaf_mm < 1 || leaf_mm > 32767) return fail(c, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (n_points > 0 && (!d_payload || !d_out)) return fail(c, PCS_ERR_INVALID_ARG, "NULL device pointer");
    if (out_shorts < (size_t)n_points * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "output holds %zu shorts; the worst case (every point its own voxel) needs %zu",
                    out_shorts, (size_t)n_points * PCS_POINT_SHORTS);
    DeviceGuard guard(c->device);
    const size_t need = voxel_workspace_bytes((uint32_t)n_points, voxel_workspace_level((uint32_t)n_points, leaf_mm, c->vox_state, false));
    int rc = ensure_voxel_ws(c, need);
    if (rc) return rc;
    HIPCHK(c, launch_voxel_grid(d_payload, (uint32_t)n_points, d_n_points, leaf_mm, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, d_out,
                                d_out_points, c->stream));
    return PCS_OK;
}

int pcs_set_voxel_tail(pcs_ctx* c, int tail)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (tail != PCS_VOXEL_TAIL_AUTO && tail != PCS_VOXEL_TAIL_BUCKET && tail != PCS_VOXEL_TAIL_LSD && tail != PCS_VOXEL_TAIL_LSD_LATCHED)
        return fail(c, PCS_ERR_INVALID_ARG, "unknown voxel tail %d", tail);
    if (tail == PCS_VOXEL_TAIL_LSD_LATCHED) {
        // (the flagged call's control block is cleared by the next call as any other's; clearing both costs one memset, once)
        c->vox_state.stalled = true; c->vox_state.clean = false; c->vox_state.spl_leaf = 0;
        c->voxel_reruns++;
        return PCS_OK;
    }
    c->vox_state.tail_pref = tail;
    c->vox_state.stalled = false;
    return PCS_OK;
}

int pcs_voxel_tail_reruns(const pcs_ctx* c) { return c ? c->voxel_reruns : 0; }

int pcs_inject_voxel_stall(int launches)
{
    pcs::inject_voxel_stall(launches);
    return PCS_OK;
}

int pcs_voxel_grid_device(pcs_ctx* c, const int16_t* d_payload, int n_points, int leaf_mm, int16_t* d_out,
                          size_t out_shorts, int32_t* d_out_points)
{
    return voxel_grid_device_impl(c, d_payload, n_points, nullptr, leaf_mm, d_out, out_shorts, d_out_points);
}

int pcs_voxel_grid_device_counted(pcs_ctx* c, const int16_t* d_payload, const int32_t* d_n_points, int max_points, int leaf_mm,
                                  int16_t* d_out, size_t out_shorts, int32_t* d_out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_n_points) return fail(c, PCS_ERR_INVALID_ARG, "d_n_points is NULL");
    return voxel_grid_device_impl(c, d_payload, max_points, d_n_points, leaf_mm, d_out, out_shorts, d_out_points);
}

int pcs_process_frames_voxel_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm,
                                    int16_t* d_out, size_t out_shorts, int32_t* d_out_points)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_depth || !d_color || !d_out) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (leaf_mm < 1 || leaf_mm > 32767) return fail(c, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    const int S = c->n_streams;
    for (int s = 0; s < S; s++) {
        if (!d_depth[s] || !d_color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
        if ((uintptr_t)d_depth[s] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: depth pointer not 2-byte aligned", s);
    }
    const size_t cap = c->max_payload_points;          // every pixel kept, no stride
    if (out_shorts < cap * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "output holds %zu shorts; the worst case (every pixel its own voxel) needs %zu",
                    out_shorts, cap * PCS_POINT_SHORTS);
    DeviceGuard guard(c->device);
    // Rasters whose width is a multiple of 8 are read in square patches and the direct route wins at every leaf
    // (16 x 1080p: 0.26 vs 0.46 ms at 50 mm, 0.52 vs 0.91 ms at 25 mm, 1.64 vs 2.57 ms at 10 mm). Other rasters are read
    // in runs of 4096 consecutive pixels; below ~36 mm (on the synthetic scene) such a run holds more voxels than a
    // workgroup's LDS table takes gracefully and the payload reader, fed by the ordered compaction, is the faster route
    // (25 mm: 0.93 vs 1.11 ms). Same result either way. PCS_VOXEL_FUSED=0/1 forces one or the other.
    static const int fused_env = [] { const char* v = getenv("PCS_VOXEL_FUSED"); return v ? atoi(v) : -1; }();
    bool all_patch = true;
    for (int s = 0; s < S; s++) all_patch &= (c->h_params[s].W & 7) == 0 && ((uintptr_t)d_depth[s] & 15u) == 0;
    const bool fused = fused_env >= 0 ? fused_env != 0 : (all_patch || leaf_mm >= 36);
    if (c->downsample != 1 || !fused) {
        // (the stride is defined on the ORDER of the kept points: build the stitched cloud, then its voxel grid)
        int rc = ensure(c, c->s_payload, c->s_payload_cap, cap * PCS_POINT_BYTES + 16);
        if (rc) return rc;
        rc = run_fused_device(c, d_depth, d_color, c->s_payload, cap * PCS_POINT_SHORTS, c->d_counts, true);
        if (rc) return rc;
        return voxel_grid_device_impl(c, c->s_payload, (int)cap, c->d_counts + S, leaf_mm, d_out, out_shorts, d_out_points);
    }
    const size_t need = voxel_workspace_bytes((uint32_t)cap, voxel_workspace_level((uint32_t)cap, leaf_mm, c->vox_state, false));
    int rc = ensure_voxel_ws(c, need);
    if (rc) return rc;
    std::pair<hipEvent_t, hipEvent_t> ev{};
    if (c->kernel_timing) {
        rc = acquire_event_pair(c, ev);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(ev.first, c->stream));
    }
    VoxelStage vs{};
    HIPCHK(c, voxel_begin((uint32_t)cap, leaf_mm, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, &vs, c->stream));
    for (int s0 = 0; s0 < S; s0 += kLaunchStreams) {
        const int nl = std::min(kLaunchStreams, S - s0);
        FramePtrs fp{};
        uint32_t mp = 0, mw = 0, mh = 0;
        bool fast = true, ident = true, rowc = true, patch_ok = true;
        for (int k = 0; k < nl; k++) {
            const StreamParams& q = c->h_params[s0 + k];
            fp.depth[k] = d_depth[s0 + k]; fp.color[k] = d_color[s0 + k];
            mp = std::max(mp, q.n_points);
            mw = std::max(mw, (uint32_t)q.W); mh = std::max(mh, q.n_points / (uint32_t)q.W);
            patch_ok &= (q.W & 7) == 0 && ((uintptr_t)d_depth[s0 + k] & 15u) == 0;
            fast &= q.cert_fast != 0; ident &= q.ident_r != 0; rowc &= q.ident_r == 2;
        }
        const MathSel sel = !fast ? MathSel::Ieee : (ident ? (rowc ? MathSel::CertRowConst : MathSel::CertIdentR) : MathSel::Cert);
        HIPCHK(c, launch_fused_voxel_partials(c->d_params, s0, nl, mp, mw, mh, patch_ok, c->any_ddist || c->any_cdist, c->flags, sel, fp, vs, c->stream));
    }
    HIPCHK(c, voxel_finish((uint32_t)cap, leaf_mm, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, d_out, d_out_points, c->stream));
    if (c->kernel_timing) {
        HIPCHK(c, hipEventRecord(ev.second, c->stream));
        c->ev_pool.push_back(ev);
    }
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_voxel_device: host allocation failed (%s)", ex.what());
}

// The pre-aggregation of this context's streams under its flags into `vs`: a caller's arrays (exchange format), or the workspace of a
// context of this device — its own, or another one's (a voxel sink, below). Fused from the rasters where that is the faster route
// (as pcs_process_frames_voxel_device decides), else through this context's own stitched cloud.
static int run_voxel_frontend(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm, const VoxelStage& vs)
{
    const int S = c->n_streams;
    const size_t cap = c->max_payload_points;
    static const int fused_env = [] { const char* v = getenv("PCS_VOXEL_FUSED"); return v ? atoi(v) : -1; }();
    bool all_patch = true;
    for (int s = 0; s < S; s++) all_patch &= (c->h_params[s].W & 7) == 0 && ((uintptr_t)d_depth[s] & 15u) == 0;
    const bool fused = fused_env >= 0 ? fused_env != 0 : (all_patch || leaf_mm >= 36);      // as pcs_process_frames_voxel_device
    if (c->downsample != 1 || !fused) {
        // the stride is defined on the ORDER of the kept points: build this GPU's stitched cloud, pre-aggregate that
        int rc = ensure(c, c->s_payload, c->s_payload_cap, cap * PCS_POINT_BYTES + 16);
        if (rc) return rc;
        rc = run_fused_device(c, d_depth, d_color, c->s_payload, cap * PCS_POINT_SHORTS, c->d_counts, true);
        if (rc) return rc;
        if (cap) HIPCHK(c, launch_payload_voxel_partials(c->s_payload, (uint32_t)cap, c->d_counts + S, vs, c->stream));
        return PCS_OK;
    }
    for (int s0 = 0; s0 < S; s0 += kLaunchStreams) {
        const int nl = std::min(kLaunchStreams, S - s0);
        FramePtrs fp{};
        uint32_t mp = 0, mw = 0, mh = 0;
        bool fast = true, ident = true, rowc = true, patch_ok = true;
        for (int k = 0; k < nl; k++) {
            const StreamParams& q = c->h_params[s0 + k];
            fp.depth[k] = d_depth[s0 + k]; fp.color[k] = d_color[s0 + k];
            mp = std::max(mp, q.n_points);
            mw = std::max(mw, (uint32_t)q.W); mh = std::max(mh, q.n_points / (uint32_t)q.W);
            patch_ok &= (q.W & 7) == 0 && ((uintptr_t)d_depth[s0 + k] & 15u) == 0;
            fast &= q.cert_fast != 0; ident &= q.ident_r != 0; rowc &= q.ident_r == 2;
        }
        const MathSel sel = !fast ? MathSel::Ieee : (ident ? (rowc ? MathSel::CertRowConst : MathSel::CertIdentR) : MathSel::Cert);
        HIPCHK(c, launch_fused_voxel_partials(c->d_params, s0, nl, mp, mw, mh, patch_ok, c->any_ddist || c->any_cdist, c->flags, sel, fp, vs, c->stream));
    }
    return PCS_OK;
}

// ---- voxel partials (exchange format of the multi-GPU voxel grid) ------------------------------------
int pcs_process_frames_voxel_partials_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                             int leaf_mm, uint64_t* d_keys, pcs_voxel_partial* d_partials, size_t capacity,
                                             int32_t* d_n_partials)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_depth || !d_color || !d_keys || !d_partials || !d_n_partials) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (leaf_mm < 1 || leaf_mm > 32767) return fail(c, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (((uintptr_t)d_keys & 7u) || ((uintptr_t)d_partials & 31u))
        return fail(c, PCS_ERR_INVALID_ARG, "d_keys must be 8-byte and d_partials 32-byte aligned");
    const int S = c->n_streams;
    for (int s = 0; s < S; s++) {
        if (!d_depth[s] || !d_color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
        if ((uintptr_t)d_depth[s] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: depth pointer not 2-byte aligned", s);
    }
    const size_t cap = c->max_payload_points;
    if (capacity < cap)
        return fail(c, PCS_ERR_CAPACITY, "partial arrays hold %zu entries; the worst case (every kept point its own partial) needs %zu",
                    capacity, cap);
    DeviceGuard guard(c->device);
    static_assert(sizeof(pcs_voxel_partial) == 32, "pcs_voxel_partial is the kernels' 32-byte VoxelPartial");
    // the caller's count word IS the append counter of the pre-aggregation (cleared here, complete when the kernels are)
    VoxelStage vs{};
    HIPCHK(c, voxel_partials_stage(leaf_mm, reinterpret_cast<unsigned long long*>(d_keys), d_partials,
                                   reinterpret_cast<unsigned int*>(d_n_partials), &vs, c->stream));
    const int rc = run_voxel_frontend(c, d_depth, d_color, leaf_mm, vs);
    if (rc) return rc;
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_voxel_partials_device: host allocation failed (%s)", ex.what());
}

int pcs_voxel_grid_from_partials_device(pcs_ctx* c, const uint64_t* d_keys, const pcs_voxel_partial* d_partials, int n_partials,
                                        const int32_t* d_n_partials, int leaf_mm, int16_t* d_out, size_t out_shorts,
                                        int32_t* d_out_points)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_partials < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_partials %d < 0", n_partials);
    if (((uintptr_t)d_n_partials & 3u) || ((uintptr_t)d_out_points & 3u))
        return fail(c, PCS_ERR_INVALID_ARG, "d_n_partials / d_out_points must be 4-byte aligned");
    if (leaf_mm < 1 || leaf_mm > 32767) return fail(c, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (n_partials > 0 && (!d_keys || !d_partials || !d_out)) return fail(c, PCS_ERR_INVALID_ARG, "NULL device pointer");
    if (((uintptr_t)d_keys & 7u) || ((uintptr_t)d_partials & 31u))
        return fail(c, PCS_ERR_INVALID_ARG, "d_keys must be 8-byte and d_partials 32-byte aligned");
    if (out_shorts < (size_t)n_partials * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "output holds %zu shorts; the worst case (every partial its own voxel) needs %zu",
                    out_shorts, (size_t)n_partials * PCS_POINT_SHORTS);
    DeviceGuard guard(c->device);
    // (a quarter more than this call's count + 64 Ki: the regions a warm call fills were sized by the PREVIOUS call's count, and a root's
    // count moves from frame-set to frame-set — launch_voxel_from_partials carves the workspace for what it holds)
    const uint32_t n_size = (uint32_t)std::min<uint64_t>((uint64_t)n_partials + (uint64_t)n_partials / 4u + 65536u,
                                                         (uint32_t)n_partials < (1u << 26) ? (1u << 26) - 1u : 0xFFFFFFF0u);
    const size_t need = voxel_workspace_bytes(n_size, voxel_workspace_level((uint32_t)n_partials, leaf_mm, c->vox_state, true));
    int rc = ensure_voxel_ws(c, need);
    if (rc) return rc;
    HIPCHK(c, launch_voxel_from_partials(reinterpret_cast<const unsigned long long*>(d_keys), d_partials, (uint32_t)n_partials,
                                         d_n_partials, leaf_mm, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, d_out, d_out_points, c->stream));
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_voxel_grid_from_partials_device: host allocation failed (%s)", ex.what());
}

// ---- voxel SINK: one context's workspace filled by the pre-aggregations of several contexts of the same device ---------
namespace {
struct SinkBlob {
    VoxelStage vs;
    uint32_t   capacity;
    int32_t    leaf, device;
    uint32_t   magic;
};
static_assert(sizeof(SinkBlob) <= sizeof(((pcs_voxel_sink*)nullptr)->opaque), "pcs_voxel_sink (include/pcs_hip.h) must hold a VoxelStage");
constexpr uint32_t kSinkMagic = 0x50435356u;      // "PCSV"
}  // namespace

int pcs_voxel_sink_begin(pcs_ctx* c, size_t capacity_points, int leaf_mm, pcs_voxel_sink* sink)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!sink) return fail(c, PCS_ERR_INVALID_ARG, "sink is NULL");
    if (leaf_mm < 1 || leaf_mm > 32767) return fail(c, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (capacity_points < 1 || capacity_points > 0xFFFFFFF0ull)
        return fail(c, PCS_ERR_INVALID_ARG, "capacity_points %zu outside 1..2^32-16", capacity_points);
    // (a sink that was opened and never finished — a pre-aggregation failed in between — is abandoned here: voxel_begin left the
    // workspace marked unclean, so this call clears its control blocks again)
    DeviceGuard guard(c->device);
    const uint32_t cap = (uint32_t)capacity_points;
    const size_t need = voxel_workspace_bytes(cap, voxel_workspace_level(cap, leaf_mm, c->vox_state, false));
    const int rc = ensure_voxel_ws(c, need);
    if (rc) return rc;
    SinkBlob b{};
    // (voxel_begin clears the control blocks of a workspace it has not seen, or whose last call was not enqueued completely, with a
    // memset on this stream: the only work a begin ever enqueues)
    const bool clears = !c->vox_state.clean || c->vox_state.base != c->s_voxel_ws;
    HIPCHK(c, voxel_begin(cap, leaf_mm, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, &b.vs, c->stream));
    b.capacity = cap; b.leaf = leaf_mm; b.device = c->device; b.magic = kSinkMagic;
    std::memset(sink, 0, sizeof *sink);
    std::memcpy(sink->opaque, &b, sizeof b);
    sink->work_enqueued = clears ? 1u : 0u;
    c->sink_open = true;
    return PCS_OK;
}

int pcs_process_frames_voxel_into_sink_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                              const pcs_voxel_sink* sink)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_depth || !d_color || !sink) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    SinkBlob b;
    std::memcpy(&b, sink->opaque, sizeof b);
    if (b.magic != kSinkMagic) return fail(c, PCS_ERR_INVALID_ARG, "not a sink pcs_voxel_sink_begin filled");
    if (b.device != c->device)
        return fail(c, PCS_ERR_INVALID_ARG, "the sink lives on device %d, this context on device %d: a sink takes contexts of its own device only "
                    "(the pre-aggregation's atomics are device-scope; other GPUs exchange partials)", b.device, c->device);
    const int S = c->n_streams;
    for (int s = 0; s < S; s++) {
        if (!d_depth[s] || !d_color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
        if ((uintptr_t)d_depth[s] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: depth pointer not 2-byte aligned", s);
    }
    if (c->max_payload_points > b.capacity)
        return fail(c, PCS_ERR_CAPACITY, "the sink was opened for %u points; this context alone can produce %zu", b.capacity, c->max_payload_points);
    DeviceGuard guard(c->device);
    return run_voxel_frontend(c, d_depth, d_color, b.leaf, b.vs);
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_voxel_into_sink_device: host allocation failed (%s)", ex.what());
}

int pcs_voxel_sink_finish(pcs_ctx* c, const pcs_voxel_sink* sink, int16_t* d_out, size_t out_shorts, int32_t* d_out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!sink || !d_out) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)d_out_points & 3u) return fail(c, PCS_ERR_INVALID_ARG, "d_out_points must be 4-byte aligned");
    SinkBlob b;
    std::memcpy(&b, sink->opaque, sizeof b);
    if (b.magic != kSinkMagic || !c->sink_open || b.device != c->device)
        return fail(c, PCS_ERR_INVALID_ARG, "not the sink this context has open");
    if (out_shorts < (size_t)b.capacity * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "output holds %zu shorts; the worst case (every point its own voxel) needs %zu",
                    out_shorts, (size_t)b.capacity * PCS_POINT_SHORTS);
    DeviceGuard guard(c->device);
    c->sink_open = false;
    HIPCHK(c, voxel_finish(b.capacity, b.leaf, c->s_voxel_ws, c->s_voxel_ws_cap, &c->vox_state, d_out, d_out_points, c->stream));
    return PCS_OK;
}

int pcs_voxel_grid(pcs_ctx* c, const int16_t* payload, int n_points, int leaf_mm, int16_t* out, size_t out_shorts,
                   int* out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_points %d < 0", n_points);
    if (n_points == 0) { if (out_points) *out_points = 0; return PCS_OK; }
    if (!payload || !out) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    DeviceGuard guard(c->device);
    const size_t bytes = (size_t)n_points * PCS_POINT_BYTES;
    int rc;
    if ((rc = ensure(c, c->s_voxel_in, c->s_voxel_in_cap, bytes))) return rc;
    if ((rc = ensure(c, c->s_voxel_out, c->s_voxel_out_cap, bytes))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s_voxel_in, payload, bytes, hipMemcpyHostToDevice, c->stream));
    rc = pcs_voxel_grid_device(c, c->s_voxel_in, n_points, leaf_mm, c->s_voxel_out, (size_t)n_points * PCS_POINT_SHORTS, c->d_counts);
    if (rc) return rc;
    int32_t nv = 0;
    HIPCHK(c, hipMemcpyAsync(&nv, c->d_counts, sizeof nv, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nv < 0) {
        // the bucket tail gave up waiting for one of its own workgroups (the device forms report the -1 as it is; see
        // include/pcs_hip.h): the input is still in s_voxel_in — once more on the LSD tail, which waits for nobody, and LSD for
        // this context from here on
        if ((rc = pcs_set_voxel_tail(c, PCS_VOXEL_TAIL_LSD_LATCHED))) return rc;
        rc = pcs_voxel_grid_device(c, c->s_voxel_in, n_points, leaf_mm, c->s_voxel_out, (size_t)n_points * PCS_POINT_SHORTS, c->d_counts);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(&nv, c->d_counts, sizeof nv, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nv < 0) return fail(c, PCS_ERR_HIP, "the voxel pipeline reported a negative count on the LSD tail");
    }
    if (out_shorts < (size_t)nv * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "output holds %zu shorts, %zu needed", out_shorts, (size_t)nv * PCS_POINT_SHORTS);
    if (nv) HIPCHK(c, hipMemcpy(out, c->s_voxel_out, (size_t)nv * PCS_POINT_BYTES, hipMemcpyDeviceToHost));
    if (out_points) *out_points = nv;
    return PCS_OK;
}

}  // extern "C"
