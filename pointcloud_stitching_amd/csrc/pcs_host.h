// pcs_host.h — what the translation units of the C ABI (pcs_capi.cpp: contexts, the a1 / a2 twins, the fused path, host forms, plumbing;
// pcs_capi_voxel.cpp: the voxel grid, partials, sinks) share: the context itself and the few helpers both sides call. Internal to
// libpcs_hip.so; nothing here is exported.
#ifndef PCS_HOST_H
#define PCS_HOST_H

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "pcs_device.h"

using namespace pcs;

struct Certificate {
    bool   fast = false, ident_r = false;
    double xb = 0, yb = 0, zb = 0;      // |X|,|Y|,|Z| upper bounds over valid depths
    double a_max[3] = {0, 0, 0};        // |P_i| upper bounds
    double p2_low = 0;                  // P2 lower bound (> 0) over valid depths
};

struct pcs_ctx {
    int                             device = 0;
    int                             n_streams = 0;
    uint32_t                        flags = 0;
    int                             downsample = 1;
    std::vector<pcs_stream_config>  cfg;
    std::vector<StreamParams>       h_params;
    StreamParams*                   d_params = nullptr;
    std::vector<float*>             d_lut;            // 2 per stream (mx, my)
    uint32_t                        total_tiles = 0;
    uint32_t*                       d_tile_counts = nullptr;
    uint32_t*                       d_tile_prefix = nullptr;
    uint32_t*                       d_stream_base = nullptr;   // n_streams + 1 (kept points per stream)
    uint32_t*                       d_arrive = nullptr;        // scan arrival counter (self-resetting)
    int32_t*                        d_counts = nullptr;        // n_streams + 1 (internal, for host APIs)
    int32_t*                        d_static_counts = nullptr; // n_streams + 1: ceil(n/downsample) per stream, total
    // batched compaction scratch (pcs_process_frames_device_batch with a predicate): one slab, rows per frame-set
    uint32_t*                       d_batch_scratch = nullptr;
    int                             batch_scratch_sets = 0;
    // single-pass compaction state (pcs_fused_compact_kernel)
    unsigned long long*             d_ticket = nullptr;        // never reset
    unsigned long long              tickets_issued = 0;
    uint64_t*                       d_desc = nullptr;          // one per tile
    uint32_t*                       d_stream_end = nullptr;    // n_streams
    uint32_t*                       d_error = nullptr;
    uint32_t                        compact_seq = 0;
    bool                            single_pass_ok = true;     // cleared if a placement wait ever timed out
    bool                            compact_tickets = false;   // tile ids by atomic ticket instead of blockIdx
    int                             compact_path = 0;          // 0 count + scan + emit (default), 1 single pass by blockIdx (opt-in)
    bool                            dense_ok = false;          // every stream has n % 8 == 0
    bool                            any_ddist = false, any_cdist = false;
    std::vector<int>                math;                      // per stream: 0 IEEE, 1 certified, 2 + identity R, 3/4 = 1/2 + no-overflow
    std::vector<Certificate>        cert;
    uint32_t                        max_points = 0;
    size_t                          max_payload_points = 0;

    hipStream_t                     own_stream = nullptr;
    hipStream_t                     stream = nullptr;
    hipEvent_t                      ev_begin = nullptr, ev_end = nullptr;
    bool                            kernel_timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;    // recorded pairs awaiting drain
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;

    // lazily sized staging for the host-pointer entry points
    // Rasters of all streams are carved from ONE slab at 256-byte granularity. Separate hipMalloc()s hand
    // out 2 MiB-aligned bases; the streams' tiles advance in lockstep, so equal offsets from power-of-two
    // aligned bases compete for the same sets of the memory-side Infinity Cache: with inputs that were just
    // written (and so sit in that cache) the 8x720p launch measured 22.7 us vs 19.0 us (tools/lab/kernel_lab.hip,
    // "allocation mode", 6-set ring). With cold inputs streamed from HBM the layout makes no difference.
    uint8_t*                        s_slab = nullptr;
    std::vector<uint16_t*>          s_depth;
    std::vector<uint8_t*>           s_color;
    int16_t*                        s_payload = nullptr;  size_t s_payload_cap = 0;   // bytes
    float*                          s_vertices = nullptr; size_t s_vertices_cap = 0;
    float*                          s_texcoords = nullptr; size_t s_texcoords_cap = 0;
    void*                           s_voxel_ws = nullptr; size_t s_voxel_ws_cap = 0;
    VoxelWsState                    vox_state;          // which control block of s_voxel_ws the next voxel call uses
    bool                            sink_open = false;  // pcs_voxel_sink_begin without its pcs_voxel_sink_finish yet
    int                             voxel_reruns = 0;   // calls that ended flagged (-1) and latched the LSD tail (pcs_voxel_tail_reruns)
    int16_t*                        s_voxel_in = nullptr; size_t s_voxel_in_cap = 0;
    int16_t*                        s_voxel_out = nullptr; size_t s_voxel_out_cap = 0;
    uint32_t*                       s_pack_counts = nullptr; uint32_t* s_pack_prefix = nullptr; size_t s_pack_tiles = 0;

    // pcs_submit_frames / pcs_collect_frames: device slots, download stream
    struct PipeSlot {
        uint8_t*               slab = nullptr;
        std::vector<uint16_t*> depth;
        std::vector<uint8_t*>  color;
        int16_t*               payload = nullptr;
        int32_t*               counts = nullptr;       // device, n_streams + 1
        hipEvent_t             done = nullptr;
        bool                   busy = false;
        int                    ticket = -1;
    };
    PipeSlot                        pipe[PCS_PIPELINE_DEPTH];
    hipStream_t                     dl_stream = nullptr;
    int                             next_ticket = 0, next_collect = 0;

    struct ZcEntry { const void* host; size_t bytes; void* dev; int verdict; };
    std::vector<ZcEntry>            zc_cache;                  // zero-copy eligibility verdicts (host_device_view)

    std::string                     err;
};

namespace pcs_host {

int fail(pcs_ctx* c, int status, const char* fmt, ...);

#define HIPCHK(c, expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail((c), PCS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));   \
    } while (0)

inline bool has_pred(uint32_t flags) { return (flags & (PCS_FLAG_CUTOFF | PCS_FLAG_DROP_INVALID)) != 0; }

template <class T>
int ensure(pcs_ctx* c, T*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap && p) return PCS_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, PCS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    p = static_cast<T*>(q);
    cap = std::max<size_t>(bytes, 256);
    return PCS_OK;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// The voxel workspace of a context, at least `need` bytes (grows with a quarter of headroom; growing waits for the stream).
int ensure_voxel_ws(pcs_ctx* c, size_t need);
int acquire_event_pair(pcs_ctx* c, std::pair<hipEvent_t, hipEvent_t>& pr);
// The fused path for device-resident rasters. Counts end up in d_counts (if non-null).
int run_fused_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color, int16_t* d_payload, size_t payload_shorts,
                     int32_t* d_counts, bool force_three_pass = false, const uint32_t* d_tile_kept = nullptr);

}  // namespace pcs_host

#endif
