// pcs_device.h — structures shared by the HIP kernels (pcs_kernels.hip) and the C-ABI host layer
// (pcs_capi.cpp). Internal; the public surface is include/pcs_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcs_hip.h"

namespace pcs {

// Geometry of one launch tile: a 256-thread workgroup owns 2048 consecutive points of one stream,
// 8 consecutive points per lane, 512 per wavefront.
constexpr int      kBlockThreads   = 256;
constexpr int      kPointsPerLane  = 8;
constexpr uint32_t kTilePoints     = kBlockThreads * kPointsPerLane;   // 2048
constexpr int      kLaunchStreams  = 16;                               // streams per launch (blockIdx.y)

// Per-stream constants. One table entry per camera, uploaded at pcs_create / pcs_set_cam_to_world.
// Every field is wave-uniform (indexed by blockIdx.y), so the kernels read them with scalar loads
// and they live in SGPRs — the CDNA counterpart of the reference's broadcast __m128 globals
// (src/pcs-camera-optimized.cpp:69-72, 390-408).
struct alignas(16) StreamParams {
    float    M[12];          // top three rows of cam_to_world (tf_mat), row-major
    float    R[9];           // depth->colour rotation, column-major (rs2_extrinsics)
    float    t[3];           // depth->colour translation
    float    depth_scale;
    float    d_ppx, d_ppy, d_fx, d_fy;
    float    c_fx, c_fy, c_ppx, c_ppy;
    float    c_w_f, c_h_f;   // (float)colour width / height
    float    c_wm1_f, c_hm1_f; // (float)(colour width - 1) / (height - 1)
    float    c_rw, c_rh;     // RN(1/c_w_f), RN(1/c_h_f) — for the verified constant-divisor quotient
    float    dk[5];          // depth distortion coefficients
    float    ck[5];          // colour distortion coefficients
    int32_t  W, H;           // depth raster
    int32_t  cW, cH;         // colour raster
    int32_t  bpp, stride;    // colour bytes per pixel / per row
    uint32_t color_bytes;    // stride * cH
    uint32_t n_points;       // W * H
    int32_t  ddist, cdist;   // distortion active (non-zero coefficients)
    uint32_t out_base;       // first output point of this stream in the stitched payload when no
                             // predicate is active: sum over earlier streams of ceil(n/downsample)
    uint32_t tile_base;      // index of this stream's first tile in the per-tile count arrays
    uint32_t w_magic, w_shift; // floor(i / W) == umulhi(i, w_magic) >> w_shift for i < 2^31 (0 = use '/')
    int32_t  cert_fast;      // host+device certified for CertMath (see pcs_capi.cpp)
    int32_t  ident_r;        // depth->colour rotation is exactly I, translation has no -0; 2: and the colour ROW of a pixel is certified
                             // independent of its depth value — my[H .. 2H) holds the row of every raster row (CertRowConst)
    int32_t  no_overflow;    // certified: no converted value (world mm, colour column/row) can reach 2^31
    int32_t  z_zero_iff_d_zero; // depth_scale finite and depth_scale*1 != 0: (z == 0) == (d == 0)
    uint32_t cut_dmax;       // -c from the Z16 word alone: in range <=> 1 <= d <= cut_dmax (0 = not certified, deproject)
    int32_t  tex_half;       // PCS_FLAG_TEXCOORD_HALF_PIXEL: u = (px + 0.5)/W (handled on the CDIST code path)
    const float* mx;         // [W]  (c - ppx) / fx   — IEEE division done once on the host
    const float* my;         // [H]  (r - ppy) / fy; behind it [H] int32: the colour row of raster row r (valid when ident_r == 2)
};

// Per-call raster pointers, passed by value in the kernarg segment (no per-frame H2D of a table).
struct FramePtrs {
    const uint16_t* depth[kLaunchStreams];
    const uint8_t*  color[kLaunchStreams];
};

struct VertexPtrs {          // a2 twin: one stream per launch
    const float*   vertices;   // n x {x,y,z}
    const float*   texcoords;  // n x {u,v}
    const uint8_t* color;
    uint32_t       n_points;
};

// Batched dense launch: K frame-sets of the same S streams in ONE launch (blockIdx.z = frame-set). A 23 us
// launch spends a large share of its life filling and draining the machine; K sets per launch amortise that
// (throughput form, pcs_process_frames_device_batch). Entry z*S + y holds the rasters of stream y in set z.
constexpr int kBatchEntries = 64;                                      // S * K <= 64
constexpr int kBatchSets    = 16;
struct BatchPtrs {
    const uint16_t* depth[kBatchEntries];
    const uint8_t*  color[kBatchEntries];
    uint8_t*        payload[kBatchSets];                               // payload base of frame-set z
};
// Where the per-stream counts [S] and the total [S] of frame-set z go (batched compaction).
struct BatchCounts {
    int32_t* counts[kBatchSets];
};

// Batched a2 twin: several cameras' rs2::points arrays in ONE launch (blockIdx.y = cloud).
constexpr int kPackBatch = 16;
struct PackBatch {
    VertexPtrs v[kPackBatch];
    uint8_t*   out[kPackBatch];
    int32_t    stream[kPackBatch];                                     // which StreamParams entry (extrinsic, colour geometry)
};

// Which arithmetic policy a launch may use (the AND over the streams of the launch).
enum class MathSel { Ieee = 0, Cert = 1, CertIdentR = 2, CertNoOvf = 3, CertIdentRNoOvf = 4,
                     CertRowConst = 5 /* voxel reader only: CertIdentR + the colour row from a per-row table (ident_r == 2) */ };

// Launchers (defined in pcs_kernels.hip). All enqueue on `st` and return the hipError of the launch.
hipError_t launch_fused_dense(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                              bool any_ddist, bool any_cdist, MathSel math, const FramePtrs& fp, int16_t* d_payload,
                              hipStream_t st);

// Generic path: predicate / downsample / unaligned payload / W % 8 != 0.
//   d_tile_counts, d_tile_prefix : one uint32 per tile of every stream (StreamParams::tile_base indexes them)
//   d_stream_kept                : n_total_streams uint32 (kept points per stream, before the stride)
//   d_arrive                     : nullptr for the fused path (the per-stream counts come from the scan, the grand total
//                                  from the first emit launch: d_total_out); the a2 twin's launch_pack_scan passes one
//                                  zero-initialised uint32 (arrival counter, self-resetting) and gets the total from the scan
//   d_counts (optional)          : n_total_streams + 1 int32 handed back to the caller
// launch_pack_scan: d_out_points receives 2 int32 (kept, total)
hipError_t launch_fused_count(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                              uint32_t flags, const FramePtrs& fp, uint32_t* d_tile_counts, hipStream_t st);
hipError_t launch_scan(const StreamParams* d_params, int n_streams, int downsample,
                       const uint32_t* d_tile_counts, uint32_t* d_tile_prefix, uint32_t* d_stream_kept,
                       int32_t* d_counts, uint32_t* d_arrive, hipStream_t st);
hipError_t launch_fused_emit(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                             uint32_t flags, int downsample, MathSel math, const FramePtrs& fp,
                             const uint32_t* d_tile_prefix, const uint32_t* d_stream_kept,
                             int16_t* d_payload, int32_t* d_total_out, int n_total_streams, hipStream_t st);
// Single-pass ordered compaction (predicate, stride 1). See pcs_fused_compact_kernel.
struct CompactLaunch {
    unsigned long long* d_ticket;       // device counter, never reset
    unsigned long long  ticket_base;    // tickets issued before this launch
    uint64_t*           d_desc;         // launch_tiles descriptors
    uint64_t*           d_stream_desc;  // all streams: per-stream totals
    uint32_t*           d_stream_end;   // all streams
    const uint32_t*     d_chain_in;     // total of earlier launches of the same frame-set, or nullptr
    uint32_t*           d_error;
    uint32_t            gen;
    uint32_t            flags;
    int32_t*            d_counts;       // n_total + 1 ints: per-stream kept counts and the total (written by the kernel)
    int32_t             n_total;        // streams of the whole frame-set
    int32_t             last_launch;    // this launch holds the frame-set's last stream
};
hipError_t launch_fused_compact(const StreamParams* d_params, int stream0, int n_launch, uint32_t launch_tiles,
                                MathSel math, const FramePtrs& fp, const CompactLaunch& cl, int16_t* d_payload,
                                hipStream_t st);

// K frame-sets per launch (dense path only).
hipError_t launch_fused_dense_batch(const StreamParams* d_params, int n_streams, int n_sets, uint32_t max_points,
                                    bool any_ddist, bool any_cdist, MathSel math, const BatchPtrs& bp, hipStream_t st);
// K frame-sets per launch, ordered compaction (stride 1): count, scan and emit each cover all K sets.
// d_tile_counts / d_tile_prefix hold n_sets * total_tiles words, d_stream_kept n_sets * n_streams.
hipError_t launch_compact_batch(const StreamParams* d_params, int n_streams, int n_sets, uint32_t max_points,
                                uint32_t total_tiles, uint32_t flags, MathSel math, const BatchPtrs& bp,
                                const BatchCounts& bc, uint32_t* d_tile_counts, uint32_t* d_tile_prefix,
                                uint32_t* d_stream_kept, hipStream_t st);
hipError_t launch_verify_div_const(float c, float rc, int32_t dim, unsigned long long* d_bad, hipStream_t st);
// CertRowConst's certificate for one stream of an uploaded parameter table: d_crow[rows] = colour row of every raster row at depth 1,
// *d_bad += (row, depth) pairs over all 65 535 depth values whose row differs (pcs_kernels.hip)
hipError_t launch_certify_color_row(const StreamParams* d_params, int stream, int rows, int32_t* d_crow, unsigned long long* d_bad, hipStream_t st);

// a2 twin.
hipError_t launch_pack_dense(const StreamParams* d_params, int stream, const VertexPtrs& vp,
                             int16_t* d_out, hipStream_t st);
hipError_t launch_pack_count(const StreamParams* d_params, int stream, const VertexPtrs& vp, uint32_t flags,
                             uint32_t* d_tile_counts, hipStream_t st);
hipError_t launch_pack_scan(uint32_t n_tiles, const uint32_t* d_tile_counts, uint32_t* d_tile_prefix,
                            int32_t* d_out_points, uint32_t* d_arrive, hipStream_t st);
hipError_t launch_pack_emit(const StreamParams* d_params, int stream, const VertexPtrs& vp, uint32_t flags,
                            const uint32_t* d_tile_prefix, int16_t* d_out, hipStream_t st);

// batched a2 twin (no predicate): n clouds in one launch; `aligned` = every out pointer is 16-byte aligned
hipError_t launch_pack_batch(const StreamParams* d_params, const PackBatch& pb, int n, uint32_t max_points, bool aligned,
                             hipStream_t st);

// a5 alone.
hipError_t launch_deproject(const StreamParams* d_params, int stream, uint32_t n_points, const uint16_t* d_depth,
                            float* d_vertices, float* d_texcoords, hipStream_t st);

// Voxel-grid downsample (pcs_voxel.hip).
struct VoxelWsState;
size_t     voxel_workspace_bytes(uint32_t n_points, int level = 2);      // level: voxel_workspace_level (0 LSD .. 2 warm bucket)
int        voxel_workspace_level(uint32_t n_points, int leaf_mm, const VoxelWsState& ws, bool from_partials);
// What the owner of a voxel workspace keeps between calls (pcs_voxel.hip: plan_for): which of the workspace's two control
// blocks the next call uses, and whether both are known to be in the state that call expects.
struct VoxelWsState {
    const void* base = nullptr;
    uint32_t    phase = 0;
    bool        clean = false;
    int         spl_leaf = 0;       // bucket tail: the leaf the workspace's splitters were made for (0: none)
    uint32_t    bkt_calls = 0;      // bucket-tail calls enqueued on this workspace (tags their published counts)
    int         tail_pref = 0;      // pcs_set_voxel_tail: 0 by the leaf, 1 bucket, 2 LSD (the environment overrides)
    bool        stalled = false;    // a bucket-tail call of this owner ended flagged (-1): LSD from here on, whatever the environment says
};
hipError_t launch_voxel_grid(const int16_t* d_payload, uint32_t n_points, const int32_t* d_n_points, int leaf_mm, void* d_ws,
                             size_t ws_bytes, VoxelWsState* ws, int16_t* d_out, int32_t* d_out_points, hipStream_t st);

// The same pipeline fed from the rasters (pcs_process_frames_voxel_device): voxel_begin carves the workspace and clears
// the counters, launch_fused_voxel_partials (pcs_kernels.hip) appends the partials, voxel_finish sorts and reduces.
struct VoxelStage {
    unsigned long long* keys;
    unsigned int*       idx;           // nullptr with idx_bits == 0: raw keys only (exchange format)
    void*               part;          // VoxelPartial[capacity]
    unsigned int*       n_runs;        // partials appended so far
    uint32_t            leaf, bits, idx_bits;
    float               div_inv, div_c;   // voxel index of a coordinate: (unsigned)fmaf(v, div_inv, div_c) (pcs_voxel_agg.h: VoxelDiv)
    uint32_t            track_bits;    // record which key bits vary (the sort may then skip a pass); 0: the host declared all of them varying
    // Warm bucket tail (pcs_voxel.hip, "regions"): the workgroup finds each partial's bucket itself (the previous call's splitters)
    // and appends it to that bucket's REGION — reg[0] buckets of reg[1] slots each, filled through cursor[bucket] — so the
    // partition kernels (histogram, column scan, scatter) are not launched. A partial whose region is full goes to keys / part as
    // before, with its bucket id in bucket_of: the tail's first workgroup sorts those few in.
    uint32_t                  regions;       // 0: every partial to keys / part
    uint32_t                  region_slots;  // capacity of keys_r / part_r (a call whose reg[] asks for more uses no regions)
    const unsigned long long* spl;           // kVoxBuckets - 1 ascending splitters (+ one unused word)
    const unsigned int*       reg;           // {buckets in use, slots per region}: written by the previous call's tail
    unsigned int*             cursor;        // [kVoxBuckets] partials offered to each bucket so far (zero at launch)
    unsigned long long*       keys_r;
    void*                     part_r;
    unsigned short*           bucket_of;
};
hipError_t voxel_begin(uint32_t capacity_points, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws, VoxelStage* stage,
                       hipStream_t st);
hipError_t voxel_finish(uint32_t capacity_points, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws, int16_t* d_out,
                        int32_t* d_out_points, hipStream_t st);
// Partials as an exchange format (multi-GPU config 5): a stage that appends (raw voxel key, sums) to caller arrays
// (d_count: one word = partials appended), and the sort + segmented mean over caller-held partials from any number of
// such stages (same leaf).
hipError_t voxel_partials_stage(int leaf_mm, unsigned long long* d_keys, void* d_partials, unsigned int* d_count, VoxelStage* stage,
                                hipStream_t st);
hipError_t launch_voxel_from_partials(const unsigned long long* d_keys, const void* d_partials, uint32_t n_partials,
                                      const int32_t* d_n_partials, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws,
                                      int16_t* d_out, int32_t* d_out_points, hipStream_t st);
// fault injection: the next `launches` bucket-tail launches end flagged (*out_points = -1), as after a stalled workgroup
void inject_voxel_stall(int launches);
// the same table fed from a 16-byte aligned payload (pcs_kernels.hip; the unaligned forms stay in pcs_voxel.hip)
hipError_t launch_payload_voxel_partials(const int16_t* d_payload, uint32_t n_points, const int32_t* d_n_points,
                                         const VoxelStage& vs, hipStream_t st);
// max_w / max_h: the largest raster of the launch; patch_ok: every raster's width is a multiple of 8 (square patches)
hipError_t launch_fused_voxel_partials(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                                       uint32_t max_w, uint32_t max_h, bool patch_ok, bool any_dist, uint32_t flags,
                                       MathSel math, const FramePtrs& fp, const VoxelStage& vs, hipStream_t st);

// Centre-side re-transform of already packed payloads (src/pcs-multicamera-optimized.cpp:226-265, 289): n clouds in one launch
// (blockIdx.y = cloud), each decoded, moved by its own 3x4 and re-packed at its camera-order offset of one stitched payload.
constexpr int kXformBatch = 16;
struct XformCloud {
    const int16_t* in;          // the camera's payload as received
    uint8_t*       out;         // where its first kept record goes
    uint32_t       n_out;       // records written = ceil(n_in / ds)
    uint32_t       ds;          // keep every ds-th record (i % downsample == 0, :236)
    float          M[12];       // top three rows of transform[i], row-major
};
struct XformBatch { XformCloud c[kXformBatch]; };
hipError_t launch_transform_payloads(const XformBatch& xb, int n, uint32_t max_out, hipStream_t st);

// a7 with stride.
hipError_t launch_stitch(const int16_t* d_src, uint32_t src_points, int downsample,
                         int16_t* d_dst, hipStream_t st);

}  // namespace pcs
