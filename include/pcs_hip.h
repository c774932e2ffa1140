/*
 * pcs_hip.h — C ABI of libpcs_hip.so, the MI355X (gfx950) implementation of the
 * per-frame hot path of conix-center/pointcloud_stitching:
 *
 *     Z16 depth raster ──deproject──► XYZ (camera frame) ──4x4 rigid──► world
 *     RGB8 colour raster ──texcoord gather──► RGB
 *     ──► [x_mm y_mm z_mm R|G<<8 B] int16 x5 records (10 B / point)
 *     ──► N cameras concatenated in camera order (the "stitched" buffer)
 *
 * Citations are relative to the reference checkout (read for behaviour only):
 *   a1  sendXYZRGBPointcloud                 src/pcs-camera-optimized.cpp:669-723
 *   a2  copyPointCloudXYZRGBToBufferSIMD     src/pcs-camera-optimized.cpp:363-616   (-m path = parity target)
 *   a4  tf_mat / transform[i]                src/pcs-camera-optimized.cpp:64-72,
 *                                            src/pcs-multicamera-optimized.cpp:417-455
 *   a5  rs2::pointcloud::calculate / map_to  call sites src/pcs-camera-optimized.cpp:198-199, 288-289
 *       (third-party librealsense2; arithmetic restated in DESIGN.md "Deprojection contract")
 *   a6  buffer layout: BUF_SIZE, header      src/pcs-camera-optimized.cpp:27, 690, 715-720
 *   a7  sendStitchToUnity (concatenate)      src/pcs-multicamera-client.cpp:373-409
 *
 * The reference has no FFI; its seam is the C++ function pair a1/a2, which reads its inputs
 * through six rs2:: accessors and takes the extrinsic, thread count and flags from globals.
 * This header is what a maintainer would bind instead: plain pointers and
 * sizes, no C++ types, no exceptions across the boundary, every call returns a pcs_status.
 *
 * Conventions
 *   - All entry points return PCS_OK (0) or a negative pcs_status. Nothing calls exit().
 *   - The caller owns every buffer it passes. The context owns its device staging buffers,
 *     its per-stream constant tables and its HIP stream.
 *   - One context per host thread (the reference's kernel is not re-entrant either,
 *     src/pcs-camera-optimized.cpp:349-360). Contexts are independent of each other.
 *   - Functions suffixed _device take DEVICE pointers, enqueue on the context's HIP stream and
 *     return without synchronising (use pcs_synchronize). All others take HOST pointers and
 *     are synchronous.
 *   - There is no CPU fallback: without a usable HIP device pcs_create fails with
 *     PCS_ERR_NO_DEVICE.
 */
#ifndef PCS_HIP_H
#define PCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCS_ABI_VERSION 1

/* a6: the wire/buffer constants of the reference (src/pcs-camera-optimized.cpp:27-30). */
#define PCS_POINT_SHORTS   5          /* x y z (R|G<<8) B                                   */
#define PCS_POINT_BYTES    10
#define PCS_HEADER_SHORTS  2          /* payload starts at buffer + 2 shorts = byte 4  :690  */
#define PCS_REF_BUF_SIZE   5000000    /* BUF_SIZE: shorts malloc'd, BYTES memset        :27,157,673 */
#define PCS_MAX_STREAMS    64

typedef enum pcs_status {
    PCS_OK               =  0,
    PCS_ERR_INVALID_ARG  = -1,
    PCS_ERR_NO_DEVICE    = -2,   /* no HIP device / device index out of range */
    PCS_ERR_HIP          = -3,   /* a HIP runtime call failed; see pcs_last_error */
    PCS_ERR_UNSUPPORTED  = -4,   /* e.g. distortion model not covered, bpp < 3 */
    PCS_ERR_CAPACITY     = -5,   /* caller's buffer too small for the result */
    PCS_ERR_NOMEM        = -6
} pcs_status;

/* Same numbering as rs2_distortion for the models covered. */
typedef enum pcs_distortion {
    PCS_DISTORTION_NONE                   = 0,
    PCS_DISTORTION_MODIFIED_BROWN_CONRADY = 1,
    PCS_DISTORTION_INVERSE_BROWN_CONRADY  = 2,
    PCS_DISTORTION_FTHETA                 = 3,  /* unsupported unless coeffs are all zero */
    PCS_DISTORTION_BROWN_CONRADY          = 4   /* unsupported unless coeffs are all zero */
} pcs_distortion;

/* Field order of rs2_intrinsics so a binding can memcpy one into the other. */
typedef struct pcs_intrinsics {
    int32_t width, height;
    float   ppx, ppy;
    float   fx, fy;
    int32_t model;          /* pcs_distortion */
    float   coeffs[5];      /* k1 k2 p1 p2 k3 */
} pcs_intrinsics;

/* rs2_extrinsics: rotation is a COLUMN-major 3x3, translation in metres. */
typedef struct pcs_extrinsics {
    float rotation[9];
    float translation[3];
} pcs_extrinsics;

/* Everything the reference takes from librealsense profile queries and from its globals,
 * for one camera stream. */
typedef struct pcs_stream_config {
    pcs_intrinsics depth;            /* depth stream geometry + pinhole model                    */
    pcs_intrinsics color;            /* colour stream geometry + pinhole model (map_to target)   */
    pcs_extrinsics depth_to_color;   /* rs2 get_extrinsics(depth -> colour)                      */
    float          depth_scale;      /* metres per Z16 unit; 0.001f on D400                      */
    int32_t        color_bpp;        /* video_frame::get_bytes_per_pixel      (>= 3)     :374    */
    int32_t        color_stride;     /* video_frame::get_stride_in_bytes                 :375    */
    float          cam_to_world[16]; /* row-major 4x4, metres, last row unused = tf_mat  :64-67  */
} pcs_stream_config;

/* pcs_config.flags */
#define PCS_FLAG_CUTOFF         0x1u  /* -c: keep iff 0<z<=1.5 && -2<x<=2 (camera frame), compacted in
                                         ascending point order (= the reference's -c -m -t1 order) :499-577 */
#define PCS_FLAG_CUTOFF_COMPAT  0x2u  /* with CUTOFF: reproduce the reference's lane-reversed mask (point k of
                                         each aligned group of 4 is gated by point 3-k)  :501-502,519 */
#define PCS_FLAG_DROP_INVALID   0x4u  /* drop depth==0 pixels (not in the reference; north-star compaction) */
#define PCS_FLAG_FORCE_IEEE     0x8u  /* never use the certified reduced-instruction arithmetic (A/B testing);
                                         results are identical either way */
#define PCS_FLAG_TEXCOORD_HALF_PIXEL 0x10u /* deprojection: u = (px + 0.5)/W, v = (py + 0.5)/H as older librealsense
                                         releases computed texture coordinates; default is
                                         u = px/W, v = py/H. The reference's +0.5 and clamp (:434-444) follow either way */

typedef struct pcs_config {
    int32_t                  device;      /* HIP device ordinal */
    int32_t                  n_streams;   /* 1..PCS_MAX_STREAMS */
    const pcs_stream_config* streams;     /* n_streams entries, copied by pcs_create */
    uint32_t                 flags;       /* PCS_FLAG_* */
    int32_t                  downsample;  /* a7: keep every downsample-th kept point per stream; >= 1
                                             (src/pcs-multicamera-client.cpp:375,388) */
} pcs_config;

typedef struct pcs_ctx pcs_ctx;

/* ---- lifecycle ------------------------------------------------------------------------- */
int          pcs_abi_version(void);
int          pcs_device_count(void);                 /* >= 0, or a negative pcs_status */
int          pcs_create(pcs_ctx** out, const pcs_config* cfg);
void         pcs_destroy(pcs_ctx* ctx);
const char*  pcs_strerror(int status);
const char*  pcs_last_error(const pcs_ctx* ctx);     /* detail of the last failure on this ctx */

/* Replace one stream's camera->world matrix (the reference edits tf_mat in source, :64-67). */
int          pcs_set_cam_to_world(pcs_ctx* ctx, int stream, const float m16[16]);

/* Which arithmetic the fused kernels use for `stream`: 0 = IEEE expansion, 1 = certified reduced-instruction
 * form, 2 = certified + identity depth->colour rotation shortcut; 3 / 4 = 1 / 2 plus the no-overflow certificate
 * (conversions cannot reach 2^31, so no running maximum is kept) (DESIGN.md "Certified arithmetic"). The
 * results are bit-identical; this is a diagnostic. */
int          pcs_stream_math(const pcs_ctx* ctx, int stream);
/* 1 when the stream's colour ROW is certified independent of the depth value (depth->colour R = I, t_y = t_z = 0, no distortion, and a
 * sweep over every raster row x every Z16 value on the device at pcs_create found no exception): the voxel reader then takes the row
 * from a per-row table instead of computing it per pixel. Same bytes either way; PCS_ROW_CONST=0 at pcs_create turns it off (A/B). */
int          pcs_stream_color_row_const(const pcs_ctx* ctx, int stream);

/* Number of points one frame of `stream` deprojects to (= depth width*height). */
int          pcs_stream_points(const pcs_ctx* ctx, int stream);
/* Upper bound of the stitched payload in shorts for the current config (no header). */
size_t       pcs_max_payload_shorts(const pcs_ctx* ctx);

/* ---- a2 twin: bit-exact replacement of copyPointCloudXYZRGBToBufferSIMD (:363-616) ------ *
 * vertices  = pts.get_vertices()            n_points x {float x,y,z}
 * texcoords = pts.get_texture_coordinates() n_points x {float u,v}
 * color     = color.get_data(); geometry (w,h,bpp,stride) and tf_mat come from stream's config
 * pc_buffer = the reference's `buffer + 2` (payload pointer); needs 5*n_points shorts
 * returns the number of points written in *out_points (n_points, or the kept count under -c).
 * Any n_points >= 0 is accepted (the reference needs n % 4 == 0, :414).                      */
int pcs_copy_pointcloud_xyzrgb_to_buffer(pcs_ctx* ctx, int stream,
                                         const float* vertices, const float* texcoords, int n_points,
                                         const uint8_t* color, int16_t* pc_buffer, int* out_points);
int pcs_copy_pointcloud_xyzrgb_to_buffer_device(pcs_ctx* ctx, int stream,
                                         const float* d_vertices, const float* d_texcoords, int n_points,
                                         const uint8_t* d_color, int16_t* d_pc_buffer, int* d_out_points);

/* Batched device form of the a2 twin: n_clouds cameras' rs2::points arrays packed by ONE launch (per group of 16)
 * instead of one latency-bound launch per camera — the reference runs copyPointCloudXYZRGBToBufferSIMD once per
 * camera process (src/pcs-camera-optimized.cpp:363); a node that hosts all cameras hands them over together.
 * `clouds` is a HOST array whose pointer fields are DEVICE pointers; cloud i is packed exactly as
 * pcs_copy_pointcloud_xyzrgb_to_buffer_device(ctx, clouds[i].stream, ...) would. d_out_points (optional, device)
 * receives n_clouds ints. Under -c (a predicate) the clouds are processed one after the other.                  */
typedef struct pcs_cloud_desc {
    int32_t        stream;      /* which stream's tf_mat / colour geometry applies                               */
    int32_t        n_points;    /* pts.size()                                                                     */
    const float*   vertices;    /* pts.get_vertices()                                                             */
    const float*   texcoords;   /* pts.get_texture_coordinates()                                                  */
    const uint8_t* color;       /* color.get_data()                                                               */
    int16_t*       pc_buffer;   /* the reference's buffer + 2; 5*n_points shorts                                  */
} pcs_cloud_desc;
int pcs_copy_pointclouds_xyzrgb_to_buffer_device(pcs_ctx* ctx, int n_clouds, const pcs_cloud_desc* clouds,
                                                 int* d_out_points);

/* ---- a1 twin: sendXYZRGBPointcloud (:669-723) without the socket ----------------------- *
 * buffer must hold buffer_shorts shorts (the reference mallocs PCS_REF_BUF_SIZE). On return:
 * payload at buffer+2 shorts; if write_header, int32 LE payload byte count at byte 0 (:718);
 * every other byte below min(PCS_REF_BUF_SIZE, 2*buffer_shorts) is zero (:673).
 * *out_size_bytes = 5*count*sizeof(short) (:697).                                            */
int pcs_send_xyzrgb_pointcloud(pcs_ctx* ctx, int stream,
                               const float* vertices, const float* texcoords, int n_points,
                               const uint8_t* color, int16_t* buffer, size_t buffer_shorts,
                               int write_header, int* out_size_bytes);

/* ---- fused a5+a2 for all streams, output in a7's stitched layout ------------------------ *
 * depth[s]  : stream s Z16 raster, depth.width*depth.height uint16, row-major, tightly packed
 * color[s]  : stream s colour raster, color.height rows of color_stride bytes
 * stitched  : [int32 payload bytes][stream 0 points][stream 1 points]...   (header only if write_header;
 *             payload always starts at stitched + 2 shorts)
 * points_per_stream[s] (optional) = points stream s contributed; *out_size_bytes = payload bytes.
 * Synchronous. Pageable buffers are staged (upload, kernel, download: 2.2 ms for 8 x 1280x720). When EVERY raster and
 * the stitched buffer are page-locked and device-addressable (pcs_host_malloc / hipHostMalloc / hipHostRegister) and the
 * stitched buffer has room for the worst case, the kernels read and write the host memory directly — both directions of the
 * link at once, no staging copies: 1.6 ms (the link's own duplex limit for these volumes is 1.47 ms). PCS_ZERO_COPY=0
 * forces the staged route. */
int pcs_process_frames(pcs_ctx* ctx, const uint16_t* const* depth, const uint8_t* const* color,
                       int16_t* stitched, size_t stitched_shorts, int write_header,
                       int* points_per_stream, int* out_size_bytes);

/* Software-pipelined form of pcs_process_frames for frame loops (same inputs, same stitched layout, same
 * bytes). pcs_submit_frames queues the uploads and the kernel(s) of one frame-set into one of
 * PCS_PIPELINE_DEPTH device slots and returns a ticket; pcs_collect_frames waits for that frame-set and
 * downloads it. Writing the loop as  submit(k+1); collect(k);  lets the upload of the next frame-set run
 * while the previous payload downloads (PCIe is full duplex; the two directions use separate HIP streams).
 * The overlap needs page-locked host buffers (pcs_host_malloc): with pageable memory the runtime copies
 * synchronously and the pair simply costs what pcs_process_frames costs. depth[s] / color[s] must stay
 * valid and unmodified until the matching collect returns. Tickets must be collected in submission order.
 * PCS_ERR_CAPACITY from submit = all slots in flight (collect first).                              */
#define PCS_PIPELINE_DEPTH 2
int pcs_submit_frames(pcs_ctx* ctx, const uint16_t* const* depth, const uint8_t* const* color, int* ticket);
int pcs_collect_frames(pcs_ctx* ctx, int ticket, int16_t* stitched, size_t stitched_shorts, int write_header,
                       int* points_per_stream, int* out_size_bytes);

/* Device-resident form: pointers are device pointers, d_payload is the PAYLOAD pointer
 * (no header), payload_shorts its capacity. d_counts (optional) receives n_streams+1 int32:
 * per-stream point counts followed by the total. Asynchronous on the context stream.
 * Without CUTOFF/DROP_INVALID the counts are known on the host (pcs_stream_points) and
 * d_counts may be NULL.                                                                       */
int pcs_process_frames_device(pcs_ctx* ctx, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                              int16_t* d_payload, size_t payload_shorts, int32_t* d_counts);

/* The same with the per-tile kept counts HANDED IN by a producer that already knows them (whatever wrote the depth image on
 * the GPU — a decoder, a filter, a simulator — can count as it writes): the count pass, and with it the second read of the
 * Z16 rasters, does not run (8 x 1280x720 with invalid-depth drop: 5.3 us of kernel time and 14.7 MB of reads per frame-set
 * less; in a back-to-back frame loop, where the next call's count pass hides behind the tail of the previous emit launch
 * anyway, 32.1 -> 31.2 us per frame-set).
 * A tile is PCS_TILE_POINTS consecutive pixels of one stream (row-major, the last tile of a stream may be short);
 * d_tile_kept[pcs_stream_tile_base(ctx, s) + t] = how many pixels of tile t of stream s the context's predicate keeps
 * (PCS_FLAG_DROP_INVALID alone: its non-zero depth words). pcs_stream_tile_base(ctx, n_streams) = entries in all. The counts
 * must be exact for the output to be the stitched cloud; WRONG counts garble the cloud but cannot write outside the payload's
 * worst-case capacity (they are clamped to a tile). Without CUTOFF / DROP_INVALID the counts are ignored.              */
#define PCS_TILE_POINTS 2048
int pcs_stream_tile_base(const pcs_ctx* ctx, int stream);
int pcs_process_frames_device_counted(pcs_ctx* ctx, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                      const uint32_t* d_tile_kept, int16_t* d_payload, size_t payload_shorts, int32_t* d_counts);

/* Throughput form: n_sets frame-sets of the SAME streams per call. d_depth / d_color hold n_sets * n_streams device
 * pointers, frame-set major (entry k*n_streams + s = stream s of frame-set k); d_payload[k] is frame-set k's payload
 * pointer (each with payload_shorts capacity), d_counts (optional) n_sets pointers as in pcs_process_frames_device.
 * Each payload receives exactly the bytes pcs_process_frames_device would write. On the dense path (no
 * CUTOFF/DROP_INVALID, downsample 1, 16-byte aligned payloads) up to 64 / n_streams frame-sets share ONE kernel
 * launch, which amortises the fill and drain of a ~23 us launch (8 x 1280x720: 58 % -> ~68 % of HBM peak). With
 * CUTOFF/DROP_INVALID (downsample 1, n_streams <= 16) the same number of frame-sets share THREE launches — count,
 * scan, emit, each covering every set — instead of three per set (8 x 1280x720: 34 -> 25 us per frame-set, 39 % ->
 * 50 %), with no assumption about workgroup dispatch order. Other configurations are processed set by set. A caller
 * that must hand a frame-set on as soon as it is complete keeps using pcs_process_frames_device: batching trades
 * latency for throughput.                                                                                       */
int pcs_process_frames_device_batch(pcs_ctx* ctx, int n_sets, const uint16_t* const* d_depth,
                                    const uint8_t* const* d_color, int16_t* const* d_payload, size_t payload_shorts,
                                    int32_t* const* d_counts);

/* Deprojection only (a5 restated): Z16 -> vertices (N x 3 float) and texcoords (N x 2 float),
 * i.e. what rs2::pointcloud::calculate + map_to hand to a2. Host pointers. For parity tests
 * and for callers that still want the intermediate rs2::points arrays.                       */
int pcs_deproject(pcs_ctx* ctx, int stream, const uint16_t* depth, float* vertices, float* texcoords);

/* ---- a7: stitch already-packed per-camera payloads (sendStitchToUnity, :373-409) -------- *
 * Keeps every downsample-th 10-byte point of each camera, cameras in index order.
 * Device pointers; *total_points is written on the HOST (counts are host-known).             */
int pcs_stitch_device(pcs_ctx* ctx, const int16_t* const* d_cam_payload, const int* cam_points, int n_cams,
                      int downsample, int16_t* d_stitched_payload, size_t stitched_shorts, int* total_points);

/* ---- the centre's re-transform of already packed payloads (src/pcs-multicamera-optimized.cpp:226-265, 289) ---------------- *
 * What the reference's pcs-multicamera-optimized does with every camera's payload before it concatenates them
 * (updateCloudXYZRGB, :268-299; send_stitchedXYZRGB, :301-318) — in contrast with pcs-multicamera-client, which copies the
 * records as they are (pcs_stitch_device). Per kept record (i % downsample == 0, :236):
 *   x,y,z = (float)int16 / 1000.0f                     CONV_RATE is `const float CONV_RATE = 1000.0` in THAT file (:46)
 *   p'    = ((m0*x + m1*y) + m2*z) + m3 per row         pcl::transformPointCloud(cloud, cloud, transform[i]) (:289): PCL 1.8
 *                                                       transforms.hpp's expression; products and sums individually rounded
 *                                                       (the target is built without -mfma). THIRD-PARTY: parity unpinned.
 *   int16 = low 16 bits of cvttss2si(p' * 1000.0f)      static_cast<short>(x * CONV_RATE) (:255-257)
 *   colour: short 3 (R | G<<8) survives bit for bit, short 4 becomes B with a zero high byte (:240-242, :258-259)
 * The round trip is LOSSY (decode, move, truncate) — it is the reference program's behaviour, not an improvement; the default
 * of the work-alike CLI stays the lossless concatenation. Cameras are written in index order into ONE stitched
 * payload (what `*stitched_cloud += *cloud_ptr[i]`, :361-364, and convertPointCloudXYZRGBToBuffer produce); all cameras of a
 * call share launches of up to 16 clouds. `cams` is a HOST array whose payload pointers are DEVICE pointers. A camera may be
 * transformed in place (its output slice starting exactly at its input, downsample 1); any other overlap of an input with an
 * output slice is refused. points_per_cam (optional, host) and *total_points are host-known: FLOOR(n_points / downsample) —
 * the reference sizes its cloud with size / downsample, rounded down (:230); when size % downsample != 0 its loop writes one
 * element past the vector (:235-246, undefined behaviour that never grows it), and transformPointCloud, += and
 * convertPointCloudXYZRGBToBuffer (`i < cloud->width`, :253) all iterate the width: the last partial stride's record is not
 * sent. (pcs_stitch_device keeps CEIL(n / downsample): that is the other program's loop, `j += 5 * downsample` while
 * j < buf_len, src/pcs-multicamera-client.cpp:388.) 20 B of HBM traffic per kept record.                                      */
typedef struct pcs_payload_desc {
    const int16_t* d_payload;      /* the camera's packed records (no header), device memory                             */
    int32_t        n_points;
    float          transform[16];  /* transform[i], row-major 4x4 in metres; the last row is not used                    */
} pcs_payload_desc;
int pcs_transform_payloads_device(pcs_ctx* ctx, int n_cams, const pcs_payload_desc* cams, int downsample,
                                  int16_t* d_stitched_payload, size_t stitched_shorts, int* points_per_cam, int* total_points);

/* ---- voxel-grid downsample of a packed payload (BASELINE config 5) ------------------------------ *
 * NOT in the reference (it includes pcl/filters/voxel_grid.h but never instantiates it). Defined in the
 * payload's integer millimetre domain: voxel = floor(coord / leaf_mm) per axis; one output point per occupied
 * voxel = integer mean of x,y,z (truncating) and of R,G,B; output sorted by (z,y,x) voxel, x fastest.
 * The output needs room for n_points points in the worst case. *d_out_points / *out_points = voxels written. The device forms
 * write -1 there if the bucket tail gave up waiting for one of its own workgroups after ~0.5 s (a stalled or preempted device; the
 * wait is bounded so that a launch can end wrong but never hang; it has not been observed outside pcs_inject_voxel_stall): the
 * bytes of such a call are NOT valid. Whoever reads a negative count calls pcs_set_voxel_tail(ctx, PCS_VOXEL_TAIL_LSD_LATCHED) and
 * runs the call again — the LSD tail waits for nobody. Every form of this library that reads the count itself does exactly that
 * (pcs_voxel_grid, libpcs_node's waits, the CLIs) and reports PCS_ERR_HIP only if the second run is negative too: a negative
 * length never reaches a caller's size arithmetic or the wire (src/pcs-multicamera-client.cpp:394-403).
 * The device form is fully asynchronous on the context stream (pre-aggregation, radix sort and segmented mean are
 * hand-written kernels that read their sizes from device memory; no host round trip in the middle).            */
int pcs_voxel_grid_device(pcs_ctx* ctx, const int16_t* d_payload, int n_points, int leaf_mm,
                          int16_t* d_out, size_t out_shorts, int32_t* d_out_points);
/* Counted form for device pipelines: the number of points is READ FROM DEVICE MEMORY when the kernels run (e.g. the total
 * pcs_process_frames_device left in d_counts[n_streams] under CUTOFF / DROP_INVALID), max_points is the capacity of the
 * payload (it sizes the workspace; the output needs room for max_points points). BASELINE config 5 — compaction, stitch,
 * voxel grid — then is two asynchronous calls with no host round trip between them.                                */
int pcs_voxel_grid_device_counted(pcs_ctx* ctx, const int16_t* d_payload, const int32_t* d_n_points, int max_points,
                                  int leaf_mm, int16_t* d_out, size_t out_shorts, int32_t* d_out_points);
/* Which tail the voxel calls of this context take behind the pre-aggregation. Both give the same bytes for every input:
 *   PCS_VOXEL_TAIL_BUCKET  one partition of the partials into <= 1024 key ranges + one workgroup per range with an LDS table
 *                          (built for the ~1 M partials of BASELINE configs[4] at leaves of a few centimetres and up). The first
 *                          call of a context for a leaf partitions in launches of its own (4 in all); every later call of
 *                          pcs_voxel_grid_device[_counted] / pcs_process_frames_voxel_device has the pre-aggregation put its
 *                          partials into the ranges' regions itself — sized, like the splitters, by the call before — and the tail
 *                          is ONE launch. A cloud that moved costs that call speed (regions overflow into a list that is gathered),
 *                          never bytes. PCS_VOXEL_REGIONS=0 (read at every call) keeps every call on the first call's chain;
 *   PCS_VOXEL_TAIL_LSD     LSD radix sort of (key, partial) + segmented mean (12 launches; the better tool for many millions of
 *                          partials, and the lighter neighbour when the tail runs BESIDE another frame-set's pre-aggregation on the
 *                          same GPU: libpcs_node picks it for its root when one GPU holds every camera);
 *   PCS_VOXEL_TAIL_AUTO    (default) by the leaf: bucket from 34 mm.
 * The environment variable PCS_VOXEL_TAIL=bucket|lsd, read at every call, overrides both (the test-suite runs every voxel
 * test under each).                                                                                                      */
#define PCS_VOXEL_TAIL_AUTO   0
#define PCS_VOXEL_TAIL_BUCKET 1
#define PCS_VOXEL_TAIL_LSD    2
/* after a flagged call (*out_points == -1): LSD from here on for this context, NOT overridable by the environment, both control
 * blocks of the workspace cleared before the next call; counted by pcs_voxel_tail_reruns. AUTO / BUCKET / LSD lift the latch. */
#define PCS_VOXEL_TAIL_LSD_LATCHED 3
int pcs_set_voxel_tail(pcs_ctx* ctx, int tail);
int pcs_voxel_tail_reruns(const pcs_ctx* ctx);     /* how often LSD_LATCHED was set on this context (by the library or the caller) */
/* Fault injection (tests; the CLIs: environment PCS_BKT_INJECT_STALL=<launches>, read once): the next `launches` bucket-tail
 * launches of this PROCESS run with their first workgroup asleep for 0.4 ms and a 20 us wait bound, i.e. they give up and end
 * flagged exactly as a launch on a stalled device would. 0 clears.                                                          */
int pcs_inject_voxel_stall(int launches);

/* Rasters -> voxel grid in one asynchronous call, WITHOUT materialising the stitched cloud: the result (bytes and
 * *d_out_points) is exactly pcs_voxel_grid_device applied to the payload pcs_process_frames_device would write for the
 * same rasters under the context's flags (CUTOFF / DROP_INVALID honoured; the voxel sums are integers, so the order of
 * the points is irrelevant and neither the ordered placement nor the 10-byte records ever touch HBM; 16 x 1920x1080 at
 * a 50 mm leaf: 0.245 ms instead of 0.37 ms for pcs_process_frames_device + pcs_voxel_grid_device_counted). The output
 * needs room for every pixel of the frame-set in the worst case (sum of the streams' n_points). With a downsample
 * stride the stitched cloud is built internally first (the stride is defined on the kept points' ORDER).             */
int pcs_process_frames_voxel_device(pcs_ctx* ctx, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                    int leaf_mm, int16_t* d_out, size_t out_shorts, int32_t* d_out_points);
int pcs_voxel_grid(pcs_ctx* ctx, const int16_t* payload, int n_points, int leaf_mm,
                   int16_t* out, size_t out_shorts, int* out_points);

/* ---- voxel PARTIALS: the exchange format of a multi-GPU voxel grid (BASELINE configs[4]: 16 streams, 2 per GPU) ---------- *
 * The voxel sums are integers, so the grid of a stitched cloud is the grid of the UNION of its cameras' points in any
 * order: each GPU pre-aggregates its own cameras into partials — one per voxel seen by a patch of pixels: the voxel's key
 * (z, y, x voxel indices packed, a function of leaf_mm alone) and the seven sums — the partials of all GPUs are
 * concatenated on the root (keys with keys, sums with sums; ~40 B per occupied voxel and patch instead of 10 B per point,
 * 7x fewer bytes over xGMI for 16 x 1920x1080 at 50 mm) and ONE sort + segmented mean there gives exactly the bytes
 * pcs_voxel_grid_device would give for the stitched cloud. The reference concatenates the cameras' full payloads on the
 * centre (src/pcs-multicamera-client.cpp:373-409) and has no voxel grid (src/pcs-multicamera-optimized.cpp:17 only
 * includes the header); libpcs_node / pointcloud_stitching_amd.stitch drive the exchange.                                */
typedef struct pcs_voxel_partial {          /* 32 bytes */
    int32_t  sx, sy, sz;                    /* sums of the int16 millimetre coordinates of the points of this partial      */
    uint32_t r, g, b;                       /* sums of the colour bytes                                                      */
    uint32_t n;                             /* points summed (>= 1)                                                          */
    uint32_t pad;
} pcs_voxel_partial;
#define PCS_VOXEL_PARTIAL_WIRE_BYTES 40     /* one uint64 key + one pcs_voxel_partial                                        */

/* Rasters -> partials of this context's streams under its flags (CUTOFF / DROP_INVALID / downsample honoured exactly as
 * pcs_process_frames_voxel_device does). d_keys / d_partials need room for `capacity` >= pcs_max_payload_shorts / 5 entries
 * (worst case: every kept point its own partial); *d_n_partials (device, 4-byte aligned) receives how many were written:
 * it is the kernels' own append counter, cleared by the call — valid once the call's work on the stream is complete, not
 * before. Asynchronous.                                                                                                   */
int pcs_process_frames_voxel_partials_device(pcs_ctx* ctx, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                             int leaf_mm, uint64_t* d_keys, pcs_voxel_partial* d_partials, size_t capacity,
                                             int32_t* d_n_partials);
/* Partials (of any number of pcs_process_frames_voxel_partials_device calls with the SAME leaf_mm, concatenated in any
 * order) -> the voxel grid. n_partials entries are read, or *d_n_partials (device, <= n_partials, which then is the
 * capacity) when that pointer is given. The output needs room for n_partials points. The partials must be ones this library
 * produced (or obey its bounds: 1 <= n <= 32 768 points per partial, i.e. colour sums < 2^23): 256 of them are summed in 32
 * bits; a voxel whose partials have n == 0 throughout is written with its sums undivided.
 * Asynchronous.                                                                                                           */
int pcs_voxel_grid_from_partials_device(pcs_ctx* ctx, const uint64_t* d_keys, const pcs_voxel_partial* d_partials,
                                        int n_partials, const int32_t* d_n_partials, int leaf_mm, int16_t* d_out,
                                        size_t out_shorts, int32_t* d_out_points);

/* ---- voxel SINK: several contexts of ONE device pre-aggregate into one context's workspace ------------------------------ *
 * pcs_process_frames_voxel_device in three steps, the middle one callable on OTHER contexts of the same device: the partials of
 * every such context land where the sink context's own would — in its buckets' regions on a warm call — so nothing is exchanged,
 * concatenated or placed, and the tail is the one launch of a warm call. libpcs_node takes this route for peers that share the
 * root's GPU (a device id that repeats); peers on OTHER GPUs exchange partials (above): the pre-aggregation's atomics are
 * device-scope and a sink refuses a context of another device.
 *   pcs_voxel_sink_begin   on the sink context's stream: workspace for `capacity_points` (the sum of every participating context's
 *                          pixels), control blocks; fills *sink (opaque; valid until the finish). One sink per context at a time — a
 *                          second begin abandons the first.
 *   pcs_process_frames_voxel_into_sink_device   on ctx's stream: ctx's rasters under ctx's flags (CUTOFF / DROP_INVALID / downsample,
 *                          as pcs_process_frames_voxel_partials_device) into the sink. The CALLER orders the streams: this launch behind
 *                          the begin (when sink->work_enqueued) and behind the sink's previous finish, the finish behind every such
 *                          launch — events, as libpcs_node does (csrc/pcs_node.cpp: enqueue_sink_ticket).
 *   pcs_voxel_sink_finish  on the sink context's stream: the tail; bytes, *d_out_points and the -1 convention exactly as
 *                          pcs_process_frames_voxel_device over the union of the participating rasters. d_out needs room for
 *                          capacity_points points.                                                                                  */
typedef struct pcs_voxel_sink {
    uint64_t opaque[23];
    uint32_t work_enqueued;   /* begin put work on the sink's stream (the clear of a workspace's control blocks: its first call, or after a
                                 call that failed half-way) that the pre-aggregations must run behind; 0: nothing to order against the begin */
    uint32_t reserved;
} pcs_voxel_sink;
int pcs_voxel_sink_begin(pcs_ctx* sink_ctx, size_t capacity_points, int leaf_mm, pcs_voxel_sink* sink);
int pcs_process_frames_voxel_into_sink_device(pcs_ctx* ctx, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                              const pcs_voxel_sink* sink);
int pcs_voxel_sink_finish(pcs_ctx* sink_ctx, const pcs_voxel_sink* sink, int16_t* d_out, size_t out_shorts, int32_t* d_out_points);

/* ---- stream / timing plumbing ----------------------------------------------------------- */
int   pcs_set_stream(pcs_ctx* ctx, void* hip_stream);   /* adopt a caller-owned hipStream_t (NULL = own stream) */
void* pcs_get_stream(pcs_ctx* ctx);
/* Two contexts of one device used IN TURN overlap — the tail of one call beside the head of the next (BASELINE configs[4]: the voxel
 * pipeline's bucket tail beside the next frame-set's pre-aggregation, 0.169 -> 0.152 ms per 16 x 1080p frame-set) — only if their streams
 * sit on different hardware queues; the runtime deals streams onto a few queues round robin, so that is luck. pcs_use_stream_beside
 * replaces ctx's OWN stream by one that is SEEN to run beside other's current stream (a no-op must finish while a 300 us spin occupies
 * the other; up to six candidates; ~2 ms, once). Returns 1 when such a stream was found and taken, 0 when none was (ctx keeps its
 * stream), a negative status on error. pcs_pick_concurrent_stream is the same probe on raw hipStream_t handles of the CURRENT device
 * (*out_stream = NULL when none overlapped; the caller owns the stream it gets). libpcs_node uses both.                              */
int   pcs_use_stream_beside(pcs_ctx* ctx, pcs_ctx* other);
int   pcs_pick_concurrent_stream(void* busy_hip_stream, void** out_stream);
int   pcs_synchronize(pcs_ctx* ctx);
/* hipEvent bracket on the context stream: begin; enqueue work; end; elapsed = GPU ms between them. */
int   pcs_timer_begin(pcs_ctx* ctx);
int   pcs_timer_end(pcs_ctx* ctx);
int   pcs_timer_elapsed_ms(pcs_ctx* ctx, float* ms);    /* synchronises on the end event */
/* Per-launch kernel timing: when enabled every fused-kernel launch is bracketed by its own
 * event pair; pcs_kernel_times_ms drains them (synchronising) into ms[0..*n).                 */
int   pcs_kernel_timing(pcs_ctx* ctx, int enable);
int   pcs_kernel_times_ms(pcs_ctx* ctx, float* ms, int capacity, int* n);

/* Page-locked host memory for the buffers that cross PCIe every frame (the reference mallocs its `buffer`
 * once, src/pcs-camera-optimized.cpp:157). When every buffer of a host call is page-locked the call runs zero copy
 * (pcs_process_frames: 1.6 instead of 2.25 ms per 8x720p frame-set; pcs_copy_pointcloud_xyzrgb_to_buffer likewise);
 * with pageable memory the calls stage through device buffers. What is really expensive is handing over a freshly
 * allocated buffer every call (first-touch page faults, 7.5 ms). */
int   pcs_host_malloc(pcs_ctx* ctx, void** h_ptr, size_t bytes);
int   pcs_host_free(pcs_ctx* ctx, void* h_ptr);
/* Page-lock memory the caller already owns (the reference's `buffer = (short*)malloc(sizeof(short) * BUF_SIZE)`,
 * src/pcs-camera-optimized.cpp:157; a capture pipeline's frame pool) so that the host entry points run zero copy on it.
 * Register once, unregister before free(). */
int   pcs_host_register(pcs_ctx* ctx, void* h_ptr, size_t bytes);
int   pcs_host_unregister(pcs_ctx* ctx, void* h_ptr);

/* Thin device-memory helpers so a C/C++ host needs no HIP headers. */
int   pcs_device_malloc(pcs_ctx* ctx, void** d_ptr, size_t bytes);
int   pcs_device_free(pcs_ctx* ctx, void* d_ptr);
int   pcs_memcpy_h2d(pcs_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int   pcs_memcpy_d2h(pcs_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* PCS_HIP_H */
