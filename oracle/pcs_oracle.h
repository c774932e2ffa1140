/*
 * pcs_oracle.h — CPU oracle for the deproject -> transform -> pack hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under pointcloud_stitching_amd/ or include/ may include,
 * link, import or execute anything in oracle/. Legitimate users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg — always as the checker / the timed CPU baseline, never as the
 * thing shipped.
 *
 * PARITY PIN STATUS
 *   pack path (a1/a2/a7): pinned by the eight known-answer vectors the surveyor printed from the
 *     compiled reference TU (`-m -t1`; SURVEY.md Appendix B -> tests/golden/kat_appendix_b.json).
 *     The reference ships no tests or golden vectors of its own, and its translation unit cannot
 *     be rebuilt here under the rules (it needs <librealsense2/rs.hpp>, absent from the image; no
 *     stand-in headers), so there is no oracle/_ref. Beyond those eight points: PARITY UNPINNED.
 *   deprojection (a5): arithmetic lives in librealsense2 (apt package, unpinned: Dockerfile:20-23),
 *     absent from /root/reference. Restated from its published pinhole model (SURVEY.md
 *     Appendix E). No reference test pins it: PARITY UNPINNED.
 *
 * Every function cites the reference lines it follows (relative to the reference checkout).
 * The POD types come from the public header so that tests drive oracle and product with the
 * same structs.
 */
#ifndef PCS_ORACLE_H
#define PCS_ORACLE_H

#include "../include/pcs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cvttss2si semantics: truncate toward zero; NaN / out-of-range -> INT32_MIN ("integer indefinite"). */
int32_t pcs_oracle_cvtt(float f);

/* a5 restated: Z16 -> vertices[N*3], texcoords[N*2]. */
void pcs_oracle_deproject(const pcs_stream_config* sc, const uint16_t* depth,
                          float* vertices, float* texcoords);
/* same, honouring PCS_FLAG_TEXCOORD_HALF_PIXEL in flags (the older librealsense texcoord formula) */
void pcs_oracle_deproject_flags(const pcs_stream_config* sc, const uint16_t* depth, uint32_t flags,
                                float* vertices, float* texcoords);

/* a2 restated: src/pcs-camera-optimized.cpp:363-616 (`-m`). flags = PCS_FLAG_*; downsample>=1
 * applies a7's stride to the kept sequence. Returns points written to out (5 shorts each). */
int pcs_oracle_pack(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                    int n_points, const uint8_t* color, uint32_t flags, int downsample, int16_t* out);

/* a3: the scalar (non -m) variant, src/pcs-camera-optimized.cpp:620-667: textbook-order affine with
 * separate roundings, then x1000 in DOUBLE. Documented variant only — NOT the parity target (it
 * differs from -m by +-1 LSB on ~2.5 % of shorts) and the reference's own result depends on the
 * compiler's contraction choices. */
int pcs_oracle_pack_scalar_variant(const pcs_stream_config* sc, const float* vertices,
                                   const float* texcoords, int n_points, const uint8_t* color, int16_t* out);

/* a1 restated: src/pcs-camera-optimized.cpp:669-723 minus the socket. Returns payload bytes. */
int pcs_oracle_send_xyzrgb_pointcloud(const pcs_stream_config* sc, const float* vertices,
                                      const float* texcoords, int n_points, const uint8_t* color,
                                      uint32_t flags, int16_t* buffer, size_t buffer_shorts, int write_header);

/* a7 restated: src/pcs-multicamera-client.cpp:373-395. Returns total points. */
int pcs_oracle_stitch(const int16_t* const* cam_payload, const int* cam_points, int n_cams,
                      int downsample, int16_t* stitched_payload);

/* The centre-side re-transform of pcs-multicamera-optimized (src/pcs-multicamera-optimized.cpp:226-265, 289): decode, PCL
 * affine, re-encode. PCL's evaluation order is third-party: parity unpinned. Returns the records written. */
int pcs_oracle_transform_payload(const int16_t* in, int n_points, int downsample, const float* m16, int16_t* out);

/* a5 + a2 + a7 composed for n_streams cameras; payload only (no header). counts[n_streams] optional.
 * scratch-free for the caller: allocates its own vertices/texcoords. Returns total points, <0 on OOM. */
int pcs_oracle_process_frames(const pcs_stream_config* streams, int n_streams,
                              const uint16_t* const* depth, const uint8_t* const* color,
                              uint32_t flags, int downsample, int16_t* payload, int* counts);

/* Voxel-grid downsample as DEFINED by this build (not in the reference; see include/pcs_hip.h). Returns the
 * number of voxels written to out (room for n_points points needed), <0 on OOM. */
int pcs_oracle_voxel_grid(const int16_t* payload, int n_points, int leaf_mm, int16_t* out);

/* ---- timed CPU baseline (pcs_oracle_simd.c): AVX2/FMA + OpenMP forms of the same arithmetic,
 * bit-identical to the functions above for flags == 0, downsample == 1. ------------------------ */
int  pcs_oracle_simd_available(void);   /* 1 if the host CPU has AVX2+FMA */
int  pcs_oracle_pack_simd_omp(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                              int n_points, const uint8_t* color, int16_t* out, int n_threads);
void pcs_oracle_deproject_omp(const pcs_stream_config* sc, const uint16_t* depth,
                              float* vertices, float* texcoords, int n_threads);
/* bracket (A) of BASELINE.md: memset(5 000 000 B) + pack, i.e. the reference's timed region :291-293.
 * buffer_shorts = capacity of `buffer`; returns -1 (nothing written) if it cannot hold the memset region and
 * 2 + 5*n_points shorts — the reference itself has no such check and overflows beyond 999 999 points. */
int  pcs_oracle_send_simd_omp(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                              int n_points, const uint8_t* color, int16_t* buffer, size_t buffer_shorts, int n_threads);
int  pcs_oracle_max_threads(void);
/* first-touch placement of a sample buffer by the team that will use it (pack loop's work-sharing); src NULL = zero */
void pcs_oracle_place_omp(void* dst, const void* src, size_t n_points, size_t bytes_per_point, int n_threads);
void pcs_oracle_team_cpus(int* cpus, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
