// pcs_kernels.hip — hand-written gfx950 (CDNA4) kernels for the deproject -> rigid transform ->
// RGB attach -> XYZRGB int16 pack hot path of conix-center/pointcloud_stitching.
//
// What is replaced (reference file:line, read for behaviour only):
//   a5 rs2::pointcloud::calculate/map_to   call sites src/pcs-camera-optimized.cpp:198-199, 288-289
//   a2 copyPointCloudXYZRGBToBufferSIMD    src/pcs-camera-optimized.cpp:363-616
//   a7 sendStitchToUnity's concatenate     src/pcs-multicamera-client.cpp:385-392
//
// Shape of the work: per point ~15 algorithmic bytes (2 B Z16 in, 3 B RGB in, 10 B out) against a few
// dozen FP32 ops -> HBM-bound, no MFMA. What matters on CDNA4:
//   * every global access is a lane-contiguous 16 B vector access: the Z16 raster is read as uint4
//     (8 pixels / lane, 1 KiB / wavefront-instruction); the 10-byte records, which no lane can store
//     on its own at a 16 B boundary, are transposed through LDS so a wavefront writes 5 x 1 KiB;
//   * per-camera constants (3x4 extrinsic, 3x3+t depth->colour, intrinsics) are wave-uniform and sit
//     in SGPRs via scalar loads; the per-column / per-row deprojection LUTs are L2-resident;
//   * the colour gather is one (possibly unaligned) dword per point through L1/L2;
//   * invalid-depth / cutoff compaction is order-preserving: per-lane popcount -> wavefront scan ->
//     4-entry LDS cross-wave scan -> tile prefix from a count pass (deterministic, = `-c -m -t1` order).
//
// Bit-exactness: compiled with -ffp-contract=off; every fused op is an explicit __fmaf_rn and every
// other product/sum/quotient is individually rounded (IEEE divide), mirroring oracle/pcs_oracle_impl.h.
// Float->int follows x86 cvttss2si including its "integer indefinite" result for NaN / out of range,
// which v_cvt_i32_f32 (saturating) does not give by itself.

// Two translation units from this one source (Makefile): pcs_kernels.o (PCS_TU_VOXEL=0: everything but the voxel readers,
// SLP vectorisation off — packed FP32 measured 1-2 % slower on the HBM-bound kernels) and pcs_kernels_voxel.o
// (PCS_TU_VOXEL=1: the raster / payload voxel readers, SLP on — that kernel is VALU-bound and v_pk_fma_f32 / v_pk_mul_f32
// take 15 % of its vector instructions away: 146 -> 140 us per 16 x 1080p frame-set). A packed op is two independent,
// individually rounded IEEE operations: the bits do not change.

#include <cstddef>

#include "pcs_device.h"

#ifndef PCS_TU_VOXEL
#define PCS_TU_VOXEL 0
#endif


#ifndef EMIT_WAVES
#define EMIT_WAVES 6      // 7 fits 72 VGPRs only with scratch spills in some instantiations and measured no faster
#endif

namespace pcs {

namespace {

constexpr uint32_t kDenseStageBytes = kTilePoints * PCS_POINT_BYTES;        // 20 480 B -> 8 workgroups / CU
constexpr uint32_t kStageBytes      = kTilePoints * PCS_POINT_BYTES + 32;   // + head skew + tail pad

// Pointers that reach a kernel through memory (the StreamParams table) have no address space the
// compiler can see and would be accessed with flat_load; they are always HBM, so say so.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class T> using gptr = const __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ gptr<T> as_global(const T* p)
{
    return (gptr<T>)(uintptr_t)p;
}

struct PointIn {
    float X, Y, Z;   // camera-frame vertex (rs2::vertex)
    float u, v;      // texture coordinate  (rs2::texture_coordinate)
};

// cvttss2si / _mm_cvttps_epi32: truncate; NaN or |f| >= 2^31 -> 0x80000000.
__device__ __forceinline__ int32_t cvtt_x86(float f)
{
    return (__builtin_fabsf(f) < 2147483648.0f) ? (int32_t)f : (int32_t)0x80000000;
}

// v_cvt_i32_f32 as the hardware does it: truncate, saturate, NaN -> 0.
__device__ __forceinline__ int32_t cvt_sat(float f)
{
    int32_t i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(f));
    return i;
}

// Float->int conversion policies. The five conversions of a point (world x,y,z; colour column,row) are
// consumed only as `& 0xFFFF` or as clamp(., 0, dim-1). Under those two uses the saturating hardware
// convert differs from cvttss2si in exactly one case: f >= 2^31 (hardware INT_MAX, x86 INT_MIN); NaN
// gives 0 vs INT_MIN, which agree both in the low 16 bits and after the clamp. FastCvt therefore uses
// the 1-instruction hardware convert and keeps a running maximum of everything it converted (v_max3
// ignores NaN); the tile code re-does a lane's points with ExactCvt in the (practically never taken)
// case that the maximum reached 2^31.
struct ExactCvt {
    [[maybe_unused]] static constexpr bool kCoordsInShort = false;       // a converted coordinate may lie outside int16: the record keeps its low 16 bits
    __device__ __forceinline__ void note(float, float, float, float, float) {}
    __device__ __forceinline__ int32_t cvt(float f) const { return cvtt_x86(f); }
    // colour column / row: clamp(cvttss2si(f), 0, dim-1)   (:438-444)
    __device__ __forceinline__ int32_t pixel(float f, int32_t dim_m1, float) const
    {
        return min(max(cvtt_x86(f), 0), dim_m1);
    }
    // One dword covers R,G,B. Never read past the raster: slide the window back at the very end of it and
    // shift the wanted bytes down.
    __device__ __forceinline__ uint32_t window(uint32_t idx, uint32_t lim, uint32_t& shift)
    {
        const uint32_t off = min(idx, lim);
        shift = (idx - off) * 8u;
        return off;
    }
    __device__ __forceinline__ bool redo() const { return false; }
};

// TRACK = false is for streams whose certificate also proves that no converted value can reach 2^31
// (pcs_capi.cpp: certify_no_overflow): then the running maximum is not needed at all.
template <bool TRACK>
struct FastCvt {
    [[maybe_unused]] static constexpr bool kCoordsInShort = false;
    float    hi = 0.0f;
    uint32_t max_idx = 0;
    uint32_t lim = 0xFFFFFFFFu;
    __device__ __forceinline__ void note(float a, float b, float c, float d, float e)
    {
        if (TRACK) {
            hi = __builtin_fmaxf(__builtin_fmaxf(hi, a), b);
            hi = __builtin_fmaxf(__builtin_fmaxf(hi, c), d);
            hi = __builtin_fmaxf(hi, e);
        }
    }
    __device__ __forceinline__ int32_t cvt(float f) const { return cvt_sat(f); }
    // Clamp in the float domain first (one v_med3_f32; NaN -> 0 like the x86 path), then convert: for
    // f < 2^31 this equals clamp(trunc(f), 0, dim-1); f >= 2^31 is the case redo() reports.
    __device__ __forceinline__ int32_t pixel(float f, int32_t, float dim_m1_f) const
    {
        return cvt_sat(__builtin_amdgcn_fmed3f(f, 0.0f, dim_m1_f));
    }
    // The window only ever slides for the raster's very last pixel: clamp the address (so nothing past the
    // raster is read), remember the largest index seen, and let redo() send the lane through the exact path
    // if any index actually needed the slide.
    __device__ __forceinline__ uint32_t window(uint32_t idx, uint32_t l, uint32_t& shift)
    {
        max_idx = max(max_idx, idx);
        lim = l;
        shift = 0u;
        return min(idx, l);
    }
    __device__ __forceinline__ bool redo() const { return (TRACK && hi >= 2147483648.0f) || max_idx > lim; }
};
using LazyCvt = FastCvt<true>;

// The voxel readers consume a point's coordinates as numbers, not as the record's 16-bit fields: when every converted
// coordinate of the lane lies inside int16 the converted value IS the record's field (no pack, no sign extension per point).
// This policy keeps a second running maximum, of |x|, |y|, |z| in millimetres (the same three instructions as FastCvt<true>'s
// one maximum over five values), and sends the lane through the exact path — whose values are then wrapped like the record's —
// when it reached 2^15. The colour coordinates keep their 2^31 check (TRACK) as in FastCvt.
template <bool TRACK>
struct VoxCvt : FastCvt<TRACK> {
    static constexpr bool kCoordsInShort = true;
    float hc = 0.0f;
    __device__ __forceinline__ void note(float a, float b, float c, float d, float e)
    {
        hc = __builtin_fmaxf(__builtin_fmaxf(hc, __builtin_fabsf(a)), __builtin_fabsf(b));
        hc = __builtin_fmaxf(hc, __builtin_fabsf(c));
        if (TRACK) this->hi = __builtin_fmaxf(__builtin_fmaxf(this->hi, d), e);
    }
    __device__ __forceinline__ bool redo() const { return hc >= 32768.0f || FastCvt<TRACK>::redo(); }
};

// Arithmetic policy of the depth->colour projection. Every policy the product launches is bit-identical
// to IeeeMath on the inputs it is launched for (see "certification" in pcs_capi.cpp and DESIGN.md);
// tools/lab/kernel_lab.hip holds the exhaustive / fuzz checks and the measurements behind each choice.
struct IeeeMath {
    static constexpr bool kIdentR = false;
    static constexpr bool kRowConst = false;
    // 0 exact conversions, 1 fast with overflow tracking, 2 fast, overflow certified impossible.
    // The tracked fast form is exact for every input (its redo path IS the exact form), so even the
    // fallback policy uses it; only the quotients stay on the IEEE expansion here.
    static constexpr int kCvtMode = 1;
    // rs2_transform_point_to_point: R column-major, products and sums individually rounded, left to right
    static __device__ __forceinline__ void d2c(const StreamParams& P, float X, float Y, float Z,
                                               float& P0, float& P1, float& P2)
    {
        P0 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.R[0], X), __fmul_rn(P.R[3], Y)), __fmul_rn(P.R[6], Z)), P.t[0]);
        P1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.R[1], X), __fmul_rn(P.R[4], Y)), __fmul_rn(P.R[7], Z)), P.t[1]);
        P2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.R[2], X), __fmul_rn(P.R[5], Y)), __fmul_rn(P.R[8], Z)), P.t[2]);
    }
    // two quotients over one denominator (rs2_project_point_to_pixel: x = P0/P2, y = P1/P2)
    static __device__ __forceinline__ void div2(float a0, float a1, float b, float& q0, float& q1)
    {
        q0 = __fdiv_rn(a0, b);
        q1 = __fdiv_rn(a1, b);
    }
    // quotient by a wave-uniform constant (pixel_to_texcoord: px / width); rc = host-computed RN(1/c)
    static __device__ __forceinline__ float div_const(float a, float c, float /*rc*/) { return __fdiv_rn(a, c); }
};

// CertMath: the same results with fewer instructions, launched only for streams whose configuration
// the host has certified (pcs_capi.cpp: certify_stream) and, for div_const, the device has verified.
//  * div2 is the IEEE-754 division expansion of this compiler (v_rcp, one Newton step on the
//    reciprocal, multiply, two fused corrections) WITHOUT v_div_scale / v_div_fmas / v_div_fixup, with
//    the refined reciprocal shared by both numerators. v_div_scale only ever rescales operands whose
//    exponents lie outside a window; the host proves from the configuration (depth scale, LUT ranges,
//    R, t) that every valid pixel's P0,P1,P2 lie inside it, and the pixels it cannot speak for (depth 0)
//    have their quotients discarded. Inside the window the two sequences are the same arithmetic (lab fuzz:
//    0 differences in 3.4e10 triples incl. adversarial mantissas) with ONE exception: for a numerator of
//    -0 this returns +0 where IEEE returns -0 (v_div_fixup restores the sign). The pack cannot observe it:
//    x = +-0 gives px = +-0*fx + ppx, and u = +-0 gives fma(u, W, .5) = .5 either way. pcs_deproject, which
//    exposes u and v themselves, always uses IeeeMath.
//  * div_const is Markstein's quotient: with y = RN(1/c), q0 = RN(a*y), r = a - c*q0 (exact in one fma),
//    q = RN(q0 + r*y). It is consumed only through trunc(fma(q, c, 0.5)) clamped to [0, c-1]; that
//    composite is compared with the IEEE one over ALL 2^32 numerators on the device when the context
//    is created (pcs_verify_div_const_kernel) and CertMath is used only if no numerator differs.
//  * IDENT_R: depth->colour rotation is exactly the identity (and t has no negative zeros), so
//    R*p + t is p + t: the dropped products are exact (1*x) or signed zeros that cannot change a sum.
template <bool IDENT_R, bool NO_OVERFLOW = false>
struct CertMath {
    static constexpr bool kIdentR = IDENT_R;
    static constexpr bool kRowConst = false;
    static constexpr int kCvtMode = NO_OVERFLOW ? 2 : 1;
    static __device__ __forceinline__ void d2c(const StreamParams& P, float X, float Y, float Z,
                                               float& P0, float& P1, float& P2)
    {
        if (IDENT_R) {
            P0 = __fadd_rn(X, P.t[0]);
            P1 = __fadd_rn(Y, P.t[1]);
            P2 = __fadd_rn(Z, P.t[2]);
        } else {
            IeeeMath::d2c(P, X, Y, Z, P0, P1, P2);
        }
    }
    static __device__ __forceinline__ void div2(float a0, float a1, float b, float& q0, float& q1)
    {
        float y = __builtin_amdgcn_rcpf(b);
        const float e = __fmaf_rn(-b, y, 1.0f);
        y = __fmaf_rn(e, y, y);
        float q = __fmul_rn(a0, y);
        float r = __fmaf_rn(-b, q, a0);
        q = __fmaf_rn(r, y, q);
        r = __fmaf_rn(-b, q, a0);
        q0 = __fmaf_rn(r, y, q);
        q = __fmul_rn(a1, y);
        r = __fmaf_rn(-b, q, a1);
        q = __fmaf_rn(r, y, q);
        r = __fmaf_rn(-b, q, a1);
        q1 = __fmaf_rn(r, y, q);
    }
    static __device__ __forceinline__ float div_const(float a, float c, float rc)
    {
        const float q0 = __fmul_rn(a, rc);
        const float r = __fmaf_rn(-c, q0, a);
        return __fmaf_rn(r, rc, q0);
    }
};

using CertNoOvf = CertMath<false, true>;
using CertIdentNoOvf = CertMath<true, true>;

// CertRowConst: CertMath<IDENT_R> for streams whose COLOUR ROW does not depend on the depth value (StreamParams::ident_r == 2). With
// R = I and t_y = t_z = 0 a pixel's colour row is trunc(fma(((z * my) / z * fy + ppy) / H, H, 0.5)) clamped — mathematically a function of
// its raster row alone, in floats almost one: the rounding of (z * my) / z moves py by ~1e-4 of a pixel, which changes the integer only for a
// row whose py lies that close to k - 0.5. Whether any row of a stream does is not argued but SWEPT when the context is created
// (pcs_certify_color_row_kernel: every row x every Z16 value 1 .. 65 535 through the IEEE chain); where none does, the row index of each
// raster row is a table (behind the my LUT) and the second quotient, its projection, texture coordinate, scale, clamp and conversion —
// 13 of the ~70 VALU instructions of a pixel — are one load per lane and one select per pixel. Used by the voxel reader (VALU-bound).
struct CertRowConst : CertMath<true, false> {
    [[maybe_unused]] static constexpr bool kRowConst = true;
};

// a2 colour lookup (src/pcs-camera-optimized.cpp:431-452, 584-585): texcoord -> byte index of the pixel.
__device__ __forceinline__ void color_coords(const StreamParams& P, float u, float v, float& xf, float& yf)
{
    xf = __fmaf_rn(u, P.c_w_f, 0.5f);
    yf = __fmaf_rn(v, P.c_h_f, 0.5f);
}

// Returns R | G<<8 | B<<16 in the low 24 bits (the top byte is whatever followed in memory): shorts 3 and 4
// of the record are its low half and byte 2.
template <class Cvt>
__device__ __forceinline__ uint32_t color_fetch(const StreamParams& P, const uint8_t* __restrict__ color,
                                                int32_t xi, int32_t yi, Cvt& cv)
{
    // xi < 2^24, bpp small, yi < 2^24, stride < 2^24: 24-bit multiplies are exact in 32 bits and full rate
    const uint32_t idx = __umul24((uint32_t)xi, (uint32_t)P.bpp) + __umul24((uint32_t)yi, (uint32_t)P.stride);
    uint32_t shift;
    const uint32_t off = cv.window(idx, P.color_bytes - 4u, shift);
    uint32_t w;
    __builtin_memcpy(&w, color + off, 4);
    return w >> shift;                   // R | G<<8 | B<<16 | (don't care)<<24
}

// a2 rigid transform + scale (src/pcs-camera-optimized.cpp:455-491).
// Order matters: x*col0 + t first, then + y*col1, then + z*col2; then a separately rounded * 1000.0f.
__device__ __forceinline__ float world_mm(const float* __restrict__ Mr, float X, float Y, float Z)
{
    float a = __fmaf_rn(X, Mr[0], Mr[3]);
    a = __fmaf_rn(Y, Mr[1], a);
    a = __fmaf_rn(Z, Mr[2], a);
    return __fmul_rn(a, 1000.0f);
}

// v_perm_b32: every result byte picks one of the 8 bytes of {hi, lo} (lo = bytes 0-3, hi = bytes 4-7).
// Two selectors cover all the 16-bit shuffles of the record packing in ONE instruction each, with no
// masks or shifts around them:
//   kLoLo: lo.lo16 | hi.lo16 << 16          kHiLo: lo.hi16 | hi.lo16 << 16
constexpr uint32_t kLoLo = 0x05040100u, kHiLo = 0x05040302u;
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

struct Record {              // one 10-byte point as three pieces
    uint32_t xy;             // x | y << 16
    uint32_t zc;             // z | (R | G<<8) << 16
    uint32_t b;              // B            (low 16 bits valid)
};

// Park one record at a 2-byte aligned LDS byte offset using ALIGNED accesses only. (gfx950 does take an
// unaligned ds_write_b64, but it stalls the LDS pipe: SQ_LDS_UNALIGNED_STALL 5.2 M per launch and 3.5 us on
// the 8x720p emit.) The 10 bytes are cut by the parity of the offset's dword phase:
//   offset % 4 == 0 :  [x y]@+0 (b32)  [z c0]@+4 (b32)  [c1]@+8 (b16)
//   offset % 4 == 2 :  [x]@+0 (b16)    [y z]@+2 (b32)   [c0 c1]@+6 (b32)
// expressed as one b16 and two b32 writes whose addresses and values are selected, not branched.
__device__ __forceinline__ void stage_record(uint8_t* lds, uint32_t off, const Record& r)
{
    const bool odd = (off & 2u) != 0u;
    const uint32_t yz = perm(r.zc, r.xy, kHiLo);
    const uint32_t cc = perm(r.b, r.zc, kHiLo);
    const uint32_t h_val = odd ? r.xy : r.b;                 // low 16 bits are what is written
    const uint32_t a_val = odd ? yz : r.xy;
    const uint32_t b_val = odd ? cc : r.zc;
    const uint32_t h_off = odd ? off : off + 8u;
    const uint32_t a_off = odd ? off + 2u : off;
    *reinterpret_cast<uint16_t*>(lds + h_off) = (uint16_t)h_val;
    *reinterpret_cast<uint32_t*>(lds + a_off) = a_val;
    *reinterpret_cast<uint32_t*>(lds + a_off + 4u) = b_val;
}

// One point -> one record. short(float) (:581-583) keeps the low 16 bits of the converted value.
template <class Cvt>
__device__ __forceinline__ Record make_record(const StreamParams& P, const uint8_t* __restrict__ color,
                                              const PointIn& p, Cvt& cv)
{
    const float ax = world_mm(P.M + 0, p.X, p.Y, p.Z);
    const float ay = world_mm(P.M + 4, p.X, p.Y, p.Z);
    const float az = world_mm(P.M + 8, p.X, p.Y, p.Z);
    float xf, yf;
    color_coords(P, p.u, p.v, xf, yf);
    cv.note(ax, ay, az, xf, yf);
    const uint32_t x = (uint32_t)cv.cvt(ax), y = (uint32_t)cv.cvt(ay), z = (uint32_t)cv.cvt(az);
    const uint32_t w = color_fetch(P, color, cv.pixel(xf, P.cW - 1, P.c_wm1_f), cv.pixel(yf, P.cH - 1, P.c_hm1_f), cv);
    Record r;
    r.xy = perm(y, x, kLoLo);                    // short(x) | short(y) << 16  — the low 16 bits of each (:581-583)
    r.zc = perm(w, z, kLoLo);                    // short(z) | (R | G<<8) << 16
    r.b  = __builtin_amdgcn_ubfe(w, 16, 8);      // B, high byte 0 (:585)
    return r;
}

// Brown-Conrady terms shared by deprojection (inverse model) and projection (modified model);
// evaluation order as in librealsense's rsutil.h, each op rounded.
__device__ __forceinline__ float bc_radial(const float* k, float r2)
{
    // 1 + k0*r2 + k1*r2*r2 + k4*r2*r2*r2, left to right
    float f = __fadd_rn(1.0f, __fmul_rn(k[0], r2));
    f = __fadd_rn(f, __fmul_rn(__fmul_rn(k[1], r2), r2));
    f = __fadd_rn(f, __fmul_rn(__fmul_rn(__fmul_rn(k[4], r2), r2), r2));
    return f;
}
// a + 2*kA*x*y + kB*(r2 + 2*a_axis*a_axis)
__device__ __forceinline__ float bc_tangential(float a, float kA, float kB, float x, float y, float r2, float axis)
{
    float s = __fadd_rn(a, __fmul_rn(__fmul_rn(__fmul_rn(2.0f, kA), x), y));
    return __fadd_rn(s, __fmul_rn(kB, __fadd_rn(r2, __fmul_rn(__fmul_rn(2.0f, axis), axis))));
}

// a5 for one pixel: depth value d, normalised ray (mx,my) from the LUTs.
template <bool DDIST, bool CDIST, class Mth>
__device__ __forceinline__ PointIn deproject_pixel(const StreamParams& P, uint32_t d, float mx, float my)
{
    const float z = __fmul_rn(P.depth_scale, (float)d);
    if (DDIST && P.ddist) {   // template gate compiles it in; the per-stream flag is wave-uniform
        const float r2 = __fadd_rn(__fmul_rn(mx, mx), __fmul_rn(my, my));
        const float f = bc_radial(P.dk, r2);
        const float ux = bc_tangential(__fmul_rn(mx, f), P.dk[2], P.dk[3], mx, my, r2, mx);
        const float uy = bc_tangential(__fmul_rn(my, f), P.dk[3], P.dk[2], mx, my, r2, my);
        mx = ux; my = uy;
    }
    PointIn p;
    p.X = __fmul_rn(z, mx);
    p.Y = __fmul_rn(z, my);
    p.Z = z;
    float P0, P1, P2;
    Mth::d2c(P, p.X, p.Y, p.Z, P0, P1, P2);
    // rs2_project_point_to_pixel
    float x, y;
    Mth::div2(P0, P1, P2, x, y);
    if (CDIST && P.cdist) {
        const float r2 = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
        const float f = bc_radial(P.ck, r2);
        x = __fmul_rn(x, f); y = __fmul_rn(y, f);
        const float dx = bc_tangential(x, P.ck[2], P.ck[3], x, y, r2, x);
        const float dy = bc_tangential(y, P.ck[3], P.ck[2], x, y, r2, y);
        x = dx; y = dy;
    }
    float px = __fadd_rn(__fmul_rn(x, P.c_fx), P.c_ppx);
    float py = __fadd_rn(__fmul_rn(y, P.c_fy), P.c_ppy);
    if (CDIST && P.tex_half) {      // older librealsense pixel_to_texcoord: (pixel + 0.5) / size. Rides on the CDIST
        px = __fadd_rn(px, 0.5f);   // instantiation (the host routes such streams there) so the common path pays nothing.
        py = __fadd_rn(py, 0.5f);
    }
    // pixel_to_texcoord; invalid depth (z == 0) -> texcoord (0,0). The quotients are computed
    // unconditionally and then selected: a conditional here becomes a divergent branch per pixel, which
    // stops the scheduler from interleaving the 8 pixels of a lane.
    const float qu = Mth::div_const(px, P.c_w_f, P.c_rw);
    const float qv = Mth::div_const(py, P.c_h_f, P.c_rh);
    const bool valid = (z != 0.0f);
    p.u = valid ? qu : 0.0f;
    p.v = valid ? qv : 0.0f;
    return p;
}

// The same pixel under CertRowConst (R = I, t_y = t_z = 0, no distortion; see the policy): X, Y, Z and u as above, and in place of v the
// pixel's colour ROW itself — `crow`, the table's entry for the raster row, 0 for an invalid pixel — carried in p.v as an INTEGER. A
// function of its own, so that the instantiations every other kernel uses keep exactly the code they had (an if inside deproject_pixel
// cost the distortion instantiation of the dense kernel 800 instructions: the two copies of its tail no longer merged).
template <class Mth>
__device__ __forceinline__ PointIn deproject_pixel_rowc(const StreamParams& P, uint32_t d, float mx, float my, int crow)
{
    const float z = __fmul_rn(P.depth_scale, (float)d);
    PointIn p;
    p.X = __fmul_rn(z, mx);
    p.Y = __fmul_rn(z, my);
    p.Z = z;
    const float P0 = __fadd_rn(p.X, P.t[0]);
    const float P2 = __fadd_rn(p.Z, P.t[2]);
    float x, y_unused;
    Mth::div2(P0, P0, P2, x, y_unused);                  // (the second quotient is dead code)
    const float px = __fadd_rn(__fmul_rn(x, P.c_fx), P.c_ppx);
    const float qu = Mth::div_const(px, P.c_w_f, P.c_rw);
    const bool valid = (z != 0.0f);
    p.u = valid ? qu : 0.0f;
    p.v = __int_as_float(valid ? crow : 0);
    return p;
}

// -c predicate on camera-frame z and x (src/pcs-camera-optimized.cpp:398-401, 504-511).
__device__ __forceinline__ bool in_range(float X, float Z)
{
    return Z > 0.0f && Z <= 1.5f && X > -2.0f && X <= 2.0f;
}

// Keep mask for a lane's 8 consecutive points (point index i0 + k, i0 % 8 == 0).
// Keep mask from the per-point predicate bits: rng = -c range test, nz = depth valid (bit k = point i0 + k).
__device__ __forceinline__ uint32_t keep_from_bits(uint32_t rng, uint32_t nz, uint32_t i0, uint32_t n, uint32_t flags)
{
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) live |= (uint32_t)(i0 + k < n) << k;
    uint32_t keep = live;
    if (flags & PCS_FLAG_CUTOFF) {
        uint32_t gate = rng;
        if (flags & PCS_FLAG_CUTOFF_COMPAT) {
            // the reference gates point k of each aligned group of four with point 3-k's test
            // (lane-reversed mask, :501-502 vs :519); groups that run past n use their own test.
            uint32_t rev = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) rev |= ((rng >> ((k & 4) | (3 - (k & 3)))) & 1u) << k;
            const uint32_t full_lo = (i0 + 3 < n) ? 0x0Fu : 0u;
            const uint32_t full_hi = (i0 + 7 < n) ? 0xF0u : 0u;
            const uint32_t full = full_lo | full_hi;
            gate = (rev & full) | (rng & ~full);
        }
        keep &= gate;
    }
    if (flags & PCS_FLAG_DROP_INVALID) keep &= nz;
    return keep;
}

// Keep mask for a lane's 8 consecutive points (point index i0 + k, i0 % 8 == 0).
__device__ __forceinline__ uint32_t keep_mask8(const PointIn (&p)[8], uint32_t i0, uint32_t n, uint32_t flags)
{
    uint32_t rng = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        rng |= (uint32_t)in_range(p[k].X, p[k].Z) << k;
        nz  |= (uint32_t)(p[k].Z != 0.0f) << k;
    }
    return keep_from_bits(rng, nz, i0, n, flags);
}

// Wavefront-wide inclusive prefix sum (64 lanes, all active) by DPP: four row_shr steps scan each row of 16 lanes,
// row_bcast:15 / row_bcast:31 carry the row totals across (the gfx9 sequence). No lane-index registers, no LDS
// crossbar (ds_bpermute, which __shfl_up compiles to) — and nothing loop-invariant for the compiler to hoist out of
// a persistent tile loop and spill.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Wavefront-wide exclusive prefix sum of a small per-lane count; wave_total is wave-uniform (an SGPR).
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t c, uint32_t& wave_total)
{
    const uint32_t inc = wave_inclusive_scan(c);
    wave_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    return inc - c;
}

// Wavefront-wide sum (wave-uniform).
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

// ------------------------------------------------------------------------------------------------
// Point sources. A source hands each lane its 8 consecutive points of the tile.
// ------------------------------------------------------------------------------------------------

// Z16 raster + LUTs -> points (the fused a5 stage).
template <bool DDIST, bool CDIST, class Mth = IeeeMath>
struct DepthSource {
    using Math = Mth;
    const uint16_t* __restrict__ depth;

    // The distortion decision is taken ONCE per lane, outside the pixel loop: a (wave-uniform) test per
    // pixel splits the lane's code into 16 basic blocks, which stops the scheduler from interleaving the
    // pixels and the compiler from packing pairs of them into v_pk_* instructions (measured: the emit
    // kernel ran 27 us with per-pixel tests vs 19 us for the dense kernel without them).
    // The fast path in two steps, for kernels that want to do something between requesting a lane's inputs and using them
    // (the single-pass compaction counts and publishes from the raw Z16 words first): fast() says whether it applies
    // (uniform over the launch's stream), fetch() issues the loads, deproject() consumes them.
    struct Raw { uint4 dv; f32x4 ma, mb; float my; };
    __device__ __forceinline__ bool fast(const StreamParams& P) const { return (P.W & 7) == 0 && ((uintptr_t)depth & 15) == 0; }
    __device__ __forceinline__ Raw fetch(const StreamParams& P, uint32_t i0) const
    {
        // all 8 pixels on one raster row; one 16-byte depth load, two 16-byte LUT loads
        // floor(i0 / W) by the host-verified multiply-shift (i0 < 2^31)
        const uint32_t r = P.w_magic ? (__umulhi(i0, P.w_magic) >> P.w_shift) : i0 / (uint32_t)P.W;
        const uint32_t c0 = i0 - r * (uint32_t)P.W;
        Raw q;
        q.dv = *reinterpret_cast<const uint4*>(depth + i0);
        const gptr<float> lut_x = as_global(P.mx);
        q.ma = *reinterpret_cast<gptr<f32x4>>(lut_x + c0);
        q.mb = *reinterpret_cast<gptr<f32x4>>(lut_x + c0 + 4);
        q.my = as_global(P.my)[r];
        return q;
    }
    template <bool DD, bool CD>
    __device__ __forceinline__ void deproject(const StreamParams& P, const Raw& q, PointIn (&p)[8]) const
    {
        const uint32_t dw[4] = {q.dv.x, q.dv.y, q.dv.z, q.dv.w};
        const float mxs[8] = {q.ma.x, q.ma.y, q.ma.z, q.ma.w, q.mb.x, q.mb.y, q.mb.z, q.mb.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
            p[k] = deproject_pixel<DD, CD, Mth>(P, d, mxs[k], q.my);
        }
    }

    template <bool DD, bool CD>
    __device__ __forceinline__ void load8_impl(const StreamParams& P, uint32_t i0, uint32_t n, PointIn (&p)[8]) const
    {
        if (fast(P)) {
            deproject<DD, CD>(P, fetch(P, i0), p);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t i = min(i0 + k, n - 1);
                const uint32_t r = i / (uint32_t)P.W;
                const uint32_t c = i - r * (uint32_t)P.W;
                p[k] = deproject_pixel<DD, CD, Mth>(P, depth[i], as_global(P.mx)[c], as_global(P.my)[r]);
            }
        }
    }

    // CertRowConst: the colour row of raster row r lies behind the H floats of the my LUT (StreamParams::my)
    __device__ __forceinline__ void load8_rowc(const StreamParams& P, uint32_t i0, uint32_t n, PointIn (&p)[8]) const
    {
        const gptr<float> lut_y = as_global(P.my);
        if (fast(P)) {
            const Raw q = fetch(P, i0);
            const uint32_t r = P.w_magic ? (__umulhi(i0, P.w_magic) >> P.w_shift) : i0 / (uint32_t)P.W;
            const int crow = __float_as_int(lut_y[(uint32_t)P.H + r]);
            const uint32_t dw[4] = {q.dv.x, q.dv.y, q.dv.z, q.dv.w};
            const float mxs[8] = {q.ma.x, q.ma.y, q.ma.z, q.ma.w, q.mb.x, q.mb.y, q.mb.z, q.mb.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
                p[k] = deproject_pixel_rowc<Mth>(P, d, mxs[k], q.my, crow);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t i = min(i0 + k, n - 1);
                const uint32_t r = i / (uint32_t)P.W;
                const uint32_t c = i - r * (uint32_t)P.W;
                p[k] = deproject_pixel_rowc<Mth>(P, depth[i], as_global(P.mx)[c], lut_y[r], __float_as_int(lut_y[(uint32_t)P.H + r]));
            }
        }
    }

    // The same with the lane's eight Z16 values already in registers (requested a round earlier: fast(P) rasters only).
    __device__ __forceinline__ Raw fetch_luts(const StreamParams& P, uint32_t i0, const uint4& dv) const
    {
        const uint32_t r = P.w_magic ? (__umulhi(i0, P.w_magic) >> P.w_shift) : i0 / (uint32_t)P.W;
        const uint32_t c0 = i0 - r * (uint32_t)P.W;
        Raw q;
        q.dv = dv;
        const gptr<float> lut_x = as_global(P.mx);
        q.ma = *reinterpret_cast<gptr<f32x4>>(lut_x + c0);
        q.mb = *reinterpret_cast<gptr<f32x4>>(lut_x + c0 + 4);
        q.my = as_global(P.my)[r];
        return q;
    }
    __device__ __forceinline__ void load8_pre(const StreamParams& P, uint32_t i0, uint32_t n, const uint4& dv, PointIn (&p)[8]) const
    {
        if (i0 >= n) {
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = PointIn{0, 0, 0, 0, 0};
            return;
        }
        const Raw q = fetch_luts(P, i0, dv);
        if constexpr (Mth::kRowConst) {
            const uint32_t r = P.w_magic ? (__umulhi(i0, P.w_magic) >> P.w_shift) : i0 / (uint32_t)P.W;
            const int crow = __float_as_int(as_global(P.my)[(uint32_t)P.H + r]);
            const uint32_t dw[4] = {q.dv.x, q.dv.y, q.dv.z, q.dv.w};
            const float mxs[8] = {q.ma.x, q.ma.y, q.ma.z, q.ma.w, q.mb.x, q.mb.y, q.mb.z, q.mb.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
                p[k] = deproject_pixel_rowc<Mth>(P, d, mxs[k], q.my, crow);
            }
        } else {
            if ((DDIST || CDIST) && (P.ddist | P.cdist | P.tex_half)) deproject<DDIST, CDIST>(P, q, p);
            else deproject<false, false>(P, q, p);
        }
    }

    __device__ __forceinline__ void load8(const StreamParams& P, uint32_t i0, uint32_t n, PointIn (&p)[8]) const
    {
        if (i0 >= n) {
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = PointIn{0, 0, 0, 0, 0};
            return;
        }
        if constexpr (Mth::kRowConst) {
            load8_rowc(P, i0, n, p);
        } else {
            if ((DDIST || CDIST) && (P.ddist | P.cdist | P.tex_half)) load8_impl<DDIST, CDIST>(P, i0, n, p);
            else load8_impl<false, false>(P, i0, n, p);
        }
    }
};

// rs2::points arrays (vertices + texcoords) -> points (the a2 twin's input), read straight into registers: a lane's 8 points
// are 96 contiguous bytes of vertices and 64 of texcoords (six + four 16-byte loads, all in flight together). A wavefront's
// k-th load touches 64 separate 16-byte pieces, but its six vertex loads cover the same 48 cache lines back to back, so the
// L1 does the merging. (Rounds 1-2 transposed the arrays through 40 KB of LDS with lane-contiguous loads — textbook
// coalescing, but the LDS round trip, its barrier and the fat workgroups cost far more than the L1 does: one 1280x720 cloud
// per launch 11.0 -> 7.1 us = 34 -> 53 % of HBM peak, eight clouds in one launch 61 -> 71 %.)
struct VertexSource {
    using Math = IeeeMath;
    const float* __restrict__ vertices;
    const float* __restrict__ texcoords;

    __device__ __forceinline__ void load8(const StreamParams&, uint32_t i0, uint32_t n, PointIn (&p)[8]) const
    {
        if (i0 + 8u <= n && ((((uintptr_t)vertices) | ((uintptr_t)texcoords)) & 15u) == 0u) {
            const float4* gv = reinterpret_cast<const float4*>(vertices + (size_t)i0 * 3);
            const float4* gt = reinterpret_cast<const float4*>(texcoords + (size_t)i0 * 2);
            float4 v[6], t[4];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = gv[k];
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = gt[k];
            const float* fv = reinterpret_cast<const float*>(v);
            const float* ft = reinterpret_cast<const float*>(t);
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = PointIn{fv[3 * k], fv[3 * k + 1], fv[3 * k + 2], ft[2 * k], ft[2 * k + 1]};
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const size_t i = (size_t)i0 + k;
                if (i < n) p[k] = PointIn{vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2], texcoords[2 * i], texcoords[2 * i + 1]};
                else p[k] = PointIn{0, 0, 0, 0, 0};
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Staged store: LDS bytes [head, head+nbytes) -> global bytes [g, g+nbytes), where (g - head) is
// 16-byte aligned. Interior goes out as lane-contiguous 16-byte stores; the ragged ends (which may
// share a 16-byte line with a neighbouring tile's bytes) go out as 2-byte stores.
// ------------------------------------------------------------------------------------------------
template <uint32_t THREADS = kBlockThreads>
__device__ __forceinline__ void store_staged(const uint8_t* lds, uint32_t head, uint32_t nbytes, uint8_t* g)
{
    uint8_t* g0 = g - head;                                  // 16-byte aligned
    const uint32_t end = head + nbytes;
    const uint32_t first_full = (head + 15u) >> 4;           // first chunk entirely inside
    const uint32_t last_full = end >> 4;                     // one past the last chunk entirely inside
    // nontemporal: the payload is written once and never re-read by this kernel; keeping it out of the
    // caches' way measured +7 % on the store-dominated stream (tools/lab/kernel_lab.hip, skeleton nt-store)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t j = first_full + threadIdx.x; j < last_full; j += THREADS)
        __builtin_nontemporal_store(reinterpret_cast<const u32x4*>(lds)[j], reinterpret_cast<u32x4*>(g0) + j);
    // ragged head: shorts in [head, min(first_full*16, end)); ragged tail: [max(last_full*16, head), end)
    const uint32_t head_end = min(first_full << 4, end);
    for (uint32_t b = head + 2u * threadIdx.x; b < head_end; b += 2u * THREADS)
        *reinterpret_cast<uint16_t*>(g0 + b) = *reinterpret_cast<const uint16_t*>(lds + b);
    if (last_full >= first_full) {
        const uint32_t tail_begin = max(last_full << 4, head_end);
        for (uint32_t b = tail_begin + 2u * threadIdx.x; b < end; b += 2u * THREADS)
            *reinterpret_cast<uint16_t*>(g0 + b) = *reinterpret_cast<const uint16_t*>(lds + b);
    }
}

// ------------------------------------------------------------------------------------------------
// DENSE tile kernel: no predicate, downsample 1, 16-byte aligned payload, every stream's point count
// a multiple of 8 (so every lane's 80-byte output run starts on a 16-byte boundary).
// Lane l of the workgroup produces 8 records = 5 x uint4 and parks them at LDS[l*80]; the workgroup
// then streams the tile's 20 480 bytes out with lane-contiguous 16-byte stores.
// ------------------------------------------------------------------------------------------------
// THREADS: lanes per workgroup = 8-point runs per tile (kBlockThreads: 2048-point tiles; 64: one wavefront per 512-point tile, what
// a launch that cannot fill the chip with 256-lane workgroups takes — launch_fused_dense).
template <class Src, uint32_t THREADS = kBlockThreads>
__device__ __forceinline__ void dense_tile(const StreamParams& P, const Src& src, const uint8_t* __restrict__ color,
                                           uint32_t tile0, uint32_t n, uint8_t* __restrict__ out_bytes,
                                           uint4* stage)
{
    const uint32_t i0 = tile0 + threadIdx.x * kPointsPerLane;
    PointIn p[8];
    src.load8(P, i0, n, p);

    uint32_t w[20];
    auto fill = [&](auto& cv) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const Record a = make_record(P, color, p[k], cv);
            const Record b = make_record(P, color, p[k + 1], cv);
            uint32_t* o = w + (k >> 1) * 5;
            o[0] = a.xy;
            o[1] = a.zc;
            o[2] = perm(b.xy, a.b, kLoLo);
            o[3] = perm(b.zc, b.xy, kHiLo);
            o[4] = perm(b.b, b.zc, kHiLo);
        }
    };
    if (Src::Math::kCvtMode == 2) {
        FastCvt<false> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else if (Src::Math::kCvtMode == 1) {
        FastCvt<true> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else {
        ExactCvt exact;
        fill(exact);
    }
    uint4* mine = stage + threadIdx.x * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    __syncthreads();

    const uint32_t pts = min(THREADS * kPointsPerLane, n - tile0);
    store_staged<THREADS>(reinterpret_cast<const uint8_t*>(stage), 0u, pts * PCS_POINT_BYTES,
                          out_bytes + (size_t)tile0 * PCS_POINT_BYTES);
}

// ------------------------------------------------------------------------------------------------
// GENERIC tile: predicate and/or downsample and/or unaligned payload. Order-preserving.
//   g0        kept-index (within the stream) of the tile's first kept point
//   out_first output point index (within the whole payload) of kept-index 0 of this stream
// A kept point with kept-index g is written iff g % ds == 0, to output point out_first + g / ds.
// ------------------------------------------------------------------------------------------------
template <class Src, bool PRED, bool DS1>
__device__ __forceinline__ void generic_tile(const StreamParams& P, const Src& src, const uint8_t* __restrict__ color,
                                             uint32_t tile0, uint32_t n, uint32_t flags, uint32_t ds,
                                             uint32_t g0, uint32_t out_first, uint8_t* __restrict__ payload_bytes,
                                             uint8_t* stage, uint32_t* wsum)
{
    if (DS1) ds = 1u;
    const uint32_t i0 = tile0 + threadIdx.x * kPointsPerLane;
    PointIn p[8];
    src.load8(P, i0, n, p);

    uint32_t keep, ex = 0;
    if (PRED) {
        keep = keep_mask8(p, i0, n, flags);
        uint32_t wave_total;
        ex = wave_exclusive_scan(__popc(keep), wave_total);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = wave_total;
    } else {
        const uint32_t mine = (i0 < n) ? min(8u, n - i0) : 0u;
        keep = (1u << mine) - 1u;
    }

    // Records for all 8 points, straight-line (a branch per point would serialise the lane's pixels and put
    // the colour gather behind it); the predicate and the stride only gate the LDS staging below. They are
    // computed BEFORE the barrier of the cross-wave scan so that the colour gathers are in flight while the
    // workgroup waits for its slowest wavefront.
    Record rec[8];
    auto fill = [&](auto& cv) {
#pragma unroll
        for (int k = 0; k < 8; k++) rec[k] = make_record(P, color, p[k], cv);
    };
    if (Src::Math::kCvtMode == 2) {
        FastCvt<false> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else if (Src::Math::kCvtMode == 1) {
        FastCvt<true> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else {
        ExactCvt exact;
        fill(exact);
    }

    uint32_t lane_first, tile_kept;
    if (PRED) {
        __syncthreads();
        const int wave = threadIdx.x >> 6;
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wsum[w];
        tile_kept = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        lane_first = before + ex;
    } else {
        const uint32_t pts = min(kTilePoints, n - tile0);
        lane_first = min(threadIdx.x * kPointsPerLane, pts);
        tile_kept = pts;
    }

    // output range of the tile, in points: q = out_first + ceil(g/ds) for g in [g0, g0 + tile_kept)
    const uint32_t q_lo = out_first + (DS1 ? g0 : (g0 + ds - 1) / ds);
    const uint32_t q_hi = out_first + (DS1 ? g0 + tile_kept : (g0 + tile_kept + ds - 1) / ds);
    uint8_t* gdst = payload_bytes + (size_t)q_lo * PCS_POINT_BYTES;
    const uint32_t head = (uint32_t)((uintptr_t)gdst & 15u);

    // kept-index of the lane's first kept point, and (for a stride) its quotient / remainder once per lane
    uint32_t g = g0 + lane_first;
    uint32_t gq = DS1 ? g : g / ds, gr = DS1 ? 0u : g - gq * ds;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((keep >> k) & 1u) {
            if (gr == 0u) {
                stage_record(stage, head + (out_first + gq - q_lo) * PCS_POINT_BYTES, rec[k]);
            }
            if (DS1) gq++;
            else if (++gr == ds) { gr = 0u; gq++; }
        }
    }
    __syncthreads();
    store_staged(stage, head, (q_hi - q_lo) * PCS_POINT_BYTES, gdst);
}

// ------------------------------------------------------------------------------------------------
// SINGLE-PASS ordered compaction (predicate active, stride 1): one launch instead of count + scan + emit,
// and the Z16 raster is read once.
//
//  * Tile ids are handed out by an atomic ticket in the order workgroups START, never by blockIdx: a tile
//    only ever waits for lower tickets, whose workgroups are already running -> no dependence on the
//    (unspecified) dispatch order, no deadlock.
//  * Every tile publishes ONE 64-bit descriptor {[63:34] launch generation, [33:32] ready, [31:0] kept
//    count} as soon as it has counted — flag and payload travel in one naturally aligned agent-scope 8-byte
//    store, stale descriptors of earlier launches read as "not ready", nothing is ever cleared.
//  * Placement is a DIRECT SUM, not a chained scan: a tile adds up the counts of its own stream's earlier
//    tiles (all 256 lanes stride over <= a few hundred descriptors) plus one total per earlier stream
//    (published by that stream's last tile). Nobody waits on anything but first-level publications, which
//    happen within the first microseconds of a workgroup's life; a chained look-back, which this replaces,
//    crawled ~64 tiles per microsecond through ~1500 resident tiles (65-70 us per frame-set).
//  * Every wait is bounded; on expiry the error word is set and the host re-runs the frame with the
//    three-pass path.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kDescReady = 1u;
constexpr uint32_t kSpinLimit = 1u << 18;

__device__ __forceinline__ uint64_t desc_pack(uint32_t gen, uint32_t status, uint32_t value)
{
    return ((uint64_t)((gen << 2) | status) << 32) | (uint64_t)value;
}

// Value of a descriptor once its writer has published it for this launch generation (bounded wait).
__device__ __forceinline__ uint32_t desc_wait(const uint64_t* __restrict__ d, uint32_t gen, uint32_t* __restrict__ error)
{
    for (uint32_t spins = 0;; spins++) {
        const uint64_t v = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t hi = (uint32_t)(v >> 32);
        if ((hi >> 2) == gen && (hi & 3u) == kDescReady) return (uint32_t)v;
        if (spins > kSpinLimit) { atomicExch(error, 1u); return 0u; }
        __builtin_amdgcn_s_sleep(2);
    }
}

struct CompactArgs {
    unsigned long long* ticket;       // never reset; ticket_base = its value when this launch was enqueued
    unsigned long long  ticket_base;
    uint64_t*           desc;         // one descriptor per tile of this launch: the tile's kept count
    uint64_t*           stream_desc;  // [stream] kept count of the whole stream (published by its last tile)
    uint32_t*           stream_end;   // [stream] inclusive global prefix at the stream's last tile (plain, for counts)
    const uint32_t*     chain_in;     // output points written by earlier launches of this frame-set (or null)
    uint32_t*           error;
    uint32_t            gen;
    uint32_t            flags;
    int32_t*            counts;       // [n_total + 1] per-stream kept counts + total, written by each launch's last tile
    int32_t             n_total;      // streams of the whole frame-set
    int32_t             last_launch;  // this launch holds the frame-set's last stream
};

#if !PCS_TU_VOXEL   // ---- kernels of the main translation unit (see the note at the top of the file) ----
template <class Mth>
__global__ __launch_bounds__(kBlockThreads)
void pcs_fused_compact_kernel(const StreamParams* __restrict__ params, int stream0, int n_launch, FramePtrs fp,
                              CompactArgs a, uint8_t* __restrict__ payload_bytes)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t psum[8];
    __shared__ uint32_t bcast[1];

    uint32_t gtile;
    if (a.ticket) {         // tile ids in workgroup START order (dispatch-order independent, but 3600 returning
        if (threadIdx.x == 0) bcast[0] = (uint32_t)(atomicAdd(a.ticket, 1ull) - a.ticket_base);   // atomics on one line)
        __syncthreads();
        gtile = bcast[0];
    } else {
        gtile = blockIdx.x; // relies on in-order dispatch for progress; the bounded waits catch anything else
    }

    // ticket -> (stream, tile): tiles are numbered stream-major
    int s = 0;
    uint32_t t = gtile, tiles_s = 0;
    for (;; s++) {
        tiles_s = (params[stream0 + s].n_points + kTilePoints - 1) / kTilePoints;
        if (t < tiles_s || s == n_launch - 1) break;
        t -= tiles_s;
    }
    const StreamParams& P = params[stream0 + s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = t * kTilePoints;
    const uint32_t i0 = tile0 + threadIdx.x * kPointsPerLane;
    const uint8_t* __restrict__ color = fp.color[s];

    DepthSource<true, true, Mth> src{fp.depth[s]};
    PointIn p[8];
    src.load8(P, i0, n, p);
    const uint32_t keep = keep_mask8(p, i0, n, a.flags);

    uint32_t wave_total;
    const uint32_t ex = wave_exclusive_scan(__popc(keep), wave_total);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum[wave] = wave_total;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += wsum[w];
    const uint32_t tile_kept = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const uint32_t lane_first = before + ex;

    // publish this tile's kept count as early as possible: later tiles only ever wait for this store
    if (threadIdx.x == 0)
        __hip_atomic_store(a.desc + gtile, desc_pack(a.gen, kDescReady, tile_kept), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);

    // records for all 8 points, straight-line (the predicate only gates the staging below); the tiles in
    // front of this one publish their counts meanwhile
    Record rec[8];
    auto fill = [&](auto& cv) {
#pragma unroll
        for (int k = 0; k < 8; k++) rec[k] = make_record(P, color, p[k], cv);
    };
    if (Mth::kCvtMode == 2) {
        FastCvt<false> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else if (Mth::kCvtMode == 1) {
        FastCvt<true> fast;
        fill(fast);
        if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
    } else {
        ExactCvt exact;
        fill(exact);
    }

    // placement: kept points of this stream's earlier tiles (<= a few hundred descriptors, all 256 lanes
    // stride over them) + the totals of the earlier streams (published once by each stream's last tile).
    // No chain: nobody waits for anything but first-level publications. (Issuing these loads before the
    // records, to hide their round trip, measured 38-42 us instead of 27-29: most of the neighbours have not
    // published yet at that point and everything is polled twice.)
    uint32_t own = 0, tot = 0;
    const uint32_t first = gtile - t;
    for (uint32_t j = first + threadIdx.x; j < gtile; j += kBlockThreads) own += desc_wait(a.desc + j, a.gen, a.error);
    if ((int)threadIdx.x < s) tot = desc_wait(a.stream_desc + stream0 + threadIdx.x, a.gen, a.error);
    const uint32_t tot_lane = tot;     // lane e < s: kept points of stream stream0 + e
    own = wave_sum(own);
    tot = wave_sum(tot);
    if (lane == 0) { psum[wave] = own; psum[4 + wave] = tot; }
    __syncthreads();
    const uint32_t own_excl = psum[0] + psum[1] + psum[2] + psum[3];
    const uint32_t chain = a.chain_in ? *a.chain_in : 0u;
    const uint32_t q_lo = chain + psum[4] + psum[5] + psum[6] + psum[7] + own_excl;   // global output point index
    if (threadIdx.x == 0 && t == tiles_s - 1) {
        __hip_atomic_store(a.stream_desc + stream0 + s, desc_pack(a.gen, kDescReady, own_excl + tile_kept), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        a.stream_end[stream0 + s] = q_lo + tile_kept;
    }
    // The launch's very last tile has every stream total of the launch in its lanes: it hands the counts back itself
    // (a separate one-workgroup kernel for this cost 4.8 us of launch latency per frame-set).
    if (a.counts && s == n_launch - 1 && t == tiles_s - 1) {
        if ((int)threadIdx.x < s) a.counts[stream0 + threadIdx.x] = (int32_t)tot_lane;
        if (threadIdx.x == 0) {
            a.counts[stream0 + s] = (int32_t)(own_excl + tile_kept);
            if (a.last_launch) a.counts[a.n_total] = (int32_t)(q_lo + tile_kept);
        }
    }

    uint8_t* gdst = payload_bytes + (size_t)q_lo * PCS_POINT_BYTES;
    const uint32_t head = (uint32_t)((uintptr_t)gdst & 15u);
    uint32_t rank = lane_first;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((keep >> k) & 1u) {
            stage_record(stage, head + rank * PCS_POINT_BYTES, rec[k]);
            rank++;
        }
    }
    __syncthreads();
    store_staged(stage, head, tile_kept * PCS_POINT_BYTES, gdst);
}

#endif  // !PCS_TU_VOXEL

// Request a stream's constants (and this launch's raster pointers) with ONE batch of scalar loads at the top of a
// kernel. Left alone, hipcc asks for them one dependent group at a time — the kernarg, then n_points for the early
// exit, then the raster pointers and the width, then the LUT pointers — four scalar round trips before the first
// Z16 load of a workgroup can be issued, paid in full by the first wave of workgroups of every launch (1.8 rounds of
// them make up an 8 x 720p launch). The empty asm only says "these are needed HERE".
// LEAN: leave out the two quads that hold nothing but distortion coefficients (dk[1..4], ck[0..3]; a stream that has any
// fetches them when it gets there) and, for IDENT_R policies, the two quads of the depth->colour rotation that p + t never
// reads. What this buys is SGPRs at the point where the most of them are live: the hardware admits a 256-lane workgroup per
// CU only while its waves' SGPR allocation allows it — 7 per CU up to 96 SGPRs, 6 from 97 (MI355X_MICROARCH.md, residency) —
// and the emit kernel's extra arguments had pushed it to 103.
template <bool LEAN = false, bool IDENT_R = false>
__device__ __forceinline__ void request_constants(const StreamParams& P, const void* a, const void* b, const void* c = nullptr)
{
    static_assert(sizeof(StreamParams) == 19 * 16, "request_constants covers the struct in 19 quads");
    static_assert(offsetof(StreamParams, dk) == 39 * 4 && offsetof(StreamParams, ck) == 44 * 4 && offsetof(StreamParams, R) == 12 * 4,
                  "quads 10, 11 = dk[1..4], ck[0..3]; quads 3, 4 = R[0..7]");
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    const u32x4s* q = reinterpret_cast<const u32x4s*>(&P);
    if (LEAN && IDENT_R)
        asm volatile("" :: "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]), "s"(q[9]),
                           "s"(q[12]), "s"(q[13]), "s"(q[14]), "s"(q[15]), "s"(q[16]), "s"(q[17]), "s"(q[18]),
                           "s"(a), "s"(b), "s"(c));
    else if (LEAN)
        asm volatile("" :: "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]), "s"(q[9]),
                           "s"(q[12]), "s"(q[13]), "s"(q[14]), "s"(q[15]), "s"(q[16]), "s"(q[17]), "s"(q[18]),
                           "s"(a), "s"(b), "s"(c));
    else
        asm volatile("" :: "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]), "s"(q[9]),
                           "s"(q[10]), "s"(q[11]), "s"(q[12]), "s"(q[13]), "s"(q[14]), "s"(q[15]), "s"(q[16]), "s"(q[17]), "s"(q[18]),
                           "s"(a), "s"(b), "s"(c));
}

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------
#if !PCS_TU_VOXEL

template <bool DDIST, bool CDIST, class Mth, uint32_t THREADS = kBlockThreads>
__global__ __launch_bounds__(THREADS, 7)     // <= 72 VGPRs for every instantiation (one landed on 73 -> 6 waves/SIMD); A/B on one box: no measurable change, 8 spills and is slower
void pcs_fused_dense_kernel(const StreamParams* __restrict__ params, int stream0, FramePtrs fp,
                            uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[THREADS * kPointsPerLane * PCS_POINT_BYTES / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[stream0 + s];
    request_constants(P, fp.depth[s], fp.color[s], payload_bytes);
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * (THREADS * kPointsPerLane);
    if (tile0 >= n) return;
    DepthSource<DDIST, CDIST, Mth> src{fp.depth[s]};
    dense_tile<DepthSource<DDIST, CDIST, Mth>, THREADS>(P, src, fp.color[s], tile0, n, payload_bytes + (size_t)P.out_base * PCS_POINT_BYTES, stage);
}

// K frame-sets of the same streams in one launch: blockIdx.z = frame-set, blockIdx.y = stream. Same tile code, same
// bytes; only the fill/drain of the launch is shared by K sets (pcs_process_frames_device_batch).
template <bool DDIST, bool CDIST, class Mth>
__global__ __launch_bounds__(kBlockThreads, 7)
void pcs_fused_dense_batch_kernel(const StreamParams* __restrict__ params, BatchPtrs bp)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const int e = blockIdx.z * gridDim.y + s;
    request_constants(P, bp.depth[e], bp.color[e], bp.payload[blockIdx.z]);
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    DepthSource<DDIST, CDIST, Mth> src{bp.depth[e]};
    dense_tile(P, src, bp.color[e], tile0, n, bp.payload[blockIdx.z] + (size_t)P.out_base * PCS_POINT_BYTES, stage);
}

// Count pass: kept points per tile. (Folding the per-stream scan into this launch through a last-arriver
// counter was measured and is slower — 450 returning atomics per counter line cost more than the separate
// 5 us scan launch)
// A workgroup counts kCountTiles consecutive tiles, their (independent) Z16 loads issued back to back with no branch
// between them. Measured on 8 x 720p (14.7 MB of Z16, rocprofv3 average): 1 tile 5.4 us, 2 tiles 5.4 us, 4 tiles
// 5.9 us, 8 tiles 6.9 us — latency, not bandwidth; 2 halves the workgroups at no cost.
constexpr int kCountTiles = 2;
template <bool DDIST, bool CDIST>
__device__ __forceinline__ void count_tiles(const StreamParams& P, const uint16_t* __restrict__ dp, uint32_t flags,
                                            uint32_t* __restrict__ tile_counts, uint32_t (*wsum)[4])
{
    const uint32_t n = P.n_points;
    const uint32_t tile_first = blockIdx.x * kCountTiles;
    if (tile_first * kTilePoints >= n) return;
    uint32_t c[kCountTiles];
    const bool cut = (flags & PCS_FLAG_CUTOFF) != 0;
    if (P.z_zero_iff_d_zero && (!cut || P.cut_dmax != 0u)) {
        // The predicate from the Z16 words alone, no deprojection:
        //  * z = depth_scale * d is zero exactly when d is (the host checked the scale: finite, and scale*1 != 0);
        //  * -c (:504-511) keeps 0 < z <= 1.5 and -2 < x <= 2. z is monotone in d, so the z test is 1 <= d <= cut_dmax
        //    (the host found the largest d with fl(depth_scale * d) <= 1.5f), and cut_dmax is only non-zero when the host
        //    has also shown that |x| = |fl(z * mx)| < 2 for every column whenever z <= 1.5 (no depth distortion,
        //    1.5 * max|mx| < 2 with slack) — true for any lens narrower than 106 degrees.
        const uint32_t dmax = cut ? P.cut_dmax : 0xFFFFu;
        uint4 dv[kCountTiles];
        const bool aligned = ((uintptr_t)dp & 15) == 0 && n >= 8;      // uniform over the launch
        if (aligned) {
            // branch-free: a lane past the end re-reads the stream's first 16 bytes (and gathers below), so the four
            // loads are issued back to back with no wait between them
#pragma unroll
            for (int q = 0; q < kCountTiles; q++) {
                const uint32_t i0 = (tile_first + q) * kTilePoints + threadIdx.x * kPointsPerLane;
                dv[q] = *reinterpret_cast<const uint4*>(dp + ((i0 + 8 <= n) ? i0 : 0u));
            }
        } else {
#pragma unroll
            for (int q = 0; q < kCountTiles; q++) dv[q] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < kCountTiles; q++) {
            const uint32_t i0 = (tile_first + q) * kTilePoints + threadIdx.x * kPointsPerLane;
            uint32_t dw[4] = {dv[q].x, dv[q].y, dv[q].z, dv[q].w};
            if (!(aligned && i0 + 8 <= n)) {            // ragged end / unaligned raster: gather the halfwords one by one
                dw[0] = dw[1] = dw[2] = dw[3] = 0u;
                for (uint32_t k = 0; k < 8 && i0 + k < n; k++) dw[k >> 1] |= (uint32_t)dp[i0 + k] << ((k & 1u) * 16u);
            }
            uint32_t rng = 0, nz = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
                nz  |= (uint32_t)(d != 0u) << k;
                rng |= (uint32_t)(d != 0u && d <= dmax) << k;
            }
            c[q] = __popc(keep_from_bits(rng, nz, i0, n, flags));
        }
    } else {
        DepthSource<DDIST, CDIST> src{dp};
#pragma unroll
        for (int q = 0; q < kCountTiles; q++) {
            const uint32_t i0 = (tile_first + q) * kTilePoints + threadIdx.x * kPointsPerLane;
            PointIn p[8];
            src.load8(P, i0, n, p);
            c[q] = __popc(keep_mask8(p, i0, n, flags));
        }
    }
#pragma unroll
    for (int q = 0; q < kCountTiles; q++) {
        const uint32_t w = wave_sum(c[q]);
        if ((threadIdx.x & 63) == 0) wsum[q][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < kCountTiles && (tile_first + threadIdx.x) * kTilePoints < n)
        tile_counts[P.tile_base + tile_first + threadIdx.x] =
            wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
}

template <bool DDIST, bool CDIST>
__global__ __launch_bounds__(kBlockThreads)
void pcs_fused_count_kernel(const StreamParams* __restrict__ params, int stream0, FramePtrs fp, uint32_t flags,
                            uint32_t* __restrict__ tile_counts)
{
    __shared__ uint32_t wsum[kCountTiles][4];
    const int s = blockIdx.y;
    // (no request_constants here: the usual route needs four words of the stream's constants, and 98 SGPRs would cap the
    // kernel at 6 workgroups per CU where 8 fit)
    count_tiles<DDIST, CDIST>(params[stream0 + s], fp.depth[s], flags, tile_counts, wsum);
}

// K frame-sets: blockIdx.z = frame-set, its tile counts at tile_counts + z * total_tiles.
template <bool DDIST, bool CDIST>
__global__ __launch_bounds__(kBlockThreads)
void pcs_fused_count_batch_kernel(const StreamParams* __restrict__ params, BatchPtrs bp, uint32_t flags,
                                  uint32_t* __restrict__ tile_counts, uint32_t total_tiles)
{
    __shared__ uint32_t wsum[kCountTiles][4];
    const int s = blockIdx.y;
    count_tiles<DDIST, CDIST>(params[s], bp.depth[blockIdx.z * gridDim.y + s], flags,
                              tile_counts + (size_t)blockIdx.z * total_tiles, wsum);
}

// (Placing the emit tiles straight from per-chunk totals of the count pass — no scan launch, four extra L2-resident loads
// per lane riding on the tile's existing barrier — was built and measured: 32.4 vs 32.5 us on 8 x 720p and 127 vs 107 us
// on 16 x 1080p. Whatever the scan launch costs, extra memory instructions in the emit tile cost at least as much.)
template <bool PRED, bool DS1, class Mth>
__global__ __launch_bounds__(kBlockThreads, EMIT_WAVES)
void pcs_fused_emit_kernel(const StreamParams* __restrict__ params, int stream0, FramePtrs fp, uint32_t flags,
                           uint32_t ds, const uint32_t* __restrict__ tile_prefix,
                           const uint32_t* __restrict__ stream_kept, uint8_t* __restrict__ payload_bytes,
                           int32_t* __restrict__ total_out, int n_total_streams)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t wsum[4];
    const int s = blockIdx.y;
    if (PRED && total_out && blockIdx.x == 0 && stream0 + s == 0 && threadIdx.x == 0) {
        // the grand total rides on the first workgroup of the first emit launch (the scan kernel used to compute it
        // with a last-arriver atomic: a third of its 4.8 us)
        uint32_t tot = 0;
        for (int e = 0; e < n_total_streams; e++) tot += DS1 ? stream_kept[e] : (stream_kept[e] + ds - 1) / ds;
        *total_out = (int32_t)tot;
    }
    const StreamParams& P = params[stream0 + s];
    request_constants<true, Mth::kIdentR>(P, fp.depth[s], fp.color[s], payload_bytes);
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    DepthSource<true, true, Mth> src{fp.depth[s]};
    const uint32_t g0 = PRED ? tile_prefix[P.tile_base + blockIdx.x] : tile0;
    uint32_t out_first = P.out_base;
    if (PRED) {     // a7: this camera starts after the strided kept points of all earlier cameras
        out_first = 0;
        for (int e = 0; e < stream0 + s; e++) out_first += DS1 ? stream_kept[e] : (stream_kept[e] + ds - 1) / ds;
    }
    generic_tile<DepthSource<true, true, Mth>, PRED, DS1>(P, src, fp.color[s], tile0, n, flags, ds, g0, out_first,
                                                  payload_bytes, stage, wsum);
}

// K frame-sets of ordered compaction (stride 1): blockIdx.z = frame-set. Prefixes at tile_prefix + z * total_tiles, the
// per-stream kept totals at stream_kept + z * S; frame-set z's total is written by its first workgroup.
template <class Mth>
__global__ __launch_bounds__(kBlockThreads, 6)
void pcs_fused_emit_batch_kernel(const StreamParams* __restrict__ params, BatchPtrs bp, uint32_t flags,
                                 const uint32_t* __restrict__ tile_prefix, const uint32_t* __restrict__ stream_kept,
                                 uint32_t total_tiles, BatchCounts bc)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t wsum[4];
    const int s = blockIdx.y, S = gridDim.y, z = blockIdx.z;
    const uint32_t* __restrict__ kept = stream_kept + z * S;
    if (blockIdx.x == 0 && s == 0 && threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int e = 0; e < S; e++) tot += kept[e];
        bc.counts[z][S] = (int32_t)tot;
    }
    const StreamParams& P = params[s];
    // (no request_constants here: with it hipcc settles on 76 VGPRs = 6 waves/SIMD instead of 72 = 7, 27.5 vs 25.0 us per
    // set; this launch is 4 x as long as a one-set launch, so its first wave of workgroups matters a quarter as much)
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    DepthSource<true, true, Mth> src{bp.depth[z * S + s]};
    const uint32_t g0 = tile_prefix[(size_t)z * total_tiles + P.tile_base + blockIdx.x];
    uint32_t out_first = 0;
    for (int e = 0; e < s; e++) out_first += kept[e];
    generic_tile<DepthSource<true, true, Mth>, true, true>(P, src, bp.color[z * S + s], tile0, n, flags, 1u, g0, out_first,
                                                           bp.payload[z], stage, wsum);
}

// Exclusive scan of the tile counts, one workgroup of 1024 lanes PER STREAM (streams scan concurrently).
// Writes the tile prefixes, the stream's kept total and its output count ceil(kept / stride); the
// stream's base in the stitched payload is summed from the totals by the emit kernel, and the grand total
// by whichever workgroup arrives last (agent-scope counter).
__global__ __launch_bounds__(1024)
void pcs_scan_kernel(const StreamParams* __restrict__ params, int stream0, int n_streams, uint32_t override_n,
                     uint32_t ds, const uint32_t* __restrict__ tile_counts, uint32_t* __restrict__ tile_prefix,
                     uint32_t* __restrict__ stream_kept, int32_t* __restrict__ counts, uint32_t* __restrict__ arrive)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int s = blockIdx.x;
    const uint32_t n = override_n ? override_n : params[stream0 + s].n_points;
    const uint32_t tiles = (n + kTilePoints - 1) / kTilePoints;
    const uint32_t tb = override_n ? 0u : params[stream0 + s].tile_base;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        // (a count can never exceed a tile: clamped, so that counts handed in by a caller — pcs_process_frames_device_counted —
        // cannot steer a tile's stores past the payload's worst-case capacity whatever they contain)
        const uint32_t c = (t < tiles) ? min(tile_counts[tb + t], min(kTilePoints, n - t * kTilePoints)) : 0u;
        uint32_t wave_total;
        const uint32_t ex = wave_exclusive_scan(c, wave_total);
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = wave_total;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wtot[w];
        if (t < tiles) tile_prefix[tb + t] = before + ex;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + ex + c;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t kept = carry_s;
        const uint32_t outc = (kept + ds - 1) / ds;
        if (stream_kept) stream_kept[stream0 + s] = kept;
        if (counts && !arrive) {
            counts[stream0 + s] = (int32_t)outc;      // the emit kernel that follows adds up the grand total
        } else if (counts) {
            // grand total: the last workgroup to arrive adds up the per-stream outputs. The per-stream
            // counts are written with agent-scope stores and read back with agent-scope loads, and the
            // arrival counter is an agent-scope atomic, so the last arriver sees every other write.
            __hip_atomic_store(counts + stream0 + s, (int32_t)outc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t ticket = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (uint32_t)n_streams - 1u) {
                    int32_t tot = 0;
                for (int e = 0; e < n_streams; e++)
                    tot += __hip_atomic_load(counts + stream0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                counts[stream0 + n_streams] = tot;
                __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
        }
    }
}

// The scan of K frame-sets: blockIdx.x = stream, blockIdx.y = frame-set (stride 1, no grand-total atomic: the emit
// launch that follows adds the totals).
__global__ __launch_bounds__(1024)
void pcs_scan_batch_kernel(const StreamParams* __restrict__ params, const uint32_t* __restrict__ tile_counts,
                           uint32_t* __restrict__ tile_prefix, uint32_t* __restrict__ stream_kept, uint32_t total_tiles,
                           BatchCounts bc)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int s = blockIdx.x, z = blockIdx.y;
    const uint32_t n = params[s].n_points;
    const uint32_t tiles = (n + kTilePoints - 1) / kTilePoints;
    const size_t tb = (size_t)z * total_tiles + params[s].tile_base;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t c = (t < tiles) ? tile_counts[tb + t] : 0u;
        uint32_t wave_total;
        const uint32_t ex = wave_exclusive_scan(c, wave_total);
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = wave_total;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wtot[w];
        if (t < tiles) tile_prefix[tb + t] = before + ex;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + ex + c;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stream_kept[z * gridDim.x + s] = carry_s;
        bc.counts[z][s] = (int32_t)carry_s;
    }
}

#endif  // !PCS_TU_VOXEL

#if PCS_TU_VOXEL
// ---- rasters -> voxel partials (config 5 without the stitched payload) ---------------------------------------------
// pcs_process_frames_voxel_device: the voxel grid of the cloud pcs_process_frames_device would stitch, without writing
// that cloud (10 B per kept point out, 10 B back in for the pre-aggregation) and without its ordered placement (count
// and scan passes): the voxel sums are integers, so the order of the points is irrelevant. A workgroup of 1024 lanes
// takes 8192 consecutive pixels of one camera; a lane deprojects, transforms and packs its 8 consecutive pixels exactly
// as the stitch kernels do (same records, bit for bit), sums runs of equal voxel keys among them in registers (8
// neighbouring pixels mostly share a voxel) and adds each run to the workgroup's LDS hash table (pcs_voxel.hip, step 1);
// one partial per occupied slot is appended to the same arrays the payload reader fills, and pcs_voxel.hip's sort and
// segmented mean run unchanged.
#include "pcs_voxel_agg.h"
#include "pcs_vox_tiling.h"

#ifndef PCS_VOX_THREADS
#define PCS_VOX_THREADS 512
#endif
#ifndef PCS_VOX_SLOTS
#define PCS_VOX_SLOTS 2048
#endif
#ifndef PCS_VOX_PREFETCH
#define PCS_VOX_PREFETCH 1
#endif
// Workgroup shape of the raster / payload readers: 512 lanes (8 wavefronts) and a 2048-slot table (72 KiB): two workgroups per
// CU, 4 wavefronts per SIMD at 120 - 126 VGPRs (~105 when these measurements were taken, with packed records). Round 5 measured the two ways to a FIFTH wavefront per SIMD (<= 96 VGPRs), 16 x 1080p
// at 50 mm, one call, A/B on one box:
//   640 lanes x 2 workgroups           0.266 vs 0.197 ms — a workgroup's 10 wavefronts are dealt to the SIMDs 3-3-2-2 and two such
//                                      workgroups need six slots on two SIMDs: only ONE fits a CU (a workgroup must be a multiple of
//                                      four wavefronts);
//   256 lanes x 5 workgroups, 896-slot 0.213 vs 0.205 ms — five wavefronts per SIMD are reached (96 VGPRs, no spill, 31.5 KiB per
//   tables (-DPCS_VOX_THREADS=256      table, same 128 x 64 patch per table in four rounds), and the launch is slower: the kernel is
//   -DPCS_VOX_SLOTS=896)               not short of wavefronts to issue from.
// Both shapes still build (the code below is generic in the two constants); neither is used.
// The same point for the voxel readers: coordinates as sign-correct integers (the record's int16 fields, widened), the colour
// dword as fetched (R | G<<8 | B<<16 | don't care). Cvt::kCoordsInShort: the policy vouches that the converted values lie in
// int16 (VoxCvt); otherwise the low 16 bits are sign-extended here, as the record would hold them.
struct VoxPoint {
    int32_t x, y, z;
    uint32_t w;
};
template <bool ROWC = false, class Cvt>
__device__ __forceinline__ VoxPoint make_vox_point(const StreamParams& P, const uint8_t* __restrict__ color,
                                                   const PointIn& p, Cvt& cv)
{
    const float ax = world_mm(P.M + 0, p.X, p.Y, p.Z);
    const float ay = world_mm(P.M + 4, p.X, p.Y, p.Z);
    const float az = world_mm(P.M + 8, p.X, p.Y, p.Z);
    float xf, yf;
    color_coords(P, p.u, p.v, xf, yf);
    cv.note(ax, ay, az, xf, ROWC ? xf : yf);
    const int32_t x = cv.cvt(ax), y = cv.cvt(ay), z = cv.cvt(az);
    // (ROWC: p.v IS the row, certified at pcs_create for every depth value — whatever conversion policy the lane runs under)
    const int32_t yi = ROWC ? __float_as_int(p.v) : cv.pixel(yf, P.cH - 1, P.c_hm1_f);
    const uint32_t w = color_fetch(P, color, cv.pixel(xf, P.cW - 1, P.c_wm1_f), yi, cv);
    if (Cvt::kCoordsInShort) return VoxPoint{x, y, z, w};
    return VoxPoint{(int32_t)(int16_t)x, (int32_t)(int16_t)y, (int32_t)(int16_t)z, w};
}
// what vox_table_round reads of a point, for both forms
__device__ __forceinline__ int pt_x(const Record& r) { return (int)(short)(r.xy & 0xFFFFu); }
__device__ __forceinline__ int pt_y(const Record& r) { return (int)(short)(r.xy >> 16); }
__device__ __forceinline__ int pt_z(const Record& r) { return (int)(short)(r.zc & 0xFFFFu); }
__device__ __forceinline__ unsigned int pt_red(const Record& r) { return (r.zc >> 16) & 0xFFu; }
__device__ __forceinline__ unsigned int pt_green(const Record& r) { return r.zc >> 24; }
__device__ __forceinline__ unsigned int pt_blue(const Record& r) { return r.b & 0xFFu; }
__device__ __forceinline__ int pt_x(const VoxPoint& r) { return r.x; }
__device__ __forceinline__ int pt_y(const VoxPoint& r) { return r.y; }
__device__ __forceinline__ int pt_z(const VoxPoint& r) { return r.z; }
__device__ __forceinline__ unsigned int pt_red(const VoxPoint& r) { return r.w & 0xFFu; }
__device__ __forceinline__ unsigned int pt_green(const VoxPoint& r) { return (r.w >> 8) & 0xFFu; }
__device__ __forceinline__ unsigned int pt_blue(const VoxPoint& r) { return (r.w >> 16) & 0xFFu; }

// Which of a lane's 8 points take part. (Eight predicates held in the condition registers instead of the mask's bits, for the
// reader without the -c test, measured: 4 % fewer VALU instructions with the wide points below, no faster — and the round's code twice.)
struct KeepBits {
    uint32_t m;
    __device__ __forceinline__ bool operator()(int k) const { return (m >> k) & 1u; }
};

constexpr int kVoxThreads = PCS_VOX_THREADS;
constexpr int kVoxSlots = PCS_VOX_SLOTS;
constexpr uint32_t kVoxRows = kVoxThreads / 8;    // a round = kVoxRows rows of 64 pixels (8 lanes x 8 pixels)
constexpr int kVoxOwn = (kVoxSlots + kVoxThreads - 1) / kVoxThreads;      // table slots a lane flushes
constexpr uint32_t kVoxRoundPoints = kVoxThreads * kPointsPerLane;      // 4096 points per round; `rounds` of them share one table

// The workgroup's LDS table: slot = key + the seven sums in three 64-bit words and one 32-bit word: (x, y), (z, count),
// (R, G), B — four LDS adds per run instead of seven. Coordinates are summed BIASED (+32768, so every term is
// non-negative and a 64-bit add never carries between its halves: <= 32 768 points x 65 535 < 2^31); the bias leaves at
// the output.
// Probe sequence in a table of kVoxSlots slots (not a power of two): start = the hash's top 16 bits scaled into the table
// with one 24-bit multiply, odd stride, wrap by a conditional subtract. A run that finds no slot within kProbe probes goes out
// as a partial of its own, so the sequence need not visit every slot.
struct VoxProbe {
    unsigned int first, step;
    __device__ __forceinline__ explicit VoxProbe(unsigned long long key)
    {
        const unsigned int lo = (unsigned int)key, hi = (unsigned int)(key >> 32);
        const unsigned int m = __umul24(lo, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbit(hi, lo, 24), 0x85EBCBu);
        if (kVoxSlots == kSlots) {                                         // the 2^11 table: VoxelProbe's sequence (pcs_voxel_agg.h)
            first = m >> 21;
            step = ((m >> 10) & (unsigned)(kSlots - 1)) | 1u;
        } else {
            first = __umul24(m >> 16, (unsigned int)kVoxSlots) >> 16;      // top 16 bits scaled into the table (the product stays below 2^32)
            step = ((m >> 3) & 0x7Fu) | 1u;                                // odd, < 128
        }
    }
    __device__ __forceinline__ unsigned int next(unsigned int h) const
    {
        if (kVoxSlots == kSlots) return (h + step) & (unsigned)(kSlots - 1);
        h += step;
        return h >= (unsigned int)kVoxSlots ? h - (unsigned int)kVoxSlots : h;
    }
};

struct VoxTable {
    unsigned long long *skey, *sxy, *szn, *srg;
    unsigned int *sbl, *wtot, *base_s, *flag, *kor;       // kor[4]: OR of the keys written (lo, hi), OR of their complements
};
#define PCS_VOX_TABLE_DECL                                                                             \
    __shared__ unsigned long long skey_[kVoxSlots];                                                       \
    __shared__ unsigned long long sxy_[kVoxSlots], szn_[kVoxSlots], srg_[kVoxSlots];                            \
    __shared__ unsigned int sbl_[kVoxSlots];                                                              \
    __shared__ unsigned int wtot_[kVoxThreads / 64];                                                   \
    __shared__ unsigned int base_s_, flag_, kor_[4];                                                   \
    const VoxTable T{skey_, sxy_, szn_, srg_, sbl_, wtot_, &base_s_, &flag_, kor_};                    \
    unsigned long long key_or = 0ull, key_orn = 0ull

// Warm bucket tail (vs.regions): the geometry of this call's regions, and one partial put into its bucket's region by a lane on
// its own (a run that found no slot in the workgroup's table: a few dozen per frame-set) — the bucket by binary search in
// the global splitters (ten dependent trips to L2: a few dozen lanes per frame-set take them), the slot by a returning add on
// the bucket's cursor. The workgroup's flush does the same for all the table's partials at once (vox_table_flush_regions).
struct VoxRegions {
    unsigned int B, cap, stride;
    __device__ __forceinline__ explicit VoxRegions(const VoxelStage& vs)
    {
        B = vs.reg[0]; cap = vs.reg[1];
        if (B == 0u || B > kVoxBuckets) B = kVoxBuckets;
        // (the tail applies the same rule. The regions hold any cloud of THIS call's capacity; the sizes come from the previous
        // call, which may have had a larger one: then nothing fits and everything goes to the general list)
        if ((unsigned long long)B * cap > vs.region_slots) cap = 0u;
        stride = kVoxBuckets / B;
    }
    // bucket j ends below splitter j (the last one is open)
    __device__ __forceinline__ unsigned long long splitter(const VoxelStage& vs, unsigned int j) const
    {
        return (j + 1u < B) ? vs.spl[(j + 1u) * stride - 1u] : kEmptyKey;
    }
};
__device__ __forceinline__ void vox_region_put(const VoxelStage& vs, unsigned long long key, const VoxelPartial& v)
{
    // (Measured and not kept: half of the splitters in LDS for this search — 4 KiB per workgroup, stored behind a barrier before the
    // first round — and one returning add per wavefront and bucket instead of one per run: +4 .. 7 us on every 16 x 1080p
    // frame-set for 0.3 ms off a call on a cloud of scattered points, whose runs mostly fail.)
    const VoxRegions R(vs);
    unsigned int b = 0;
#pragma unroll 1
    for (unsigned int step = kVoxBuckets / 2; step; step >>= 1)
        if (R.splitter(vs, b + step - 1u) <= key) b += step;
    const unsigned int at = atomicAdd(&vs.cursor[b], 1u);
    if (at < R.cap) {
        const size_t dst = (size_t)b * R.cap + at;
        vs.keys_r[dst] = key;
        static_cast<VoxelPartial*>(vs.part_r)[dst] = v;
    } else {
        const unsigned int e = atomicAdd(vs.n_runs, 1u);
        vs.keys[e] = key;
        static_cast<VoxelPartial*>(vs.part)[e] = v;
        vs.bucket_of[e] = (unsigned short)b;
    }
}

__device__ __forceinline__ void vox_table_init(const VoxTable& T)
{
    for (int j = threadIdx.x; j < kVoxSlots; j += kVoxThreads) {
        T.skey[j] = kEmptyKey;
        T.sxy[j] = T.szn[j] = T.srg[j] = 0ull;
        T.sbl[j] = 0u;
    }
    if (threadIdx.x == 0) *T.flag = 0u;
    if (threadIdx.x < 4) T.kor[threadIdx.x] = 0u;
    __syncthreads();
}

// One round: a lane's 8 consecutive records (bit k of `keep`: record k takes part) -> the table.
template <class Pt, class Keep>
__device__ __forceinline__ void vox_table_round(const VoxTable& T, const VoxelStage& vs, const Pt (&rec)[8], const Keep& keep,
                                                bool crowded, unsigned long long& key_or, unsigned long long& key_orn)
{
    unsigned long long* const skey = T.skey; unsigned long long* const sxy = T.sxy; unsigned long long* const szn = T.szn;
    unsigned long long* const srg = T.srg; unsigned int* const sbl = T.sbl;
    const VoxelDiv dv{vs.div_inv, vs.div_c};
    const unsigned int bits = vs.bits, idx_bits = vs.idx_bits;
    VoxelPartial* __restrict__ part = static_cast<VoxelPartial*>(vs.part);
    const int lane = threadIdx.x & 63;
    auto key_of = [&](const Pt& r) { return voxel_key(dv, pt_x(r), pt_y(r), pt_z(r), bits); };
    // runs of equal keys among the lane's 8 pixels: summed in registers, the run's LAST point adds them to the table
    unsigned int ax = 0, ay = 0, az = 0;                     // biased: sums of (coordinate + 32768)
    unsigned int ar = 0, ag = 0, ab = 0, an = 0, failed = 0;
    bool cont = false;                                       // point k continues the run of point k-1
    unsigned long long kcur = key_of(rec[0]);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const bool live = keep(k);
        const bool live_next = k < 7 && keep(k < 7 ? k + 1 : 7);
        const unsigned long long knext = k < 7 ? key_of(rec[k + 1]) : 0ull;
        const int x = pt_x(rec[k]), y = pt_y(rec[k]), z = pt_z(rec[k]);
        if (!cont) { ax = ay = az = 0u; ar = ag = ab = an = 0u; }
        ax += (unsigned int)(x + 32768); ay += (unsigned int)(y + 32768); az += (unsigned int)(z + 32768);
        ar += pt_red(rec[k]); ag += pt_green(rec[k]); ab += pt_blue(rec[k]); an += 1u;
        const bool same_next = live_next && knext == kcur;
        const bool actor = live && !same_next;
        if (actor) {
            const VoxProbe pr(kcur);
            unsigned int h = pr.first;
            // first probe straight-line (it succeeds for all but a few per cent of the runs), the rest in a loop
            unsigned long long old = atomicCAS(&skey[h], kEmptyKey, kcur);
            bool placed = old == kEmptyKey || old == kcur;
            if (__builtin_expect(!placed, 0)) {
                for (int t = 1; t < kProbe; t++) {
                    h = pr.next(h);
                    old = atomicCAS(&skey[h], kEmptyKey, kcur);
                    if (old == kEmptyKey || old == kcur) { placed = true; break; }
                }
            }
            if (placed) {
                atomicAdd(&sxy[h], (unsigned long long)ax | ((unsigned long long)ay << 32));
                atomicAdd(&szn[h], (unsigned long long)az | ((unsigned long long)an << 32));
                atomicAdd(&srg[h], (unsigned long long)ar | ((unsigned long long)ag << 32));
                atomicAdd(&sbl[h], ab);
            } else {
                failed |= 1u << k;                           // the run ending at k goes out as a partial of its own
            }
        }
        cont = live && same_next;
        kcur = knext;
    }
    // Runs that found no slot (more voxels under this table than it can take: leaves of a few pixels) are appended as
    // partials of their own. The sums are rebuilt from the records: a failed run is the maximal stretch of kept points with
    // the same key that ends at its bit. Where they go is reserved
    //  * crowded (the launcher expects full tables: leaves below 30 mm): with ONE returning global atomic per workgroup and
    //    round, and only when it happens — the workgroup learns that with one barrier per round (3 - 8 % of the kernel
    //    when nothing ever fails, hence the switch);
    //  * otherwise: with one atomic per wavefront that has a failed run — free when there is none, but serialised on the
    //    counter's line when most wavefronts have one (65 k per 16 x 1080p frame-set at 10 mm: 490 us of the kernel's 627).
    const unsigned long long any_failed = __ballot(failed != 0u);    // (all lanes: not inside a short-circuit)
    unsigned int pos = 0;
    bool emit = false;
    const bool to_regions = vs.regions != 0u;                         // uniform over the launch: every failed run finds its own place
    if (to_regions) {
        emit = any_failed != 0ull;
    } else if (crowded) {                                             // uniform over the launch
        if (lane == 0 && any_failed) *T.flag = 1u;
        __syncthreads();
        if (*T.flag) {                                                // workgroup-uniform
            const int wave = threadIdx.x >> 6;
            const unsigned int c = __popc(failed);
            const unsigned int inc = wave_inclusive_scan(c);
            if (lane == 63) T.wtot[wave] = inc;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned int tot = 0;
                for (int w = 0; w < kVoxThreads / 64; w++) { const unsigned int t = T.wtot[w]; T.wtot[w] = tot; tot += t; }
                *T.base_s = atomicAdd(vs.n_runs, tot);
                *T.flag = 0u;
            }
            __syncthreads();
            pos = *T.base_s + T.wtot[wave] + inc - c;
            emit = true;
        }
    } else if (any_failed) {
        const unsigned int c = __popc(failed);
        const unsigned int inc = wave_inclusive_scan(c);
        unsigned int base = 0;
        if (lane == 63) base = atomicAdd(vs.n_runs, inc);
        pos = (unsigned int)__builtin_amdgcn_readlane((int)base, 63) + inc - c;
        emit = true;
    }
    if (emit) {
        int sx = 0, sy = 0, sz = 0;
        unsigned int r = 0, g = 0, b = 0, cnt = 0;
        unsigned long long kprev = 0ull;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool live = keep(k);
            const int x = pt_x(rec[k]), y = pt_y(rec[k]), z = pt_z(rec[k]);
            const unsigned long long key = voxel_key(dv, x, y, z, bits);
            const bool joins = k > 0 && live && keep(k > 0 ? k - 1 : 0) && key == kprev;
            if (!joins) { sx = sy = sz = 0; r = g = b = cnt = 0u; }
            sx += x; sy += y; sz += z; r += pt_red(rec[k]); g += pt_green(rec[k]); b += pt_blue(rec[k]); cnt += 1u;
            kprev = key;
            if ((failed >> k) & 1u) {
                const VoxelPartial v{sx, sy, sz, r, g, b, cnt, 0u};
                if (to_regions) {
                    vox_region_put(vs, key, v);
                } else {
                    if (idx_bits) vs.keys[pos] = (key << idx_bits) | pos;
                    else { vs.keys[pos] = key; if (vs.idx) vs.idx[pos] = pos; }
                    part[pos] = v;
                    key_or |= key; key_orn |= ~key;
                    pos++;
                }
            }
        }
    }
    if (crowded) __syncthreads();         // wtot / base_s are free again before the next round (or the flush) uses them
}

// End of the workgroup on a WARM bucket tail (vs.regions): every partial straight into its bucket's region. The key array of the
// table is dead once each lane holds its slots' keys, and becomes: this call's splitters (8 KiB) | the workgroup's count per
// bucket (4 KiB) | where its partials start in each region (4 KiB). Ten dependent LDS reads per key find the bucket (the lane's
// keys side by side), a returning LDS add ranks the partial among the workgroup's for that bucket, ONE returning global add per
// bucket the workgroup touches (a 128 x 64-pixel patch touches a few dozen) reserves the slots. A partial that finds its region
// full goes to the general list (keys / part / bucket_of) with a returning add of its own.
__device__ __forceinline__ void vox_table_flush_regions(const VoxTable& T, const VoxelStage& vs)
{
    if (kVoxSlots < 2 * (int)kVoxBuckets) __builtin_trap();                 // (the lab shapes with small tables: no room for the arrays below)
    unsigned long long* const spl = T.skey;
    unsigned int* const hist = reinterpret_cast<unsigned int*>(T.skey + kVoxBuckets);
    unsigned int* const rbase = hist + kVoxBuckets;
    VoxelPartial* __restrict__ part = static_cast<VoxelPartial*>(vs.part);
    VoxelPartial* __restrict__ part_r = static_cast<VoxelPartial*>(vs.part_r);
    const VoxRegions R(vs);
    const unsigned int cap = R.cap;
    // (requesting the splitters when the workgroup starts — two dependent trips to L2 that the rounds would hide — costs four
    // registers through the rounds and measured no gain at 40 / 50 mm, 2 us more at 100 / 200 mm)
    constexpr int kSplPer = ((int)kVoxBuckets + kVoxThreads - 1) / kVoxThreads;
    unsigned long long sp[kSplPer];
#pragma unroll
    for (int i = 0; i < kSplPer; i++) sp[i] = R.splitter(vs, threadIdx.x + (unsigned int)(i * kVoxThreads));
    __syncthreads();                                                        // the last round's adds are in
    unsigned long long key[kVoxOwn];
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) { const int j = threadIdx.x * kVoxOwn + q; key[q] = j < kVoxSlots ? T.skey[j] : kEmptyKey; }
    __syncthreads();                                                        // every key is in a register: the array is free
#pragma unroll
    for (int i = 0; i < kSplPer; i++) {
        const unsigned int j = threadIdx.x + (unsigned int)(i * kVoxThreads);
        if (j < kVoxBuckets) { spl[j] = sp[i]; hist[j] = 0u; }
    }
    __syncthreads();
    unsigned int bk[kVoxOwn], rk[kVoxOwn];
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) bk[q] = 0u;
#pragma unroll
    for (unsigned int step = kVoxBuckets / 2; step; step >>= 1) {
#pragma unroll
        for (int q = 0; q < kVoxOwn; q++)
            if (spl[bk[q] + step - 1u] <= key[q]) bk[q] += step;
    }
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) {
        rk[q] = 0u;
        if (key[q] != kEmptyKey) rk[q] = atomicAdd(&hist[bk[q]], 1u);
    }
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < kVoxBuckets; j += kVoxThreads) {
        const unsigned int c = hist[j];
        rbase[j] = c ? atomicAdd(&vs.cursor[j], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) {
        const int j = threadIdx.x * kVoxOwn + q;
        if (key[q] == kEmptyKey) continue;
        const unsigned long long xy = T.sxy[j], zn = T.szn[j], rg = T.srg[j];
        const unsigned int cnt = (unsigned int)(zn >> 32);
        const int bias = (int)(cnt << 15);
        const VoxelPartial v{(int)(unsigned int)xy - bias, (int)(unsigned int)(xy >> 32) - bias, (int)(unsigned int)zn - bias,
                             (unsigned int)rg, (unsigned int)(rg >> 32), T.sbl[j], cnt, 0u};
        const unsigned int at = rbase[bk[q]] + rk[q];
        if (at < cap) {
            const size_t dst = (size_t)bk[q] * cap + at;
            vs.keys_r[dst] = key[q];
            part_r[dst] = v;
        } else {
            // (the region is full: one returning add for all the lanes of the wavefront that are here)
            const unsigned long long peers = __ballot(1);
            const unsigned int rank = (unsigned int)__popcll(peers & ((1ull << (threadIdx.x & 63u)) - 1ull));
            unsigned int e = 0;
            if (rank == 0u) e = atomicAdd(vs.n_runs, (unsigned int)__popcll(peers));
            e = (unsigned int)__shfl((int)e, __ffsll((long long)peers) - 1) + rank;
            vs.keys[e] = key[q];
            part[e] = v;
            vs.bucket_of[e] = (unsigned short)bk[q];
        }
    }
}

// End of the workgroup: one partial per occupied slot, one returning global atomic for all of them.
__device__ __forceinline__ void vox_table_flush(const VoxTable& T, const VoxelStage& vs, unsigned long long key_or,
                                                unsigned long long key_orn)
{
    if (vs.regions) { vox_table_flush_regions(T, vs); return; }            // (uniform over the launch)
    unsigned long long* const skey = T.skey; unsigned long long* const sxy = T.sxy; unsigned long long* const szn = T.szn;
    unsigned long long* const srg = T.srg; unsigned int* const sbl = T.sbl; unsigned int* const wtot = T.wtot;
    unsigned int& base_s = *T.base_s;
    const unsigned int idx_bits = vs.idx_bits;
    VoxelPartial* __restrict__ part = static_cast<VoxelPartial*>(vs.part);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    // every lane owns four slots; one partial per occupied slot
    unsigned int c = 0;
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) { const int j = threadIdx.x * kVoxOwn + q; c += j < kVoxSlots && skey[j] != kEmptyKey; }
    const unsigned int inc = wave_inclusive_scan(c);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int w = 0; w < kVoxThreads / 64; w++) { const unsigned int t = wtot[w]; wtot[w] = tot; tot += t; }
        base_s = tot ? atomicAdd(vs.n_runs, tot) : 0u;
    }
    __syncthreads();
    unsigned int pos = base_s + wtot[wave] + inc - c;
#pragma unroll
    for (int q = 0; q < kVoxOwn; q++) {
        const int j = threadIdx.x * kVoxOwn + q;
        if (j < kVoxSlots && skey[j] != kEmptyKey) {
            if (idx_bits) vs.keys[pos] = (skey[j] << idx_bits) | pos;
            else { vs.keys[pos] = skey[j]; if (vs.idx) vs.idx[pos] = pos; }
            const unsigned long long xy = sxy[j], zn = szn[j], rg = srg[j];
            const unsigned int cnt = (unsigned int)(zn >> 32);
            const int bias = (int)(cnt << 15);                   // count x 32768 (count <= 32 768)
            part[pos] = VoxelPartial{(int)(unsigned int)xy - bias, (int)(unsigned int)(xy >> 32) - bias, (int)(unsigned int)zn - bias,
                                     (unsigned int)rg, (unsigned int)(rg >> 32), sbl[j], cnt, 0u};
            key_or |= skey[j]; key_orn |= ~skey[j];
            pos++;
        }
    }
    // which key bits vary at all (pcs_voxel.hip's sort drops the others): OR of every key this workgroup wrote and of
    // every complement -> wavefront (DPP) -> LDS -> global ORs. Only when the launcher asked for it: the extra barrier and
    // the read of the global words at the very end of every workgroup cost the 16 x 1080p launch 9 us (6 %), which a
    // skipped pass repays three times over — where one can be skipped (pcs_voxel.hip: plan_for).
    if (!vs.track_bits) return;
    unsigned int v[4] = {(unsigned int)key_or, (unsigned int)(key_or >> 32), (unsigned int)key_orn, (unsigned int)(key_orn >> 32)};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned int x = v[q];
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15
        x |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31
        if (lane == 63 && x) atomicOr(&T.kor[q], x);
    }
    __syncthreads();
    // The words live on their own 128-byte line (kVoxCtlOr), away from the partial counter every workgroup adds to, and a
    // workgroup only issues an atomic if it has a bit the word does not show yet (a stale read costs a redundant OR,
    // nothing else): after the first few workgroups almost none do. (Four unconditional ORs per workgroup on the
    // counter's own line cost the 16 x 1080p launch 10 us.)
    if (threadIdx.x < 4) {
        unsigned int* g = vs.n_runs + kVoxCtlOr + threadIdx.x;
        const unsigned int mine = T.kor[threadIdx.x];
        if (mine & ~__builtin_nontemporal_load(g)) atomicOr(g, mine);
    }
}

// 512 lanes x several rounds rather than 1024 x 1: the raster reader needs > 100 VGPRs (the stitch kernels' 8 points in
// flight plus the table phase), which leaves room for one 1024-lane workgroup per CU — its load phase and its LDS phase
// then have nothing to overlap with. Two 512-lane workgroups fit, and there is no barrier between the rounds.
template <bool DD, bool CD, class Mth>
__global__ __launch_bounds__(kVoxThreads) __attribute__((amdgpu_waves_per_eu(4)))     // two workgroups per CU: <= 128 VGPRs
void pcs_fused_voxel_partials_kernel(const StreamParams* __restrict__ params, int stream0, FramePtrs fp, uint32_t flags,
                                     VoxelStage vs, int rounds, VoxTiling tl, int crowded)
{
    PCS_VOX_TABLE_DECL;
    int s = blockIdx.y;
    uint32_t sq_x0 = 0, sq_y0 = 0, nrx = 1, nry = (uint32_t)rounds;          // rx == 0: `rounds` runs of 4096 consecutive pixels
    if (tl.rx) {
        const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;
        PCS_VOX_TILING_DECODE(tl, lin, gridDim.y, s, sq_x0, sq_y0, nrx, nry);
    }
    const StreamParams& P = params[stream0 + s];
    request_constants(P, fp.depth[s], fp.color[s]);
    const uint32_t n = P.n_points;
    const uint32_t W = (uint32_t)P.W, Hh = n / W;
    const uint32_t tile0 = blockIdx.x * (kVoxRoundPoints * (uint32_t)rounds);
    if (tl.rx ? (sq_y0 * kVoxRows >= Hh || sq_x0 * 64u >= W) : (tile0 >= n)) return;      // (a smaller raster than the launch's largest)
    const uint8_t* __restrict__ color = fp.color[s];
    DepthSource<DD, CD, Mth> src{fp.depth[s]};

    // The rounds of this workgroup, in the order (yy, xx) of its patch. A square-row below the raster ends the workgroup, a square
    // beside the raster ends its square-row — uniform over the workgroup, so the barriers of a crowded round stay matched. Where a
    // round has no barrier, a WAVEFRONT whose 8 rows lie below the raster (the last 8 of 1080 = 16 x 64 + 56) passes too.
    // PCS_VOX_PREFETCH: the lane's Z16 quad of the NEXT round is requested before this round's deprojection and table phase (rasters
    // read in 16-byte quads only): the one load of a round that comes from HBM then has a whole round to arrive in.
    auto lane_index = [&](uint32_t yy, uint32_t xx) -> uint32_t {         // this lane's first pixel of round (yy, xx); n: nothing to read
        if (!tl.rx) return tile0 + yy * kVoxRoundPoints + threadIdx.x * kPointsPerLane;
        const uint32_t row = (sq_y0 + yy) * kVoxRows + (threadIdx.x >> 3);
        const uint32_t col = (sq_x0 + xx) * 64u + (threadIdx.x & 7u) * 8u;
        return (row < Hh && col < W) ? row * W + col : n;                // W % 8 == 0: a lane is inside the row or outside it
    };
    const bool prefetch = PCS_VOX_PREFETCH != 0 && src.fast(P);          // uniform over the launch's stream
    uint32_t yy = 0, xx = 0;
    bool have = nry > 0u && nrx > 0u && !(tl.rx && (sq_y0 * kVoxRows >= Hh || sq_x0 * 64u >= W));
    uint32_t i0 = have ? lane_index(0u, 0u) : n;
    uint4 dv = make_uint4(0u, 0u, 0u, 0u);
    if (prefetch && i0 < n) dv = *reinterpret_cast<const uint4*>(fp.depth[s] + i0);      // (on its way while the table is cleared)
    vox_table_init(T);
    while (have) {
        uint32_t ny = yy, nx = xx + 1u;
        if (nx >= nrx || (tl.rx && (sq_x0 + nx) * 64u >= W)) { nx = 0u; ny = yy + 1u; }
        const bool have_next = ny < nry && !(tl.rx && (sq_y0 + ny) * kVoxRows >= Hh);
        const uint32_t i0n = have_next ? lane_index(ny, nx) : n;
        uint4 dvn = make_uint4(0u, 0u, 0u, 0u);
        if (prefetch && i0n < n) dvn = *reinterpret_cast<const uint4*>(fp.depth[s] + i0n);
        const uint32_t row0 = (sq_y0 + yy) * kVoxRows;
        if (!(tl.rx && !crowded && row0 + ((threadIdx.x >> 6) << 3) >= Hh)) {
            PointIn p[8];
            if (prefetch) src.load8_pre(P, i0, n, dv, p);
            else src.load8(P, i0, n, p);
            const KeepBits keep{keep_mask8(p, i0, n, flags)};

            VoxPoint rec[8];
            auto fill = [&](auto& cv) {
#pragma unroll
                for (int k = 0; k < 8; k++) rec[k] = make_vox_point<Mth::kRowConst>(P, color, p[k], cv);
            };
            if (Mth::kCvtMode == 2) {
                VoxCvt<false> fast;
                fill(fast);
                if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
            } else if (Mth::kCvtMode == 1) {
                VoxCvt<true> fast;
                fill(fast);
                if (__builtin_expect(fast.redo(), 0)) { ExactCvt exact; fill(exact); }
            } else {
                ExactCvt exact;
                fill(exact);
            }
            vox_table_round(T, vs, rec, keep, crowded != 0, key_or, key_orn);
        }
        yy = ny; xx = nx; i0 = i0n; dv = dvn; have = have_next;
    }
    vox_table_flush(T, vs, key_or, key_orn);
}

// The same table fed from a packed payload (16-byte aligned): a lane's 8 consecutive records are 80 contiguous bytes,
// five 16-byte loads. (Requested one round ahead like the raster reader's Z16 quad — 20 registers carried over the table phase, 116
// VGPRs — the voxel grid of the 16 x 1080p cloud measured 0.199-0.201 vs 0.202-0.203 ms at 50 mm, 0.096-0.097 vs 0.094 at 200 mm: not kept.) The point count comes from the host or (counted form) from device memory.
__global__ __launch_bounds__(kVoxThreads)
void pcs_payload_voxel_partials_kernel(const int16_t* __restrict__ payload, uint32_t n_host, const int32_t* __restrict__ n_dev,
                                       VoxelStage vs, int rounds, int crowded)
{
    PCS_VOX_TABLE_DECL;
    const uint32_t n = n_dev ? (uint32_t)max(*n_dev, 0) : n_host;
    const uint32_t tile0 = blockIdx.x * (kVoxRoundPoints * (uint32_t)rounds);
    if (tile0 >= n) return;
    vox_table_init(T);
    for (int round = 0; round < rounds; round++) {
        const uint32_t i0 = tile0 + round * kVoxRoundPoints + threadIdx.x * kPointsPerLane;
        uint32_t w[20];
        if (i0 + 8u <= n) {
            const uint4* q = reinterpret_cast<const uint4*>(payload + (size_t)i0 * PCS_POINT_SHORTS);
#pragma unroll
            for (int j = 0; j < 5; j++) { const uint4 v = q[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
        } else {
#pragma unroll
            for (int j = 0; j < 20; j++) {                        // ragged end: halfword by halfword, zeros past the end
                const uint32_t h0 = (uint32_t)(2 * j), h1 = h0 + 1u;
                const size_t base = (size_t)i0 * PCS_POINT_SHORTS;
                const uint32_t lo = (i0 < n && base + h0 < (size_t)n * PCS_POINT_SHORTS) ? (uint16_t)payload[base + h0] : 0u;
                const uint32_t hi = (i0 < n && base + h1 < (size_t)n * PCS_POINT_SHORTS) ? (uint16_t)payload[base + h1] : 0u;
                w[j] = lo | (hi << 16);
            }
        }
        Record rec[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {                          // two records = five dwords: xy zc b|x' y'|z' c'|b'
            const uint32_t* o = w + (k >> 1) * 5;
            rec[k].xy = o[0]; rec[k].zc = o[1]; rec[k].b = o[2] & 0xFFFFu;
            rec[k + 1].xy = perm(o[3], o[2], kHiLo); rec[k + 1].zc = perm(o[4], o[3], kHiLo); rec[k + 1].b = o[4] >> 16;
        }
        uint32_t keep = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) keep |= (uint32_t)(i0 + k < n) << k;
        vox_table_round(T, vs, rec, KeepBits{keep}, crowded != 0, key_or, key_orn);
    }
    vox_table_flush(T, vs, key_or, key_orn);
}

#endif  // PCS_TU_VOXEL

#if !PCS_TU_VOXEL
// ---- a2 twin -----------------------------------------------------------------------------------
template <uint32_t THREADS = kBlockThreads>
__global__ __launch_bounds__(THREADS)
void pcs_pack_dense_kernel(const StreamParams* __restrict__ params, int stream, VertexPtrs vp,
                                  uint8_t* __restrict__ out_bytes)
{
    __shared__ uint4 stage[THREADS * kPointsPerLane * PCS_POINT_BYTES / 16];
    const StreamParams& P = params[stream];
    const uint32_t n = vp.n_points;
    const uint32_t tile0 = blockIdx.x * (THREADS * kPointsPerLane);
    if (tile0 >= n) return;
    VertexSource src{vp.vertices, vp.texcoords};
    dense_tile<VertexSource, THREADS>(P, src, vp.color, tile0, n, out_bytes, stage);
}

__global__ __launch_bounds__(kBlockThreads)
void pcs_pack_count_kernel(const StreamParams* __restrict__ params, int stream, VertexPtrs vp, uint32_t flags,
                           uint32_t* __restrict__ tile_counts)
{
    __shared__ uint32_t wsum[4];
    const uint32_t n = vp.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    const uint32_t i0 = tile0 + threadIdx.x * kPointsPerLane;
    // the predicate needs x and z only; read them straight from global (12-byte stride)
    PointIn p[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t i = i0 + k;
        p[k] = PointIn{0, 0, 0, 0, 0};
        if (i < n) { p[k].X = vp.vertices[3 * (size_t)i]; p[k].Z = vp.vertices[3 * (size_t)i + 2]; }
    }
    uint32_t c = __popc(keep_mask8(p, i0, n, flags));
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    (void)params; (void)stream;
}

template <bool PRED>
__global__ __launch_bounds__(kBlockThreads)
void pcs_pack_emit_kernel(const StreamParams* __restrict__ params, int stream, VertexPtrs vp, uint32_t flags,
                          const uint32_t* __restrict__ tile_prefix, uint8_t* __restrict__ out_bytes)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t wsum[4];
    const StreamParams& P = params[stream];
    const uint32_t n = vp.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    VertexSource src{vp.vertices, vp.texcoords};
    const uint32_t g0 = PRED ? tile_prefix[blockIdx.x] : tile0;
    generic_tile<VertexSource, PRED, true>(P, src, vp.color, tile0, n, flags, 1u, g0, 0u, out_bytes, stage, wsum);
}

// Batched a2 twin (no predicate): blockIdx.y = cloud. One launch for all cameras of a frame-set instead of one
// latency-bound launch per camera (the reference calls copyPointCloudXYZRGBToBufferSIMD once per camera process).
template <bool ALIGNED>
__global__ __launch_bounds__(kBlockThreads)
void pcs_pack_batch_kernel(const StreamParams* __restrict__ params, PackBatch pb)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t wsum[4];
    const int e = blockIdx.y;
    const VertexPtrs vp = pb.v[e];
    const StreamParams& P = params[pb.stream[e]];
    const uint32_t n = vp.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    VertexSource src{vp.vertices, vp.texcoords};
    if (ALIGNED)
        dense_tile(P, src, vp.color, tile0, n, pb.out[e], reinterpret_cast<uint4*>(stage));
    else
        generic_tile<VertexSource, false, true>(P, src, vp.color, tile0, n, 0u, 1u, tile0, 0u, pb.out[e], stage, wsum);
}

// ---- a5 alone ----------------------------------------------------------------------------------
template <bool DDIST, bool CDIST>
__global__ __launch_bounds__(kBlockThreads)
void pcs_deproject_kernel(const StreamParams* __restrict__ params, int stream, const uint16_t* __restrict__ depth,
                          float* __restrict__ vertices, float* __restrict__ texcoords)
{
    const StreamParams& P = params[stream];
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= P.n_points) return;
    const uint32_t r = i / (uint32_t)P.W;
    const uint32_t c = i - r * (uint32_t)P.W;
    const PointIn p = deproject_pixel<DDIST, CDIST, IeeeMath>(P, depth[i], as_global(P.mx)[c], as_global(P.my)[r]);
    vertices[3 * (size_t)i + 0] = p.X;
    vertices[3 * (size_t)i + 1] = p.Y;
    vertices[3 * (size_t)i + 2] = p.Z;
    texcoords[2 * (size_t)i + 0] = p.u;
    texcoords[2 * (size_t)i + 1] = p.v;
}

// ---- the centre's re-transform of packed payloads (src/pcs-multicamera-optimized.cpp:226-265, 289) -------------------------
// What pcs-multicamera-optimized does to every camera's payload before it concatenates them: int16 millimetres -> float metres
// (`(float)buffer[..] / CONV_RATE`, CONV_RATE a `const float` 1000.0 in that file, :46, :237-239), pcl::transformPointCloud with
// transform[thread_num] (:289), float metres -> int16 millimetres (`static_cast<short>(x * CONV_RATE)`, :255-257), colour bytes
// re-packed (:240-242, :258-259: R | G<<8 survives as it is, the high byte of the B short is cleared). The affine is evaluated in
// the order of PCL 1.8's transforms.hpp (Ubuntu 18.04's libpcl-dev, Dockerfile:1,24) — ((m0*x + m1*y) + m2*z) + m3, every product
// and sum individually rounded (the target is built without -mfma, src/CMakeLists.txt) — third-party, so "parity unpinned".
// HBM-bound: 10 B in + 10 B out per kept record. One tile = 2048 output records: the tile's input bytes come in as
// lane-contiguous 16-byte loads into LDS at the input's 16-byte phase, every lane takes its 8 records into registers, and the
// results are parked at the OUTPUT's phase and leave as 16-byte nontemporal stores (store_staged) — the same LDS buffer twice.
__device__ __forceinline__ void load_staged(uint8_t* lds, uint32_t head, uint32_t nbytes, const uint8_t* g)
{
    const uint8_t* g0 = g - head;                            // 16-byte aligned
    const uint32_t end = head + nbytes;
    const uint32_t first_full = (head + 15u) >> 4, last_full = end >> 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t j = first_full + threadIdx.x; j < last_full; j += kBlockThreads)
        reinterpret_cast<u32x4*>(lds)[j] = reinterpret_cast<const u32x4*>(g0)[j];
    const uint32_t head_end = min(first_full << 4, end);     // ragged ends as 2-byte loads: never a byte outside [g, g + nbytes)
    for (uint32_t b = head + 2u * threadIdx.x; b < head_end; b += 2u * kBlockThreads)
        *reinterpret_cast<uint16_t*>(lds + b) = *reinterpret_cast<const uint16_t*>(g0 + b);
    if (last_full >= first_full) {
        const uint32_t tail_begin = max(last_full << 4, head_end);
        for (uint32_t b = tail_begin + 2u * threadIdx.x; b < end; b += 2u * kBlockThreads)
            *reinterpret_cast<uint16_t*>(lds + b) = *reinterpret_cast<const uint16_t*>(g0 + b);
    }
}

// stage_record read backwards: one record from a 2-byte aligned LDS offset with aligned accesses only.
__device__ __forceinline__ Record unstage_record(const uint8_t* lds, uint32_t off)
{
    const bool odd = (off & 2u) != 0u;
    const uint32_t h_off = odd ? off : off + 8u;
    const uint32_t a_off = odd ? off + 2u : off;
    const uint32_t h = *reinterpret_cast<const uint16_t*>(lds + h_off);
    const uint32_t a = *reinterpret_cast<const uint32_t*>(lds + a_off);
    const uint32_t b = *reinterpret_cast<const uint32_t*>(lds + a_off + 4u);
    Record r;
    r.xy = odd ? perm(a, h, kLoLo) : a;
    r.zc = odd ? perm(b, a, kHiLo) : b;
    r.b = odd ? (b >> 16) : h;
    return r;
}

__device__ __forceinline__ uint32_t retransform_mm(const float* __restrict__ Mr, float x, float y, float z)
{
    const float a = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Mr[0], x), __fmul_rn(Mr[1], y)), __fmul_rn(Mr[2], z)), Mr[3]);
    return (uint32_t)cvtt_x86(__fmul_rn(a, 1000.0f)) & 0xFFFFu;          // static_cast<short>: cvttss2si, low 16 bits
}

__device__ __forceinline__ Record retransform_record(const float* __restrict__ M, const Record& r)
{
    const float x = (float)(int32_t)(int16_t)(r.xy & 0xFFFFu) / 1000.0f;  // correctly rounded division (TU flag)
    const float y = (float)((int32_t)r.xy >> 16) / 1000.0f;
    const float z = (float)(int32_t)(int16_t)(r.zc & 0xFFFFu) / 1000.0f;
    Record o;
    o.xy = retransform_mm(M + 0, x, y, z) | (retransform_mm(M + 4, x, y, z) << 16);
    o.zc = retransform_mm(M + 8, x, y, z) | (r.zc & 0xFFFF0000u);
    o.b = r.b & 0xFFu;
    return o;
}

template <bool DS1>
__global__ __launch_bounds__(kBlockThreads)
void pcs_transform_payload_kernel(XformBatch xb)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    const XformCloud& C = xb.c[blockIdx.y];
    const uint32_t n_out = C.n_out;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n_out) return;
    const uint32_t pts = min(kTilePoints, n_out - tile0);
    uint8_t* gdst = C.out + (size_t)tile0 * PCS_POINT_BYTES;
    const uint32_t ohead = (uint32_t)((uintptr_t)gdst & 15u);
    Record rec[kPointsPerLane];
    if (DS1 && pts == kTilePoints && ohead == 0u &&
        (((uintptr_t)C.in + (size_t)tile0 * PCS_POINT_BYTES) & 15u) == 0u) {
        // FULL, 16-byte aligned tile (every tile but the last of a camera whose payload starts on a 16-byte boundary — what
        // hipMalloc and the stitched offsets of standard rasters give): a lane reads its own 8 consecutive records, 80 contiguous
        // bytes, as five 16-byte loads straight into registers (the a2 twin's reader: the L1 merges the wavefront's 64 x 5 pieces),
        // transforms them there and parks 5 x 16 bytes at LDS[lane * 80] (conflict-free); one LDS trip and one barrier instead
        // of two of each.
        const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(C.in) + (size_t)tile0 * PCS_POINT_BYTES) +
                         threadIdx.x * 5u;
        uint32_t w[20];
#pragma unroll
        for (int j = 0; j < 5; j++) { const uint4 v = q[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
#pragma unroll
        for (int k = 0; k < 8; k += 2) {                          // two records = five dwords: xy zc b|x' y'|z' c'|b'
            uint32_t* o = w + (k >> 1) * 5;
            Record a, b;
            a.xy = o[0]; a.zc = o[1]; a.b = o[2] & 0xFFFFu;
            b.xy = perm(o[3], o[2], kHiLo); b.zc = perm(o[4], o[3], kHiLo); b.b = o[4] >> 16;
            a = retransform_record(C.M, a);
            b = retransform_record(C.M, b);
            o[0] = a.xy; o[1] = a.zc;
            o[2] = perm(b.xy, a.b, kLoLo);
            o[3] = perm(b.zc, b.xy, kHiLo);
            o[4] = perm(b.b, b.zc, kHiLo);
        }
        uint4* mine = reinterpret_cast<uint4*>(stage) + threadIdx.x * 5u;
#pragma unroll
        for (int j = 0; j < 5; j++) mine[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
        __syncthreads();
        store_staged(stage, 0u, kTilePoints * PCS_POINT_BYTES, gdst);
        return;
    }
    if (DS1) {
        const uint8_t* gsrc = reinterpret_cast<const uint8_t*>(C.in) + (size_t)tile0 * PCS_POINT_BYTES;
        const uint32_t ihead = (uint32_t)((uintptr_t)gsrc & 15u);
        load_staged(stage, ihead, pts * PCS_POINT_BYTES, gsrc);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPointsPerLane; k++) {
            const uint32_t j = threadIdx.x + (uint32_t)k * kBlockThreads;
            if (j < pts) rec[k] = unstage_record(stage, ihead + j * PCS_POINT_BYTES);
        }
        __syncthreads();                                     // everybody holds its records: the buffer may be overwritten
    } else {
        const uint32_t ds = C.ds;
#pragma unroll
        for (int k = 0; k < kPointsPerLane; k++) {
            const uint32_t j = threadIdx.x + (uint32_t)k * kBlockThreads;
            if (j < pts) {
                const uint16_t* s = reinterpret_cast<const uint16_t*>(C.in) + (size_t)(tile0 + j) * ds * PCS_POINT_SHORTS;
                rec[k].xy = (uint32_t)s[0] | ((uint32_t)s[1] << 16);
                rec[k].zc = (uint32_t)s[2] | ((uint32_t)s[3] << 16);
                rec[k].b = s[4];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPointsPerLane; k++) {
        const uint32_t j = threadIdx.x + (uint32_t)k * kBlockThreads;
        if (j < pts) stage_record(stage, ohead + j * PCS_POINT_BYTES, retransform_record(C.M, rec[k]));
    }
    __syncthreads();
    store_staged(stage, ohead, pts * PCS_POINT_BYTES, gdst);
}

// ---- a7 with stride ----------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads)
void pcs_stitch_kernel(const uint16_t* __restrict__ src, uint32_t out_points, uint32_t ds, uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= out_points) return;
    const uint32_t pts = min(kTilePoints, out_points - tile0);
    uint8_t* gdst = dst + (size_t)tile0 * PCS_POINT_BYTES;
    const uint32_t head = (uint32_t)((uintptr_t)gdst & 15u);
    for (uint32_t j = threadIdx.x; j < pts; j += kBlockThreads) {
        const uint16_t* s = src + (size_t)(tile0 + j) * ds * PCS_POINT_SHORTS;
        uint16_t* o = reinterpret_cast<uint16_t*>(stage + head + j * PCS_POINT_BYTES);
#pragma unroll
        for (int k = 0; k < PCS_POINT_SHORTS; k++) o[k] = s[k];
    }
    __syncthreads();
    store_staged(stage, head, pts * PCS_POINT_BYTES, gdst);
}

// Certificate of CertRowConst for one stream: row r's colour row through the IEEE chain for EVERY Z16 value 1 .. 65 535 (one workgroup
// per raster row). crow[r] = the row for d = 1; *bad counts the (row, depth) pairs that give another one. The chain is the product's
// own code (deproject_pixel + color_coords + the exact conversion), so the sweep cannot disagree with what the kernels would compute.
__global__ __launch_bounds__(256)
void pcs_certify_color_row_kernel(const StreamParams* __restrict__ params, int stream, int32_t* __restrict__ crow,
                                  unsigned long long* __restrict__ bad)
{
    const StreamParams& P = params[stream];
    const uint32_t r = blockIdx.x;
    if (r >= (uint32_t)P.H) return;
    const float my = as_global(P.my)[r];
    auto row_of = [&](uint32_t d) {
        const PointIn p = deproject_pixel<false, false, IeeeMath>(P, d, 0.0f, my);
        float xf, yf;
        color_coords(P, p.u, p.v, xf, yf);
        ExactCvt ex;
        return ex.pixel(yf, P.cH - 1, P.c_hm1_f);
    };
    const int32_t want = row_of(1u);
    uint32_t local = 0;
    for (uint32_t d = 1u + threadIdx.x; d < 65536u; d += 256u) local += row_of(d) != want;
    if (local) atomicAdd(bad, (unsigned long long)local);
    if (threadIdx.x == 0) crow[r] = want;
}


// Device-side certificate for CertMath::div_const: over ALL 2^32 numerators a, the pixel coordinate that
// the pack derives from a quotient by the raster dimension c — clamp(cvttss2si(fma(a/c, c, 0.5)), 0, c-1)
// — is the same with Markstein's quotient as with the IEEE one. Run once per distinct dimension when a
// context is created (about a millisecond); any difference disables CertMath for that stream.
__global__ __launch_bounds__(256)
void pcs_verify_div_const_kernel(float c, float rc, int32_t dim, unsigned long long* __restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t local = 0;
    for (uint64_t bits = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < (1ull << 32); bits += stride) {
        const float a = __uint_as_float((uint32_t)bits);
        const float f1 = __fmaf_rn(IeeeMath::div_const(a, c, rc), c, 0.5f);
        const float f2 = __fmaf_rn(CertMath<false>::div_const(a, c, rc), c, 0.5f);
        const int32_t i1 = min(max(cvtt_x86(f1), 0), dim - 1);
        const int32_t i2 = min(max(cvtt_x86(f2), 0), dim - 1);
        // the lazy convert must also agree whenever it is the one used (f2 < 2^31 or NaN)
        const int32_t i3 = (f2 >= 2147483648.0f) ? i2 : min(max(cvt_sat(f2), 0), dim - 1);
        local += (i1 != i2) | (i2 != i3);
    }
    if (local) atomicAdd(bad, (unsigned long long)local);
}

#endif  // !PCS_TU_VOXEL

// The small-launch shape of the dense kernels (dense_tile<Src, 64>: one wavefront per 512-point tile) — an A/B knob, NOT taken by
// default. One 1280 x 720 stream is 450 workgroups of 2048 points on a chip that holds 1 792; four times as many one-wavefront
// workgroups were measured on it (round 6, rocprofv3 average over 3 000 cold launches): fused 6.95 -> 7.44 us, two streams 8.54 ->
// 8.30 us, the a2 twin 7.28 -> 7.77 us; hipEvent period of back-to-back launches 6.09 -> 6.22 us. The lone launch is a chain of
// dependent round trips (constants, Z16 + LUT, the colour gather, the store drain: ~5 us before the first stream's bytes count,
// ~2.3 us per further stream), which the tile size does not shorten. PCS_SMALL_TILES=1 forces it (the parity suite runs under it).
constexpr uint32_t kSmallThreads = 64, kSmallTilePoints = kSmallThreads * kPointsPerLane;
inline bool small_launch(uint32_t, int)
{
    const char* v = getenv("PCS_SMALL_TILES");      // (read at every call: bench.py times both shapes in one process)
    return v && v[0] == '1';
}

inline dim3 tile_grid(uint32_t max_points, int n_launch)
{
    return dim3((max_points + kTilePoints - 1) / kTilePoints, (unsigned)n_launch, 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------

#if !PCS_TU_VOXEL
hipError_t launch_fused_dense(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                              bool any_ddist, bool any_cdist, MathSel math, const FramePtrs& fp, int16_t* d_payload,
                              hipStream_t st)
{
    if (n_launch <= 0 || max_points == 0) return hipSuccess;
    // (PCS_SMALL_TILES=1: one wavefront per 512-point tile — measured no faster on the launches it was meant for, see small_launch)
    const bool small = small_launch(max_points, n_launch);
    const dim3 grid = small ? dim3((max_points + kSmallTilePoints - 1) / kSmallTilePoints, (unsigned)n_launch, 1) : tile_grid(max_points, n_launch);
#define L(DD, CD, M) do { if (small) hipLaunchKernelGGL((pcs_fused_dense_kernel<DD, CD, M, kSmallThreads>), grid, dim3(kSmallThreads), 0, st, \
                                        d_params, stream0, fp, reinterpret_cast<uint8_t*>(d_payload)); \
                          else hipLaunchKernelGGL((pcs_fused_dense_kernel<DD, CD, M>), grid, dim3(kBlockThreads), 0, st, \
                                        d_params, stream0, fp, reinterpret_cast<uint8_t*>(d_payload)); } while (0)
    if (math != MathSel::Ieee && !any_ddist) {
        const bool ident = (math == MathSel::CertIdentR || math == MathSel::CertIdentRNoOvf);
        const bool noovf = (math == MathSel::CertNoOvf || math == MathSel::CertIdentRNoOvf) && !any_cdist;
        if (noovf)      { if (ident) L(false, false, CertIdentNoOvf); else L(false, false, CertNoOvf); }
        else if (ident) { if (any_cdist) L(false, true, CertMath<true>); else L(false, false, CertMath<true>); }
        else            { if (any_cdist) L(false, true, CertMath<false>); else L(false, false, CertMath<false>); }
    } else {
        if (any_ddist) { if (any_cdist) L(true, true, IeeeMath); else L(true, false, IeeeMath); }
        else           { if (any_cdist) L(false, true, IeeeMath); else L(false, false, IeeeMath); }
    }
#undef L
    return hipGetLastError();
}

hipError_t launch_fused_count(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                              uint32_t flags, const FramePtrs& fp, uint32_t* d_tile_counts, hipStream_t st)
{
    if (n_launch <= 0 || max_points == 0) return hipSuccess;
    const uint32_t tiles = (max_points + kTilePoints - 1) / kTilePoints;
    const dim3 grid((tiles + kCountTiles - 1) / kCountTiles, (unsigned)n_launch, 1);
    // the predicate depends on depth-side distortion only through x; always run the general form
    hipLaunchKernelGGL((pcs_fused_count_kernel<true, false>), grid, dim3(kBlockThreads), 0, st,
                       d_params, stream0, fp, flags, d_tile_counts);
    return hipGetLastError();
}

hipError_t launch_scan(const StreamParams* d_params, int n_streams, int downsample,
                       const uint32_t* d_tile_counts, uint32_t* d_tile_prefix, uint32_t* d_stream_kept,
                       int32_t* d_counts, uint32_t* d_arrive, hipStream_t st)
{
    hipLaunchKernelGGL(pcs_scan_kernel, dim3((unsigned)n_streams), dim3(1024), 0, st, d_params, 0, n_streams, 0u,
                       (uint32_t)downsample, d_tile_counts, d_tile_prefix, d_stream_kept, d_counts, d_arrive);
    return hipGetLastError();
}

hipError_t launch_fused_emit(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                             uint32_t flags, int downsample, MathSel math, const FramePtrs& fp,
                             const uint32_t* d_tile_prefix, const uint32_t* d_stream_kept,
                             int16_t* d_payload, int32_t* d_total_out, int n_total_streams, hipStream_t st)
{
    if (n_launch <= 0 || max_points == 0) return hipSuccess;
    const dim3 grid = tile_grid(max_points, n_launch);
    const bool pred = (flags & (PCS_FLAG_CUTOFF | PCS_FLAG_DROP_INVALID)) != 0;
    const bool ds1 = downsample == 1;
    uint8_t* out = reinterpret_cast<uint8_t*>(d_payload);
#define L(PR, D1, M) hipLaunchKernelGGL((pcs_fused_emit_kernel<PR, D1, M>), grid, dim3(kBlockThreads), 0, st, d_params, \
                                        stream0, fp, flags, (uint32_t)downsample, d_tile_prefix, d_stream_kept, out, \
                                        d_total_out, n_total_streams)
#define LM(M) do { if (pred) { if (ds1) L(true, true, M); else L(true, false, M); } \
                   else      { if (ds1) L(false, true, M); else L(false, false, M); } } while (0)
    const bool ident = (math == MathSel::CertIdentR || math == MathSel::CertIdentRNoOvf);
    if (math == MathSel::Ieee) LM(IeeeMath); else if (ident) LM(CertMath<true>); else LM(CertMath<false>);
#undef LM
#undef L
    return hipGetLastError();
}

hipError_t launch_fused_compact(const StreamParams* d_params, int stream0, int n_launch, uint32_t launch_tiles,
                                MathSel math, const FramePtrs& fp, const CompactLaunch& cl, int16_t* d_payload,
                                hipStream_t st)
{
    if (n_launch <= 0 || launch_tiles == 0) return hipSuccess;
    CompactArgs a;
    a.ticket = cl.d_ticket; a.ticket_base = cl.ticket_base; a.desc = cl.d_desc; a.stream_desc = cl.d_stream_desc;
    a.stream_end = cl.d_stream_end;
    a.chain_in = cl.d_chain_in; a.error = cl.d_error; a.gen = cl.gen & 0x3FFFFFFFu; a.flags = cl.flags;
    a.counts = cl.d_counts; a.n_total = cl.n_total; a.last_launch = cl.last_launch;
    uint8_t* out = reinterpret_cast<uint8_t*>(d_payload);
    if (math != MathSel::Ieee)
        hipLaunchKernelGGL((pcs_fused_compact_kernel<CertMath<false>>), dim3(launch_tiles), dim3(kBlockThreads), 0, st,
                           d_params, stream0, n_launch, fp, a, out);
    else
        hipLaunchKernelGGL((pcs_fused_compact_kernel<IeeeMath>), dim3(launch_tiles), dim3(kBlockThreads), 0, st,
                           d_params, stream0, n_launch, fp, a, out);
    return hipGetLastError();
}

hipError_t launch_fused_dense_batch(const StreamParams* d_params, int n_streams, int n_sets, uint32_t max_points,
                                    bool any_ddist, bool any_cdist, MathSel math, const BatchPtrs& bp, hipStream_t st)
{
    if (n_streams <= 0 || n_sets <= 0 || max_points == 0) return hipSuccess;
    if (n_streams * n_sets > kBatchEntries || n_sets > kBatchSets) return hipErrorInvalidValue;
    const dim3 grid((max_points + kTilePoints - 1) / kTilePoints, (unsigned)n_streams, (unsigned)n_sets);
#define L(DD, CD, M) hipLaunchKernelGGL((pcs_fused_dense_batch_kernel<DD, CD, M>), grid, dim3(kBlockThreads), 0, st, d_params, bp)
    if (math != MathSel::Ieee && !any_ddist) {
        const bool ident = (math == MathSel::CertIdentR || math == MathSel::CertIdentRNoOvf);
        const bool noovf = (math == MathSel::CertNoOvf || math == MathSel::CertIdentRNoOvf) && !any_cdist;
        if (noovf)      { if (ident) L(false, false, CertIdentNoOvf); else L(false, false, CertNoOvf); }
        else if (ident) { if (any_cdist) L(false, true, CertMath<true>); else L(false, false, CertMath<true>); }
        else            { if (any_cdist) L(false, true, CertMath<false>); else L(false, false, CertMath<false>); }
    } else {
        if (any_ddist) { if (any_cdist) L(true, true, IeeeMath); else L(true, false, IeeeMath); }
        else           { if (any_cdist) L(false, true, IeeeMath); else L(false, false, IeeeMath); }
    }
#undef L
    return hipGetLastError();
}

hipError_t launch_compact_batch(const StreamParams* d_params, int n_streams, int n_sets, uint32_t max_points,
                                uint32_t total_tiles, uint32_t flags, MathSel math, const BatchPtrs& bp,
                                const BatchCounts& bc, uint32_t* d_tile_counts, uint32_t* d_tile_prefix,
                                uint32_t* d_stream_kept, hipStream_t st)
{
    if (n_streams <= 0 || n_sets <= 0 || max_points == 0) return hipSuccess;
    if (n_streams * n_sets > kBatchEntries || n_sets > kBatchSets) return hipErrorInvalidValue;
    const uint32_t tiles = (max_points + kTilePoints - 1) / kTilePoints;
    hipLaunchKernelGGL((pcs_fused_count_batch_kernel<true, false>),
                       dim3((tiles + kCountTiles - 1) / kCountTiles, (unsigned)n_streams, (unsigned)n_sets),
                       dim3(kBlockThreads), 0, st, d_params, bp, flags, d_tile_counts, total_tiles);
    hipLaunchKernelGGL(pcs_scan_batch_kernel, dim3((unsigned)n_streams, (unsigned)n_sets), dim3(1024), 0, st, d_params,
                       d_tile_counts, d_tile_prefix, d_stream_kept, total_tiles, bc);
    const dim3 grid(tiles, (unsigned)n_streams, (unsigned)n_sets);
#define L(M) hipLaunchKernelGGL((pcs_fused_emit_batch_kernel<M>), grid, dim3(kBlockThreads), 0, st, d_params, bp, flags, \
                                d_tile_prefix, d_stream_kept, total_tiles, bc)
    const bool ident = (math == MathSel::CertIdentR || math == MathSel::CertIdentRNoOvf);
    if (math == MathSel::Ieee) L(IeeeMath); else if (ident) L(CertMath<true>); else L(CertMath<false>);
#undef L
    return hipGetLastError();
}

#endif  // !PCS_TU_VOXEL

#if PCS_TU_VOXEL
hipError_t launch_fused_voxel_partials(const StreamParams* d_params, int stream0, int n_launch, uint32_t max_points,
                                       uint32_t max_w, uint32_t max_h, bool patch_ok, bool any_dist, uint32_t flags,
                                       MathSel math, const FramePtrs& fp, const VoxelStage& vs, hipStream_t st)
{
    if (n_launch <= 0 || max_points == 0) return hipSuccess;
    static const int env_rounds = [] { const char* v = getenv("PCS_VOXEL_ROUNDS"); return v ? atoi(v) : 0; }();
    static const int env_patch = [] { const char* v = getenv("PCS_VOXEL_PATCH"); return v ? atoi(v) : 1; }();
    // Rounds (4096 pixels each) that share one 2048-slot table. More rounds = fewer partials for the sort, as long as
    // the voxels under one table stay below its slots: a leaf spans leaf / (depth / focal) pixels, so the voxels per
    // round fall roughly with the square of the leaf. A wrong guess costs speed, never correctness (runs that find no
    // slot go out as partials of their own). Capped so that the launch still fills the chip twice over. The packed sums
    // of the table hold at most 8 rounds.
    // (`rounds` below counts SQUARES of 4096 pixels, the unit these measurements were taken in; a workgroup of kVoxThreads lanes
    // covers one in kSub rounds of kVoxRoundPoints pixels — converted just before the launch)
    constexpr int kSub = 4096 / (int)kVoxRoundPoints;
    const uint64_t launch_tiles = (uint64_t)((max_points + 4095u) / 4096u) * (uint64_t)n_launch;
    const uint64_t fill_cap = std::max<uint64_t>(1, (launch_tiles + launch_tiles / 32) / 1024);   // (3 % slack: 16 x 1080p = 8112 squares, 8 per table still fill the chip twice)
    int rounds, rx = 0;
    if (patch_ok && env_patch) {
        // square patches: 64 x 64 per round, (rx x ry) rounds per workgroup. 16 x 1080p, ms per frame-set with 1 / 2 / 4 /
        // 8 rounds: 10 mm 1.64 / 1.84 / - / -, 25 mm 0.52 / 0.53 / - / -, 36 mm 0.33 / 0.32 / 0.34 / 0.46, 50 mm 0.30 / 0.26 /
        // 0.28 / 0.33, 100 mm 0.26 / 0.23 / 0.23 / 0.25, 200 mm 0.26 / 0.23 / 0.23 / 0.24: two squares (128 x 64) per table
        // from 30 mm up, one below. (Voxels per 128 x 64 patch on the synthetic scene: 780 / 240 / 80 / 30 at 25 / 50 /
        // 100 / 200 mm.)
        // Re-measured after the workgroups were re-ordered (VoxTiling: the chip's last round no longer waits for full patches
        // behind short ones, which is what had made more squares per table lose), 16 x 1080p, ms per call with 2 / 4 / 8 squares:
        //   warm bucket tail   36 mm 0.230 / 0.236 / -, 40 mm 0.217 / 0.210 / -, 45 mm 0.199 / 0.183 / 0.250, 50 mm 0.187 / 0.174 / 0.216,
        //                      100 mm 0.160 / 0.144 / 0.159, 150 mm 0.149 / 0.138 / 0.135, 200 mm 0.147 / 0.136 / 0.135, 300 mm 0.147 / 0.137 / 0.133
        //   its cold chain     40 mm 0.243 / 0.246 / 0.309, 50 mm 0.197 / 0.193 / 0.221, 100 mm 0.167 / 0.158 / 0.163, 200 mm 0.154 / 0.148 / 0.149
        //   LSD tail           30 mm 0.378 / 0.402, 36 mm 0.269 / 0.277, 50 mm 0.227 / 0.222, 100 mm 0.193 / 0.183, 200 mm 0.176 / 0.169
        // (25 mm: one square 0.456, two 0.453; 30 mm: 0.404 / 0.375.) A warm call ends every workgroup with a dearer flush (bucket
        // search, a returning add per bucket it touches, scattered writes), so it gains most from fewer, larger tables.
        const VoxPatchShape shape = vox_patch_shape(vs.leaf, vs.regions != 0u, launch_tiles, env_rounds);      // pcs_vox_tiling.h
        rounds = shape.squares; rx = shape.rx;
    } else {
        const uint64_t by_leaf = std::min<uint64_t>(8, std::max<uint64_t>(3, ((uint64_t)vs.leaf * vs.leaf) / 625u));
        rounds = (int)std::min<uint64_t>(by_leaf, fill_cap);
        if (env_rounds > 0) rounds = std::min(env_rounds, 8);
    }
    rounds *= kSub;
    dim3 grid;
    VoxTiling tl{};
    if (rx) {
        // Two tiers (VoxTiling): the head in (rx x ry) patches; the square-rows that do not fill a head patch (1080 rows = 16
        // square-rows + 56 rows: one of 17 with ry = 2) in (rxb x 1) patches, dealt after all the head patches. Moving MORE of
        // the raster into the tail of small items does not pay — every workgroup costs a table clear and a flush — 16 x 1080p,
        // one warm call, ms with 0 / 25 / 40 / 60 / 100 % of the square-rows in the tail (one box): 50 mm 0.174 / 0.176 / 0.178 /
        // 0.183 / 0.187, 100 mm 0.146 / 0.149 / 0.151 / 0.155 / 0.161; PCS_VOXEL_TAILPCT keeps the knob for the lab.
        static const int env_tail = [] { const char* v = getenv("PCS_VOXEL_TAILPCT"); return v ? atoi(v) : 0; }();
        tl = vox_tiling_make(max_w, max_h, kVoxRows, (unsigned)rx, (unsigned)rounds / (unsigned)rx, env_tail);
        grid = dim3((unsigned)(tl.na + tl.nb), (unsigned)n_launch, 1);
    } else {
        const uint32_t tile_points = kVoxRoundPoints * (uint32_t)rounds;
        grid = dim3((max_points + tile_points - 1) / tile_points, (unsigned)n_launch, 1);
    }
    // no stream of the context has a distortion model (or the half-pixel texture convention): the instantiation without
    // their (uniform, but not free in a VALU-bound kernel) tests
#define L(D, M) hipLaunchKernelGGL((pcs_fused_voxel_partials_kernel<D, D, M>), grid, dim3(kVoxThreads), 0, st, d_params, stream0, fp, flags, vs, rounds, tl, vs.leaf < 30u ? 1 : 0)
    const bool ident = (math == MathSel::CertIdentR || math == MathSel::CertIdentRNoOvf || math == MathSel::CertRowConst);
    if (math == MathSel::Ieee) L(true, IeeeMath);
    else if (math == MathSel::CertRowConst && !any_dist) L(false, CertRowConst);
    else if (any_dist) { if (ident) L(true, CertMath<true>); else L(true, CertMath<false>); }
    else               { if (ident) L(false, CertMath<true>); else L(false, CertMath<false>); }
#undef L
    return hipGetLastError();
}

hipError_t launch_payload_voxel_partials(const int16_t* d_payload, uint32_t n_points, const int32_t* d_n_points,
                                         const VoxelStage& vs, hipStream_t st)
{
    if (n_points == 0) return hipSuccess;
    // Rounds of 4096 consecutive records per table (the stitched order is all a payload offers: no square patches).
    // 29.8 M-point config-5 cloud, ms for the voxel grid with 2 / 3 / 4 / 6 rounds: 36 mm 0.48 / 0.40 / 0.46 / 0.60,
    // 50 mm 0.35 / 0.31 / 0.26 / 0.26, 100 mm 0.22 / 0.19 / 0.18 / 0.17 (the 1024-lane reader of pcs_voxel.hip: 0.52 / 0.36 /
    // 0.25). Below 30 mm the tables are crowded and runs are passed through per workgroup (vox_table_round): 10 mm 1.97 ms
    // with 2 rounds (that reader: 2.43), 15 mm 1.35 (1.56), 25 mm 0.72 with 3 rounds (0.81).
    static const int env_rounds = [] { const char* v = getenv("PCS_VOXEL_ROUNDS"); return v ? atoi(v) : 0; }();
    const uint64_t tiles = (n_points + 4095u) / 4096u;                   // (rounds in units of 4096 records, as measured)
    const uint64_t by_leaf = vs.leaf >= 80 ? 6 : vs.leaf >= 44 ? 4 : vs.leaf >= 23 ? 3 : 2;
    int rounds = (int)std::min<uint64_t>(by_leaf, std::max<uint64_t>(1, tiles / 1024));
    if (env_rounds > 0) rounds = std::min(env_rounds, 8);
    rounds *= 4096 / (int)kVoxRoundPoints;
    const uint32_t tile_points = kVoxRoundPoints * (uint32_t)rounds;
    hipLaunchKernelGGL(pcs_payload_voxel_partials_kernel, dim3((n_points + tile_points - 1) / tile_points), dim3(kVoxThreads), 0, st,
                       d_payload, n_points, d_n_points, vs, rounds, vs.leaf < 30u ? 1 : 0);
    return hipGetLastError();
}

#endif  // PCS_TU_VOXEL

#if !PCS_TU_VOXEL
hipError_t launch_pack_batch(const StreamParams* d_params, const PackBatch& pb, int n, uint32_t max_points, bool aligned,
                             hipStream_t st)
{
    if (n <= 0 || max_points == 0) return hipSuccess;
    if (n > kPackBatch) return hipErrorInvalidValue;
    const dim3 grid = tile_grid(max_points, n);
    if (aligned) hipLaunchKernelGGL((pcs_pack_batch_kernel<true>), grid, dim3(kBlockThreads), 0, st, d_params, pb);
    else         hipLaunchKernelGGL((pcs_pack_batch_kernel<false>), grid, dim3(kBlockThreads), 0, st, d_params, pb);
    return hipGetLastError();
}

hipError_t launch_certify_color_row(const StreamParams* d_params, int stream, int rows, int32_t* d_crow, unsigned long long* d_bad, hipStream_t st)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(pcs_certify_color_row_kernel, dim3((unsigned)rows), dim3(256), 0, st, d_params, stream, d_crow, d_bad);
    return hipGetLastError();
}

hipError_t launch_verify_div_const(float c, float rc, int32_t dim, unsigned long long* d_bad, hipStream_t st)
{
    hipLaunchKernelGGL(pcs_verify_div_const_kernel, dim3(8192), dim3(256), 0, st, c, rc, dim, d_bad);
    return hipGetLastError();
}

hipError_t launch_pack_dense(const StreamParams* d_params, int stream, const VertexPtrs& vp,
                             int16_t* d_out, hipStream_t st)
{
    if (vp.n_points == 0) return hipSuccess;
    if (small_launch(vp.n_points, 1))       // (A/B knob only)
        hipLaunchKernelGGL((pcs_pack_dense_kernel<kSmallThreads>), dim3((vp.n_points + kSmallTilePoints - 1) / kSmallTilePoints),
                           dim3(kSmallThreads), 0, st, d_params, stream, vp, reinterpret_cast<uint8_t*>(d_out));
    else
        hipLaunchKernelGGL((pcs_pack_dense_kernel<kBlockThreads>), tile_grid(vp.n_points, 1), dim3(kBlockThreads), 0, st,
                           d_params, stream, vp, reinterpret_cast<uint8_t*>(d_out));
    return hipGetLastError();
}

hipError_t launch_pack_count(const StreamParams* d_params, int stream, const VertexPtrs& vp, uint32_t flags,
                             uint32_t* d_tile_counts, hipStream_t st)
{
    if (vp.n_points == 0) return hipSuccess;
    hipLaunchKernelGGL(pcs_pack_count_kernel, tile_grid(vp.n_points, 1), dim3(kBlockThreads), 0, st,
                       d_params, stream, vp, flags, d_tile_counts);
    return hipGetLastError();
}

hipError_t launch_pack_scan(uint32_t n_tiles, const uint32_t* d_tile_counts, uint32_t* d_tile_prefix,
                            int32_t* d_out_points, uint32_t* d_arrive, hipStream_t st)
{
    // override_n makes the scan kernel ignore the params table (one segment of n_tiles tiles at index 0);
    // counts[0] = kept, counts[1] = total — d_out_points must hold 2 ints.
    hipLaunchKernelGGL(pcs_scan_kernel, dim3(1), dim3(1024), 0, st, (const StreamParams*)nullptr, 0, 1,
                       n_tiles * kTilePoints, 1u, d_tile_counts, d_tile_prefix, (uint32_t*)nullptr, d_out_points, d_arrive);
    return hipGetLastError();
}

hipError_t launch_pack_emit(const StreamParams* d_params, int stream, const VertexPtrs& vp, uint32_t flags,
                            const uint32_t* d_tile_prefix, int16_t* d_out, hipStream_t st)
{
    if (vp.n_points == 0) return hipSuccess;
    const bool pred = (flags & (PCS_FLAG_CUTOFF | PCS_FLAG_DROP_INVALID)) != 0;
    uint8_t* out = reinterpret_cast<uint8_t*>(d_out);
    if (pred)
        hipLaunchKernelGGL((pcs_pack_emit_kernel<true>), tile_grid(vp.n_points, 1), dim3(kBlockThreads), 0, st,
                           d_params, stream, vp, flags, d_tile_prefix, out);
    else
        hipLaunchKernelGGL((pcs_pack_emit_kernel<false>), tile_grid(vp.n_points, 1), dim3(kBlockThreads), 0, st,
                           d_params, stream, vp, flags, d_tile_prefix, out);
    return hipGetLastError();
}

hipError_t launch_deproject(const StreamParams* d_params, int stream, uint32_t n_points,
                            const uint16_t* d_depth, float* d_vertices, float* d_texcoords, hipStream_t st)
{
    if (n_points == 0) return hipSuccess;
    const dim3 grid((n_points + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL((pcs_deproject_kernel<true, true>), grid, dim3(kBlockThreads), 0, st,
                       d_params, stream, d_depth, d_vertices, d_texcoords);
    return hipGetLastError();
}

hipError_t launch_transform_payloads(const XformBatch& xb, int n, uint32_t max_out, hipStream_t st)
{
    if (n <= 0 || max_out == 0) return hipSuccess;
    if (n > kXformBatch) return hipErrorInvalidValue;
    bool ds1 = true;
    for (int i = 0; i < n; i++) ds1 = ds1 && xb.c[i].ds == 1u;
    const dim3 grid = tile_grid(max_out, n);
    if (ds1) hipLaunchKernelGGL((pcs_transform_payload_kernel<true>), grid, dim3(kBlockThreads), 0, st, xb);
    else     hipLaunchKernelGGL((pcs_transform_payload_kernel<false>), grid, dim3(kBlockThreads), 0, st, xb);
    return hipGetLastError();
}

hipError_t launch_stitch(const int16_t* d_src, uint32_t src_points, int downsample,
                         int16_t* d_dst, hipStream_t st)
{
    const uint32_t ds = downsample < 1 ? 1u : (uint32_t)downsample;
    const uint32_t out_points = (src_points + ds - 1) / ds;
    if (out_points == 0) return hipSuccess;
    hipLaunchKernelGGL(pcs_stitch_kernel, tile_grid(out_points, 1), dim3(kBlockThreads), 0, st,
                       reinterpret_cast<const uint16_t*>(d_src), out_points, ds, reinterpret_cast<uint8_t*>(d_dst));
    return hipGetLastError();
}

#endif  // !PCS_TU_VOXEL

}  // namespace pcs
