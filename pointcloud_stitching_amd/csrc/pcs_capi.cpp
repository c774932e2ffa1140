// pcs_capi.cpp — the C ABI of libpcs_hip.so (include/pcs_hip.h) over the gfx950 kernels.
//
// Host-side twin of the reference's seam: sendXYZRGBPointcloud / copyPointCloudXYZRGBToBufferSIMD
// (src/pcs-camera-optimized.cpp:669-723, 363-616) and sendStitchToUnity's concatenate
// (src/pcs-multicamera-client.cpp:373-395). No CPU compute fallback lives here: if HIP is not usable
// pcs_create fails and nothing else can be called.

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "pcs_host.h"

using namespace pcs_host;

namespace pcs_host {

thread_local std::string g_create_err;

int fail(pcs_ctx* c, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return status;
}

}  // namespace pcs_host

namespace {

bool coeffs_nonzero(const pcs_intrinsics& in)
{
    for (int k = 0; k < 5; k++) if (in.coeffs[k] != 0.0f) return true;
    return false;
}

int validate_stream(const pcs_stream_config& s, int idx)
{
    const auto bad = [&](const char* what) {
        return fail(nullptr, PCS_ERR_INVALID_ARG, "stream %d: %s", idx, what);
    };
    if (s.depth.width <= 0 || s.depth.height <= 0) return bad("depth width/height must be positive");
    if (s.color.width <= 0 || s.color.height <= 0) return bad("colour width/height must be positive");
    if (s.depth.width >= (1 << 24) || s.depth.height >= (1 << 24) ||
        s.color.width >= (1 << 24) || s.color.height >= (1 << 24)) return bad("raster dimension too large");
    if ((uint64_t)s.depth.width * (uint64_t)s.depth.height > 0x7FFFFFF8ull) return bad("too many depth pixels");
    if (s.color_bpp < 3)
        return fail(nullptr, PCS_ERR_UNSUPPORTED, "stream %d: colour bytes-per-pixel %d < 3 (the pack reads bytes "
                    "idx, idx+1, idx+2 of a pixel)", idx, s.color_bpp);
    if (s.color_bpp > 16) return bad("colour bytes-per-pixel too large");
    if ((int64_t)s.color_stride < (int64_t)s.color_bpp * s.color.width) return bad("colour stride < bpp*width");
    if (s.color_stride >= (1 << 24)) return bad("colour stride too large");
    if ((uint64_t)s.color_stride * (uint64_t)s.color.height > 0xFFFFFFF0ull) return bad("colour raster too large");
    if ((uint64_t)s.color_stride * (uint64_t)s.color.height < 4) return bad("colour raster smaller than 4 bytes");
    if (!(s.depth.fx != 0.0f) || !(s.depth.fy != 0.0f)) return bad("depth fx/fy must be non-zero");
    // Distortion coverage: a model with all-zero coefficients is the identity.
    if (coeffs_nonzero(s.depth) && s.depth.model != PCS_DISTORTION_INVERSE_BROWN_CONRADY)
        return fail(nullptr, PCS_ERR_UNSUPPORTED, "stream %d: depth distortion model %d with non-zero "
                    "coefficients is not covered (only INVERSE_BROWN_CONRADY)", idx, s.depth.model);
    if (coeffs_nonzero(s.color) && s.color.model != PCS_DISTORTION_MODIFIED_BROWN_CONRADY &&
        s.color.model != PCS_DISTORTION_INVERSE_BROWN_CONRADY)
        return fail(nullptr, PCS_ERR_UNSUPPORTED, "stream %d: colour distortion model %d with non-zero "
                    "coefficients is not covered (only MODIFIED/INVERSE_BROWN_CONRADY)", idx, s.color.model);
    return PCS_OK;
}

void fill_params(const pcs_stream_config& s, StreamParams& p)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) p.M[4 * r + c] = s.cam_to_world[4 * r + c];
    for (int k = 0; k < 9; k++) p.R[k] = s.depth_to_color.rotation[k];
    for (int k = 0; k < 3; k++) p.t[k] = s.depth_to_color.translation[k];
    p.depth_scale = s.depth_scale;
    p.d_ppx = s.depth.ppx; p.d_ppy = s.depth.ppy; p.d_fx = s.depth.fx; p.d_fy = s.depth.fy;
    p.c_fx = s.color.fx; p.c_fy = s.color.fy; p.c_ppx = s.color.ppx; p.c_ppy = s.color.ppy;
    p.c_w_f = (float)s.color.width; p.c_h_f = (float)s.color.height;
    p.c_wm1_f = (float)(s.color.width - 1); p.c_hm1_f = (float)(s.color.height - 1);
    p.c_rw = (float)(1.0 / (double)p.c_w_f); p.c_rh = (float)(1.0 / (double)p.c_h_f);
    for (int k = 0; k < 5; k++) { p.dk[k] = s.depth.coeffs[k]; p.ck[k] = s.color.coeffs[k]; }
    p.W = s.depth.width; p.H = s.depth.height;
    p.cW = s.color.width; p.cH = s.color.height;
    p.bpp = s.color_bpp; p.stride = s.color_stride;
    p.color_bytes = (uint32_t)((uint64_t)s.color_stride * (uint64_t)s.color.height);
    p.n_points = (uint32_t)s.depth.width * (uint32_t)s.depth.height;
    p.z_zero_iff_d_zero = (std::isfinite(s.depth_scale) && (s.depth_scale * 1.0f) != 0.0f) ? 1 : 0;
    p.ddist = (s.depth.model != PCS_DISTORTION_NONE && coeffs_nonzero(s.depth)) ? 1 : 0;
    p.cdist = (s.color.model != PCS_DISTORTION_NONE && coeffs_nonzero(s.color)) ? 1 : 0;
}


// ------------------------------------------------------------------------------------------------
// Certification of the reduced-instruction arithmetic (CertMath in pcs_kernels.hip) for one stream.
//
// CertMath::div2 drops v_div_scale / v_div_fmas / v_div_fixup from the IEEE division expansion. That is
// the same arithmetic as long as no operand would have been rescaled, i.e. for every VALID pixel
// (depth d in 1..65535):  2^-40 <= |P2| < 2^30,  |P0|,|P1| < 2^30,  and P0,P1 are either exactly zero or
// >= 2^-70 in magnitude (then every quotient is zero or in [2^-100, 2^70], the exact remainders of the
// fused corrections are representable, and 1/P2 is normal). Everything is a conservative bound computed
// from the configuration; if any bound fails, or anything is non-finite, the stream keeps IeeeMath.
//
//   Z = depth_scale*d in [zmin, zmax];  |X| <= Z*mxmax,  |Y| <= Z*mymax   (LUT maxima, inflated by 2^-20)
//   P_i = fl(fl(fl(R_i0 X + R_i3 Y) + R_i6 Z) + t_i)
//   upper:   |P_i| <= (|R_i0| mxmax + |R_i3| mymax + |R_i6|) zmax + |t_i|                 (+ rounding slack)
//   P2 low:  P2 >= (R8 - |R2| mxmax - |R5| mymax) Z + t2 - 2^-21 (sigma Z + |t2|),  minimised at Z = zmin
//   P_i nonzero low: a float sum is an integer multiple of the smallest ulp among its addends, so a
//            non-zero P_i is >= 2^-24 * (smallest non-zero addend magnitude), with addends
//            |R_i0| xmin, |R_i3| ymin, |R_i6| zmin, |t_i| (xmin = zmin * smallest non-zero |mx|, ...).
// ------------------------------------------------------------------------------------------------

Certificate certify_stream(const pcs_stream_config& s, const std::vector<float>& mx, const std::vector<float>& my)
{
    Certificate c;
    const auto fin = [](double v) { return std::isfinite(v); };
    const float* R = s.depth_to_color.rotation;
    const float* t = s.depth_to_color.translation;
    for (int k = 0; k < 9; k++) if (!fin(R[k])) return c;
    for (int k = 0; k < 3; k++) if (!fin(t[k])) return c;
    if (!fin(s.depth_scale) || !(s.depth_scale > 0.0f)) return c;
    if (s.depth.model != PCS_DISTORTION_NONE && coeffs_nonzero(s.depth)) return c;   // rays not separable
    if (!fin(s.color.fx) || !fin(s.color.fy) || !fin(s.color.ppx) || !fin(s.color.ppy)) return c;
    for (int k = 0; k < 5; k++) if (!fin(s.color.coeffs[k])) return c;

    double mxmax = 0, mymax = 0, mxmin = INFINITY, mymin = INFINITY;   // min over non-zero magnitudes
    for (float v : mx) { if (!fin(v)) return c; const double a = std::fabs((double)v); mxmax = std::max(mxmax, a); if (a > 0) mxmin = std::min(mxmin, a); }
    for (float v : my) { if (!fin(v)) return c; const double a = std::fabs((double)v); mymax = std::max(mymax, a); if (a > 0) mymin = std::min(mymin, a); }
    const double infl = 1.0 + std::ldexp(1.0, -20);
    mxmax *= infl; mymax *= infl;
    const double zmin = (double)s.depth_scale * (1.0 - std::ldexp(1.0, -22));
    const double zmax = (double)s.depth_scale * 65535.0 * infl;
    if (!(zmin >= std::ldexp(1.0, -40)) || !(zmax < std::ldexp(1.0, 28))) return c;
    const double xmin = zmin * mxmin * (1.0 - std::ldexp(1.0, -22));
    const double ymin = zmin * mymin * (1.0 - std::ldexp(1.0, -22));

    const double lim_hi = std::ldexp(1.0, 30), lim_num = std::ldexp(1.0, -70), lim_den = std::ldexp(1.0, -40);
    for (int i = 0; i < 3; i++) {
        const double r0 = std::fabs((double)R[i]), r3 = std::fabs((double)R[i + 3]), r6 = std::fabs((double)R[i + 6]);
        const double ti = std::fabs((double)t[i]);
        const double sigma = r0 * mxmax + r3 * mymax + r6;
        if (!(sigma * zmax + ti < lim_hi * 0.5)) return c;
        c.a_max[i] = (sigma * zmax + ti) * infl;
        // smallest non-zero addend (an addend whose coefficient is zero is an exact zero and drops out)
        double small = INFINITY;
        if (r0 > 0 && std::isfinite(xmin)) small = std::min(small, r0 * xmin);
        if (r3 > 0 && std::isfinite(ymin)) small = std::min(small, r3 * ymin);
        if (r6 > 0) small = std::min(small, r6 * zmin);
        if (ti > 0) small = std::min(small, ti);
        if (std::isfinite(small)) {
            if (!(small * (1.0 - std::ldexp(1.0, -20)) >= std::ldexp(1.0, -46))) return c;   // no product underflows
            if (i < 2 && !(std::ldexp(small, -24) >= lim_num)) return c;
        }
        if (i == 2) {
            const double kappa = (double)R[8] - std::fabs((double)R[2]) * mxmax - std::fabs((double)R[5]) * mymax;
            const double slack = std::ldexp(1.0, -21);
            const double kp = kappa - slack * sigma;
            if (!(kp > 0)) return c;
            const double low = kp * zmin + (double)t[2] - slack * ti;
            if (!(low >= lim_den)) return c;
            c.p2_low = low;
        }
    }
    c.fast = true;
    c.xb = zmax * mxmax; c.yb = zmax * mymax; c.zb = zmax;
    // identity shortcut: R == I exactly, translation entries are not negative zero
    static const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool ident = true;
    for (int k = 0; k < 9; k++) {
        uint32_t a, b; std::memcpy(&a, &R[k], 4); std::memcpy(&b, &I9[k], 4);
        if (a != b) ident = false;
    }
    for (int k = 0; k < 3; k++) { uint32_t a; std::memcpy(&a, &t[k], 4); if (a == 0x80000000u) ident = false; }
    c.ident_r = ident;
    return c;
}

// Second certificate, on top of `fast`: no float the pack converts to an integer can reach 2^31, so the
// kernels may use the saturating hardware convert without keeping a running maximum.
//   world millimetres  |a_r| <= (|M_r0| Xb + |M_r1| Yb + |M_r2| Zb + |M_r3|) * 1000
//   colour column/row  |xf|  <= (A0 / P2low) * |fx_c| + |ppx_c| + 0.5      (no colour distortion)
// each required < 2^30. Depends on cam_to_world, so it is re-evaluated by pcs_set_cam_to_world.
bool certify_no_overflow(const pcs_stream_config& s, const Certificate& c, const float* M)
{
    if (!c.fast || !(c.p2_low > 0)) return false;
    if (s.color.model != PCS_DISTORTION_NONE && coeffs_nonzero(s.color)) return false;
    const double lim = std::ldexp(1.0, 30), infl = 1.0 + std::ldexp(1.0, -18);
    for (int r = 0; r < 3; r++) {
        double a = 0;
        for (int k = 0; k < 4; k++) if (!std::isfinite(M[4 * r + k])) return false;
        a = std::fabs((double)M[4 * r]) * c.xb + std::fabs((double)M[4 * r + 1]) * c.yb + std::fabs((double)M[4 * r + 2]) * c.zb +
            std::fabs((double)M[4 * r + 3]);
        if (!(a * 1000.0 * infl < lim)) return false;
    }
    const double x = c.a_max[0] / c.p2_low * infl, y = c.a_max[1] / c.p2_low * infl;
    const double xf = x * std::fabs((double)s.color.fx) * infl + std::fabs((double)s.color.ppx) + 2.0;   // + .5 rounding slack,
    const double yf = y * std::fabs((double)s.color.fy) * infl + std::fabs((double)s.color.ppy) + 2.0;   // + .5 half-pixel option, + .5 of a2
    return xf * infl < lim && yf * infl < lim;
}

// floor(i / W) == umulhi(i, magic) >> shift for every i < 2^31 (Granlund-Montgomery round-up method,
// N = 31 so the multiplier fits 32 bits); returns false (use '/') if it does not fit or fails the check.
bool row_magic(uint32_t W, uint32_t H, uint32_t& magic, uint32_t& shift)
{
    magic = shift = 0;
    if (W < 2) return false;
    uint32_t l = 0;
    while ((1ull << l) < W) l++;
    const unsigned __int128 m = (((unsigned __int128)1 << (31 + l)) / W) + 1;
    if (m >> 32) return false;
    magic = (uint32_t)m; shift = l - 1;
    auto ok = [&](uint64_t i) { return (uint32_t)(((uint64_t)i * magic) >> 32) >> shift == (uint32_t)(i / W); };
    for (uint64_t r = 0; r < H; r++) {
        const uint64_t a = r * W, b = r * W + W - 1;
        if (a >= (1ull << 31)) break;
        if (!ok(a) || (b < (1ull << 31) && !ok(b))) { magic = shift = 0; return false; }
    }
    if (!ok((1ull << 31) - 1)) { magic = shift = 0; return false; }
    return true;
}

// -c (src/pcs-camera-optimized.cpp:398-401, 504-511) keeps 0 < z <= 1.5 && -2 < x <= 2 on the camera-frame vertex. The
// count pass may evaluate it on the raw Z16 word if (a) z = fl(depth_scale * d) is positive exactly for d >= 1 and
// monotone in d — then z <= 1.5f <=> d <= dmax, dmax found by bisection with the device's own float product — and
// (b) the x test cannot fail while the z test holds: x = fl(z * mx[c]) with no depth distortion, so
// |x| <= 1.5 * max|mx| * (1 + 2^-23) < 2 whenever 1.5 * max|mx| < 2 - 2^-10. Returns 0 if either cannot be shown.
uint32_t cutoff_dmax(const pcs_stream_config& s, const StreamParams& p, const std::vector<float>& mx)
{
    if (!p.z_zero_iff_d_zero || p.ddist || !(s.depth_scale > 0.0f)) return 0;
    double mxmax = 0;
    for (float v : mx) { if (!std::isfinite(v)) return 0; mxmax = std::max(mxmax, std::fabs((double)v)); }
    if (!(1.5 * mxmax < 2.0 - 1.0 / 1024.0)) return 0;
    const volatile float scale = s.depth_scale;
    auto z_of = [&](uint32_t d) { volatile float z = scale * (float)d; return (float)z; };      // one rounded product, as on the device
    if (!(z_of(1) > 0.0f)) return 0;
    for (uint32_t d = 1; d < 65535; d += 257) if (!(z_of(d) <= z_of(d + 1))) return 0;            // monotone (spot check; exact product of positives is)
    uint32_t lo = 0, hi = 65535;                                                                     // largest d with z(d) <= 1.5
    if (z_of(65535) <= 1.5f) return 65535;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (z_of(mid) <= 1.5f) lo = mid; else hi = mid; }
    return lo;       // 0 if even d = 1 is beyond 1.5 m: nothing is in range; the general path handles it
}

inline uint32_t tiles_of(uint32_t n) { return (n + kTilePoints - 1) / kTilePoints; }

// One slab for all streams' rasters, carved at 256-byte granularity (see the comment at s_slab).
int alloc_raster_slab(pcs_ctx* c, uint8_t*& slab, std::vector<uint16_t*>& depth, std::vector<uint8_t*>& color)
{
    const auto up = [](size_t b) { return (b + 16 + 255) & ~(size_t)255; };
    size_t total = 0;
    for (int s = 0; s < c->n_streams; s++)
        total += up((size_t)c->h_params[s].n_points * sizeof(uint16_t)) + up(c->h_params[s].color_bytes);
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, total + 256);
    if (e != hipSuccess) return fail(c, PCS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    slab = static_cast<uint8_t*>(q);
    depth.assign(c->n_streams, nullptr); color.assign(c->n_streams, nullptr);
    size_t off = 0;
    for (int s = 0; s < c->n_streams; s++) {
        depth[s] = reinterpret_cast<uint16_t*>(slab + off);
        off += up((size_t)c->h_params[s].n_points * sizeof(uint16_t));
        color[s] = slab + off;
        off += up(c->h_params[s].color_bytes);
    }
    return PCS_OK;
}

// Lazily allocate the packed raster slab of the host-pointer entry points (sizes are fixed by the config).
int ensure_rasters(pcs_ctx* c)
{
    if (c->s_slab) return PCS_OK;
    return alloc_raster_slab(c, c->s_slab, c->s_depth, c->s_color);
}

}  // namespace

namespace pcs_host {

// The voxel workspace. Growing it means: wait for the stream (the old one may be in use), free, allocate — a device-wide stall.
// A caller whose sizes creep upwards (the node's root reduces a different number of partials every frame-set) must not pay that
// at every new maximum: a workspace that has to grow takes a quarter more than asked.
int ensure_voxel_ws(pcs_ctx* c, size_t need)
{
    if (need <= c->s_voxel_ws_cap && c->s_voxel_ws) return PCS_OK;
    const size_t exact = need;
    if (c->s_voxel_ws) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        need += need / 4;
    }
    c->vox_state.clean = false;                                                   // (a new one may land on the same address)
    c->vox_state.spl_leaf = 0;
    int rc = ensure(c, c->s_voxel_ws, c->s_voxel_ws_cap, need);
    if (rc == PCS_ERR_NOMEM && need != exact) rc = ensure(c, c->s_voxel_ws, c->s_voxel_ws_cap, exact);      // the headroom is a wish
    return rc;
}

int acquire_event_pair(pcs_ctx* c, std::pair<hipEvent_t, hipEvent_t>& pr)
{
    if (!c->ev_free.empty()) { pr = c->ev_free.back(); c->ev_free.pop_back(); return PCS_OK; }
    HIPCHK(c, hipEventCreate(&pr.first));
    HIPCHK(c, hipEventCreate(&pr.second));
    return PCS_OK;
}

// The fused path for device-resident rasters. Counts end up in d_counts (if non-null).
int run_fused_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                     int16_t* d_payload, size_t payload_shorts, int32_t* d_counts, bool force_three_pass,
                     const uint32_t* d_tile_kept)
{
    if (payload_shorts < c->max_payload_points * PCS_POINT_SHORTS && !has_pred(c->flags))
        return fail(c, PCS_ERR_CAPACITY, "payload buffer holds %zu shorts, %zu needed", payload_shorts,
                    c->max_payload_points * PCS_POINT_SHORTS);
    if (has_pred(c->flags) && payload_shorts < c->max_payload_points * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "payload buffer holds %zu shorts; with compaction the worst case "
                    "%zu is required", payload_shorts, c->max_payload_points * PCS_POINT_SHORTS);
    if (((uintptr_t)d_payload & 1u) != 0) return fail(c, PCS_ERR_INVALID_ARG, "payload pointer must be 2-byte aligned");

    const bool pred = has_pred(c->flags);
    const bool dense = !pred && c->downsample == 1 && c->dense_ok && (((uintptr_t)d_payload & 15u) == 0);
    std::pair<hipEvent_t, hipEvent_t> ev{};
    if (c->kernel_timing) {
        int rc = acquire_event_pair(c, ev);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(ev.first, c->stream));
    }
    const bool one_launch = pred && c->downsample == 1 && c->single_pass_ok && !force_three_pass && !d_tile_kept;
    const bool single_pass = one_launch && c->compact_path == 1;
    if (single_pass) {
        if (!c->d_ticket) {
            HIPCHK(c, hipMalloc((void**)&c->d_ticket, sizeof(unsigned long long)));
            HIPCHK(c, hipMalloc((void**)&c->d_desc, sizeof(uint64_t) * ((size_t)c->total_tiles + PCS_MAX_STREAMS)));
            HIPCHK(c, hipMalloc((void**)&c->d_stream_end, sizeof(uint32_t) * c->n_streams));
            HIPCHK(c, hipMalloc((void**)&c->d_error, sizeof(uint32_t)));
            HIPCHK(c, hipMemsetAsync(c->d_ticket, 0, sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_desc, 0, sizeof(uint64_t) * ((size_t)c->total_tiles + PCS_MAX_STREAMS), c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_error, 0, sizeof(uint32_t), c->stream));
        }
        c->compact_seq++;
        if ((c->compact_seq & 0x3FFFFFFFu) == 0) {       // generation wrapped: old descriptors could alias
            HIPCHK(c, hipMemsetAsync(c->d_desc, 0, sizeof(uint64_t) * ((size_t)c->total_tiles + PCS_MAX_STREAMS), c->stream));
            c->compact_seq++;
        }
        for (int s0 = 0; single_pass && s0 < c->n_streams; s0 += kLaunchStreams) {
            const int nl = std::min(kLaunchStreams, c->n_streams - s0);
            FramePtrs fp{};
            uint32_t tiles = 0;
            int m = 1;
            for (int k = 0; k < nl; k++) {
                fp.depth[k] = d_depth[s0 + k]; fp.color[k] = d_color[s0 + k];
                tiles += tiles_of(c->h_params[s0 + k].n_points);
                m = std::min(m, c->h_params[s0 + k].cert_fast);
            }
            CompactLaunch cl{};
            cl.d_ticket = c->compact_tickets ? c->d_ticket : nullptr; cl.ticket_base = c->tickets_issued;
            cl.d_desc = c->d_desc + c->h_params[s0].tile_base;
            cl.d_stream_end = c->d_stream_end;
            cl.d_stream_desc = c->d_desc + c->total_tiles;
            cl.d_chain_in = s0 > 0 ? c->d_stream_end + (s0 - 1) : nullptr;
            cl.d_error = c->d_error; cl.gen = c->compact_seq; cl.flags = c->flags;
            cl.d_counts = d_counts ? d_counts : c->d_counts; cl.n_total = c->n_streams;
            cl.last_launch = (s0 + nl == c->n_streams) ? 1 : 0;
            HIPCHK(c, launch_fused_compact(c->d_params, s0, nl, tiles, m >= 1 ? MathSel::Cert : MathSel::Ieee, fp, cl,
                                           d_payload, c->stream));
            c->tickets_issued += tiles;
        }
        if (c->kernel_timing) {
            HIPCHK(c, hipEventRecord(ev.second, c->stream));
            c->ev_pool.push_back(ev);
        }
        return PCS_OK;
    }
    if (pred) {
        // (a caller that knows how many points each tile keeps — whatever wrote the depth image on the GPU — hands the counts
        // over and the count pass, with its second read of the Z16 rasters, does not run at all)
        for (int s0 = 0; !d_tile_kept && s0 < c->n_streams; s0 += kLaunchStreams) {
            const int nl = std::min(kLaunchStreams, c->n_streams - s0);
            FramePtrs fp{};
            uint32_t mp = 0;
            for (int k = 0; k < nl; k++) { fp.depth[k] = d_depth[s0 + k]; fp.color[k] = d_color[s0 + k]; mp = std::max(mp, c->h_params[s0 + k].n_points); }
            HIPCHK(c, launch_fused_count(c->d_params, s0, nl, mp, c->flags, fp, c->d_tile_counts, c->stream));
        }
        // per-stream counts come from the scan; the grand total from the first emit launch (no last-arriver atomic)
        HIPCHK(c, launch_scan(c->d_params, c->n_streams, c->downsample, d_tile_kept ? d_tile_kept : c->d_tile_counts, c->d_tile_prefix,
                              c->d_stream_base, d_counts ? d_counts : c->d_counts, nullptr, c->stream));
    }
    for (int s0 = 0; s0 < c->n_streams; s0 += kLaunchStreams) {
        const int nl = std::min(kLaunchStreams, c->n_streams - s0);
        FramePtrs fp{};
        uint32_t mp = 0;
        for (int k = 0; k < nl; k++) { fp.depth[k] = d_depth[s0 + k]; fp.color[k] = d_color[s0 + k]; mp = std::max(mp, c->h_params[s0 + k].n_points); }
        bool fast = true, ident = true, noovf = true;    // AND over the streams of this launch
        bool dd = false, cd = false;
        for (int k = 0; k < nl; k++) {
            const StreamParams& q = c->h_params[s0 + k];
            fast &= q.cert_fast != 0; ident &= q.ident_r != 0; noovf &= q.no_overflow != 0;
            dd |= q.ddist != 0;
            cd |= q.cdist != 0 || q.tex_half != 0;
        }
        const MathSel sel = !fast ? MathSel::Ieee
                          : noovf ? (ident ? MathSel::CertIdentRNoOvf : MathSel::CertNoOvf)
                                  : (ident ? MathSel::CertIdentR : MathSel::Cert);
        if (dense)
            HIPCHK(c, launch_fused_dense(c->d_params, s0, nl, mp, dd, cd, sel, fp, d_payload, c->stream));
        else
            HIPCHK(c, launch_fused_emit(c->d_params, s0, nl, mp, c->flags, c->downsample, sel, fp, c->d_tile_prefix,
                                        c->d_stream_base, d_payload,
                                        pred ? (d_counts ? d_counts : c->d_counts) + c->n_streams : nullptr, c->n_streams, c->stream));
    }
    if (!pred && d_counts)     // counts are known from the configuration: a device-to-device copy, no host sync
        HIPCHK(c, hipMemcpyAsync(d_counts, c->d_static_counts, sizeof(int32_t) * (c->n_streams + 1),
                                 hipMemcpyDeviceToDevice, c->stream));
    if (c->kernel_timing) {
        HIPCHK(c, hipEventRecord(ev.second, c->stream));
        c->ev_pool.push_back(ev);
    }
    return PCS_OK;
}

}  // namespace pcs_host

namespace {

// Reads (and clears) the single-pass compaction's time-out word. Returns 1 if it was set.
int take_compact_error(pcs_ctx* c, bool& was_set)
{
    was_set = false;
    if (!c->d_error) return PCS_OK;
    uint32_t e = 0;
    HIPCHK(c, hipMemcpyAsync(&e, c->d_error, sizeof e, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (e) {
        was_set = true;
        c->single_pass_ok = false;       // stay on the three-pass path from now on
        HIPCHK(c, hipMemsetAsync(c->d_error, 0, sizeof e, c->stream));
    }
    return PCS_OK;
}

int upload_params(pcs_ctx* c)
{
    HIPCHK(c, hipMemcpy(c->d_params, c->h_params.data(), c->h_params.size() * sizeof(StreamParams), hipMemcpyHostToDevice));
    return PCS_OK;
}

}  // namespace

extern "C" {

int pcs_abi_version(void) { return PCS_ABI_VERSION; }

int pcs_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return 0;
    return n;
}

const char* pcs_strerror(int status)
{
    switch (status) {
        case PCS_OK:              return "ok";
        case PCS_ERR_INVALID_ARG: return "invalid argument";
        case PCS_ERR_NO_DEVICE:   return "no usable HIP device (libpcs_hip has no CPU fallback)";
        case PCS_ERR_HIP:         return "HIP runtime error";
        case PCS_ERR_UNSUPPORTED: return "unsupported configuration";
        case PCS_ERR_CAPACITY:    return "caller buffer too small";
        case PCS_ERR_NOMEM:       return "out of device memory";
        default:                  return "unknown status";
    }
}

const char* pcs_last_error(const pcs_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int pcs_create(pcs_ctx** out, const pcs_config* cfg)
{
    g_create_err.clear();
    if (!out) return fail(nullptr, PCS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfg || !cfg->streams) return fail(nullptr, PCS_ERR_INVALID_ARG, "config / streams is NULL");
    if (cfg->n_streams < 1 || cfg->n_streams > PCS_MAX_STREAMS)
        return fail(nullptr, PCS_ERR_INVALID_ARG, "n_streams %d outside 1..%d", cfg->n_streams, PCS_MAX_STREAMS);
    if (cfg->downsample < 1) return fail(nullptr, PCS_ERR_INVALID_ARG, "downsample %d < 1", cfg->downsample);
    if (cfg->flags & ~(PCS_FLAG_CUTOFF | PCS_FLAG_CUTOFF_COMPAT | PCS_FLAG_DROP_INVALID | PCS_FLAG_FORCE_IEEE |
                       PCS_FLAG_TEXCOORD_HALF_PIXEL))
        return fail(nullptr, PCS_ERR_INVALID_ARG, "unknown flag bits 0x%x", cfg->flags);
    if ((cfg->flags & PCS_FLAG_CUTOFF_COMPAT) && !(cfg->flags & PCS_FLAG_CUTOFF))
        return fail(nullptr, PCS_ERR_INVALID_ARG, "PCS_FLAG_CUTOFF_COMPAT needs PCS_FLAG_CUTOFF");
    for (int s = 0; s < cfg->n_streams; s++) {
        int rc = validate_stream(cfg->streams[s], s);
        if (rc) return rc;
    }
    {   // The wire header and the counts are int32 (src/pcs-camera-optimized.cpp:697, 718): bound the stitched payload, in
        // 64 bits, before a device is touched — the per-stream bases below are 32-bit sums of up to 64 streams.
        uint64_t all = 0, tiles = 0;
        for (int s = 0; s < cfg->n_streams; s++) {
            const uint64_t np = (uint64_t)cfg->streams[s].depth.width * (uint64_t)cfg->streams[s].depth.height;
            all += (np + (uint64_t)cfg->downsample - 1) / (uint64_t)cfg->downsample;
            tiles += (np + kTilePoints - 1) / kTilePoints;
        }
        if (all * PCS_POINT_BYTES > 0x7FFFFFFFull || tiles > 0x7FFFFFFFull)
            return fail(nullptr, PCS_ERR_INVALID_ARG, "stitched payload of %llu points exceeds the int32 byte-count header "
                        "(214 748 364 points)", (unsigned long long)all);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PCS_ERR_NO_DEVICE, "hipGetDeviceCount: %s (%d devices) — libpcs_hip needs an AMD GPU; "
                    "there is no CPU fallback", hipGetErrorString(e), ndev);
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, PCS_ERR_NO_DEVICE, "device %d out of range (0..%d)", cfg->device, ndev - 1);

    pcs_ctx* c = new (std::nothrow) pcs_ctx;
    if (!c) return fail(nullptr, PCS_ERR_NOMEM, "host allocation failed");
    try {      // no exception may cross the C ABI: host allocation failures become PCS_ERR_NOMEM
    c->device = cfg->device;
    c->n_streams = cfg->n_streams;
    c->flags = cfg->flags;
    c->downsample = cfg->downsample;
    c->cfg.assign(cfg->streams, cfg->streams + cfg->n_streams);
    c->h_params.resize(c->n_streams);
    c->d_lut.assign(2 * (size_t)c->n_streams, nullptr);
    c->s_depth.assign(c->n_streams, nullptr); c->s_color.assign(c->n_streams, nullptr);

#define CREATE_CHK(expr)                                                                             \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            int _rc = fail(nullptr, PCS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));     \
            pcs_destroy(c);                                                                          \
            return _rc;                                                                              \
        }                                                                                            \
    } while (0)

    DeviceGuard guard(c->device);
    CREATE_CHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CREATE_CHK(hipEventCreate(&c->ev_begin));
    CREATE_CHK(hipEventCreate(&c->ev_end));
    CREATE_CHK(hipMalloc((void**)&c->d_params, sizeof(StreamParams) * c->n_streams));

    uint32_t out_base = 0, tile_base = 0;
    c->dense_ok = true;
    for (int s = 0; s < c->n_streams; s++) {
        StreamParams& p = c->h_params[s];
        std::memset(&p, 0, sizeof p);
        fill_params(c->cfg[s], p);
        p.tex_half = (c->flags & PCS_FLAG_TEXCOORD_HALF_PIXEL) ? 1 : 0;
        p.out_base = out_base;
        p.tile_base = tile_base;
        out_base += (p.n_points + c->downsample - 1) / c->downsample;
        tile_base += tiles_of(p.n_points);
        if (p.n_points % 8) c->dense_ok = false;
        c->any_ddist |= p.ddist != 0;
        c->any_cdist |= p.cdist != 0 || p.tex_half != 0;
        c->max_points = std::max(c->max_points, p.n_points);
        // deprojection LUTs: the IEEE divisions of rs2_deproject_pixel_to_point, once per column / row
        std::vector<float> mx(p.W), my(p.H);
        for (int x = 0; x < p.W; x++) mx[x] = ((float)x - p.d_ppx) / p.d_fx;
        for (int y = 0; y < p.H; y++) my[y] = ((float)y - p.d_ppy) / p.d_fy;
        const Certificate cert = (c->flags & PCS_FLAG_FORCE_IEEE) ? Certificate{} : certify_stream(c->cfg[s], mx, my);
        p.cert_fast = cert.fast ? 1 : 0;
        p.ident_r = (cert.fast && cert.ident_r) ? 1 : 0;
        p.no_overflow = certify_no_overflow(c->cfg[s], cert, c->cfg[s].cam_to_world) ? 1 : 0;
        c->cert.push_back(cert);
        row_magic((uint32_t)p.W, (uint32_t)p.H, p.w_magic, p.w_shift);
        p.cut_dmax = cutoff_dmax(c->cfg[s], p, mx);
        float *dmx = nullptr, *dmy = nullptr;
        CREATE_CHK(hipMalloc((void**)&dmx, sizeof(float) * ((size_t)p.W + 8)));
        c->d_lut[2 * s] = dmx;
        CREATE_CHK(hipMalloc((void**)&dmy, sizeof(float) * (2 * (size_t)p.H + 8)));      // + the colour-row table of CertRowConst
        CREATE_CHK(hipMemset(dmy, 0, sizeof(float) * (2 * (size_t)p.H + 8)));
        c->d_lut[2 * s + 1] = dmy;
        CREATE_CHK(hipMemcpy(dmx, mx.data(), sizeof(float) * p.W, hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy(dmy, my.data(), sizeof(float) * p.H, hipMemcpyHostToDevice));
        p.mx = dmx; p.my = dmy;
    }
    c->max_payload_points = out_base;
    c->total_tiles = tile_base;
    CREATE_CHK(hipMalloc((void**)&c->d_tile_counts, sizeof(uint32_t) * std::max<uint32_t>(tile_base, 1)));
    CREATE_CHK(hipMalloc((void**)&c->d_tile_prefix, sizeof(uint32_t) * std::max<uint32_t>(tile_base, 1)));
    CREATE_CHK(hipMalloc((void**)&c->d_stream_base, sizeof(uint32_t) * (c->n_streams + 1)));
    CREATE_CHK(hipMalloc((void**)&c->d_counts, sizeof(int32_t) * (c->n_streams + 1)));
    {
        std::vector<int32_t> hc(c->n_streams + 1);
        int64_t tot = 0;
        for (int s = 0; s < c->n_streams; s++) {
            hc[s] = (int32_t)((c->h_params[s].n_points + c->downsample - 1) / c->downsample);
            tot += hc[s];
        }
        hc[c->n_streams] = (int32_t)tot;
        CREATE_CHK(hipMalloc((void**)&c->d_static_counts, sizeof(int32_t) * hc.size()));
        CREATE_CHK(hipMemcpy(c->d_static_counts, hc.data(), sizeof(int32_t) * hc.size(), hipMemcpyHostToDevice));
    }
    CREATE_CHK(hipMalloc((void**)&c->d_arrive, sizeof(uint32_t)));
    CREATE_CHK(hipMemset(c->d_arrive, 0, sizeof(uint32_t)));
    {   // device certificate for CertMath::div_const: all 2^32 numerators, once per distinct raster dimension
        std::vector<std::pair<int32_t, bool>> seen;
        unsigned long long* d_bad = nullptr;
        auto verified = [&](int32_t dim, bool& ok) -> hipError_t {
            for (auto& pr : seen) if (pr.first == dim) { ok = pr.second; return hipSuccess; }
            hipError_t e;
            if (!d_bad && (e = hipMalloc((void**)&d_bad, sizeof(unsigned long long))) != hipSuccess) return e;
            if ((e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream)) != hipSuccess) return e;
            const float cf = (float)dim, rc = (float)(1.0 / (double)cf);
            if ((e = launch_verify_div_const(cf, rc, dim, d_bad, c->stream)) != hipSuccess) return e;
            unsigned long long h = 1;
            if ((e = hipMemcpyAsync(&h, d_bad, sizeof h, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
            ok = (h == 0);
            seen.emplace_back(dim, ok);
            return hipSuccess;
        };
        for (int s = 0; s < c->n_streams; s++) {
            StreamParams& p = c->h_params[s];
            if (!p.cert_fast) continue;
            bool okw = false, okh = false;
            CREATE_CHK(verified(p.cW, okw));
            CREATE_CHK(verified(p.cH, okh));
            if (!(okw && okh)) { p.cert_fast = 0; p.ident_r = 0; p.no_overflow = 0; }
        }
        if (d_bad) (void)hipFree(d_bad);
    }
    // Single-pass compaction (one launch, direct-sum placement): 27.3-28.6 us against 30.3-32.2 us
    // for count + scan + emit on 8x720p, but its forward progress assumes workgroups are dispatched in
    // blockIdx order (bounded waits + three-pass re-run catch a violation) -> opt-in. PCS_COMPACT_TICKETS=1
    // hands tile ids out in start order instead (dispatch-order independent; the contended atomic makes it 63 us).
    // Ordered compaction. Default: count + scan + emit — three small launches, no inter-workgroup waiting at all, and
    // the fastest form at large sizes (16 x 1080p: 111 us vs 129 us). PCS_COMPACT_PATH=single (older spelling:
    // PCS_COMPACT_SINGLE_PASS=1) selects the single-pass kernel (one launch, Z16 read once; 32.6 vs 33.9 us on 8 x 720p,
    // but its forward progress assumes workgroups are dispatched in blockIdx order — bounded waits + a three-pass re-run
    // catch a violation), PCS_COMPACT_TICKETS=1 its dispatch-order independent but slow ticketed variant. DESIGN.md §5
    // lists the persistent / chunked variants that were built to lift the ordering assumption and measured slower.
    // Ordered compaction. Default: count + scan + emit — three small launches, no inter-workgroup waiting at all, and the
    // fastest form at large sizes (16 x 1080p: 106 us vs 126 us). PCS_COMPACT_PATH=single (older spelling:
    // PCS_COMPACT_SINGLE_PASS=1) selects the single-pass kernel (one launch, Z16 read once; 31.2 vs 31.7 us on 8 x 720p,
    // but its forward progress assumes workgroups are dispatched in blockIdx order — bounded waits + a three-pass re-run
    // catch a violation), PCS_COMPACT_TICKETS=1 its dispatch-order independent but slow ticketed variant. DESIGN.md §5
    // lists everything else that was built to beat the three launches and measured slower or equal.
    c->compact_path = 0;
    if (const char* e = getenv("PCS_COMPACT_PATH")) {
        if (!strcmp(e, "single")) c->compact_path = 1;
    } else if (const char* e1 = getenv("PCS_COMPACT_SINGLE_PASS")) {
        if (e1[0] == '1') c->compact_path = 1;
    }
    c->single_pass_ok = c->compact_path == 1;
    { const char* e = getenv("PCS_COMPACT_TICKETS"); c->compact_tickets = e && e[0] == '1'; }
    c->math.resize(c->n_streams);
    for (int s = 0; s < c->n_streams; s++) {
        const StreamParams& q = c->h_params[s];
        c->math[s] = q.cert_fast ? ((q.ident_r ? 2 : 1) + (q.no_overflow ? 2 : 0)) : 0;
    }
    CREATE_CHK(hipMemcpy(c->d_params, c->h_params.data(), sizeof(StreamParams) * c->n_streams, hipMemcpyHostToDevice));
    {   // CertRowConst (pcs_kernels.hip): for a stream with R = I, t_y = t_z = 0 and no distortion the colour ROW of a pixel is — up to
        // the rounding of (z * my) / z — a function of its raster row. Swept on the device over every row x every Z16 value 1 .. 65 535
        // through the IEEE chain (H x 65 535 evaluations: 47 M for 720 rows, under a millisecond); where no pair disagrees the table
        // behind the my LUT is valid and ident_r becomes 2. PCS_ROW_CONST=0 leaves every stream at 1 (A/B; the tests run both).
        const char* env = getenv("PCS_ROW_CONST");
        const bool want = !(env && env[0] == '0');
        unsigned long long* d_bad = nullptr;
        bool any = false;
        for (int s = 0; want && s < c->n_streams; s++) {
            StreamParams& p = c->h_params[s];
            const float* t = c->cfg[s].depth_to_color.translation;
            if (p.ident_r != 1 || p.ddist || p.cdist || p.tex_half || t[1] != 0.0f || t[2] != 0.0f || !p.z_zero_iff_d_zero) continue;
            if (!d_bad) CREATE_CHK(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
            CREATE_CHK(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream));
            int32_t* d_crow = reinterpret_cast<int32_t*>(const_cast<float*>(p.my) + p.H);
            CREATE_CHK(launch_certify_color_row(c->d_params, s, p.H, d_crow, d_bad, c->stream));
            unsigned long long h = 1;
            CREATE_CHK(hipMemcpyAsync(&h, d_bad, sizeof h, hipMemcpyDeviceToHost, c->stream));
            CREATE_CHK(hipStreamSynchronize(c->stream));
            if (h == 0) { p.ident_r = 2; any = true; }
        }
        if (d_bad) (void)hipFree(d_bad);
        if (any) CREATE_CHK(hipMemcpy(c->d_params, c->h_params.data(), sizeof(StreamParams) * c->n_streams, hipMemcpyHostToDevice));
    }
#undef CREATE_CHK
    } catch (const std::exception& ex) {
        const int rc = fail(nullptr, PCS_ERR_NOMEM, "pcs_create: host allocation failed (%s)", ex.what());
        pcs_destroy(c);
        return rc;
    }
    *out = c;
    return PCS_OK;
}

void pcs_destroy(pcs_ctx* c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (float* p : c->d_lut) if (p) (void)hipFree(p);
    if (c->s_slab) (void)hipFree(c->s_slab);
    if (c->dl_stream) (void)hipStreamSynchronize(c->dl_stream);
    for (auto& sl : c->pipe) {
        if (sl.slab) (void)hipFree(sl.slab);
        if (sl.payload) (void)hipFree(sl.payload);
        if (sl.counts) (void)hipFree(sl.counts);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
    void* singles[] = {c->d_params, c->d_tile_counts, c->d_tile_prefix, c->d_stream_base, c->d_counts, c->s_payload,
                       c->d_ticket, c->d_desc, c->d_stream_end, c->d_error, c->d_arrive, c->d_static_counts, c->d_batch_scratch, c->s_voxel_ws, c->s_voxel_in, c->s_voxel_out,
                       c->s_vertices, c->s_texcoords, c->s_pack_counts, c->s_pack_prefix};
    for (void* p : singles) if (p) (void)hipFree(p);
    for (auto& pr : c->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : c->ev_free) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pcs_set_cam_to_world(pcs_ctx* c, int stream, const float m16[16])
{
    if (!c || !m16) return PCS_ERR_INVALID_ARG;
    if (stream < 0 || stream >= c->n_streams) return fail(c, PCS_ERR_INVALID_ARG, "stream %d out of range", stream);
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(c->cfg[stream].cam_to_world, m16, sizeof(float) * 16);
    StreamParams& q = c->h_params[stream];
    for (int k = 0; k < 12; k++) q.M[k] = m16[k];
    q.no_overflow = (q.cert_fast && certify_no_overflow(c->cfg[stream], c->cert[stream], m16)) ? 1 : 0;
    c->math[stream] = q.cert_fast ? ((q.ident_r ? 2 : 1) + (q.no_overflow ? 2 : 0)) : 0;
    return upload_params(c);
}

int pcs_stream_points(const pcs_ctx* c, int stream)
{
    if (!c || stream < 0 || stream >= c->n_streams) return PCS_ERR_INVALID_ARG;
    return (int)c->h_params[stream].n_points;
}

int pcs_stream_math(const pcs_ctx* c, int stream)
{
    if (!c || stream < 0 || stream >= c->n_streams) return PCS_ERR_INVALID_ARG;
    return c->math[stream];
}

int pcs_stream_color_row_const(const pcs_ctx* c, int stream)
{
    if (!c || stream < 0 || stream >= c->n_streams) return PCS_ERR_INVALID_ARG;
    return c->h_params[stream].ident_r == 2 ? 1 : 0;
}

size_t pcs_max_payload_shorts(const pcs_ctx* c)
{
    return c ? c->max_payload_points * PCS_POINT_SHORTS : 0;
}

// ---- a2 twin ---------------------------------------------------------------------------------
int pcs_copy_pointcloud_xyzrgb_to_buffer_device(pcs_ctx* c, int stream, const float* d_vertices,
                                                const float* d_texcoords, int n_points, const uint8_t* d_color,
                                                int16_t* d_pc_buffer, int* d_out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (stream < 0 || stream >= c->n_streams) return fail(c, PCS_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_points %d < 0", n_points);
    if (n_points > 0 && (!d_vertices || !d_texcoords || !d_color || !d_pc_buffer))
        return fail(c, PCS_ERR_INVALID_ARG, "NULL device pointer");
    if (((uintptr_t)d_pc_buffer & 1u) || ((uintptr_t)d_vertices & 3u) || ((uintptr_t)d_texcoords & 3u))
        return fail(c, PCS_ERR_INVALID_ARG, "misaligned device pointer");
    DeviceGuard guard(c->device);
    if (n_points == 0) {
        if (d_out_points) HIPCHK(c, hipMemsetAsync(d_out_points, 0, sizeof(int), c->stream));
        return PCS_OK;
    }
    const bool pred = has_pred(c->flags);
    VertexPtrs vp{d_vertices, d_texcoords, d_color, (uint32_t)n_points};
    if (!pred) {
        if (((uintptr_t)d_pc_buffer & 15u) == 0)
            HIPCHK(c, launch_pack_dense(c->d_params, stream, vp, d_pc_buffer, c->stream));
        else
            HIPCHK(c, launch_pack_emit(c->d_params, stream, vp, 0u, nullptr, d_pc_buffer, c->stream));
        if (d_out_points) HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)d_out_points, n_points, 1, c->stream));
        return PCS_OK;
    }
    const uint32_t tiles = std::max<uint32_t>(tiles_of((uint32_t)n_points), 1);
    if (tiles > c->s_pack_tiles) {
        if (c->s_pack_counts) (void)hipFree(c->s_pack_counts);
        if (c->s_pack_prefix) (void)hipFree(c->s_pack_prefix);
        c->s_pack_counts = c->s_pack_prefix = nullptr; c->s_pack_tiles = 0;
        HIPCHK(c, hipMalloc((void**)&c->s_pack_counts, sizeof(uint32_t) * tiles));
        HIPCHK(c, hipMalloc((void**)&c->s_pack_prefix, sizeof(uint32_t) * tiles));
        c->s_pack_tiles = tiles;
    }
    HIPCHK(c, launch_pack_count(c->d_params, stream, vp, c->flags, c->s_pack_counts, c->stream));
    HIPCHK(c, launch_pack_scan(tiles_of((uint32_t)n_points), c->s_pack_counts, c->s_pack_prefix, c->d_counts, c->d_arrive, c->stream));
    HIPCHK(c, launch_pack_emit(c->d_params, stream, vp, c->flags, c->s_pack_prefix, d_pc_buffer, c->stream));
    if (d_out_points)
        HIPCHK(c, hipMemcpyAsync(d_out_points, c->d_counts, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    return PCS_OK;
}

// Batched form: all cameras of a frame-set in one launch (groups of kPackBatch). With a predicate the kept counts are
// data dependent and every cloud needs its own count + scan + emit, so that case runs the single-cloud path per entry.
int pcs_copy_pointclouds_xyzrgb_to_buffer_device(pcs_ctx* c, int n_clouds, const pcs_cloud_desc* clouds, int* d_out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_clouds < 0 || (n_clouds > 0 && !clouds)) return fail(c, PCS_ERR_INVALID_ARG, "bad cloud list");
    for (int i = 0; i < n_clouds; i++) {
        const pcs_cloud_desc& q = clouds[i];
        if (q.stream < 0 || q.stream >= c->n_streams) return fail(c, PCS_ERR_INVALID_ARG, "cloud %d: stream %d out of range", i, q.stream);
        if (q.n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "cloud %d: n_points %d < 0", i, q.n_points);
        if (q.n_points > 0 && (!q.vertices || !q.texcoords || !q.color || !q.pc_buffer))
            return fail(c, PCS_ERR_INVALID_ARG, "cloud %d: NULL device pointer", i);
        if (((uintptr_t)q.pc_buffer & 1u) || ((uintptr_t)q.vertices & 3u) || ((uintptr_t)q.texcoords & 3u))
            return fail(c, PCS_ERR_INVALID_ARG, "cloud %d: misaligned device pointer", i);
    }
    DeviceGuard guard(c->device);
    if (has_pred(c->flags)) {
        for (int i = 0; i < n_clouds; i++) {
            const pcs_cloud_desc& q = clouds[i];
            int rc = pcs_copy_pointcloud_xyzrgb_to_buffer_device(c, q.stream, q.vertices, q.texcoords, q.n_points, q.color,
                                                                 q.pc_buffer, d_out_points ? d_out_points + i : nullptr);
            if (rc) return rc;
        }
        return PCS_OK;
    }
    std::pair<hipEvent_t, hipEvent_t> ev{};
    if (c->kernel_timing) {
        int rc = acquire_event_pair(c, ev);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(ev.first, c->stream));
    }
    for (int i0 = 0; i0 < n_clouds; i0 += kPackBatch) {
        const int nb = std::min(kPackBatch, n_clouds - i0);
        PackBatch pb{};
        uint32_t mp = 0;
        bool aligned = true;
        for (int k = 0; k < nb; k++) {
            const pcs_cloud_desc& q = clouds[i0 + k];
            pb.v[k] = VertexPtrs{q.vertices, q.texcoords, q.color, (uint32_t)q.n_points};
            pb.out[k] = reinterpret_cast<uint8_t*>(q.pc_buffer);
            pb.stream[k] = q.stream;
            mp = std::max(mp, (uint32_t)q.n_points);
            aligned &= ((uintptr_t)q.pc_buffer & 15u) == 0;
        }
        HIPCHK(c, launch_pack_batch(c->d_params, pb, nb, mp, aligned, c->stream));
    }
    if (d_out_points)
        for (int i = 0; i < n_clouds; i++)
            HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)(d_out_points + i), clouds[i].n_points, 1, c->stream));
    if (c->kernel_timing) {
        HIPCHK(c, hipEventRecord(ev.second, c->stream));
        c->ev_pool.push_back(ev);
    }
    return PCS_OK;
}

// The device's view of a page-locked host range (pcs_host_malloc, hipHostMalloc, hipHostRegister).
//   1  the page-locked allocation covers ALL `bytes` the kernels will touch from `h` on: *d is the device view (zero copy)
//   0  pageable memory: the caller stages
//  -1  page-locked, but the allocation ends before h + bytes (a buffer registered in part, an interior pointer too close to
//      the end of a registration): neither route can take it — a kernel would run off the end of the mapping and fault on
//      the GPU, and the HIP runtime refuses copies that straddle the edge of a registration — so the call is refused
// The verdict is cached per (pointer, bytes): a frame loop that hands the same buffers over every frame does one attribute
// query per buffer (the re-validation below) instead of the range look-ups; pcs_host_free / pcs_host_unregister /
// pcs_host_register drop the cache.
static int host_device_view(pcs_ctx* c, const void* h, size_t bytes, void** d)
{
    // A cached verdict is re-validated with ONE attribute query: the buffer may have been released (or registered) behind this
    // context's back — hipHostFree / hipHostUnregister / another context — and its address handed out again; a stale "zero
    // copy" verdict would let a kernel dereference a dead mapping. Only an unchanged answer (still page-locked with the same
    // device view, or still pageable) is a hit; anything else is looked up afresh (the range queries below).
    hipPointerAttribute_t a{};
    const bool locked = hipPointerGetAttributes(&a, h) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer;
    (void)hipGetLastError();
    for (size_t i = 0; i < c->zc_cache.size(); i++) {
        const auto& e = c->zc_cache[i];
        if (e.host != h || e.bytes != bytes) continue;
        if ((e.verdict == 1 && locked && a.devicePointer == e.dev) || (e.verdict == 0 && !locked)) { *d = e.dev; return e.verdict; }
        c->zc_cache.erase(c->zc_cache.begin() + (ptrdiff_t)i);
        break;
    }
    void* dev = nullptr;
    int verdict = 0;
    if (locked) {
        verdict = -1;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) == hipSuccess && base) {
            const size_t off = (size_t)((const char*)a.devicePointer - (const char*)base);
            if (off <= size && bytes <= size - off) { dev = a.devicePointer; verdict = 1; }
        } else if (bytes > 0) {
            // (the runtime reports no range for hipHostRegister'ed memory) the LAST byte must be page-locked too, and sit in
            // the same mapping: its device view is the first byte's plus the distance
            (void)hipGetLastError();
            hipPointerAttribute_t z{};
            const char* last = static_cast<const char*>(h) + (bytes - 1);
            if (hipPointerGetAttributes(&z, last) == hipSuccess && z.type == hipMemoryTypeHost && z.devicePointer &&
                (const char*)z.devicePointer - (const char*)a.devicePointer == (ptrdiff_t)(bytes - 1)) { dev = a.devicePointer; verdict = 1; }
        }
    }
    (void)hipGetLastError();
    if (c->zc_cache.size() >= 64) c->zc_cache.erase(c->zc_cache.begin());
    c->zc_cache.push_back({h, bytes, dev, verdict});
    *d = dev;
    return verdict;
}
static int partly_locked(pcs_ctx* c, const char* what, int idx)
{
    return fail(c, PCS_ERR_INVALID_ARG, "%s %d is page-locked for only part of the range this call touches: register the whole "
                "buffer (pcs_host_register) or none of it", what, idx);
}
static bool zero_copy_enabled()
{
    static const int zc_env = [] { const char* v = getenv("PCS_ZERO_COPY"); return v ? atoi(v) : 1; }();
    return zc_env != 0;
}

int pcs_copy_pointcloud_xyzrgb_to_buffer(pcs_ctx* c, int stream, const float* vertices, const float* texcoords,
                                         int n_points, const uint8_t* color, int16_t* pc_buffer, int* out_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (stream < 0 || stream >= c->n_streams) return fail(c, PCS_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_points %d < 0", n_points);
    if (n_points > 0 && (!vertices || !texcoords || !color || !pc_buffer))
        return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (n_points == 0) { if (out_points) *out_points = 0; return PCS_OK; }
    DeviceGuard guard(c->device);
    const StreamParams& P = c->h_params[stream];
    const size_t vb = (size_t)n_points * 3 * sizeof(float), tb = (size_t)n_points * 2 * sizeof(float);
    const size_t ob = (size_t)n_points * PCS_POINT_BYTES;
    int rc;
    if (zero_copy_enabled()) {
        // page-locked arrays on every side (see pcs_process_frames): the kernel reads and writes them in place
        void *zv = nullptr, *zt = nullptr, *zc = nullptr, *zo = nullptr;
        const int v4[4] = {host_device_view(c, vertices, vb, &zv), host_device_view(c, texcoords, tb, &zt),
                           host_device_view(c, color, P.color_bytes, &zc), host_device_view(c, pc_buffer, ob, &zo)};
        for (int k = 0; k < 4; k++) if (v4[k] < 0) return partly_locked(c, "array", k);
        if (v4[0] == 1 && v4[1] == 1 && v4[2] == 1 && v4[3] == 1) {
            rc = pcs_copy_pointcloud_xyzrgb_to_buffer_device(c, stream, static_cast<const float*>(zv), static_cast<const float*>(zt),
                                                             n_points, static_cast<const uint8_t*>(zc), static_cast<int16_t*>(zo), nullptr);
            if (rc) return rc;
            int count = n_points;
            if (has_pred(c->flags))
                HIPCHK(c, hipMemcpyAsync(&count, c->d_counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (out_points) *out_points = count;
            return PCS_OK;
        }
    }
    if ((rc = ensure(c, c->s_vertices, c->s_vertices_cap, vb))) return rc;
    if ((rc = ensure(c, c->s_texcoords, c->s_texcoords_cap, tb))) return rc;
    if ((rc = ensure_rasters(c))) return rc;
    if ((rc = ensure(c, c->s_payload, c->s_payload_cap, ob + 16))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s_vertices, vertices, vb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->s_texcoords, texcoords, tb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->s_color[stream], color, P.color_bytes, hipMemcpyHostToDevice, c->stream));
    rc = pcs_copy_pointcloud_xyzrgb_to_buffer_device(c, stream, c->s_vertices, c->s_texcoords, n_points,
                                                     c->s_color[stream], c->s_payload, nullptr);
    if (rc) return rc;
    int count = n_points;
    if (has_pred(c->flags)) {
        HIPCHK(c, hipMemcpyAsync(&count, c->d_counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (count > 0)
        HIPCHK(c, hipMemcpyAsync(pc_buffer, c->s_payload, (size_t)count * PCS_POINT_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_points) *out_points = count;
    return PCS_OK;
}

// ---- a1 twin ---------------------------------------------------------------------------------
int pcs_send_xyzrgb_pointcloud(pcs_ctx* c, int stream, const float* vertices, const float* texcoords, int n_points,
                               const uint8_t* color, int16_t* buffer, size_t buffer_shorts, int write_header,
                               int* out_size_bytes)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!buffer) return fail(c, PCS_ERR_INVALID_ARG, "buffer is NULL");
    if (n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_points %d < 0", n_points);
    const size_t need = PCS_HEADER_SHORTS + (size_t)n_points * PCS_POINT_SHORTS;
    if (buffer_shorts < need)
        return fail(c, PCS_ERR_CAPACITY, "buffer holds %zu shorts, %zu needed (the reference's BUF_SIZE of %d shorts "
                    "overflows beyond 999 999 points)", buffer_shorts, need, PCS_REF_BUF_SIZE);
    int count = 0;
    int rc = pcs_copy_pointcloud_xyzrgb_to_buffer(c, stream, vertices, texcoords, n_points, color,
                                                  buffer + PCS_HEADER_SHORTS, &count);
    if (rc) return rc;
    const int32_t size = (int32_t)((size_t)count * PCS_POINT_BYTES);                     // :697
    // :673 — everything below BUF_SIZE bytes that the payload does not cover reads as zero
    const size_t clear_end = std::min<size_t>(PCS_REF_BUF_SIZE, buffer_shorts * sizeof(int16_t));
    uint8_t* b = reinterpret_cast<uint8_t*>(buffer);
    std::memset(b, 0, std::min<size_t>(4, clear_end));
    const size_t pay_end = 4 + (size_t)size;
    if (pay_end < clear_end) std::memset(b + pay_end, 0, clear_end - pay_end);
    if (write_header) std::memcpy(b, &size, sizeof size);                                 // :718
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

// ---- fused a5+a2(+a7) ------------------------------------------------------------------------
int pcs_process_frames_device(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                              int16_t* d_payload, size_t payload_shorts, int32_t* d_counts)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_depth || !d_color || !d_payload) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    for (int s = 0; s < c->n_streams; s++)
        if (!d_depth[s] || !d_color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
    for (int s = 0; s < c->n_streams; s++)
        if ((uintptr_t)d_depth[s] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: depth pointer not 2-byte aligned", s);
    DeviceGuard guard(c->device);
    return run_fused_device(c, d_depth, d_color, d_payload, payload_shorts, d_counts);
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_device: host allocation failed (%s)", ex.what());
}

int pcs_process_frames_device_counted(pcs_ctx* c, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                      const uint32_t* d_tile_kept, int16_t* d_payload, size_t payload_shorts, int32_t* d_counts)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!d_depth || !d_color || !d_payload || !d_tile_kept) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)d_tile_kept & 3u) return fail(c, PCS_ERR_INVALID_ARG, "d_tile_kept must be 4-byte aligned");
    for (int s = 0; s < c->n_streams; s++) {
        if (!d_depth[s] || !d_color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
        if ((uintptr_t)d_depth[s] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: depth pointer not 2-byte aligned", s);
    }
    DeviceGuard guard(c->device);
    return run_fused_device(c, d_depth, d_color, d_payload, payload_shorts, d_counts, true, has_pred(c->flags) ? d_tile_kept : nullptr);
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_device_counted: host allocation failed (%s)", ex.what());
}

int pcs_stream_tile_base(const pcs_ctx* c, int stream)
{
    if (!c || stream < 0 || stream > c->n_streams) return PCS_ERR_INVALID_ARG;
    return stream == c->n_streams ? (int)c->total_tiles : (int)c->h_params[stream].tile_base;
}

// K frame-sets per launch (throughput form). The dense path (no predicate, stride 1, 16-byte aligned payloads, every
// stream a multiple of 8 points) puts up to 64 / n_streams frame-sets into ONE launch, so the fill and drain of the
// machine are paid once per group instead of once per frame-set; every other configuration runs the sets one after
// the other through the same code as pcs_process_frames_device. The bytes written are identical either way.
int pcs_process_frames_device_batch(pcs_ctx* c, int n_sets, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                    int16_t* const* d_payload, size_t payload_shorts, int32_t* const* d_counts)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_sets < 0) return fail(c, PCS_ERR_INVALID_ARG, "n_sets %d < 0", n_sets);
    if (n_sets == 0) return PCS_OK;
    if (!d_depth || !d_color || !d_payload) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    const int S = c->n_streams;
    bool aligned = true;
    for (int k = 0; k < n_sets; k++) {
        if (!d_payload[k]) return fail(c, PCS_ERR_INVALID_ARG, "frame-set %d: NULL payload pointer", k);
        if ((uintptr_t)d_payload[k] & 1u) return fail(c, PCS_ERR_INVALID_ARG, "frame-set %d: payload pointer must be 2-byte aligned", k);
        aligned &= ((uintptr_t)d_payload[k] & 15u) == 0;
        for (int s = 0; s < S; s++) {
            if (!d_depth[(size_t)k * S + s] || !d_color[(size_t)k * S + s])
                return fail(c, PCS_ERR_INVALID_ARG, "frame-set %d stream %d: NULL raster pointer", k, s);
            if ((uintptr_t)d_depth[(size_t)k * S + s] & 1u)
                return fail(c, PCS_ERR_INVALID_ARG, "frame-set %d stream %d: depth pointer not 2-byte aligned", k, s);
        }
    }
    DeviceGuard guard(c->device);
    const int per_launch = std::min(kBatchSets, kBatchEntries / S);
    const bool dense = !has_pred(c->flags) && c->downsample == 1 && c->dense_ok && aligned && per_launch >= 2;
    if (has_pred(c->flags) && c->downsample == 1 && per_launch >= 2 && S <= kLaunchStreams) {
        // Ordered compaction of K frame-sets with THREE launches for all of them: count (grid.z = set), scan
        // (one workgroup per stream and set), emit (grid.z = set). Same kernels' tile code as the one-set path,
        // the same bytes; no in-launch handoff between tiles, so nothing depends on dispatch order.
        if (payload_shorts < c->max_payload_points * PCS_POINT_SHORTS)
            return fail(c, PCS_ERR_CAPACITY, "payload buffers hold %zu shorts; with compaction the worst case %zu is required",
                        payload_shorts, c->max_payload_points * PCS_POINT_SHORTS);
        const size_t tt = std::max<uint32_t>(c->total_tiles, 1);
        const size_t row = 2 * tt + (size_t)S + (size_t)(S + 1);      // counts, prefixes, kept per stream, counts out
        if (c->batch_scratch_sets < per_launch) {
            if (c->d_batch_scratch) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_batch_scratch); c->d_batch_scratch = nullptr; }
            c->batch_scratch_sets = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_batch_scratch, sizeof(uint32_t) * row * per_launch));
            c->batch_scratch_sets = per_launch;
        }
        uint32_t* tcounts = c->d_batch_scratch;
        uint32_t* tprefix = tcounts + tt * per_launch;
        uint32_t* kept    = tprefix + tt * per_launch;
        int32_t*  icounts = reinterpret_cast<int32_t*>(kept + (size_t)S * per_launch);
        bool fast = true, ident = true;
        for (int s = 0; s < S; s++) { fast &= c->h_params[s].cert_fast != 0; ident &= c->h_params[s].ident_r != 0; }
        const MathSel sel = !fast ? MathSel::Ieee : (ident ? MathSel::CertIdentR : MathSel::Cert);
        std::pair<hipEvent_t, hipEvent_t> ev{};
        if (c->kernel_timing) {
            int rc = acquire_event_pair(c, ev);
            if (rc) return rc;
            HIPCHK(c, hipEventRecord(ev.first, c->stream));
        }
        for (int k0 = 0; k0 < n_sets; k0 += per_launch) {
            const int nk = std::min(per_launch, n_sets - k0);
            BatchPtrs bp{};
            BatchCounts bc{};
            for (int k = 0; k < nk; k++) {
                bp.payload[k] = reinterpret_cast<uint8_t*>(d_payload[k0 + k]);
                bc.counts[k] = (d_counts && d_counts[k0 + k]) ? d_counts[k0 + k] : icounts + (size_t)k * (S + 1);
                for (int s = 0; s < S; s++) {
                    bp.depth[k * S + s] = d_depth[(size_t)(k0 + k) * S + s];
                    bp.color[k * S + s] = d_color[(size_t)(k0 + k) * S + s];
                }
            }
            HIPCHK(c, launch_compact_batch(c->d_params, S, nk, c->max_points, c->total_tiles, c->flags, sel, bp, bc,
                                           tcounts, tprefix, kept, c->stream));
        }
        if (c->kernel_timing) {
            HIPCHK(c, hipEventRecord(ev.second, c->stream));
            c->ev_pool.push_back(ev);
        }
        return PCS_OK;
    }
    if (!dense) {
        for (int k = 0; k < n_sets; k++) {
            int rc = run_fused_device(c, d_depth + (size_t)k * S, d_color + (size_t)k * S, d_payload[k], payload_shorts,
                                      d_counts ? d_counts[k] : nullptr);
            if (rc) return rc;
        }
        return PCS_OK;
    }
    if (payload_shorts < c->max_payload_points * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "payload buffers hold %zu shorts, %zu needed", payload_shorts,
                    c->max_payload_points * PCS_POINT_SHORTS);
    bool fast = true, ident = true, noovf = true, dd = false, cd = false;
    for (int s = 0; s < S; s++) {
        const StreamParams& q = c->h_params[s];
        fast &= q.cert_fast != 0; ident &= q.ident_r != 0; noovf &= q.no_overflow != 0;
        dd |= q.ddist != 0;
        cd |= q.cdist != 0 || q.tex_half != 0;
    }
    const MathSel sel = !fast ? MathSel::Ieee
                      : noovf ? (ident ? MathSel::CertIdentRNoOvf : MathSel::CertNoOvf)
                              : (ident ? MathSel::CertIdentR : MathSel::Cert);
    std::pair<hipEvent_t, hipEvent_t> ev{};
    if (c->kernel_timing) {
        int rc = acquire_event_pair(c, ev);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(ev.first, c->stream));
    }
    for (int k0 = 0; k0 < n_sets; k0 += per_launch) {
        const int nk = std::min(per_launch, n_sets - k0);
        BatchPtrs bp{};
        for (int k = 0; k < nk; k++) {
            bp.payload[k] = reinterpret_cast<uint8_t*>(d_payload[k0 + k]);
            for (int s = 0; s < S; s++) {
                bp.depth[k * S + s] = d_depth[(size_t)(k0 + k) * S + s];
                bp.color[k * S + s] = d_color[(size_t)(k0 + k) * S + s];
            }
        }
        HIPCHK(c, launch_fused_dense_batch(c->d_params, S, nk, c->max_points, dd, cd, sel, bp, c->stream));
    }
    if (d_counts)
        for (int k = 0; k < n_sets; k++)
            if (d_counts[k])
                HIPCHK(c, hipMemcpyAsync(d_counts[k], c->d_static_counts, sizeof(int32_t) * (S + 1), hipMemcpyDeviceToDevice, c->stream));
    if (c->kernel_timing) {
        HIPCHK(c, hipEventRecord(ev.second, c->stream));
        c->ev_pool.push_back(ev);
    }
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames_device_batch: host allocation failed (%s)", ex.what());
}

int pcs_process_frames(pcs_ctx* c, const uint16_t* const* depth, const uint8_t* const* color, int16_t* stitched,
                       size_t stitched_shorts, int write_header, int* points_per_stream, int* out_size_bytes)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!depth || !color || !stitched) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    DeviceGuard guard(c->device);
    int rc;
    for (int s = 0; s < c->n_streams; s++)
        if (!depth[s] || !color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
    {
        // Zero copy: when every raster and the stitched buffer are page-locked host memory the device can address
        // (pcs_host_malloc, hipHostMalloc, hipHostRegister), the kernels read the rasters and write the payload over PCIe
        // themselves — no staging copies, both directions of the link busy at once: 8 x 1280x720 synchronous 2.17 -> 1.58 ms
        // (staged: 0.82 H2D + kernel + 1.30 D2H; the link's own duplex limit for these volumes is 1.47 ms). The payload starts
        // 4 bytes into the buffer, so the alignment-agnostic emit kernel does the writing. PCS_ZERO_COPY=0 disables.
        const size_t need = PCS_HEADER_SHORTS + c->max_payload_points * PCS_POINT_SHORTS;
        std::vector<const uint16_t*> zd(c->n_streams);
        std::vector<const uint8_t*> zcol(c->n_streams);
        void* zout = nullptr;
        bool zero_copy = zero_copy_enabled() && stitched_shorts >= need && ((uintptr_t)stitched & 3u) == 0;
        // (every buffer is looked at even when zero copy is already off the table: a partly page-locked one is refused here
        // with a reason instead of failing inside a staging copy)
        for (int s = 0; s < c->n_streams; s++) {
            void *a = nullptr, *b = nullptr;
            const int va = host_device_view(c, depth[s], (size_t)c->h_params[s].n_points * sizeof(uint16_t), &a);
            const int vb = host_device_view(c, color[s], c->h_params[s].color_bytes, &b);
            if (va < 0 || vb < 0) return partly_locked(c, va < 0 ? "depth raster" : "colour raster", s);
            zero_copy = zero_copy && va == 1 && vb == 1;
            zd[s] = static_cast<const uint16_t*>(a); zcol[s] = static_cast<const uint8_t*>(b);
        }
        {
            const int vo = host_device_view(c, stitched, std::min(stitched_shorts, need) * sizeof(int16_t), &zout);
            if (vo < 0) return partly_locked(c, "stitched buffer", 0);
            zero_copy = zero_copy && vo == 1;
        }
        if (zero_copy) {
            int16_t* zpay = static_cast<int16_t*>(zout) + PCS_HEADER_SHORTS;
            rc = run_fused_device(c, zd.data(), zcol.data(), zpay, c->max_payload_points * PCS_POINT_SHORTS, c->d_counts, true);
            if (rc) return rc;
            std::vector<int32_t> h(c->n_streams + 1);
            HIPCHK(c, hipMemcpyAsync(h.data(), c->d_counts, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int32_t size = (int32_t)((size_t)h[c->n_streams] * PCS_POINT_BYTES);
            if (write_header) std::memcpy(stitched, &size, sizeof size);   // src/pcs-multicamera-client.cpp:394-395
            if (points_per_stream) for (int s = 0; s < c->n_streams; s++) points_per_stream[s] = h[s];
            if (out_size_bytes) *out_size_bytes = size;
            return PCS_OK;
        }
    }
    if ((rc = ensure_rasters(c))) return rc;
    for (int s = 0; s < c->n_streams; s++) {
        const StreamParams& P = c->h_params[s];
        const size_t db = (size_t)P.n_points * sizeof(uint16_t);
        HIPCHK(c, hipMemcpyAsync(c->s_depth[s], depth[s], db, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->s_color[s], color[s], P.color_bytes, hipMemcpyHostToDevice, c->stream));
    }
    const size_t max_bytes = c->max_payload_points * PCS_POINT_BYTES;
    if ((rc = ensure(c, c->s_payload, c->s_payload_cap, max_bytes + 16))) return rc;
    rc = run_fused_device(c, c->s_depth.data(), c->s_color.data(), c->s_payload, c->max_payload_points * PCS_POINT_SHORTS,
                          c->d_counts);
    if (rc) return rc;
    bool timed_out = false;
    if ((rc = take_compact_error(c, timed_out))) return rc;
    if (timed_out) {      // a placement wait expired: redo this frame-set with the count + scan + emit passes
        rc = run_fused_device(c, c->s_depth.data(), c->s_color.data(), c->s_payload,
                              c->max_payload_points * PCS_POINT_SHORTS, c->d_counts, true);
        if (rc) return rc;
    }
    std::vector<int32_t> h(c->n_streams + 1);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->d_counts, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t total = (size_t)h[c->n_streams];
    if (stitched_shorts < PCS_HEADER_SHORTS + total * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "stitched buffer holds %zu shorts, %zu needed", stitched_shorts,
                    PCS_HEADER_SHORTS + total * PCS_POINT_SHORTS);
    if (total)
        HIPCHK(c, hipMemcpy(stitched + PCS_HEADER_SHORTS, c->s_payload, total * PCS_POINT_BYTES, hipMemcpyDeviceToHost));
    const int32_t size = (int32_t)(total * PCS_POINT_BYTES);
    if (write_header) std::memcpy(stitched, &size, sizeof size);   // src/pcs-multicamera-client.cpp:394-395
    if (points_per_stream) for (int s = 0; s < c->n_streams; s++) points_per_stream[s] = h[s];
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_process_frames: host allocation failed (%s)", ex.what());
}

// ---- software-pipelined host form ------------------------------------------------------------
int pcs_submit_frames(pcs_ctx* c, const uint16_t* const* depth, const uint8_t* const* color, int* ticket)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!depth || !color || !ticket) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    for (int s = 0; s < c->n_streams; s++)
        if (!depth[s] || !color[s]) return fail(c, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", s);
    DeviceGuard guard(c->device);
    pcs_ctx::PipeSlot* sl = nullptr;
    for (auto& cand : c->pipe) if (!cand.busy) { sl = &cand; break; }
    if (!sl) return fail(c, PCS_ERR_CAPACITY, "all %d pipeline slots are in flight: collect a frame-set first", PCS_PIPELINE_DEPTH);
    if (!c->dl_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
    if (!sl->slab) {      // allocate into locals; the slot is committed only when everything succeeded
        uint8_t* slab = nullptr; std::vector<uint16_t*> dv; std::vector<uint8_t*> cv;
        int16_t* payload = nullptr; int32_t* counts = nullptr; hipEvent_t done = nullptr;
        int rc = alloc_raster_slab(c, slab, dv, cv);
        if (rc) return rc;
        const size_t max_bytes = c->max_payload_points * PCS_POINT_BYTES;
        hipError_t e = hipMalloc((void**)&payload, max_bytes + 256);
        if (e == hipSuccess) e = hipMalloc((void**)&counts, sizeof(int32_t) * (c->n_streams + 1));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (payload) (void)hipFree(payload);
            if (counts) (void)hipFree(counts);
            (void)hipFree(slab);
            return fail(c, PCS_ERR_NOMEM, "pipeline slot allocation failed: %s", hipGetErrorString(e));
        }
        sl->slab = slab; sl->depth = std::move(dv); sl->color = std::move(cv);
        sl->payload = payload; sl->counts = counts; sl->done = done;
    }
    // (Letting the kernel read page-locked rasters itself here — zero copy, as pcs_process_frames does — was measured: the
    // kernel's reads over PCIe and the previous frame-set's download DMA get in each other's way, 1.93 vs 1.64 ms per 8 x 720p
    // frame-set. The staged upload stays.)
    for (int s = 0; s < c->n_streams; s++) {
        const StreamParams& P = c->h_params[s];
        HIPCHK(c, hipMemcpyAsync(sl->depth[s], depth[s], (size_t)P.n_points * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(sl->color[s], color[s], P.color_bytes, hipMemcpyHostToDevice, c->stream));
    }
    // the three-pass compaction only: its result never needs a host-side retry
    int rc = run_fused_device(c, sl->depth.data(), sl->color.data(), sl->payload, c->max_payload_points * PCS_POINT_SHORTS,
                              sl->counts, true);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(sl->done, c->stream));
    sl->busy = true;
    sl->ticket = c->next_ticket++;
    *ticket = sl->ticket;
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_submit_frames: host allocation failed (%s)", ex.what());
}

int pcs_collect_frames(pcs_ctx* c, int ticket, int16_t* stitched, size_t stitched_shorts, int write_header,
                       int* points_per_stream, int* out_size_bytes)
try {
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!stitched) return fail(c, PCS_ERR_INVALID_ARG, "stitched is NULL");
    pcs_ctx::PipeSlot* sl = nullptr;
    for (auto& cand : c->pipe) if (cand.busy && cand.ticket == ticket) { sl = &cand; break; }
    if (!sl) return fail(c, PCS_ERR_INVALID_ARG, "ticket %d is not in flight", ticket);
    if (ticket != c->next_collect) return fail(c, PCS_ERR_INVALID_ARG, "tickets are collected in submission order: %d is next", c->next_collect);
    DeviceGuard guard(c->device);
    // whatever happens below, the slot is released and the ticket consumed: a failed frame-set is dropped, it must
    // not wedge the pipeline (later tickets could otherwise never be collected)
    // ... and on a FAILED exit both streams are drained first: the slot's kernels or its download may still be in flight, and
    // the next submit would reuse its rasters, payload and counts under them
    struct Release {
        pcs_ctx* c; pcs_ctx::PipeSlot* sl; bool ok = false;
        ~Release()
        {
            if (!ok) { (void)hipStreamSynchronize(c->stream); if (c->dl_stream) (void)hipStreamSynchronize(c->dl_stream); }
            sl->busy = false; c->next_collect++;
        }
    } release{c, sl};
    HIPCHK(c, hipStreamWaitEvent(c->dl_stream, sl->done, 0));
    std::vector<int32_t> h(c->n_streams + 1);
    size_t total;
    if (has_pred(c->flags)) {        // the payload size is data dependent: counts first
        HIPCHK(c, hipMemcpyAsync(h.data(), sl->counts, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->dl_stream));
        HIPCHK(c, hipStreamSynchronize(c->dl_stream));
        total = (size_t)h[c->n_streams];
    } else {
        for (int s = 0; s < c->n_streams; s++) h[s] = (int32_t)((c->h_params[s].n_points + c->downsample - 1) / c->downsample);
        total = c->max_payload_points;
        h[c->n_streams] = (int32_t)total;
    }
    if (stitched_shorts < PCS_HEADER_SHORTS + total * PCS_POINT_SHORTS) {
        (void)hipStreamSynchronize(c->dl_stream);     // the frame-set is dropped; the slot is usable again
        return fail(c, PCS_ERR_CAPACITY, "stitched buffer holds %zu shorts, %zu needed", stitched_shorts,
                    PCS_HEADER_SHORTS + total * PCS_POINT_SHORTS);
    }
    if (total)
        HIPCHK(c, hipMemcpyAsync(stitched + PCS_HEADER_SHORTS, sl->payload, total * PCS_POINT_BYTES, hipMemcpyDeviceToHost, c->dl_stream));
    HIPCHK(c, hipStreamSynchronize(c->dl_stream));
    const int32_t size = (int32_t)(total * PCS_POINT_BYTES);
    if (write_header) std::memcpy(stitched, &size, sizeof size);
    if (points_per_stream) for (int s = 0; s < c->n_streams; s++) points_per_stream[s] = h[s];
    if (out_size_bytes) *out_size_bytes = size;
    release.ok = true;
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_collect_frames: host allocation failed (%s)", ex.what());
}

int pcs_deproject(pcs_ctx* c, int stream, const uint16_t* depth, float* vertices, float* texcoords)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (stream < 0 || stream >= c->n_streams) return fail(c, PCS_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (!depth || !vertices || !texcoords) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    DeviceGuard guard(c->device);
    const StreamParams& P = c->h_params[stream];
    const size_t n = P.n_points;
    int rc;
    if ((rc = ensure_rasters(c))) return rc;
    if ((rc = ensure(c, c->s_vertices, c->s_vertices_cap, n * 3 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->s_texcoords, c->s_texcoords_cap, n * 2 * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s_depth[stream], depth, n * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_deproject(c->d_params, stream, P.n_points, c->s_depth[stream], c->s_vertices, c->s_texcoords, c->stream));
    HIPCHK(c, hipMemcpyAsync(vertices, c->s_vertices, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(texcoords, c->s_texcoords, n * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PCS_OK;
}

// ---- a7 --------------------------------------------------------------------------------------
int pcs_stitch_device(pcs_ctx* c, const int16_t* const* d_cam_payload, const int* cam_points, int n_cams,
                      int downsample, int16_t* d_stitched_payload, size_t stitched_shorts, int* total_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_cams < 0 || (n_cams > 0 && (!d_cam_payload || !cam_points || !d_stitched_payload)))
        return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (downsample < 1) return fail(c, PCS_ERR_INVALID_ARG, "downsample %d < 1", downsample);
    DeviceGuard guard(c->device);
    size_t need = 0;
    for (int i = 0; i < n_cams; i++) {
        if (cam_points[i] < 0) return fail(c, PCS_ERR_INVALID_ARG, "camera %d: negative point count", i);
        need += ((size_t)cam_points[i] + downsample - 1) / downsample;
    }
    if (stitched_shorts < need * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "stitched payload holds %zu shorts, %zu needed", stitched_shorts, need * PCS_POINT_SHORTS);
    size_t out = 0;
    for (int i = 0; i < n_cams; i++) {
        const size_t kept = ((size_t)cam_points[i] + downsample - 1) / downsample;
        if (!kept) continue;
        if (downsample == 1)
            HIPCHK(c, hipMemcpyAsync(d_stitched_payload + out * PCS_POINT_SHORTS, d_cam_payload[i],
                                     kept * PCS_POINT_BYTES, hipMemcpyDeviceToDevice, c->stream));
        else
            HIPCHK(c, launch_stitch(d_cam_payload[i], (uint32_t)cam_points[i], downsample,
                                    d_stitched_payload + out * PCS_POINT_SHORTS, c->stream));
        out += kept;
    }
    if (total_points) *total_points = (int)out;
    return PCS_OK;
}

// ---- the centre's re-transform of packed payloads (src/pcs-multicamera-optimized.cpp:226-265, 289) ----------
int pcs_transform_payloads_device(pcs_ctx* c, int n_cams, const pcs_payload_desc* cams, int downsample,
                                  int16_t* d_stitched_payload, size_t stitched_shorts, int* points_per_cam, int* total_points)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (n_cams < 0 || (n_cams > 0 && (!cams || !d_stitched_payload))) return fail(c, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (downsample < 1) return fail(c, PCS_ERR_INVALID_ARG, "downsample %d < 1", downsample);
    size_t need = 0;
    std::vector<size_t> first(n_cams > 0 ? n_cams : 0), kept(n_cams > 0 ? n_cams : 0);
    for (int i = 0; i < n_cams; i++) {
        if (cams[i].n_points < 0) return fail(c, PCS_ERR_INVALID_ARG, "camera %d: negative point count", i);
        if (cams[i].n_points > 0 && !cams[i].d_payload) return fail(c, PCS_ERR_INVALID_ARG, "camera %d: NULL payload", i);
        first[i] = need;
        // floor: the reference sizes the decoded cloud with size / downsample (src/pcs-multicamera-optimized.cpp:230) and every
        // later step iterates that width; pcs_stitch_device's ceiling is the OTHER program's loop (pcs-multicamera-client.cpp:388)
        kept[i] = (size_t)cams[i].n_points / (size_t)downsample;
        need += kept[i];
    }
    if (need * PCS_POINT_BYTES > 0x7FFFFFFFull)
        return fail(c, PCS_ERR_INVALID_ARG, "stitched payload of %zu points exceeds the int32 byte-count header", need);
    if (stitched_shorts < need * PCS_POINT_SHORTS)
        return fail(c, PCS_ERR_CAPACITY, "stitched payload holds %zu shorts, %zu needed", stitched_shorts, need * PCS_POINT_SHORTS);
    // a tile reads all of its input before it writes, and tiles of one camera cover disjoint ranges: a camera may be transformed
    // in place (same start, stride 1); any other meeting of an input with an output slice would be a race between workgroups
    for (int i = 0; i < n_cams; i++) {
        const uintptr_t o0 = (uintptr_t)(d_stitched_payload + first[i] * PCS_POINT_SHORTS), o1 = o0 + kept[i] * PCS_POINT_BYTES;
        for (int j = 0; j < n_cams; j++) {
            const uintptr_t i0 = (uintptr_t)cams[j].d_payload, i1 = i0 + (size_t)cams[j].n_points * PCS_POINT_BYTES;
            if (o0 < i1 && i0 < o1 && !(i == j && i0 == o0 && downsample == 1))
                return fail(c, PCS_ERR_INVALID_ARG, "camera %d's output slice overlaps camera %d's input (only an exact in-place "
                            "transform with downsample 1 is allowed)", i, j);
        }
    }
    DeviceGuard guard(c->device);
    for (int base = 0; base < n_cams; base += kXformBatch) {
        const int n = std::min(kXformBatch, n_cams - base);
        XformBatch xb;
        std::memset(&xb, 0, sizeof xb);
        uint32_t max_out = 0;
        for (int k = 0; k < n; k++) {
            const int i = base + k;
            XformCloud& x = xb.c[k];
            x.in = cams[i].d_payload;
            x.out = reinterpret_cast<uint8_t*>(d_stitched_payload + first[i] * PCS_POINT_SHORTS);
            x.n_out = (uint32_t)kept[i];
            x.ds = (uint32_t)downsample;
            std::memcpy(x.M, cams[i].transform, sizeof x.M);
            max_out = std::max(max_out, x.n_out);
        }
        HIPCHK(c, launch_transform_payloads(xb, n, max_out, c->stream));
    }
    if (points_per_cam) for (int i = 0; i < n_cams; i++) points_per_cam[i] = (int)kept[i];
    if (total_points) *total_points = (int)need;
    return PCS_OK;
}

// ---- plumbing --------------------------------------------------------------------------------
int pcs_set_stream(pcs_ctx* c, void* hip_stream)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return PCS_OK;
}

void* pcs_get_stream(pcs_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

// ---- a stream that really runs beside another one ---------------------------------------------------------------------------
// The HIP runtime maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), round robin as they are created,
// and two streams on one queue do not overlap at all: two contexts used in turn then run exactly as one. So a stream is not assumed
// to be concurrent, it is SEEN to be: a no-op launched on the candidate must finish while a 300 us spin still occupies the other.
namespace {
__global__ void pcs_spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();                    // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void pcs_nop_kernel() {}
}  // namespace

int pcs_pick_concurrent_stream(void* busy_stream, void** out_stream)
{
    if (!out_stream) return PCS_ERR_INVALID_ARG;
    *out_stream = nullptr;
    hipStream_t busy = static_cast<hipStream_t>(busy_stream);
    hipEvent_t spun = nullptr, done = nullptr;
    if (hipEventCreateWithFlags(&spun, hipEventDisableTiming) != hipSuccess) return PCS_ERR_HIP;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(spun); return PCS_ERR_HIP; }
    std::vector<hipStream_t> rejected;
    hipStream_t found = nullptr;
    for (int attempt = 0; attempt < 6 && !found; attempt++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        hipLaunchKernelGGL(pcs_spin_kernel, dim3(1), dim3(64), 0, busy, 30000ll);       // ~300 us
        (void)hipEventRecord(spun, busy);
        hipLaunchKernelGGL(pcs_nop_kernel, dim3(1), dim3(64), 0, cand);
        (void)hipEventRecord(done, cand);
        (void)hipEventSynchronize(done);
        const bool beside = hipEventQuery(spun) == hipErrorNotReady;      // the spin was still going when the other stream finished
        (void)hipEventSynchronize(spun);
        (void)hipGetLastError();
        if (beside) found = cand; else rejected.push_back(cand);           // (kept alive until the search ends: the next stream then takes another queue)
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipEventDestroy(spun); (void)hipEventDestroy(done);
    *out_stream = found;
    return PCS_OK;
}

int pcs_use_stream_beside(pcs_ctx* c, pcs_ctx* other)
{
    if (!c || !other || c == other) return PCS_ERR_INVALID_ARG;
    if (c->device != other->device) return fail(c, PCS_ERR_INVALID_ARG, "the two contexts are on different devices (%d, %d)", c->device, other->device);
    if (c->stream != c->own_stream) return fail(c, PCS_ERR_INVALID_ARG, "the context runs on an adopted stream (pcs_set_stream): its owner picks it");
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    void* found = nullptr;
    const int rc = pcs_pick_concurrent_stream(other->stream, &found);
    if (rc != PCS_OK) return fail(c, rc, "could not probe for a concurrent stream");
    if (!found) return 0;                                   // none of six candidates overlapped: the context keeps its stream
    (void)hipStreamDestroy(c->own_stream);
    c->own_stream = static_cast<hipStream_t>(found);
    c->stream = c->own_stream;
    return 1;
}

int pcs_synchronize(pcs_ctx* c)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bool timed_out = false;
    int rc = take_compact_error(c, timed_out);
    if (rc) return rc;
    if (timed_out)
        return fail(c, PCS_ERR_HIP, "single-pass compaction: a placement wait expired; the payload of the frame-set(s) since "
                    "the last pcs_synchronize is invalid — re-submit them (the context now uses the three-pass path)");
    return PCS_OK;
}

int pcs_timer_begin(pcs_ctx* c)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipEventRecord(c->ev_begin, c->stream));
    return PCS_OK;
}

int pcs_timer_end(pcs_ctx* c)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipEventRecord(c->ev_end, c->stream));
    return PCS_OK;
}

int pcs_timer_elapsed_ms(pcs_ctx* c, float* ms)
{
    if (!c || !ms) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipEventSynchronize(c->ev_end));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev_begin, c->ev_end));
    return PCS_OK;
}

int pcs_kernel_timing(pcs_ctx* c, int enable)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    c->kernel_timing = enable != 0;
    return PCS_OK;
}

int pcs_kernel_times_ms(pcs_ctx* c, float* ms, int capacity, int* n)
try {
    if (!c || !n || (capacity > 0 && !ms)) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int k = 0;
    for (auto& pr : c->ev_pool) {
        if (k < capacity) HIPCHK(c, hipEventElapsedTime(&ms[k], pr.first, pr.second));
        k++;
        c->ev_free.push_back(pr);
    }
    c->ev_pool.clear();
    *n = k;
    return PCS_OK;
} catch (const std::exception& ex) {
    return fail(c, PCS_ERR_NOMEM, "pcs_kernel_times_ms: host allocation failed (%s)", ex.what());
}

int pcs_host_malloc(pcs_ctx* c, void** h_ptr, size_t bytes)
{
    if (!c || !h_ptr) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    hipError_t e = hipHostMalloc(h_ptr, std::max<size_t>(bytes, 16), hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, PCS_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return PCS_OK;
}

int pcs_host_free(pcs_ctx* c, void* h_ptr)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->dl_stream) HIPCHK(c, hipStreamSynchronize(c->dl_stream));
    c->zc_cache.clear();
    HIPCHK(c, hipHostFree(h_ptr));
    return PCS_OK;
}

int pcs_host_register(pcs_ctx* c, void* h_ptr, size_t bytes)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    if (!h_ptr || !bytes) return fail(c, PCS_ERR_INVALID_ARG, "pcs_host_register: NULL pointer or zero size");
    DeviceGuard guard(c->device);
    hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, PCS_ERR_HIP, "hipHostRegister(%p, %zu) failed: %s", h_ptr, bytes, hipGetErrorString(e));
    }
    c->zc_cache.clear();       // verdicts about this range ("pageable", "locked in part") are out of date
    return PCS_OK;
}

int pcs_host_unregister(pcs_ctx* c, void* h_ptr)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->dl_stream) HIPCHK(c, hipStreamSynchronize(c->dl_stream));     // a pipelined download may still target the range
    c->zc_cache.clear();
    HIPCHK(c, hipHostUnregister(h_ptr));
    return PCS_OK;
}

int pcs_device_malloc(pcs_ctx* c, void** d_ptr, size_t bytes)
{
    if (!c || !d_ptr) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    hipError_t e = hipMalloc(d_ptr, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return fail(c, PCS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return PCS_OK;
}

int pcs_device_free(pcs_ctx* c, void* d_ptr)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d_ptr));
    return PCS_OK;
}

int pcs_memcpy_h2d(pcs_ctx* c, void* d_dst, const void* h_src, size_t bytes)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PCS_OK;
}

int pcs_memcpy_d2h(pcs_ctx* c, void* h_dst, const void* d_src, size_t bytes)
{
    if (!c) return PCS_ERR_INVALID_ARG;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PCS_OK;
}

}  // extern "C"
