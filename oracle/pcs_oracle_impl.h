/*
 * pcs_oracle_impl.h — per-pixel / per-point bodies shared by pcs_oracle.c (scalar oracle) and
 * pcs_oracle_simd.c (timed CPU baseline). TEST INFRASTRUCTURE ONLY — see pcs_oracle.h.
 * Must be compiled with -ffp-contract=off.
 */
#ifndef PCS_ORACLE_IMPL_H
#define PCS_ORACLE_IMPL_H
#include "pcs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* _mm_cvttps_epi32 / cvttss2si (:438-439 and the short() casts :581-583): truncate toward zero;
 * NaN and anything outside [-2^31, 2^31) give 0x80000000. */
static inline int32_t pcs_o_cvtt(float f)
{
    if (f >= -2147483648.0f && f < 2147483648.0f) return (int32_t)f;
    return INT32_MIN;
}

static inline int pcs_o_distortion_active(const pcs_intrinsics* in)
{
    if (in->model == PCS_DISTORTION_NONE) return 0;
    for (int k = 0; k < 5; k++) if (in->coeffs[k] != 0.0f) return 1;
    return 0;   /* a model with all-zero coefficients is the identity on finite input */
}

/* ------------------------------------------------------------------------------------------ *
 * a5 - rs2::pointcloud::calculate + map_to (call sites :198-199, :288-289). Third-party
 * librealsense2; restated from SURVEY.md Appendix E (rsutil.h rs2_deproject_pixel_to_point,
 * rs2_transform_point_to_point, rs2_project_point_to_pixel; pointcloud.cpp pixel_to_texcoord).
 * All products and sums are individually rounded, evaluated left to right. One raster row.
 * ------------------------------------------------------------------------------------------ */
static inline void pcs_o_deproject_row_ex(const pcs_stream_config* sc, const uint16_t* depth, int r,
                                          float* vertices, float* texcoords, int half_pixel)
{
    const pcs_intrinsics* di = &sc->depth;
    const pcs_intrinsics* ci = &sc->color;
    const float* R = sc->depth_to_color.rotation;      /* column-major */
    const float* t = sc->depth_to_color.translation;
    const int W = di->width;
    const int ddist = pcs_o_distortion_active(di), cdist = pcs_o_distortion_active(ci);
    const float wc = (float)ci->width, hc = (float)ci->height;
    const float my0 = ((float)r - di->ppy) / di->fy;

    for (int c = 0; c < W; c++) {
        const size_t i = (size_t)r * (size_t)W + (size_t)c;
        float z = sc->depth_scale * (float)depth[i];
        float mx = ((float)c - di->ppx) / di->fx;
        float my = my0;
        if (ddist) {   /* INVERSE_BROWN_CONRADY branch of rs2_deproject_pixel_to_point */
            const float* k = di->coeffs;
            float r2 = mx * mx + my * my;
            float f = 1 + k[0] * r2 + k[1] * r2 * r2 + k[4] * r2 * r2 * r2;
            float ux = mx * f + 2 * k[2] * mx * my + k[3] * (r2 + 2 * mx * mx);
            float uy = my * f + 2 * k[3] * mx * my + k[2] * (r2 + 2 * my * my);
            mx = ux; my = uy;
        }
        float X = z * mx, Y = z * my, Z = z;
        float u = 0.0f, v = 0.0f;
        if (Z != 0.0f) {
            /* rs2_transform_point_to_point */
            float P0 = R[0] * X + R[3] * Y + R[6] * Z + t[0];
            float P1 = R[1] * X + R[4] * Y + R[7] * Z + t[1];
            float P2 = R[2] * X + R[5] * Y + R[8] * Z + t[2];
            /* rs2_project_point_to_pixel */
            float x = P0 / P2, y = P1 / P2;
            if (cdist) {
                const float* k = ci->coeffs;
                float r2 = x * x + y * y;
                float f = 1 + k[0] * r2 + k[1] * r2 * r2 + k[4] * r2 * r2 * r2;
                x *= f; y *= f;
                float dx = x + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
                float dy = y + 2 * k[3] * x * y + k[2] * (r2 + 2 * y * y);
                x = dx; y = dy;
            }
            float px = x * ci->fx + ci->ppx;
            float py = y * ci->fy + ci->ppy;
            /* pixel_to_texcoord; older librealsense releases: (pixel + 0.5) / size  (PCS_FLAG_TEXCOORD_HALF_PIXEL) */
            if (half_pixel) { px = px + 0.5f; py = py + 0.5f; }
            u = px / wc;
            v = py / hc;
        }
        vertices[3 * i + 0] = X; vertices[3 * i + 1] = Y; vertices[3 * i + 2] = Z;
        texcoords[2 * i + 0] = u; texcoords[2 * i + 1] = v;
    }
}

static inline void pcs_o_deproject_row(const pcs_stream_config* sc, const uint16_t* depth, int r,
                                       float* vertices, float* texcoords)
{
    pcs_o_deproject_row_ex(sc, depth, r, vertices, texcoords, 0);
}

/* colour lookup :431-452 - x = fma(u, W, .5); truncate; clamp; idx = x*bpp + y*stride */
static inline size_t pcs_o_color_index(const pcs_stream_config* sc, const float* uv)
{
    const int W = sc->color.width, H = sc->color.height;
    float xf = fmaf(uv[0], (float)W, 0.5f);
    float yf = fmaf(uv[1], (float)H, 0.5f);
    int32_t xi = pcs_o_cvtt(xf), yi = pcs_o_cvtt(yf);
    if (xi < 0) xi = 0;
    if (yi < 0) yi = 0;
    if (xi > W - 1) xi = W - 1;
    if (yi > H - 1) yi = H - 1;
    return (size_t)xi * (size_t)sc->color_bpp + (size_t)yi * (size_t)sc->color_stride;
}

/* One point of a2. Writes 5 shorts. */
static inline void pcs_o_pack_point(const pcs_stream_config* sc, const float* vtx, const float* uv,
                                    const uint8_t* color, int16_t* o)
{
    const float* M = sc->cam_to_world;
    const size_t idx = pcs_o_color_index(sc, uv);

    /* rigid transform :455-485 - x*col0 + translation FIRST, then + y*col1, then + z*col2 (3 FMAs),
     * then a separate float multiply by 1000.0f :488-491, then short(float) :581-583 = cvttss2si
     * followed by keeping the low 16 bits. */
    for (int r = 0; r < 3; r++) {
        float a = fmaf(vtx[0], M[4 * r + 0], M[4 * r + 3]);
        a = fmaf(vtx[1], M[4 * r + 1], a);
        a = fmaf(vtx[2], M[4 * r + 2], a);
        a = a * 1000.0f;
        o[r] = (int16_t)(uint16_t)((uint32_t)pcs_o_cvtt(a) & 0xFFFFu);
    }
    /* colour pack :584-585 - short3 = R + (G << 8), short4 = B */
    o[3] = (int16_t)(uint16_t)(color[idx] + ((unsigned)color[idx + 1] << 8));
    o[4] = (int16_t)color[idx + 2];
}

#endif
