// pcs_vox_tiling.h — how a launch of the voxel pipeline's raster reader (pcs_kernels.hip: pcs_fused_voxel_partials_kernel) cuts
// the rasters into workgroups. Plain C++ without a HIP header, so that the host logic — every square of every raster is visited
// exactly once, whatever the geometry — is tested on the CPU (tests/test_vox_tiling.py compiles this file with g++); the kernel and
// its launcher #include it inside their namespace.
#pragma once
#include <stdint.h>

// How a launch of the raster reader cuts the rasters into workgroups, in SQUARES of 64 x 64 pixels (one round of the 512
// lanes: 8 lanes x 8 pixels per row, 64 rows; a wavefront's loads cover 8 full 128-byte lines of Z16). A voxel of a few dozen
// pixels across lies inside one square patch but in three or four 8-row strips, so square patches leave 2.2x (50 mm) to 4x
// (200 mm) fewer partials behind than runs of consecutive pixels; more squares per table, fewer still. Two tiers:
//   head: square-rows [0, ya) of every stream in patches of rx x ry squares (one table each),
//   tail: square-rows [ya, ..) in patches of rxb x 1 squares — smaller work items, dealt LAST.
// Workgroups start in the order of their linear id: all streams' head patches come first, all tail patches after them. The
// kernel is VALU-bound and a head workgroup lives for a fifth of the launch, so the chip's last round of workgroups decides
// when the launch ends: short items dealt last level it, short items in the middle of the order (where each stream's remainder
// row used to sit, and at the price of a full patch: its two rounds below the raster ran on nothing) do not. 16 x 1080p at
// 50 mm, one call: 0.184 -> 0.178 ms; passing over the empty rounds alone, in the old order: 0.185.
struct VoxTiling {
    int rx, ry;          // head patch, in squares (rx == 0: consecutive pixels, `rounds` x 4096 per workgroup; any raster)
    int rxb;             // tail patch: rxb x 1 squares
    int ya;              // first square-row of the tail (a multiple of ry)
    int gxa, gxb;        // patches per square-row in the head / tail (from the launch's widest raster)
    int na, nb;          // head / tail patches per stream; gridDim.x == na + nb
};

// The tiling of a launch whose largest raster is max_w x max_h pixels, squares of 64 x `rows` pixels, head patches of rx x ry
// squares (rx >= 1). The square-rows that do not fill a head patch go to the tail; tail_pct moves more of the raster's square-rows
// there (in whole head patch rows). Tail patches are rx x 1 squares under head patches of two or more rows, (rx / 2) x 1 otherwise;
// a head of one square per table has no tail.
inline VoxTiling vox_tiling_make(unsigned max_w, unsigned max_h, unsigned rows, unsigned rx, unsigned ry, int tail_pct)
{
    VoxTiling tl{};
    const unsigned sx = (max_w + 63u) / 64u, sy = (max_h + rows - 1u) / rows;
    const unsigned rxb = ry >= 2u ? rx : (rx / 2u > 1u ? rx / 2u : 1u);
    unsigned tail_rows = sy % ry;
    const unsigned pct = tail_pct < 0 ? 0u : tail_pct > 100 ? 100u : (unsigned)tail_pct;
    unsigned want = (sy * pct + 50u) / 100u;
    if (want > sy) want = sy;
    while (tail_rows < want && tail_rows + ry <= sy) tail_rows += ry;
    if (ry == 1u && rxb == rx) tail_rows = 0;                          // (one square per table: nothing smaller to deal)
    tl.rx = (int)rx; tl.ry = (int)ry; tl.rxb = (int)rxb; tl.ya = (int)(sy - tail_rows);
    tl.gxa = (int)((sx + rx - 1u) / rx); tl.gxb = (int)((sx + rxb - 1u) / rxb);
    tl.na = tl.gxa * (tl.ya / (int)ry); tl.nb = tl.gxb * (int)tail_rows;
    return tl;
}

// Squares (4096 pixels) that share one 2048-slot table on the patch route, and the head patch's width in squares: by the leaf
// (the voxels under a table fall roughly with the square of the leaf; a wrong guess costs speed, never bits — runs that find no slot
// go out as partials of their own), by the tail that follows (`regions`: a warm bucket call's workgroups end with the dearer flush
// and gain most from fewer, larger tables) and capped so that the launch still fills the chip twice (launch_squares = the launch's
// pixels / 4096; 3 % slack: 16 x 1080p are 8112 squares and take 8 per table). The measurements behind the thresholds:
// pcs_kernels.hip, launch_fused_voxel_partials. force > 0 (PCS_VOXEL_ROUNDS) overrides the rule.
struct VoxPatchShape { int squares, rx; };
inline VoxPatchShape vox_patch_shape(unsigned leaf_mm, bool regions, unsigned long long launch_squares, int force)
{
    unsigned long long cap = (launch_squares + launch_squares / 32u) / 1024u;
    if (cap < 1u) cap = 1u;
    const unsigned long long by_leaf = leaf_mm < 30u ? 1u : regions ? (leaf_mm >= 150u ? 8u : leaf_mm >= 40u ? 4u : 2u) : (leaf_mm >= 45u ? 4u : 2u);
    unsigned long long r = by_leaf < cap ? by_leaf : cap;
    if (force > 0) r = (unsigned long long)force;
    const int squares = r >= 8u ? 8 : r >= 4u ? 4 : r >= 2u ? 2 : 1;
    return VoxPatchShape{squares, squares == 8 ? 4 : squares >= 2 ? 2 : 1};
}

// Workgroup `lin` (= blockIdx.y * gridDim.x + blockIdx.x) of a launch over n_streams streams (= gridDim.y): its stream, the
// first square of its patch and the patch's extent in squares. All streams' head patches come first, then all tail patches.
// A macro, so that the kernel's code is this text itself (a function taking references compiled to a different register
// allocation of the whole kernel; the measured build is the one with the statements in place) and the CPU test runs the same text.
#define PCS_VOX_TILING_DECODE(tl, lin, n_streams, s, sq_x0, sq_y0, nrx, nry)                                              \
    {                                                                                                                     \
        const uint32_t head_ = (uint32_t)(tl).na * (n_streams);                                                           \
        if ((lin) < head_) {                                                                                              \
            s = (int)((lin) / (uint32_t)(tl).na);                                                                         \
            const uint32_t q_ = (lin) % (uint32_t)(tl).na;                                                                \
            sq_x0 = (q_ % (uint32_t)(tl).gxa) * (uint32_t)(tl).rx; sq_y0 = (q_ / (uint32_t)(tl).gxa) * (uint32_t)(tl).ry; \
            nrx = (uint32_t)(tl).rx; nry = (uint32_t)(tl).ry;                                                             \
        } else {                                                                                                          \
            const uint32_t r_ = (lin) - head_;                                                                            \
            s = (int)(r_ / (uint32_t)(tl).nb);                                                                            \
            const uint32_t q_ = r_ % (uint32_t)(tl).nb;                                                                   \
            sq_x0 = (q_ % (uint32_t)(tl).gxb) * (uint32_t)(tl).rxb; sq_y0 = (uint32_t)(tl).ya + q_ / (uint32_t)(tl).gxb;  \
            nrx = (uint32_t)(tl).rxb; nry = 1u;                                                                           \
        }                                                                                                                 \
    }
