// pcs_voxel_agg.h — the pieces of the voxel pre-aggregation that both its sources share: the payload reader
// (pcs_voxel.hip) and the raster reader that never materialises the stitched payload (pcs_kernels.hip).
// Device code; #included INSIDE each translation unit's anonymous namespace.

// What one workgroup of the pre-aggregation kernel knows about one voxel: the sums over its (<= 8192) points
// that fall into it (|sum| <= 8192 * 32768 = 2^28 fits an int).
struct alignas(32) VoxelPartial {      // 32 B: one sector per gathered partial in the segmented mean
    int          sx, sy, sz;
    unsigned int r, g, b, n, pad;
};

// Bits one axis needs: voxel indices run over 0 .. floor(32767/leaf) + ceil(32768/leaf) <= 65536/leaf + 1.
// Packing the three axes into 3*bits (instead of a fixed 51) saves whole radix passes for realistic leaves.
inline unsigned int axis_bits(int leaf)
{
    const unsigned int max_index = 32767u / (unsigned)leaf + (32768u + (unsigned)leaf - 1u) / (unsigned)leaf;
    unsigned int b = 1;
    while ((1u << b) <= max_index) b++;
    return b;
}

// floor(v / leaf) + bias, bias = ceil(32768 / leaf), in THREE full-rate instructions: convert, fused multiply-add, truncate —
//     k = (unsigned)fmaf((float)v, inv, c),   inv = fl(1 / leaf),   c = fl((bias * leaf + 0.5) / leaf).
// The exact value t = (v + 0.5) / leaf + bias is positive and never closer than 0.5 / leaf to an integer, while the
// float result is off by less than 0.008 / leaf (|v| * ulp(inv)/2 + ulp(c)/2 + ulp(t)/2 with t < 2^17 / leaf), so the
// truncation is floor(v / leaf) + bias for every int16 v and every leaf in 1 .. 32767 (tested exhaustively over all
// 2^31 pairs; the host re-checks the 65 536 values of the leaf in use before a launch, pcs_voxel.hip: div_for).
// (Rounds 1-3 used umulhi(v + bias * leaf, ceil(2^32 / leaf)): v_mul_hi_u32 issues at a quarter of the rate, and the
// pre-aggregation is VALU-bound — three of them per point were 9 % of its cycles.)
struct VoxelDiv {
    float inv, c;
    __device__ __forceinline__ unsigned int operator()(int v) const { return (unsigned int)__builtin_fmaf((float)v, inv, c); }
};

__device__ __forceinline__ unsigned long long voxel_key(const VoxelDiv& dv, int x, int y, int z, unsigned int bits)
{
    // z major, x fastest: (z,y,x) voxel order. bits <= 16, so the x and y fields share one 32-bit word.
    const unsigned int xy = dv(x) | (dv(y) << bits);
    return ((unsigned long long)dv(z) << (2u * bits)) | xy;
}

// LDS hash table of one pre-aggregation workgroup
constexpr int kSlots = 2048, kProbe = 12;
// control words of a voxel call (unsigned int ctl[64]): [0] partials, [1] voxels; on their own line: [32..33] OR of the
// voxel keys written, [34..35] OR of their complements (which key bits vary at all: pcs_voxel.hip's sort drops the others)
constexpr unsigned int kVoxCtlOr = 32, kVoxCtlOrn = 34;
constexpr unsigned long long kEmptyKey = ~0ull;
// The bucket tail's partition (pcs_voxel.hip): at most kVoxBuckets key ranges. The pre-aggregation kernels know the number because,
// on a warm workspace, they place their partials into the buckets' regions themselves (VoxelStage::regions).
constexpr unsigned int kVoxBuckets = 1024;

// Probe sequence of a key: double hashing (start and an odd stride from one multiplicative hash), so a crowded table
// costs 1/(1 - load) probes on average instead of linear probing's clusters — at 2/3 load a key fails to find a slot
// within kProbe probes 1 time in 100, not 1 in 4. The hash multiplies the key's two 24-bit halves (3 * bits <= 48) by odd
// 24-bit constants with v_mul_u32_u24 / v_mad_u32_u24 (full rate; the 64-bit multiply of rounds 1-3 was three
// quarter-rate instructions per run) and takes the TOP bits of the low word, which depend on every input bit.
struct VoxelProbe {
    unsigned int first, step;
    __device__ __forceinline__ explicit VoxelProbe(unsigned long long key)
    {
        const unsigned int lo = (unsigned int)key, hi = (unsigned int)(key >> 32);
        const unsigned int m = __umul24(lo, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbit(hi, lo, 24), 0x85EBCBu);
        first = m >> 21;                                                   // 11 bits
        step = ((m >> 10) & (unsigned)(kSlots - 1)) | 1u;                  // odd: visits every slot of the 2^11 table
    }
    __device__ __forceinline__ unsigned int next(unsigned int h) const { return (h + step) & (unsigned)(kSlots - 1); }
};
