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

// floor(v / leaf) + bias, bias = ceil(32768 / leaf): the biased numerator u = v + bias*leaf is in [0, 2^17), and
// floor(u / leaf) == (u * magic) >> 32 with magic = ceil(2^32 / leaf): the error magic*leaf - 2^32 is below leaf <= 2^15,
// so u * error < 2^32 for every u < 2^17 (the host still checks all 65 536 of them before the launch). leaf = 1 has no
// 32-bit magic: magic = 0 and the quotient is u itself (`pass` = all ones). No branch: a (uniform) test per division put
// three branches and a 13-instruction divide into every point of the unrolled loops.
struct VoxelDiv {
    unsigned int leaf, bias_leaf, magic;
    __device__ __forceinline__ unsigned int operator()(int v) const
    {
        const unsigned int u = (unsigned int)(v + (int)bias_leaf);
        const unsigned int pass = magic ? 0u : ~0u;          // uniform: an SGPR select, hoisted out of the loops
        return __umulhi(u, magic) + (u & pass);
    }
};

__device__ __forceinline__ unsigned long long voxel_key(const VoxelDiv& dv, int x, int y, int z, unsigned int bits)
{
    const unsigned long long kx = dv(x), ky = dv(y), kz = dv(z);
    return (kz << (2 * bits)) | (ky << bits) | kx;      // z major, x fastest: (z,y,x) voxel order
}

// LDS hash table of one pre-aggregation workgroup
constexpr int kSlots = 2048, kProbe = 12;
// control words of a voxel call (unsigned int ctl[64]): [0] partials, [1] voxels; on their own line: [32..33] OR of the
// voxel keys written, [34..35] OR of their complements (which key bits vary at all: pcs_voxel.hip's sort drops the others)
constexpr unsigned int kVoxCtlOr = 32, kVoxCtlOrn = 34;
constexpr unsigned long long kEmptyKey = ~0ull;

// Probe sequence of a key: double hashing (start and an odd stride from two multiplicative hashes), so a crowded table
// costs 1/(1 - load) probes on average instead of linear probing's clusters — at 2/3 load a key fails to find a slot
// within kProbe probes 1 time in 100, not 1 in 4.
struct VoxelProbe {
    unsigned int first, step;
    __device__ __forceinline__ explicit VoxelProbe(unsigned long long key)
    {
        const unsigned long long m = key * 0x9E3779B97F4A7C15ull;
        first = (unsigned int)(m >> 53);                                   // 11 bits
        step = ((unsigned int)(m >> 42) & (unsigned)(kSlots - 1)) | 1u;    // odd: visits every slot of the 2^11 table
    }
    __device__ __forceinline__ unsigned int next(unsigned int h) const { return (h + step) & (unsigned)(kSlots - 1); }
};
