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
// floor(u / leaf) == (u * magic) >> 32 with magic = ceil(2^32 / leaf) — checked by the host over every u before the
// launch (magic = 0: use '/'). Three variable-divisor integer divisions per point otherwise cost ~75 instructions.
struct VoxelDiv {
    unsigned int leaf, bias_leaf, magic;
    __device__ __forceinline__ unsigned int operator()(int v) const
    {
        const unsigned int u = (unsigned int)(v + (int)bias_leaf);
        return magic ? __umulhi(u, magic) : u / leaf;
    }
};

__device__ __forceinline__ unsigned long long voxel_key(const VoxelDiv& dv, int x, int y, int z, unsigned int bits)
{
    const unsigned long long kx = dv(x), ky = dv(y), kz = dv(z);
    return (kz << (2 * bits)) | (ky << bits) | kx;      // z major, x fastest: (z,y,x) voxel order
}

// LDS hash table of one pre-aggregation workgroup
constexpr int kSlots = 2048, kProbe = 12;
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
