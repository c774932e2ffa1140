// pcs_voxel.hip — voxel-grid downsample of a stitched XYZRGB payload (BASELINE.json configs[4]).
//
// NOT in the reference: pcl/filters/voxel_grid.h is #included there (src/pcs-multicamera-optimized.cpp:17,
// src/pcs-multicamera-client.cpp:17) but VoxelGrid is never instantiated; the reference's only downsample
// is the integer stride (implemented in the pack kernels). This op is therefore DEFINED here, in the
// payload's own integer millimetre domain, so that it is exactly reproducible:
//
//   voxel of a point : (floor(x / leaf), floor(y / leaf), floor(z / leaf)),  x,y,z the int16 mm coordinates,
//                      floor toward -inf so voxels tile space uniformly and are anchored at the origin
//   one output point per occupied voxel:
//        x,y,z = trunc(sum / count) per axis (integer sums, C division)         -> the centroid, in mm
//        R,G,B = sum / count per channel (integer)                              -> the mean colour
//   output order     : ascending (z-voxel, y-voxel, x-voxel), x fastest — the order PCL's VoxelGrid
//                      emits (index = ix + iy*dx + iz*dx*dy); PCL itself averages in float, so against
//                      PCL 1.8 this is "parity unpinned" (+-1 LSB per field expected).
//
// Pipeline — every kernel hand-written, every size read from device memory, nothing waits for the host:
//   1. pre-aggregation  a workgroup accumulates 8192 consecutive points (a few image rows of one camera, which fall
//                       into few voxels) in an LDS hash table and appends one PARTIAL (key + 7 sums) per voxel it saw;
//                       m = number of partials (3.0 M for the 29.8 M-point config-5 cloud at 50 mm)
//   2. radix sort       of (key, partial index) over the m partials, least significant digit first, 11 bits per pass
//                       (3 passes at 50 mm: the key is packed to 3 * ceil(log2(65536/leaf)) bits); per pass:
//                       per-chunk histograms -> per-digit column scan -> stable scatter
//   3. segmented mean   heads of equal-key runs -> voxel ordinals (block counts + scan) -> every head sums its run's
//                       partials (short runs by one lane, long runs by the whole wavefront) and writes the record
// Integer sums make the result independent of the order of the partials and of any reduction order.
// (Round 1 used rocPRIM's radix_sort_pairs + reduce_by_key for steps 2-3: 0.56 ms of library kernels per call at 50 mm,
// plus a stream synchronisation in the middle of the call to learn m.)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pcs_device.h"

namespace pcs {

namespace {

#include "pcs_voxel_agg.h"

// ---- wavefront helpers (64 lanes, all active) ---------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_from(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }

__device__ __forceinline__ unsigned int wave_incl_scan(unsigned int x)
{
    unsigned int v = x;
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Segmented inclusive scan step over the wavefront for the sums of a run of points: lanes whose span already contains
// a run head keep their value; the others add the value `ctrl` lanes below (or the row's carry) and inherit its flag.
// A lane without a valid source (start of a row for row_shr) receives zeros / "no head", i.e. it simply keeps going.
// Within a wavefront the colour sums (<= 64 * 255) and the count (<= 64) fit 16 bits: r|g<<16 and b|n<<16 travel as
// two words instead of four.
struct SegAcc { int x, y, z; unsigned int rg, bn; int head; };
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(SegAcc& a)
{
    const int ux = dpp_from<CTRL, ROW_MASK>(a.x), uy = dpp_from<CTRL, ROW_MASK>(a.y), uz = dpp_from<CTRL, ROW_MASK>(a.z);
    const int urg = dpp_from<CTRL, ROW_MASK>((int)a.rg), ubn = dpp_from<CTRL, ROW_MASK>((int)a.bn);
    const int uh = dpp_from<CTRL, ROW_MASK>(a.head);
    if (!a.head) {
        a.x += ux; a.y += uy; a.z += uz;
        a.rg += (unsigned int)urg; a.bn += (unsigned int)ubn;
        a.head = uh;
    }
}

// One packed record (10 bytes at byte offset 10*i) from ONE 12-byte load instead of five 2-byte loads: an even record
// starts on a dword, an odd one two bytes after one (the window then starts two bytes early).
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u32x3 load_record_window(const int16_t* __restrict__ payload, unsigned int i)
{
    return *reinterpret_cast<const u32x3*>(payload + (size_t)i * PCS_POINT_SHORTS - (i & 1u));
}
__device__ __forceinline__ void unpack_record(const u32x3 d, unsigned int i, int& x, int& y, int& z, unsigned int& col, unsigned int& blue)
{
    const bool odd = (i & 1u) != 0u;
    const unsigned int a = odd ? __builtin_amdgcn_alignbit(d.y, d.x, 16) : d.x;     // x | y << 16
    const unsigned int b = odd ? __builtin_amdgcn_alignbit(d.z, d.y, 16) : d.y;     // z | colour << 16
    const unsigned int c = odd ? d.z >> 16 : d.z;                                   // blue (low 16)
    x = (int)(short)(a & 0xFFFFu); y = (int)(short)(a >> 16);
    z = (int)(short)(b & 0xFFFFu); col = b >> 16; blue = c & 0xFFu;
}

// ------------------------------------------------------------------------------------------------
// 1. Pre-aggregation. A workgroup (1024 lanes) takes 8192 consecutive points of the payload and accumulates them in
// an LDS hash table (2048 slots, open addressing, 64-bit compare-and-swap on the key, 32-bit LDS adds on the seven
// sums). Neighbouring lanes hold neighbouring pixels, which mostly share a voxel: 64 lanes adding to the same LDS
// words serialise, so runs of equal keys across the wavefront are summed first (segmented scan over the lanes, by DPP:
// the ds_bpermute shuffles of the first version kept the LDS pipe busier than the table itself) and only the LAST lane
// of a run touches the table. Points that find no slot within kProbe probes (dense tiny leaves) are passed through
// as single-point partials. One returning global atomic per workgroup reserves its slice of the partial arrays; the
// order of the partials does not matter (they are sorted next, and the sums are integers).
// ------------------------------------------------------------------------------------------------
constexpr int kAggThreads = 1024, kAggPerLane = 8;      // kSlots, kProbe, kEmptyKey: pcs_voxel_agg.h

// WIDE: the payload is 4-byte aligned and holds >= 2 points: every lane requests its eight records with eight 12-byte
// loads up front, branch-free (indices clamped to the last record whose window stays inside the payload; an even LAST
// record, whose window would read two bytes past the end, is passed through as a single-point partial instead).
template <bool WIDE>
__global__ __launch_bounds__(kAggThreads)
void pcs_voxel_partials_kernel(const int16_t* __restrict__ payload, unsigned int n_host, const int32_t* __restrict__ n_dev,
                               VoxelDiv dv, unsigned int bits,
                               unsigned int idx_bits, unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx,
                               VoxelPartial* __restrict__ part, unsigned int* __restrict__ n_runs)
{
    __shared__ unsigned long long skey[kSlots];
    __shared__ int          ssx[kSlots], ssy[kSlots], ssz[kSlots];
    __shared__ unsigned int sr[kSlots], sg[kSlots], sb[kSlots], sn[kSlots];
    __shared__ unsigned int wtot[kAggThreads / 64];
    __shared__ unsigned int base_s;

    // the point count comes from the host, or (counted form) from device memory — e.g. the total a compaction launch
    // left behind; the grid then covers the buffer's capacity and the surplus workgroups leave here
    const unsigned int n = n_dev ? (unsigned int)max(*n_dev, 0) : n_host;
    const unsigned int tile0 = blockIdx.x * (unsigned)(kAggThreads * kAggPerLane);
    if (tile0 >= n) return;
    u32x3 raw[kAggPerLane];
    // WIDE: last record with an in-bounds window (the host guarantees room for two records; a lone record is passed through)
    const unsigned int last_safe = n < 2u ? 0u : (((n - 1u) & 1u) ? n - 1u : n - 2u);
    if (WIDE) {
#pragma unroll
        for (int k = 0; k < kAggPerLane; k++) {
            const unsigned int i = tile0 + (unsigned)k * kAggThreads + threadIdx.x;      // lane-contiguous records
            raw[k] = load_record_window(payload, i < last_safe ? i : last_safe);
        }
    }
    for (int j = threadIdx.x; j < kSlots; j += kAggThreads) {
        skey[j] = kEmptyKey;
        ssx[j] = ssy[j] = ssz[j] = 0;
        sr[j] = sg[j] = sb[j] = sn[j] = 0u;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    unsigned int failed = 0;                                  // bit k: point k of this lane found no slot
#pragma unroll
    for (int k = 0; k < kAggPerLane; k++) {
        const unsigned int i = tile0 + (unsigned)k * kAggThreads + threadIdx.x;
        const bool exists = i < n;
        const bool live = exists && (!WIDE || (i <= last_safe && n >= 2u));
        int x = 0, y = 0, z = 0;
        unsigned int col = 0, blue = 0;
        if (WIDE) {
            unpack_record(raw[k], i, x, y, z, col, blue);
        } else if (live) {
            const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
            x = p[0]; y = p[1]; z = p[2];
            col = (unsigned short)p[3]; blue = (unsigned short)p[4] & 0xFFu;
        }
        // dead lanes (ragged last wavefront) get a key no live point has, so they form their own runs and never act
        const unsigned long long key = live ? voxel_key(dv, x, y, z, bits) : (kEmptyKey - 1ull - (unsigned long long)lane);
        // Neighbouring lanes hold neighbouring pixels, which mostly share a voxel: runs of equal keys across the
        // wavefront are summed first and only the LAST lane of a run touches the table.
        SegAcc a{x, y, z, (col & 0xFFu) | ((col >> 8) << 16), blue | (1u << 16), 0};
        const unsigned long long prev = ((unsigned long long)(unsigned int)dpp_from<0x138, 0xf>((int)(key >> 32)) << 32) |
                                        (unsigned int)dpp_from<0x138, 0xf>((int)key);       // wave_shr:1
        const unsigned long long next = ((unsigned long long)(unsigned int)dpp_from<0x130, 0xf>((int)(key >> 32)) << 32) |
                                        (unsigned int)dpp_from<0x130, 0xf>((int)key);       // wave_shl:1
        // runs are merged inside aligned groups of 8 lanes only: a run that crosses a group border simply becomes several
        // actors (a few more table accesses, to the same slot) and three row_shr steps replace the six of a wavefront-wide
        // segmented scan — the merge was half of this VALU-bound kernel (50 mm: 188 us wavefront-wide, 158 us per row of
        // 16, 149 us per group of 8; 200 mm: 137 / 118 / 111 us)
        a.head = ((lane & 7) == 0 || prev != key) ? 1 : 0;
        bool actor = live;
        // merging pays when the wavefront holds few runs; with many (small leaves) the scan costs more than the
        // conflicts it avoids
        const bool merge = __popcll(__ballot(a.head != 0)) <= 32;
        if (merge) {
            actor = live && ((lane & 7) == 7 || next != key);  // last lane of its run within the group of 8
            // (Measured and dropped: a wave-uniform early exit once every lane's span holds its run's head — any branch
            // inside this unrolled loop stops the compiler from overlapping the eight iterations: 254 vs 188 us at 50 mm,
            // taken or not; and a split into a branch-free phase for all eight records followed by the table phase —
            // 128 VGPRs, one workgroup per CU: 312 us.)
            seg_step<0x111, 0xf>(a);                             // row_shr:1
            seg_step<0x112, 0xf>(a);                             // row_shr:2
            seg_step<0x114, 0xf>(a);                             // row_shr:4
        }
        const VoxelProbe pr(key);
        unsigned int h = pr.first;
        bool placed = false;
        if (actor) {
            // (Peeling the first probe out of the loop, which helps the raster reader, costs this kernel a third: 143 -> 195 us.
            // The extra branch ends the overlap of the eight iterations — the same effect as every other branch tried here.)
            for (int t = 0; t < kProbe; t++) {
                const unsigned long long old = atomicCAS(&skey[h], kEmptyKey, key);
                if (old == kEmptyKey || old == key) { placed = true; break; }
                h = pr.next(h);
            }
            if (placed) {
                atomicAdd(&ssx[h], a.x); atomicAdd(&ssy[h], a.y); atomicAdd(&ssz[h], a.z);
                atomicAdd(&sr[h], a.rg & 0xFFFFu); atomicAdd(&sg[h], a.rg >> 16); atomicAdd(&sb[h], a.bn & 0xFFFFu);
                atomicAdd(&sn[h], a.bn >> 16);
            }
        }
        if (merge) {
            // a run's verdict is its last lane's: walk it back to every lane of the run (the failed ones pass their
            // own point through). Runs are contiguous, so "the next actor at or after me" decides.
            const unsigned long long actors = __ballot(actor), ok = __ballot(actor && placed);
            const unsigned long long at_or_after = actors & (~0ull << lane);
            const int mine_actor = __ffsll((long long)at_or_after) - 1;            // -1 only for dead lanes
            placed = mine_actor >= 0 && ((ok >> mine_actor) & 1ull);
        }
        if (exists && !placed) failed |= 1u << k;
    }
    __syncthreads();

    // every lane owns two slots; count what this workgroup will append: occupied slots + passed-through points
    unsigned int c = __popc(failed);
#pragma unroll
    for (int q = 0; q < kSlots / kAggThreads; q++) c += skey[threadIdx.x * (kSlots / kAggThreads) + q] != kEmptyKey;
    const int wave = threadIdx.x >> 6;
    const unsigned int inc = wave_incl_scan(c);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int w = 0; w < kAggThreads / 64; w++) { const unsigned int t = wtot[w]; wtot[w] = tot; tot += t; }
        base_s = tot ? atomicAdd(n_runs, tot) : 0u;
    }
    __syncthreads();
    unsigned int pos = base_s + wtot[wave] + inc - c;
#pragma unroll
    for (int q = 0; q < kSlots / kAggThreads; q++) {
        const int j = threadIdx.x * (kSlots / kAggThreads) + q;
        if (skey[j] != kEmptyKey) {
            if (idx_bits) keys[pos] = (skey[j] << idx_bits) | pos;      // packed: the partial's index rides in the low bits
            else { keys[pos] = skey[j]; idx[pos] = pos; }
            part[pos] = VoxelPartial{ssx[j], ssy[j], ssz[j], sr[j], sg[j], sb[j], sn[j], 0u};
            pos++;
        }
    }
#pragma unroll
    for (int k = 0; k < kAggPerLane; k++) {
        if (!((failed >> k) & 1u)) continue;
        const unsigned int i = tile0 + (unsigned)k * kAggThreads + threadIdx.x;
        const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
        const int x = p[0], y = p[1], z = p[2];
        const unsigned int col = (unsigned short)p[3], blue = (unsigned short)p[4] & 0xFFu;
        if (idx_bits) keys[pos] = (voxel_key(dv, x, y, z, bits) << idx_bits) | pos;
        else { keys[pos] = voxel_key(dv, x, y, z, bits); idx[pos] = pos; }
        part[pos] = VoxelPartial{x, y, z, col & 0xFFu, col >> 8, blue, 1u, 0u};
        pos++;
    }
}

// ------------------------------------------------------------------------------------------------
// 2. LSD radix sort of (key, index) pairs, 11 bits per pass, element count read from device memory.
//    Chunks of 8192 elements handled by 512-lane workgroups (half as many chunks = half as many strided accesses in the
//    column scan, same number of wavefronts in flight); persistent grids (the kernels loop over chunks), so a launch costs the same whether the
//    pre-aggregation left 90 k or 17 M partials and needs no size from the host.
// ------------------------------------------------------------------------------------------------
constexpr int kRadixBits = 11, kRadix = 1 << kRadixBits;
constexpr unsigned int kSortChunk = 8192, kSortThreads = 512, kSortWaves = kSortThreads / 64, kSortGrid = 512, kSortBatch = 8;
constexpr unsigned int kSortMinChunk = 1024, kSortMinRows = 256;
constexpr unsigned int kSuperShift = 6, kSuperStride = 32;      // voxel ordinals: groups of 64 blocks, one counter per 128-byte line

// Elements per chunk (= per workgroup pass), chosen ON THE DEVICE from the number of partials so that a small sort still
// spreads over 128 - 256 workgroups: 8192 from 1 M elements up, halving down to 1024 below 256 k (the 125 k partials
// of a 200 mm grid were 15 chunks of 8192: three 27 us passes on 15 CUs). More, smaller chunks cost the column scan more
// than they save the scatter (0.9 M elements as 440 chunks of 2048: scatter 16.5 -> 12.9 us, column scan 5.8 -> 14.2 us).
__device__ __forceinline__ unsigned int sort_chunk_of(unsigned int m)
{
    return m >= (1u << 20) ? 8192u : m >= (512u << 10) ? 4096u : m >= (256u << 10) ? 2048u : 1024u;
}

// Which bits of the keys actually differ, learnt on the device: the pre-aggregation ORs every key it writes into
// ctl[32..33] and every complement into ctl[34..35]; a bit varies iff it is set in both. Within each axis field the bits
// above the highest varying one are the same in all keys, so dropping them changes neither order nor equality: the
// passes sort the COMPACT key (the three fields' low w0 / w1 / w2 bits, concatenated), 11 bits at a time, and a cloud
// that spans 8 m at a 10 mm leaf needs 3 passes instead of the 4 its 39-bit key would (200 mm: 2 instead of 3). The
// host enqueues P = ceil(3 * bits / 11) passes; the first P - R of them return at once, real pass e reads buffer A when
// e is even, B when odd, and the segmented mean reads whichever buffer the last real pass wrote.
constexpr unsigned int kCtlOr = kVoxCtlOr, kCtlOrn = kVoxCtlOrn;
struct SortPass {
    bool skip, in_a, plain;          // plain: no pass is saved, the digit is a plain bit field of the key
    unsigned int lo, w0, w1, p0, p1, p2;
    unsigned long long m0, m1, m2;
    __device__ __forceinline__ unsigned int digit(unsigned long long key) const
    {
        if (plain) return (unsigned int)(key >> (p0 + lo)) & (unsigned int)(kRadix - 1);       // uniform
        const unsigned long long c = ((key >> p0) & m0) | (((key >> p1) & m1) << w0) | (((key >> p2) & m2) << (w0 + w1));
        return (unsigned int)(c >> lo) & (unsigned int)(kRadix - 1);
    }
};
// `bits` carries the launcher's tracking decision in bit 31 (kTrackFlag): without it nobody recorded the varying bits and
// every bit counts.
constexpr unsigned int kTrackFlag = 1u << 31;
__device__ __forceinline__ unsigned int real_passes(const unsigned int* __restrict__ ctl, unsigned int bits_and_flag, unsigned int& w0,
                                                    unsigned int& w1, unsigned int& w2)
{
    const unsigned int bits = bits_and_flag & ~kTrackFlag;
    if (!(bits_and_flag & kTrackFlag)) {
        w0 = w1 = w2 = bits;
        return (3u * bits + (unsigned int)kRadixBits - 1u) / (unsigned int)kRadixBits;
    }
    const unsigned long long var = (ctl[kCtlOr] | ((unsigned long long)ctl[kCtlOr + 1] << 32)) &
                                   (ctl[kCtlOrn] | ((unsigned long long)ctl[kCtlOrn + 1] << 32));
    const unsigned int fm = (1u << bits) - 1u;
    const unsigned int v0 = (unsigned int)var & fm, v1 = (unsigned int)(var >> bits) & fm, v2 = (unsigned int)(var >> (2u * bits)) & fm;
    w0 = v0 ? 32u - (unsigned int)__clz((int)v0) : 0u;
    w1 = v1 ? 32u - (unsigned int)__clz((int)v1) : 0u;
    w2 = v2 ? 32u - (unsigned int)__clz((int)v2) : 0u;
    return (w0 + w1 + w2 + (unsigned int)kRadixBits - 1u) / (unsigned int)kRadixBits;
}
__device__ __forceinline__ SortPass sort_pass(const unsigned int* __restrict__ ctl, unsigned int bits, unsigned int idx_bits,
                                              unsigned int p, unsigned int P)
{
    SortPass sp;
    unsigned int w2;
    const unsigned int R = real_passes(ctl, bits, sp.w0, sp.w1, w2);
    bits &= ~kTrackFlag;
    const unsigned int skipped = P - R;                     // R <= P: the widths never exceed the fields
    sp.plain = skipped == 0u;
    sp.skip = p < skipped;
    const unsigned int e = p - skipped;
    sp.in_a = (e & 1u) == 0u;
    sp.lo = e * (unsigned int)kRadixBits;
    sp.p0 = idx_bits; sp.p1 = idx_bits + bits; sp.p2 = idx_bits + 2u * bits;
    sp.m0 = (1ull << sp.w0) - 1ull; sp.m1 = (1ull << sp.w1) - 1ull; sp.m2 = (1ull << w2) - 1ull;
    return sp;
}
// true: the sorted keys are in buffer A
__device__ __forceinline__ bool sorted_in_a(const unsigned int* __restrict__ ctl, unsigned int bits)
{
    unsigned int w0, w1, w2;
    return (real_passes(ctl, bits, w0, w1, w2) & 1u) == 0u;
}

// Partials that arrive from outside the pre-aggregation of this call (other GPUs' pre-aggregations, gathered to the root:
// BASELINE configs[4]) bring RAW voxel keys; the sort wants (key << idx_bits) | index, or key and index side by side. The
// first pass reads them as they are and forms its elements on the fly — keys == nullptr everywhere else. (Until round 3 a
// separate import kernel rewrote them first: one more launch and 14 MB more traffic on the root.)
struct RawKeys {
    const unsigned long long* keys;      // nullptr: the elements are in the sort's own buffers
    const int32_t*            m_dev;     // their number: read from the device if given (at most m_host), else m_host
    unsigned int              m_host;
    __device__ __forceinline__ unsigned int count() const
    {
        const unsigned int m = m_dev ? (unsigned int)max(*m_dev, 0) : m_host;
        return min(m, m_host);
    }
    __device__ __forceinline__ unsigned long long element(unsigned int e, unsigned int idx_bits) const
    {
        const unsigned long long k = keys[e];
        return idx_bits ? (k << idx_bits) | e : k;
    }
};

// table[chunk][digit] = occurrences of the digit in the chunk
__global__ __launch_bounds__(kSortThreads)
void pcs_voxel_hist_kernel(const unsigned long long* __restrict__ keys_a, const unsigned long long* __restrict__ keys_b,
                           unsigned int* __restrict__ ctl, unsigned int bits, unsigned int idx_bits, unsigned int pass,
                           unsigned int n_passes, unsigned int* __restrict__ table, unsigned int* __restrict__ super, RawKeys raw)
{
    __shared__ unsigned int hist[kRadix];
    const bool from_raw = raw.keys != nullptr && pass == 0;        // (raw input never skips a pass: nobody tracked its key bits)
    const SortPass sp = sort_pass(ctl, bits, idx_bits, pass, n_passes);
    // raw input: this launch also publishes the element count for the device-driven kernels that follow
    const unsigned int m = from_raw ? raw.count() : ctl[0];
    if (from_raw && blockIdx.x == 0 && threadIdx.x == 0) ctl[0] = m;
    // the launch of the LAST pass (which is never skipped while there is anything to sort) clears the group counters the heads
    // kernel adds to: one per 128-byte line, as many as m needs
    if (pass + 1u == n_passes && blockIdx.x == 0) {
        const unsigned int groups = ((((m + 255u) >> 8) + (1u << kSuperShift) - 1u) >> kSuperShift) + 1u;
        for (unsigned int j = threadIdx.x; j < groups; j += kSortThreads) super[(size_t)j * kSuperStride] = 0u;
    }
    if (sp.skip) return;
    const unsigned long long* __restrict__ keys = sp.in_a ? keys_a : keys_b;
    const unsigned int csize = sort_chunk_of(m), per_thread = csize / kSortThreads;
    const unsigned int chunks = (m + csize - 1) / csize;
    for (unsigned int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        for (unsigned int j = threadIdx.x; j < kRadix; j += kSortThreads) hist[j] = 0u;
        __syncthreads();
        const unsigned int c0 = chunk * csize, c1 = min(c0 + csize, m);
        for (unsigned int it0 = 0; it0 < per_thread; it0 += kSortBatch) {
            unsigned long long k[kSortBatch];       // the batch's loads are in flight together
#pragma unroll
            for (unsigned int q = 0; q < kSortBatch; q++) {
                const unsigned int e = c0 + (it0 + q) * kSortThreads + threadIdx.x;
                k[q] = e < c1 ? (from_raw ? raw.element(e, idx_bits) : keys[e]) : 0ull;
            }
#pragma unroll
            for (unsigned int q = 0; q < kSortBatch; q++) {
                const unsigned int e = c0 + (it0 + q) * kSortThreads + threadIdx.x;
                if (e < c1) atomicAdd(&hist[sp.digit(k[q])], 1u);
            }
        }
        __syncthreads();
        for (unsigned int j = threadIdx.x; j < kRadix; j += kSortThreads) table[(size_t)chunk * kRadix + j] = hist[j];
        __syncthreads();
    }
}

// One wavefront per digit: exclusive scan of the digit's column over the chunks (in place) + the digit's total. The
// column is strided (one row per chunk), so up to 16 x 64 entries are requested before any is used.
__global__ __launch_bounds__(256)
void pcs_voxel_colscan_kernel(unsigned int* __restrict__ table, const unsigned int* __restrict__ ctl,
                              unsigned int* __restrict__ digit_total, unsigned int bits, unsigned int pass, unsigned int n_passes)
{
    constexpr unsigned int kCols = 16;
    __shared__ unsigned int tot[16][16];
    {
        unsigned int w0, w1, w2;
        if (pass < n_passes - real_passes(ctl, bits, w0, w1, w2)) return;       // a skipped pass
    }
    const unsigned int m = ctl[0];
    const unsigned int csize = sort_chunk_of(m);
    const unsigned int chunks = (m + csize - 1) / csize;
    if (chunks <= 256u) {
        // Few rows (every sort below 2 M elements): 16 digits per workgroup, 16 row slots (4 per wavefront, 16 lanes = 64
        // contiguous bytes each) of <= 16 consecutive rows. A lane requests all its rows at once, keeps them in registers,
        // the slots' totals cross through LDS, and the prefixes are written from the registers: one round trip to memory
        // instead of one per 64 rows of a strided column (220 rows: 9.5 -> ~4 us per pass).
        if (blockIdx.x >= kRadix / 16) return;
        const unsigned int d = blockIdx.x * 16u + (threadIdx.x & 15u), slot = threadIdx.x >> 4;
        const unsigned int rps = (chunks + 15u) / 16u, r0 = slot * rps;
        unsigned int v[16], sum = 0;
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) {
            const unsigned int r = r0 + j;
            v[j] = (j < rps && r < chunks) ? table[(size_t)r * kRadix + d] : 0u;
        }
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) sum += v[j];
        tot[slot][threadIdx.x & 15u] = sum;
        __syncthreads();
        unsigned int run = 0, all = 0;
#pragma unroll
        for (unsigned int q = 0; q < 16; q++) { const unsigned int t = tot[q][threadIdx.x & 15u]; run += q < slot ? t : 0u; all += t; }
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) {
            const unsigned int r = r0 + j;
            if (j < rps && r < chunks) table[(size_t)r * kRadix + d] = run;
            run += v[j];
        }
        if (slot == 0) digit_total[d] = all;
        return;
    }
    const unsigned int digit = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned int carry = 0;
    for (unsigned int c0 = 0; c0 < chunks; c0 += 64 * kCols) {
        unsigned int v[kCols];
#pragma unroll
        for (unsigned int q = 0; q < kCols; q++) {
            const unsigned int c = c0 + q * 64 + lane;
            v[q] = c < chunks ? table[(size_t)c * kRadix + digit] : 0u;
        }
#pragma unroll
        for (unsigned int q = 0; q < kCols; q++) {
            const unsigned int c = c0 + q * 64 + lane;
            if (c0 + q * 64 < chunks) {                           // wave-uniform
                const unsigned int inc = wave_incl_scan(v[q]);
                if (c < chunks) table[(size_t)c * kRadix + digit] = carry + inc - v[q];
                carry += (unsigned int)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
    }
    if (lane == 0) digit_total[digit] = carry;
}

// Stable scatter. Wavefront w of the workgroup owns the w-th eighth of the chunk and walks it 64 elements at a time,
// so "earlier in memory" = (lower wavefront, lower round, lower lane):
//   destination = digit base (all smaller digits) + this digit in earlier chunks (column scan)
//               + this digit in earlier wavefronts of the chunk + in earlier rounds of this wavefront + in lower lanes.
// The last three come from per-(wavefront, digit) LDS counters: counted first (LDS adds), turned into starting offsets,
// then advanced round by round by the lowest lane of each group of equal digits (found with 11 ballots).
// ONE trip to memory per chunk: a lane requests its (at most 16) elements, its four digit totals and its four entries of
// the chunk's table row together and keeps all of them in registers — the elements are counted and later placed from
// there. (Rounds 2-3 read the totals, the elements, the table row and the elements again one after the other: four
// dependent round trips of ~2 us each in a workgroup that runs alone on its CU.)
// PACKED: the element is (key << idx_bits) | partial index in ONE 64-bit word (possible when 3*bits + idx_bits <= 64, i.e.
// for every leaf >= 8 mm on the 30 M-point cloud): one 8-byte scattered store per element instead of 8 + 4, and no
// index arrays at all.
template <bool PACKED>
__global__ __launch_bounds__(kSortThreads)
void pcs_voxel_scatter_kernel(unsigned long long* __restrict__ keys_a, unsigned int* __restrict__ idx_a,
                              unsigned long long* __restrict__ keys_b, unsigned int* __restrict__ idx_b,
                              const unsigned int* __restrict__ ctl, unsigned int bits, unsigned int idx_bits, unsigned int pass,
                              unsigned int n_passes, const unsigned int* __restrict__ table,
                              const unsigned int* __restrict__ digit_total, RawKeys raw)
{
    __shared__ __attribute__((aligned(16))) unsigned int cnt[kSortWaves][kRadix];      // 64 KiB
    const bool from_raw = raw.keys != nullptr && pass == 0;
    __shared__ unsigned int wsum[kSortWaves];
    const SortPass sp = sort_pass(ctl, bits, idx_bits, pass, n_passes);
    if (sp.skip) return;
    const unsigned long long* __restrict__ keys_in = sp.in_a ? keys_a : keys_b;
    const unsigned int* __restrict__ idx_in = sp.in_a ? idx_a : idx_b;
    unsigned long long* __restrict__ keys_out = sp.in_a ? keys_b : keys_a;
    unsigned int* __restrict__ idx_out = sp.in_a ? idx_b : idx_a;
    const unsigned int m = ctl[0];
    const unsigned int csize = sort_chunk_of(m);
    const unsigned int chunks = (m + csize - 1) / csize;
    if (blockIdx.x >= chunks) return;
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    constexpr unsigned int kPer = kRadix / kSortThreads;      // 4 consecutive digits per thread, the same four throughout
    constexpr unsigned int kMaxRounds = kSortChunk / kSortWaves / 64;      // 16
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    static_assert(kPer == 4, "a thread's digits travel as one 16-byte vector");
    const unsigned int kRounds = csize / kSortWaves / 64;                  // 2 .. 16

    const u32x4 tot4 = *reinterpret_cast<const u32x4*>(digit_total + threadIdx.x * kPer);
    unsigned int dbase[kPer];
    bool have_base = false;

    for (unsigned int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const unsigned int w0 = chunk * csize + wave * (csize / kSortWaves);
        const unsigned int w1 = min(w0 + csize / kSortWaves, m);               // this wavefront's elements: [w0, w1)
        // everything this chunk needs from memory, requested at once
        unsigned long long k[kMaxRounds];
        unsigned int id[PACKED ? 1 : kMaxRounds];
#pragma unroll
        for (unsigned int r = 0; r < kMaxRounds; r++) {
            const unsigned int e = w0 + r * 64 + lane;
            k[r] = 0ull;
            if (!PACKED) id[r] = 0u;
            if (r < kRounds && e < w1) {
                if (from_raw) { k[r] = raw.element(e, idx_bits); if (!PACKED) id[r] = e; }
                else { k[r] = keys_in[e]; if (!PACKED) id[r] = idx_in[e]; }
            }
        }
        const u32x4 row4 = *reinterpret_cast<const u32x4*>(table + (size_t)chunk * kRadix + threadIdx.x * kPer);
        {   // clear the counters: 16-byte LDS stores, lane-contiguous
            u32x4* c4 = reinterpret_cast<u32x4*>(&cnt[0][0]);
#pragma unroll
            for (unsigned int j = 0; j < kSortWaves * kRadix / 4 / kSortThreads; j++) c4[j * kSortThreads + threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        // count: this wavefront's occurrences of every digit
#pragma unroll
        for (unsigned int r = 0; r < kMaxRounds; r++) {
            const unsigned int e = w0 + r * 64 + lane;
            if (r < kRounds && e < w1) atomicAdd(&cnt[wave][sp.digit(k[r])], 1u);
        }
        if (!have_base) {   // digit bases: exclusive scan of the 2048 digit totals (once per workgroup)
            const unsigned int s4 = tot4.x + tot4.y + tot4.z + tot4.w;
            const unsigned int inc = wave_incl_scan(s4);
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            unsigned int run = inc - s4;
            for (unsigned int w = 0; w < wave; w++) run += wsum[w];
            dbase[0] = run; dbase[1] = run + tot4.x; dbase[2] = dbase[1] + tot4.y; dbase[3] = dbase[2] + tot4.z;
            have_base = true;
        } else {
            __syncthreads();
        }
        // starting offsets per (wavefront, digit): a thread's four digits are one 16-byte LDS word per wavefront row
        {
            u32x4 start = u32x4{dbase[0], dbase[1], dbase[2], dbase[3]} + row4;
#pragma unroll
            for (unsigned int w = 0; w < kSortWaves; w++) {
                u32x4* slot = reinterpret_cast<u32x4*>(&cnt[w][threadIdx.x * kPer]);
                const u32x4 c = *slot;
                *slot = start;
                start += c;
            }
        }
        __syncthreads();
        // place: one round of 64 elements at a time, in order, from the registers
#pragma unroll
        for (unsigned int r = 0; r < kMaxRounds; r++) {
            if (r < kRounds) {                                                  // uniform
                const unsigned int e = w0 + r * 64 + lane;
                const bool live = e < w1;
                const unsigned int d = sp.digit(k[r]);
                unsigned long long peers = __ballot(live);
#pragma unroll
                for (int b = 0; b < kRadixBits; b++) {
                    const bool bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    peers &= bit ? bal : ~bal;
                }
                if (live) {
                    const unsigned int below = __popcll(peers & ((1ull << lane) - 1ull));
                    const unsigned int start = cnt[wave][d];
                    if (below == 0) cnt[wave][d] = start + __popcll(peers);       // the group's lowest lane advances the counter
                    const unsigned int dst = start + below;
                    keys_out[dst] = k[r];
                    if (!PACKED) idx_out[dst] = id[r];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// 3. Segmented mean over the sorted partials. Every lane gathers ONE partial (all loads in flight at once); equal-key
//    runs are summed by a segmented scan across the wavefront (DPP) and across the block's four wavefronts (LDS
//    carries). A run that lies inside one block of 256 sorted elements — almost all of them — is finished there.
//    Pieces of runs that cross block boundaries are left per block as {lead: the part of a run begun earlier, trail:
//    the open run at the block's end}; a second, tiny kernel adds trail[b] + lead[b+1] + ... and writes those voxels.
//    (Tried in round 3: the block a run STARTS in reading on past its end with all its lanes, so that the second kernel
//    disappears — every third block then pays two more dependent round trips: reduce 18.6 -> 29 us for the 5 us saved.)
//    No lane ever walks a run serially (the first version did: 231 us at 50 mm, most lanes idle, the rest latency-bound).
// ------------------------------------------------------------------------------------------------
constexpr unsigned int kSegThreads = 256, kSegGrid = 4096;
constexpr unsigned int kCtlWords = 64;                            // one call's control block

// heads[b] = runs that START in block b (block = 256 consecutive sorted elements); super[g] = the same for the group of 64
// blocks g — added with non-returning atomics, one counter per 128-byte line (64 adds to a line serialise at ~12 ns each; all
// 3 500 of them on two lines took 23 us), cleared by the last pass's histogram launch. A block's first voxel ordinal is then
// the sum of the groups before its own + the blocks before it in its group: <= 2 loads per lane in the reduce kernel. (Until
// round 4 a separate single-workgroup launch scanned heads[] in place: one more dependent dispatch.)
__global__ __launch_bounds__(kSegThreads)
void pcs_voxel_heads_kernel(const unsigned long long* __restrict__ keys_a, const unsigned long long* __restrict__ keys_b,
                            const unsigned int* __restrict__ ctl, unsigned int bits, unsigned int idx_bits,
                            unsigned int* __restrict__ heads, unsigned int* __restrict__ super)
{
    __shared__ unsigned int wsum[4];
    const unsigned long long* __restrict__ keys = sorted_in_a(ctl, bits) ? keys_a : keys_b;
    const unsigned int m = ctl[0];
    const unsigned int blocks = (m + kSegThreads - 1) / kSegThreads;
    for (unsigned int b = blockIdx.x; b < blocks; b += gridDim.x) {
        const unsigned int g = b * kSegThreads + threadIdx.x;
        const bool head = g < m && (g == 0 || (keys[g] >> idx_bits) != (keys[g - 1] >> idx_bits));
        const unsigned int c = __popcll(__ballot(head));
        if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            heads[b] = t;
            if (t) atomicAdd(&super[(size_t)(b >> kSuperShift) * kSuperStride], t);
        }
        __syncthreads();
    }
}

// Sums of a piece of a run inside one block: 256 partials of <= 32 768 points each -> the colour sums (< 2^31) and the
// count fit 32 bits, the coordinate sums need 64.
struct SegSum {
    long long x, y, z;
    unsigned int r, g, b, n;
    int head;                       // a run head lies inside the span summed so far
};
__device__ __forceinline__ void seg_add(SegSum& a, const SegSum& u)
{
    a.x += u.x; a.y += u.y; a.z += u.z; a.r += u.r; a.g += u.g; a.b += u.b; a.n += u.n;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_from64(long long v)
{
    const unsigned int lo = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step64(SegSum& a)
{
    SegSum u;
    u.x = dpp_from64<CTRL, ROW_MASK>(a.x); u.y = dpp_from64<CTRL, ROW_MASK>(a.y); u.z = dpp_from64<CTRL, ROW_MASK>(a.z);
    u.r = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)a.r); u.g = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)a.g);
    u.b = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)a.b); u.n = (unsigned int)dpp_from<CTRL, ROW_MASK>((int)a.n);
    u.head = dpp_from<CTRL, ROW_MASK>(a.head);
    if (!a.head) { seg_add(a, u); a.head = u.head; }
}

// What a block leaves behind for runs that cross its borders (64-bit throughout: a run may span thousands of blocks).
struct BlockPiece {
    long long x, y, z;
    unsigned long long r, g, b;
    unsigned int n;
    unsigned int flag;              // lead: 1 = the run goes on into the next block; trail: 1 = there is an open run
    unsigned int ordinal;           // trail: the voxel the open run belongs to
    unsigned int pad;
};

// trunc(s / n) for a coordinate sum: |s| < 2^52 and n < 2^32 are exact doubles, the correctly rounded quotient is off by at
// most 2^-38 (|s / n| <= 32 768) while a quotient that is not an integer is at least 1 / n > 2^-32 away from one, so truncating
// the double quotient IS the integer division — in ~10 instructions instead of the ~100 of a 64-bit integer divide, six of
// which per voxel made the reduce kernel VALU-bound (8.3 M wave-instructions: 13.5 of its 18.8 us). Sums that large cannot
// come out of this library's own pre-aggregation (|s| < 2^44); caller-made partials that exceed the bound take the integer path.
__device__ __forceinline__ long long div_sum(long long s, unsigned int n)
{
    if (__builtin_expect((unsigned long long)(s + (1ll << 52)) >> 53, 0)) return s / (long long)n;
    return (long long)((double)s / (double)n);
}
__device__ __forceinline__ unsigned long long div_sum(unsigned long long s, unsigned int n)
{
    if (__builtin_expect(s >> 52, 0)) return s / n;
    return (unsigned long long)((double)s / (double)n);
}

__device__ __forceinline__ void write_voxel(int16_t* __restrict__ out, unsigned int ordinal, long long sx, long long sy,
                                            long long sz, unsigned long long r, unsigned long long g, unsigned long long b,
                                            unsigned int n)
{
    int16_t* o = out + (size_t)ordinal * PCS_POINT_SHORTS;
    // (n == 0 cannot come out of this library's pre-aggregation; caller-made partials that sum to no points at all get their
    // sums written undivided instead of a division by zero)
    n = n ? n : 1u;
    o[0] = (int16_t)div_sum(sx, n);
    o[1] = (int16_t)div_sum(sy, n);
    o[2] = (int16_t)div_sum(sz, n);
    o[3] = (int16_t)(unsigned short)(div_sum(r, n) | (div_sum(g, n) << 8));
    o[4] = (int16_t)div_sum(b, n);
}

__global__ __launch_bounds__(kSegThreads)
void pcs_voxel_reduce_kernel(const unsigned long long* __restrict__ keys_a, const unsigned int* __restrict__ idx_a,
                             const unsigned long long* __restrict__ keys_b, const unsigned int* __restrict__ idx_b,
                             const VoxelPartial* __restrict__ part, const unsigned int* __restrict__ ctl, unsigned int bits,
                             unsigned int idx_bits, const unsigned int* __restrict__ heads,
                             const unsigned int* __restrict__ super, int16_t* __restrict__ out,
                             BlockPiece* __restrict__ lead, BlockPiece* __restrict__ trail)
{
    __shared__ SegSum wv[4];
    __shared__ unsigned int wheads[4], wbase[4];
    const bool in_a = sorted_in_a(ctl, bits);
    const unsigned long long* __restrict__ keys = in_a ? keys_a : keys_b;
    const unsigned int* __restrict__ idx = in_a ? idx_a : idx_b;
    const unsigned int m = ctl[0];
    const unsigned int blocks = (m + kSegThreads - 1) / kSegThreads;
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (unsigned int b = blockIdx.x; b < blocks; b += gridDim.x) {
        const unsigned int g = b * kSegThreads + threadIdx.x;
        const bool live = g < m;
        // the block's first voxel ordinal: runs begun in the groups before this block's + in the blocks before it in its group
        unsigned int obase = 0;
        for (unsigned int j = threadIdx.x; j < (b >> kSuperShift); j += kSegThreads) obase += super[(size_t)j * kSuperStride];
        if (threadIdx.x < (b & ((1u << kSuperShift) - 1u))) obase += heads[(b >> kSuperShift << kSuperShift) + threadIdx.x];
        unsigned long long key = 0, kprev = 0, knext = 0;
        VoxelPartial q{0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
            const unsigned long long e = keys[g];
            key = e >> idx_bits;
            kprev = g ? keys[g - 1] >> idx_bits : ~key;
            knext = g + 1 < m ? keys[g + 1] >> idx_bits : ~key;
            q = part[idx_bits ? (unsigned int)(e & ((1ull << idx_bits) - 1ull)) : idx[g]];
        }
        const bool head = !live || kprev != key;          // dead lanes (past m) are heads of empty runs: they absorb nothing
        const bool tail = live && knext != key;
        SegSum a{q.sx, q.sy, q.sz, q.r, q.g, q.b, q.n, head ? 1 : 0};
        seg_step64<0x111, 0xf>(a);                        // row_shr:1
        seg_step64<0x112, 0xf>(a);                        // row_shr:2
        seg_step64<0x114, 0xf>(a);                        // row_shr:4
        seg_step64<0x118, 0xf>(a);                        // row_shr:8
        seg_step64<0x142, 0xa>(a);                        // row_bcast:15 -> rows 1, 3
        seg_step64<0x143, 0xc>(a);                        // row_bcast:31 -> rows 2, 3
        const unsigned long long hb = __ballot(head && live);
        obase = wave_incl_scan(obase);
        if (lane == 63) { wv[wave] = a; wbase[wave] = obase; }
        if (lane == 0) wheads[wave] = __popcll(hb);
        __syncthreads();
        // carry from the earlier wavefronts of the block: back to (and including) the nearest one that holds a head
        SegSum c{0, 0, 0, 0, 0, 0, 0, 0};
        unsigned int heads_before = __popcll(hb & ((2ull << lane) - 1ull));       // heads at positions <= mine, this wavefront
        for (int u = (int)wave - 1; u >= 0; u--) {
            heads_before += wheads[u];
            if (!c.head) { seg_add(c, wv[u]); c.head = wv[u].head; }
        }
        if (!a.head) { seg_add(a, c); a.head = c.head; }
        // a.head now says: my run's head lies inside this block
        const unsigned int ordinal = wbase[0] + wbase[1] + wbase[2] + wbase[3] + heads_before - 1u;
        const unsigned int last = (m - b * kSegThreads < kSegThreads ? m - b * kSegThreads : kSegThreads) - 1u;   // last live lane
        if (tail && a.head) write_voxel(out, ordinal, a.x, a.y, a.z, a.r, a.g, a.b, a.n);
        if (live && !a.head && (tail || threadIdx.x == last))       // the run begun in an earlier block: its piece in here
            lead[b] = BlockPiece{a.x, a.y, a.z, a.r, a.g, a.b, a.n, tail ? 0u : 1u, 0u, 0u};
        if (threadIdx.x == 0 && head) lead[b] = BlockPiece{0, 0, 0, 0, 0, 0, 0, 0u, 0u, 0u};       // nothing reaches into this block
        if (threadIdx.x == last)
            trail[b] = (!tail && a.head) ? BlockPiece{a.x, a.y, a.z, a.r, a.g, a.b, a.n, 1u, ordinal, 0u}
                                         : BlockPiece{0, 0, 0, 0, 0, 0, 0, 0u, 0u, 0u};
        __syncthreads();
    }
}

// one lane per block with an open trailing run: add the pieces the following blocks hold of it. Its first wavefront also
// totals the voxels (the sum of the group counters) and clears the control words of the NEXT call on this workspace
// (plan_for), so that no call needs a memset launch of its own.
__global__ __launch_bounds__(256)
void pcs_voxel_fixup_kernel(unsigned int* __restrict__ ctl, const BlockPiece* __restrict__ lead,
                            const BlockPiece* __restrict__ trail, int16_t* __restrict__ out, const unsigned int* __restrict__ super,
                            int32_t* __restrict__ out_points, unsigned int* __restrict__ zero_next)
{
    const unsigned int m = ctl[0];
    const unsigned int blocks = (m + kSegThreads - 1) / kSegThreads;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 64) {
            const unsigned int groups = (blocks + (1u << kSuperShift) - 1u) >> kSuperShift;
            unsigned int t = 0;
            for (unsigned int j = threadIdx.x; j < groups; j += 64) t += super[(size_t)j * kSuperStride];
            t = wave_incl_scan(t);
            if (threadIdx.x == 63) {
                ctl[1] = t;
                if (out_points) *out_points = (int32_t)t;
            }
        } else if (threadIdx.x < 64 + kCtlWords) {
            if (zero_next) zero_next[threadIdx.x - 64] = 0u;
        }
    }
    for (unsigned int b = blockIdx.x * blockDim.x + threadIdx.x; b < blocks; b += gridDim.x * blockDim.x) {
        BlockPiece t = trail[b];
        if (!t.flag) continue;
        for (unsigned int bb = b + 1; bb < blocks; bb++) {
            const BlockPiece l = lead[bb];
            t.x += l.x; t.y += l.y; t.z += l.z; t.r += l.r; t.g += l.g; t.b += l.b; t.n += l.n;
            if (!l.flag) break;
        }
        write_voxel(out, t.ordinal, t.x, t.y, t.z, t.r, t.g, t.b, t.n);
    }
}


// ================================================================================================================
// The BUCKET tail (round 5): the same result as steps 2-3 above in 5 launches instead of 12.
//
// The LSD sort above is a chain of 3 x (histogram, column scan, scatter) + heads + reduce + fix-up = 12 dependent
// launches whose cost does not depend on their size at ~1 M partials: 79-85 us of which the memory work is a fraction.
// What the result needs is weaker than a sort of the partials: (a) all partials of a voxel in one place, (b) the voxels
// in key order. So:
//   P1  partition histogram   the partials are split ONCE into kBkt = 1024 contiguous KEY RANGES by kBkt - 1 splitters
//                             (upper_bound by binary search in LDS); per-chunk bucket counts, every element's bucket id
//   P2  column scan           of the per-chunk counts (as above, 1024 columns)
//   P3  scatter               (key, partial) to its bucket's range: 40 B per element, placed with returning LDS adds on a
//                             per-chunk cursor — unstable, which integer sums do not care about
//   G1  bucket reduce         one workgroup per bucket: LDS hash table over the bucket's ~870 partials (64-bit LDS
//                             compare-and-swap + 64-bit LDS adds), bitonic sort of the occupied (key, slot) words in LDS, one
//                             finished RECORD per voxel, in key order, parked at the bucket's own offset
//   W   write                 exclusive scan of the buckets' voxel counts (every workgroup for itself: 1024 words), records
//                             copied to their final place, voxel total, next call's control block cleared
// Buckets are key ranges, so bucket order + key order inside a bucket IS the (z, y, x) output order.
// SPLITTERS are quantiles of the key distribution. They only decide balance, never the result, so they come from the
// previous call on this workspace: G1 has every bucket's sorted keys and rewrites, in place, the splitters that fall into
// its share of the partials — consecutive frame-sets of a camera rig differ by sensor noise. A workspace without splitters
// for this leaf runs a one-workgroup sample sort first (S0: 4096 evenly spaced keys, bitonic in LDS).
// WARM CALLS: the splitters of a call are the previous call's, known before it starts. From a workspace's second bucket call on
// the pre-aggregation (pcs_kernels.hip: vox_table_flush_regions) therefore does P1 - P3 itself: every workgroup finds its
// partials' buckets and appends them to the buckets' REGIONS (B regions of `cap` slots in keys_r / part_r, filled through one
// cursor per bucket; B and cap left behind by the previous call's G1 together with zeroed cursors — two sets, used alternately).
// The tail is then G1 alone. A partial that finds its region full goes to the general list (keys_a / part, tagged with its bucket
// in bucket_of), and the bucket's workgroup gathers its own from there first (bkt_gather): regions only ever cost speed.
// SKEW / STALE SPLITTERS cost speed, never bits: a bucket whose distinct voxels do not fit the 1024-slot table is worked
// off in several passes over key sub-ranges [L, T): when the table fills up, T drops to the median of the keys seen so far
// and the pass restarts; every pass emits its voxels in key order behind the previous pass's. A hot voxel (thousands of
// partials of one key) is no skew at all: the bucket is streamed, only DISTINCT keys take slots.
// ================================================================================================================
#ifndef PCS_BKT
#define PCS_BKT 1024
#endif
constexpr unsigned int kBkt = PCS_BKT, kBktSample = 4096;      // buckets: 256 .. 1024, a power of two
static_assert(kBkt == kVoxBuckets, "the pre-aggregation's region flush (pcs_kernels.hip) partitions into kVoxBuckets ranges");
constexpr unsigned int kBktChunk = 4096, kBktThreads = 512, kBktPer = kBktChunk / kBktThreads;     // 8 elements per lane
constexpr unsigned int kBktGrid = 512;
#ifndef PCS_BKT_MIN_PER
#define PCS_BKT_MIN_PER 700
#endif
constexpr unsigned long long kBktMinPer = PCS_BKT_MIN_PER;       // fewest partials per bucket before a call halves its bucket count
constexpr unsigned int kBktSlots = 1024, kBktProbe = 48;           // G1's LDS table
constexpr unsigned long long kBktInf = ~0ull;

template <unsigned int N, unsigned int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* s)
{
    for (unsigned int size = 2; size <= N; size <<= 1)
        for (unsigned int stride = size >> 1; stride; stride >>= 1) {
            for (unsigned int t = threadIdx.x; t < N / 2; t += THREADS) {
                const unsigned int lo = 2u * t - (t & (stride - 1u)), hi = lo + stride;
                const bool up = (lo & size) == 0u;
                const unsigned long long a = s[lo], b = s[hi];
                if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
}

// S0: splitters from a regular sample of the keys (cold start of a workspace / a new leaf).
__global__ __launch_bounds__(1024)
void pcs_vox_bkt_sample_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ ctl, RawKeys raw,
                               unsigned long long* __restrict__ spl)
{
    __shared__ unsigned long long s[kBktSample];
    const unsigned int m = raw.keys ? raw.count() : ctl[0];
    const unsigned long long* __restrict__ k = raw.keys ? raw.keys : keys;
    for (unsigned int i = threadIdx.x; i < kBktSample; i += 1024u) {
        unsigned long long v = kBktInf;
        if (m) {
            const unsigned int e = (unsigned int)(((unsigned long long)i * m + (m >> 1)) / kBktSample);
            v = k[e < m ? e : m - 1u];
        }
        s[i] = v;
    }
    __syncthreads();
    bitonic_sort_lds<kBktSample, 1024u>(s);
    for (unsigned int j = threadIdx.x + 1u; j < kBkt; j += 1024u) spl[j - 1u] = s[j * (kBktSample / kBkt)];
    if (threadIdx.x == 0) spl[kBkt - 1u] = kBktInf;
}

// P1: table[chunk][bucket] = elements of the chunk in the bucket; bucket_of[e]
__global__ __launch_bounds__(kBktThreads)
void pcs_vox_bkt_hist_kernel(const unsigned long long* __restrict__ keys, unsigned int* __restrict__ ctl, RawKeys raw,
                             const unsigned long long* __restrict__ spl_g, unsigned int* __restrict__ table,
                             unsigned short* __restrict__ bucket_of)
{
    __shared__ unsigned long long spl[kBkt];
    __shared__ unsigned int hist[kBkt];
    const bool from_raw = raw.keys != nullptr;
    // raw input: this launch also publishes the element count for the device-driven kernels that follow
    const unsigned int m = from_raw ? raw.count() : ctl[0];
    if (from_raw && blockIdx.x == 0 && threadIdx.x == 0) ctl[0] = m;
    const unsigned long long* __restrict__ k = from_raw ? raw.keys : keys;
    const unsigned int chunks = (m + kBktChunk - 1u) / kBktChunk;
    if (blockIdx.x >= chunks) return;
    // How many buckets this call uses: a power of two with ~700 - 1400 partials each (a bucket is one workgroup's LDS table:
    // too many partials and it overflows, too few and 1024 workgroups queue up for nothing). The workspace keeps kBkt - 1
    // splitters (the 1/1024 quantiles); a call with B buckets takes every (kBkt / B)-th. Buckets B .. kBkt - 1 stay empty:
    // every later kernel sees zero counts for them and needs no special case.
    unsigned int B = kBkt;
    while (B > 32u && (unsigned long long)B * kBktMinPer > m) B >>= 1;
    const unsigned int stride = kBkt / B;
    for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) spl[j] = j + 1u < B ? spl_g[(j + 1u) * stride - 1u] : kBktInf;
    __syncthreads();
    // the splitters must ascend, or bucket order would not be key order. They do by construction (S0 sorts, G1 rewrites
    // all of them monotonically); if they ever did not, every workgroup sees the same array and takes the same way out: one
    // bucket for everything (slow, exact — and G1 then rewrites every splitter from sorted keys).
    bool bad = false;
    for (unsigned int j = threadIdx.x + 1u; j < kBkt; j += kBktThreads) bad |= spl[j - 1u] > spl[j];
    if (__syncthreads_or(bad ? 1 : 0)) {
        for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) spl[j] = kBktInf;
        __syncthreads();
    }
    for (unsigned int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) hist[j] = 0u;
        __syncthreads();
        const unsigned int c0 = chunk * kBktChunk;
        unsigned long long kk[kBktPer];
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const unsigned int e = c0 + q * kBktThreads + threadIdx.x;
            kk[q] = e < m ? k[e] : 0ull;
        }
        // the eight binary searches side by side (ten dependent LDS reads each)
        unsigned int lo[kBktPer];
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) lo[q] = 0u;
#pragma unroll
        for (unsigned int step = kBkt / 2; step; step >>= 1) {
#pragma unroll
            for (unsigned int q = 0; q < kBktPer; q++)
                if (spl[lo[q] + step - 1u] <= kk[q]) lo[q] += step;
        }
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const unsigned int e = c0 + q * kBktThreads + threadIdx.x;
            if (e < m) {
                atomicAdd(&hist[lo[q]], 1u);
                bucket_of[e] = (unsigned short)lo[q];
            }
        }
        __syncthreads();
        for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) table[(size_t)chunk * kBkt + j] = hist[j];
        __syncthreads();
    }
}

// P2: exclusive scan of every bucket's column over the chunks (in place) + the bucket's total (the scheme of
// pcs_voxel_colscan_kernel, kBkt columns).
__global__ __launch_bounds__(256)
void pcs_vox_bkt_colscan_kernel(unsigned int* __restrict__ table, const unsigned int* __restrict__ ctl, unsigned int* __restrict__ total)
{
    constexpr unsigned int kCols = 16;
    __shared__ unsigned int tot[16][16];
    const unsigned int m = ctl[0];
    const unsigned int chunks = (m + kBktChunk - 1u) / kBktChunk;
    if (chunks <= 256u) {
        if (blockIdx.x >= kBkt / 16u) return;
        const unsigned int d = blockIdx.x * 16u + (threadIdx.x & 15u), slot = threadIdx.x >> 4;
        const unsigned int rps = (chunks + 15u) / 16u, r0 = slot * rps;
        unsigned int v[16], sum = 0;
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) {
            const unsigned int r = r0 + j;
            v[j] = (j < rps && r < chunks) ? table[(size_t)r * kBkt + d] : 0u;
        }
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) sum += v[j];
        tot[slot][threadIdx.x & 15u] = sum;
        __syncthreads();
        unsigned int run = 0, all = 0;
#pragma unroll
        for (unsigned int q = 0; q < 16; q++) { const unsigned int t = tot[q][threadIdx.x & 15u]; run += q < slot ? t : 0u; all += t; }
#pragma unroll
        for (unsigned int j = 0; j < 16; j++) {
            const unsigned int r = r0 + j;
            if (j < rps && r < chunks) table[(size_t)r * kBkt + d] = run;
            run += v[j];
        }
        if (slot == 0) total[d] = all;
        return;
    }
    const unsigned int digit = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (digit >= kBkt) return;
    unsigned int carry = 0;
    for (unsigned int c0 = 0; c0 < chunks; c0 += 64 * kCols) {
        unsigned int v[kCols];
#pragma unroll
        for (unsigned int q = 0; q < kCols; q++) {
            const unsigned int c = c0 + q * 64 + lane;
            v[q] = c < chunks ? table[(size_t)c * kBkt + digit] : 0u;
        }
#pragma unroll
        for (unsigned int q = 0; q < kCols; q++) {
            const unsigned int c = c0 + q * 64 + lane;
            if (c0 + q * 64 < chunks) {                           // wave-uniform
                const unsigned int inc = wave_incl_scan(v[q]);
                if (c < chunks) table[(size_t)c * kBkt + digit] = carry + inc - v[q];
                carry += (unsigned int)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
    }
    if (lane == 0) total[digit] = carry;
}

// P3: every (key, partial) to its bucket's range. boff[b] = first element of bucket b (boff[kBkt] = m), written by block 0.
__global__ __launch_bounds__(kBktThreads)
void pcs_vox_bkt_scatter_kernel(const unsigned long long* __restrict__ keys, const VoxelPartial* __restrict__ part,
                                const unsigned short* __restrict__ bucket_of, const unsigned int* __restrict__ ctl, RawKeys raw,
                                const unsigned int* __restrict__ table, const unsigned int* __restrict__ total,
                                unsigned long long* __restrict__ keys_s, VoxelPartial* __restrict__ part_s,
                                unsigned int* __restrict__ boff)
{
    __shared__ unsigned int cur[kBkt];
    __shared__ unsigned int wsum[kBktThreads / 64];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned int m = ctl[0];
    const unsigned long long* __restrict__ k = raw.keys ? raw.keys : keys;
    const unsigned int chunks = (m + kBktChunk - 1u) / kBktChunk;
    if (blockIdx.x >= chunks && blockIdx.x != 0) return;
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // bucket bases: exclusive scan of the kBkt totals, kOwn consecutive buckets per lane
    constexpr unsigned int kOwn = (kBkt + kBktThreads - 1u) / kBktThreads;
    const bool owner = threadIdx.x * kOwn < kBkt;
    unsigned int tt[kOwn], s2 = 0;
#pragma unroll
    for (unsigned int q = 0; q < kOwn; q++) { tt[q] = owner ? total[threadIdx.x * kOwn + q] : 0u; s2 += tt[q]; }
    const unsigned int inc = wave_incl_scan(s2);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int run = inc - s2;
    for (unsigned int w = 0; w < wave; w++) run += wsum[w];
    unsigned int base[kOwn];
#pragma unroll
    for (unsigned int q = 0; q < kOwn; q++) { base[q] = run; run += tt[q]; }
    if (blockIdx.x == 0 && owner) {
#pragma unroll
        for (unsigned int q = 0; q < kOwn; q++) boff[threadIdx.x * kOwn + q] = base[q];
        if (threadIdx.x * kOwn + kOwn == kBkt) boff[kBkt] = run;
    }
    for (unsigned int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const unsigned int c0 = chunk * kBktChunk;
        // everything the chunk needs from memory, requested together
        unsigned long long kk[kBktPer];
        unsigned int bb[kBktPer];
        u32x4 pa[kBktPer], pb[kBktPer];
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const unsigned int e = c0 + q * kBktThreads + threadIdx.x;
            const bool live = e < m;
            kk[q] = live ? k[e] : 0ull;
            bb[q] = live ? bucket_of[e] : 0u;
            const u32x4* p4 = reinterpret_cast<const u32x4*>(part + (live ? e : 0u));
            pa[q] = p4[0]; pb[q] = p4[1];
        }
        if (owner) {
#pragma unroll
            for (unsigned int q = 0; q < kOwn; q++)
                cur[threadIdx.x * kOwn + q] = base[q] + table[(size_t)chunk * kBkt + threadIdx.x * kOwn + q];
        }
        __syncthreads();
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const unsigned int e = c0 + q * kBktThreads + threadIdx.x;
            if (e < m) {
                const unsigned int dst = atomicAdd(&cur[bb[q]], 1u);
                keys_s[dst] = kk[q];
                u32x4* o4 = reinterpret_cast<u32x4*>(part_s + dst);
                o4[0] = pa[q]; o4[1] = pb[q];
            }
        }
        __syncthreads();
    }
}

// PLACE: the warm form of P1 - P3 for CALLER-HELD partials (the root of a multi-GPU voxel grid: pcs_voxel_grid_from_partials_device on
// a workspace whose previous call left splitters, region sizes and zeroed cursors): every (key, partial) of the list straight into its
// bucket's region, exactly what the raster reader's flush does for its own partials (pcs_kernels.hip: vox_table_flush_regions) — the
// bucket by binary search in the splitters (8 KiB of LDS per workgroup, ten dependent LDS reads per key, a lane's keys side by side), a
// returning LDS add ranks the element among its chunk's for that bucket, ONE returning global add per bucket a chunk touches reserves
// the slots. An element that finds its region full goes to the general list (keys_l / part_l, its bucket in bucket_of, counted in
// ctl[0]) — regions only ever cost speed. One launch in place of histogram + column scan + scatter; the tail is then G1 alone.
__global__ __launch_bounds__(kBktThreads)
void pcs_vox_bkt_place_kernel(RawKeys raw, const VoxelPartial* __restrict__ part_in, const unsigned long long* __restrict__ spl_g,
                              const unsigned int* __restrict__ reg, unsigned int* __restrict__ cursor,
                              unsigned long long* __restrict__ keys_r, VoxelPartial* __restrict__ part_r, const unsigned int region_slots,
                              unsigned long long* __restrict__ keys_l, VoxelPartial* __restrict__ part_l,
                              unsigned short* __restrict__ bucket_of, unsigned int* __restrict__ ctl)
{
    __shared__ unsigned long long spl[kBkt];
    __shared__ unsigned int hist[kBkt], rbase[kBkt];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned int m = raw.count();
    const unsigned int chunks = (m + kBktChunk - 1u) / kBktChunk;
    if (blockIdx.x >= chunks) return;
    unsigned int B = reg[0], cap = reg[1];
    if (B == 0u || B > kBkt) B = kBkt;
    if ((unsigned long long)B * cap > region_slots) cap = 0u;              // (as the raster reader and the tail decide)
    const unsigned int stride = kBkt / B;
    for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) {
        spl[j] = (j + 1u < B) ? spl_g[(j + 1u) * stride - 1u] : kEmptyKey;   // bucket j ends below splitter j (the last one is open)
        hist[j] = 0u;
    }
    __syncthreads();
    for (unsigned int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const unsigned int c0 = chunk * kBktChunk;
        unsigned long long kk[kBktPer];
        u32x4 pa[kBktPer], pb[kBktPer];
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const unsigned int e = c0 + q * kBktThreads + threadIdx.x;
            const bool live = e < m;
            kk[q] = live ? raw.keys[e] : kEmptyKey;
            const u32x4* p4 = reinterpret_cast<const u32x4*>(part_in + (live ? e : 0u));
            pa[q] = p4[0]; pb[q] = p4[1];
        }
        unsigned int bk[kBktPer], rk[kBktPer];
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) bk[q] = 0u;
#pragma unroll
        for (unsigned int step = kBkt / 2; step; step >>= 1) {
#pragma unroll
            for (unsigned int q = 0; q < kBktPer; q++)
                if (spl[bk[q] + step - 1u] <= kk[q]) bk[q] += step;
        }
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            const bool live = c0 + q * kBktThreads + threadIdx.x < m;
            rk[q] = live ? atomicAdd(&hist[bk[q]], 1u) : 0u;
        }
        __syncthreads();
        for (unsigned int j = threadIdx.x; j < kBkt; j += kBktThreads) {
            const unsigned int c = hist[j];
            rbase[j] = c ? atomicAdd(&cursor[j], c) : 0u;
            hist[j] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (unsigned int q = 0; q < kBktPer; q++) {
            if (!(c0 + q * kBktThreads + threadIdx.x < m)) continue;
            const unsigned int at = rbase[bk[q]] + rk[q];
            if (at < cap) {
                const size_t dst = (size_t)bk[q] * cap + at;
                keys_r[dst] = kk[q];
                u32x4* o4 = reinterpret_cast<u32x4*>(part_r + dst);
                o4[0] = pa[q]; o4[1] = pb[q];
            } else {
                const unsigned int e = atomicAdd(ctl, 1u);
                keys_l[e] = kk[q];
                u32x4* o4 = reinterpret_cast<u32x4*>(part_l + e);
                o4[0] = pa[q]; o4[1] = pb[q];
                bucket_of[e] = (unsigned short)bk[q];
            }
        }
        __syncthreads();                                                   // rbase is rewritten by the next chunk
    }
}

// G1's table: key + six 64-bit sums per slot (a voxel may collect every point of the cloud: 2^25 points x 2^16 overflow 32
// bits in every field); blue and the count share a word (b < 2^34, n < 2^30).
__device__ __forceinline__ void bkt_hash(unsigned long long key, unsigned int& first, unsigned int& step)
{
    const unsigned int lo = (unsigned int)key, hi = (unsigned int)(key >> 32);
    const unsigned int h = __umul24(lo, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbit(hi, lo, 24), 0x85EBCBu);
    first = h >> 22;                                                       // 10 bits
    step = ((h >> 11) & (kBktSlots - 1u)) | 1u;                            // odd: the probe sequence visits every slot
}

// First output voxel of bucket b = the voxel counts of the buckets before it, which they publish (tagged with this call's
// generation: nobody clears the words) as soon as they know them. Waiting on LOWER-numbered workgroups only is safe on hardware
// that starts workgroups in order (what rocPRIM's look-back scan relies on too); the wait is bounded all the same: after ~0.5 s
// the workgroup gives up, flags the call (ctl[3] -> *out_points = -1) and carries on, so a launch can end wrong but never hang.
// The host forms that read the count re-run a flagged call on the LSD tail (pcs_capi_voxel.cpp, pcs_node.cpp). `bound`: the wait in
// 100 MHz ticks (kBktWaitTicks; a launch with an injected stall, pcs_inject_voxel_stall, waits 20 us only).
constexpr long long kBktWaitTicks = 50000000ll, kBktStallWaitTicks = 2000ll, kBktStallTicks = 40000ll;
__device__ __forceinline__ unsigned int bkt_base(const unsigned int* pub, unsigned int b, unsigned int gen, unsigned int* ctl, unsigned int* wsum,
                                                 const long long bound)
{
    unsigned int sum = 0;
    bool gave_up = false;
    for (unsigned int t = threadIdx.x; t < b; t += kBktThreads) {
        unsigned int v = __hip_atomic_load(pub + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 26) != gen) {
            const long long t0 = wall_clock64();                           // 100 MHz
            do {
                __builtin_amdgcn_s_sleep(4);
                v = __hip_atomic_load(pub + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((v >> 26) != gen && wall_clock64() - t0 < bound);
            if ((v >> 26) != gen) { gave_up = true; v = 0u; }
        }
        sum += v & ((1u << 26) - 1u);
    }
    if (gave_up) atomicOr(ctl + 3, 2u);
    const unsigned int inc = wave_incl_scan(sum);
    __syncthreads();                                                       // wsum may still be read as the dense scan's scratch
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned int total = 0;
    for (unsigned int w = 0; w < kBktThreads / 64; w++) total += wsum[w];
    __syncthreads();
    return total;
}

// sum of one word per lane over the workgroup (wsum: kBktThreads / 64 words of scratch)
__device__ __forceinline__ unsigned int bkt_block_sum(unsigned int v, unsigned int* wsum)
{
    const unsigned int inc = wave_incl_scan(v);
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned int total = 0;
    for (unsigned int w = 0; w < kBktThreads / 64; w++) total += wsum[w];
    __syncthreads();
    return total;
}

// two sums at once (both below 2^32): low and high half of one 64-bit word per lane
__device__ __forceinline__ unsigned long long bkt_block_sum2(unsigned int lo, unsigned int hi, unsigned long long* wsum2)
{
    const unsigned int slo = wave_incl_scan(lo), shi = wave_incl_scan(hi);      // (DPP: lane 63 holds the wavefront's sums)
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) wsum2[threadIdx.x >> 6] = (unsigned long long)slo | ((unsigned long long)shi << 32);
    __syncthreads();
    unsigned long long total = 0;
    for (unsigned int w = 0; w < kBktThreads / 64; w++) total += wsum2[w];
    return total;
}

// rank of v among NRUNS sorted runs of 64 words (padded with the sentinel) = its place in its own run + the words below it in
// every other run: the binary searches run side by side
template <unsigned int NRUNS>
__device__ __forceinline__ unsigned int bkt_rank(const unsigned long long* runs, unsigned long long v, unsigned int my_run, unsigned int lane)
{
    constexpr unsigned int G = NRUNS < 8u ? NRUNS : 8u;       // searches side by side (more would only cost registers)
    unsigned int r = 0;
    for (unsigned int g0 = 0; g0 < NRUNS; g0 += G) {
        unsigned int lo[G];
#pragma unroll
        for (unsigned int q = 0; q < G; q++) lo[q] = 0u;
#pragma unroll
        for (unsigned int step = 32; step; step >>= 1) {
#pragma unroll
            for (unsigned int q = 0; q < G; q++)
                if (runs[(g0 + q) * 64u + lo[q] + step - 1u] < v) lo[q] += step;
        }
#pragma unroll
        for (unsigned int q = 0; q < G; q++) {
            if (lo[q] == 63u && runs[(g0 + q) * 64u + 63u] < v) lo[q] = 64u;
            r += g0 + q == my_run ? lane : lo[q];
        }
    }
    return r;
}

// one wavefront's 64 words, ascending, in registers: bitonic network over cross-lane shuffles (21 steps)
__device__ __forceinline__ unsigned long long bkt_wave_sort(unsigned long long v, unsigned int lane)
{
    // (the 21 lane masks are recomputed at every call: hoisted out of the bucket loop they occupy 42 SGPRs for the whole kernel,
    // which then spills ~200 SGPRs and, through them, vector registers)
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (unsigned int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (unsigned int j = k >> 1; j > 0; j >>= 1) {
            const bool keep_min = ((lane & j) == 0u) == ((lane & k) == 0u);
            const unsigned long long o = __shfl_xor(v, (int)j);
            v = ((v < o) == keep_min) ? v : o;
        }
    }
    return v;
}

// lab only (-DPCS_BKT_TRACE): thread 0 of every reduce workgroup stamps wall_clock64 (100 MHz) at phase boundaries into the
// buffer whose address is in PCS_BKT_TRACE_PTR (tools/lab/bkt_trace.py); compiled out of the product
#ifndef PCS_BKT_TRACE
#define PCS_BKT_TRACE 0
#endif
#if PCS_BKT_TRACE
#define BKT_STAMP(i) do { if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 16u + (i)] = wall_clock64(); } while (0)
#else
#define BKT_STAMP(i) do { } while (0)
#endif
constexpr unsigned int kBktGiant = 8192, kBktSub = 256;       // a bucket beyond kBktGiant partials is first split into <= kBktSub key ranges

// A GIANT bucket (stale splitters: the cloud moved into one of the previous call's key ranges; or far more voxels than 1024
// tables hold). Passes over key sub-ranges would each re-read the whole bucket: n^2 / 768 partial reads, half a second for 400 k
// partials. Instead the workgroup splits the bucket ONCE more — up to 255 splitters from a sorted sample of 256 of its keys, a
// counting pass, a scatter into the (by now dead) pre-aggregation arrays — and the caller runs the ranges one after the other:
// three reads of the bucket instead of hundreds. Returns the number of ranges; soff[0 .. n_sub] = their bounds (relative to o0).
// The bucket's partials are keys_s / part_s [o0, o0 + n_a) and, on a warm call whose region overflowed, a second stretch
// ov.keys / ov.part [ov.r0, ov.r0 + ov.n) (bkt_gather); the ranges go to kscr / pscr [t0, t0 + n).
struct BktOverflow {
    const unsigned long long* keys;
    const VoxelPartial* part;
    unsigned int r0, n;
};
// A warm call's bucket whose region was full (its share of the cloud more than doubled since the previous call — the cloud moved):
// the partials that did not fit sit in the general list, in no order, tagged with their bucket. The workgroup copies its own
// to gk / gp from g0 on (cursor[b] - cap of them): eight tags per lane and step (one 16-byte load), places by a returning LDS add. Every
// overflowing bucket reads all tags — a stale call pays for that once, like the cold path's split of a giant bucket.
__device__ __forceinline__ void bkt_gather(const unsigned long long* __restrict__ lk, const VoxelPartial* __restrict__ lp,
                                        const unsigned short* __restrict__ ids, unsigned int n_list, unsigned int b,
                                        unsigned long long* __restrict__ gk, VoxelPartial* __restrict__ gp, unsigned int g0,
                                        unsigned int* cur)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (threadIdx.x == 0) *cur = 0u;
    __syncthreads();
    for (unsigned int base = 0; base < n_list; base += kBktThreads * 8u) {
        const unsigned int e0 = base + threadIdx.x * 8u;
        unsigned int match = 0;
        if (e0 + 8u <= n_list) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(ids + e0);      // (the list starts 256-byte aligned, e0 is a multiple of 8)
            const unsigned int w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (unsigned int k = 0; k < 4; k++) {
                match |= ((w[k] & 0xFFFFu) == b ? 1u : 0u) << (2u * k);
                match |= ((w[k] >> 16) == b ? 1u : 0u) << (2u * k + 1u);
            }
        } else {
            for (unsigned int k = 0; k < 8u; k++)
                if (e0 + k < n_list && ids[e0 + k] == b) match |= 1u << k;
        }
        if (match) {
            unsigned int dst = g0 + atomicAdd(cur, (unsigned int)__popc(match));
            while (match) {
                const unsigned int k = (unsigned int)__ffs((int)match) - 1u;
                match &= match - 1u;
                const unsigned int e = e0 + k;
                const u32x4* p4 = reinterpret_cast<const u32x4*>(lp + e);
                const u32x4 pa = p4[0], pb = p4[1];
                gk[dst] = lk[e];
                u32x4* o4 = reinterpret_cast<u32x4*>(gp + dst);
                o4[0] = pa; o4[1] = pb;
                dst++;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
}
__device__ __forceinline__ unsigned int bkt_presplit(const unsigned long long* __restrict__ keys_s, const VoxelPartial* __restrict__ part_s,
                                                  unsigned long long* __restrict__ kscr, VoxelPartial* __restrict__ pscr,
                                                  unsigned int o0, unsigned int n_a, const BktOverflow ov, unsigned int t0, unsigned int n,
                                                  unsigned long long* dl, unsigned long long* srt,
                                                  unsigned int* scur, unsigned int* soff, unsigned int* wcnt)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // as many ranges as give ~1500 partials each (a range costs a fixed ~6 us on top of its partials): every
    // (kBktSub / n_sub)-th of the 255 sample splitters
    unsigned int n_sub = 1;
    while (n_sub < kBktSub && n_sub * 1536u < n) n_sub <<= 1;
    const unsigned int sstride = kBktSub / n_sub;
    if (threadIdx.x < kBktSub) {
        const unsigned int at = (unsigned int)(((unsigned long long)threadIdx.x * n) / kBktSub);      // (over both stretches)
        const unsigned long long k = at < n_a ? keys_s[o0 + at] : ov.keys[ov.r0 + (at - n_a)];
        const unsigned long long v = bkt_wave_sort((k << 8) | threadIdx.x, lane);      // (distinct words: the index rides below the key)
        dl[wave * 64u + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < kBktSub) srt[bkt_rank<kBktSub / 64u>(dl, dl[threadIdx.x], wave, lane)] = dl[threadIdx.x];
    if (threadIdx.x < kBktSub) scur[threadIdx.x] = 0u;
    __syncthreads();
    auto sub_of = [&](unsigned long long key) {            // number of splitters srt[sstride * (1 .. n_sub - 1)] (>> 8) that are <= key
        unsigned int lo = 0;
        for (unsigned int step = n_sub >> 1; step; step >>= 1)
            if ((srt[(lo + step) * sstride] >> 8) <= key) lo += step;
        return lo;
    };
    for (unsigned int e = threadIdx.x; e < n_a; e += kBktThreads) atomicAdd(&scur[sub_of(keys_s[o0 + e])], 1u);
    for (unsigned int e = threadIdx.x; e < ov.n; e += kBktThreads) atomicAdd(&scur[sub_of(ov.keys[ov.r0 + e])], 1u);
    __syncthreads();
    {
        const unsigned int c = threadIdx.x < kBktSub ? scur[threadIdx.x] : 0u;
        const unsigned int inc = wave_incl_scan(c);
        if (lane == 63) wcnt[wave] = inc;
        __syncthreads();
        unsigned int excl = inc - c;
        for (unsigned int w = 0; w < wave; w++) excl += wcnt[w];
        if (threadIdx.x < kBktSub) { soff[threadIdx.x] = excl; scur[threadIdx.x] = excl; }
        if (threadIdx.x == 0) soff[n_sub] = n;
    }
    __syncthreads();
    for (unsigned int e = threadIdx.x; e < n_a + ov.n; e += kBktThreads) {
        const bool a = e < n_a;
        const unsigned long long key = a ? keys_s[o0 + e] : ov.keys[ov.r0 + (e - n_a)];
        const u32x4* p4 = reinterpret_cast<const u32x4*>(a ? part_s + o0 + e : ov.part + ov.r0 + (e - n_a));
        const u32x4 pa = p4[0], pb = p4[1];
        const unsigned int dst = t0 + atomicAdd(&scur[sub_of(key)], 1u);
        kscr[dst] = key;
        u32x4* o4 = reinterpret_cast<u32x4*>(pscr + dst);
        o4[0] = pa; o4[1] = pb;
    }
    __threadfence_block();
    __syncthreads();
    return n_sub;
}

__global__ __launch_bounds__(kBktThreads) __attribute__((amdgpu_waves_per_eu(4)))      // two workgroups per CU (LDS): <= 128 VGPRs
void pcs_vox_bkt_reduce_kernel(const unsigned long long* __restrict__ keys_s, const VoxelPartial* __restrict__ part_s,
                               const unsigned int* __restrict__ boff, unsigned int* __restrict__ ctl,
                               int16_t* __restrict__ out, int16_t* __restrict__ tmp_rec, unsigned int* __restrict__ pub, unsigned int gen,
                               unsigned long long* __restrict__ spl, int32_t* __restrict__ out_points, unsigned int* __restrict__ zero_next,
                               unsigned long long* __restrict__ kscr, VoxelPartial* __restrict__ pscr, long long* __restrict__ trace,
                               const unsigned int regions, const unsigned long long* __restrict__ keys_r, const VoxelPartial* __restrict__ part_r,
                               const unsigned int region_slots, const unsigned int* __restrict__ reg, unsigned int* __restrict__ reg_next,
                               const unsigned int* __restrict__ cursor, unsigned int* __restrict__ cursor_next,
                               const unsigned long long* __restrict__ kscr_list, const VoxelPartial* __restrict__ pscr_list,
                               const unsigned short* __restrict__ ov_ids, unsigned long long* __restrict__ gath_k,
                               VoxelPartial* __restrict__ gath_p, const unsigned int stall)
{
    __shared__ unsigned long long tkey[kBktSlots];
    __shared__ unsigned long long tx[kBktSlots], ty[kBktSlots], tz[kBktSlots], tr[kBktSlots], tg[kBktSlots], tbn[kBktSlots];
    __shared__ unsigned long long dl[kBktSlots];           // the occupied slots as (key << 10 | slot): dense, then sorted runs
    __shared__ unsigned long long srt[kBktSlots];          // ... in key order
    __shared__ unsigned int wcnt[kBktThreads / 64];
    __shared__ unsigned long long smp[64];
    __shared__ unsigned int soff[kBktSub + 1], scur[kBktSub];
    __shared__ unsigned long long wsum2[kBktThreads / 64];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    unsigned int m = ctl[0];
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // fault injection (pcs_inject_voxel_stall; the tests walk the give-up path with it): bucket 0's workgroup sleeps 0.4 ms before
    // it does anything, everybody else waits 20 us for a count — every later bucket gives up, the call ends flagged
    const long long wait_bound = stall ? kBktStallWaitTicks : kBktWaitTicks;
    if (stall && blockIdx.x == 0u) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < kBktStallTicks) __builtin_amdgcn_s_sleep(64);
    }
    // WARM call (regions): the pre-aggregation put bucket b's partials into region b — reg[1] slots at b * reg[1], cursor[b] of them
    // offered — and ctl[0] counts the partials that found their region full. Every workgroup sums the counts for itself (1024
    // words): the partials before its bucket are what the splitter refresh needs, their total what the next call is sized by.
    unsigned int reg_cap = 0, m_over = 0;
    if (regions) {
        reg_cap = reg[1];
        unsigned int reg_b = reg[0];
        if (reg_b == 0u || reg_b > kBkt) reg_b = kBkt;
        if ((unsigned long long)reg_b * reg_cap > region_slots) reg_cap = 0u;      // (as the pre-aggregation decided)
        m_over = m;
    }
    const unsigned long long* __restrict__ Kb = regions ? keys_r : keys_s;
    const VoxelPartial* __restrict__ Pb = regions ? part_r : part_s;
    for (unsigned int b = blockIdx.x; b < kBkt; b += gridDim.x) {
        BKT_STAMP(0);
        // o0: where the bucket's partials lie, n_a of them; n: all of the bucket's partials (a warm call: + those that found the
        // region full and sit in the general list); rank0: the partials of the buckets before it — also where, in the scratch
        // arrays, its parked records and split ranges go
        unsigned int o0, n_a, n, rank0, over_here = 0;
        if (regions) {
            constexpr unsigned int kPer = kBkt / kBktThreads;
            unsigned int below = 0, all = 0, over_below = 0;
#pragma unroll
            for (unsigned int q = 0; q < kPer; q++) {
                const unsigned int j = threadIdx.x * kPer + q, c = cursor[j];
                all += min(c, reg_cap);
                below += j < b ? min(c, reg_cap) : 0u;
                over_below += j < b ? c - min(c, reg_cap) : 0u;
            }
            const unsigned int cb = cursor[b];
            const unsigned long long t = bkt_block_sum2(below, all, wsum2);
            if (m_over) over_below = (unsigned int)bkt_block_sum2(over_below, 0u, wsum2);      // (uniform)
            o0 = b * reg_cap;
            n_a = min(cb, reg_cap);
            n = cb;
            rank0 = (unsigned int)t + over_below;
            over_here = over_below;
            m = (unsigned int)(t >> 32) + m_over;
        } else {
            o0 = boff[b]; n_a = n = boff[b + 1u] - o0; rank0 = o0;
        }
        // (over_below: the overflow of the buckets before this one = where, in the gather space, its own goes)
        if (n > n_a) bkt_gather(kscr_list, pscr_list, ov_ids, m_over, b, gath_k, gath_p, over_here, scur);      // (uniform)
        const BktOverflow ov{gath_k, gath_p, over_here, n - n_a};
        unsigned int emitted = 0, base = 0, fed = 0;
        bool have_base = false, published = false;

        // ---- one key range of the bucket: the rn partials at K / P [r0, r0 + rn), in passes over key sub-ranges [L, T) ----------
        // (+ a second stretch ovr; rn counts both)
        auto run = [&](const unsigned long long* __restrict__ K, const VoxelPartial* __restrict__ P, const unsigned int r0,
                       const unsigned int rn_a, const BktOverflow ovr, const unsigned int rn, const bool whole_bucket) {
            unsigned long long L = 0ull, T = kBktInf;              // this pass takes the keys in [L, T)
            bool more = rn != 0u;
            const bool big = rn > 2u * kBktSlots;                   // far more partials than slots: several passes
            if (big) {
                // 64 evenly spaced keys, sorted by the first wavefront: every pass takes an upper bound T from them that lets
                // about 3/4 of a table's worth of partials in, instead of finding out by overflowing
                if (wave == 0) {
                    const unsigned int at = (unsigned int)(((unsigned long long)lane * rn) >> 6);
                    smp[lane] = bkt_wave_sort(at < rn_a ? K[r0 + at] : ovr.keys[ovr.r0 + (at - rn_a)], lane);
                }
                __syncthreads();
            }
            while (more) {                                          // workgroup-uniform
                if (big && T == kBktInf) {
                    const unsigned int want = max(1u, (64u * 768u) / rn);       // sample keys per pass
                    unsigned int below = 0;
                    for (unsigned int i = 0; i < 64u; i++) below += smp[i] < L;
                    if (below + want < 64u) { const unsigned long long t = smp[below + want]; if (t > L) T = t; }
                }
                // the first batch's loads go out before the table is cleared
                unsigned long long key_n = 0ull;
                u32x4 pa_n = u32x4{0u, 0u, 0u, 0u}, pb_n = pa_n;
                if (threadIdx.x < rn_a) {
                    key_n = K[r0 + threadIdx.x];
                    const u32x4* p4 = reinterpret_cast<const u32x4*>(P + r0 + threadIdx.x);
                    pa_n = p4[0]; pb_n = p4[1];
                }
                {
                    u32x4* z4;
                    const u32x4 zero{0u, 0u, 0u, 0u}, ones{~0u, ~0u, ~0u, ~0u};
                    z4 = reinterpret_cast<u32x4*>(tkey); z4[threadIdx.x] = ones;                     // kEmptyKey = ~0
                    z4 = reinterpret_cast<u32x4*>(tx);  z4[threadIdx.x] = zero;
                    z4 = reinterpret_cast<u32x4*>(ty);  z4[threadIdx.x] = zero;
                    z4 = reinterpret_cast<u32x4*>(tz);  z4[threadIdx.x] = zero;
                    z4 = reinterpret_cast<u32x4*>(tr);  z4[threadIdx.x] = zero;
                    z4 = reinterpret_cast<u32x4*>(tg);  z4[threadIdx.x] = zero;
                    z4 = reinterpret_cast<u32x4*>(tbn); z4[threadIdx.x] = zero;
                }
                __syncthreads();
                BKT_STAMP(1);      // boff + first batch requested, table cleared
                bool restart = false, over_any = false;
                unsigned int my_in = 0;                              // partials this lane fed into the table in this pass
                // one batch (a partial per lane) meets the table; true (workgroup-uniform): somebody found the table crowded
                auto feed = [&](const bool live, const unsigned long long key, const u32x4 pa, const u32x4 pb) -> bool {
                    const bool over = live && key >= T;
                    const bool in = live && key >= L && !over;
                    int slot = -1;
                    if (in) {
                        unsigned int h, step;
                        bkt_hash(key, h, step);
                        // a key that finds no slot within kBktProbe probes calls the table crowded: the pass then restarts below
                        // the median. (Probing a full table to the end cost 1024 dependent LDS round trips per lane. With 24 probes
                        // a bucket of ~850 distinct voxels — one or two of 1024 on the config-5 scene, a different one every call —
                        // gave up, took two passes and published its count 10 us late, which every later bucket waits for: 48 probes
                        // took 6 us off the call at 50 mm and 24 us at 40 mm.)
                        for (unsigned int t = 0; t < kBktProbe; t++) {
                            const unsigned long long old = atomicCAS(&tkey[h], kEmptyKey, key);
                            if (old == kEmptyKey || old == key) { slot = (int)h; break; }
                            h = (h + step) & (kBktSlots - 1u);
                        }
                    }
                    over_any |= over;
                    my_in += in ? 1u : 0u;
                    if (__syncthreads_or((in && slot < 0) ? 1 : 0)) return true;
                    if (in) {
                        atomicAdd(&tx[slot], (unsigned long long)(long long)(int)pa.x);
                        atomicAdd(&ty[slot], (unsigned long long)(long long)(int)pa.y);
                        atomicAdd(&tz[slot], (unsigned long long)(long long)(int)pa.z);
                        atomicAdd(&tr[slot], (unsigned long long)pa.w);
                        atomicAdd(&tg[slot], (unsigned long long)pb.x);
                        atomicAdd(&tbn[slot], (unsigned long long)pb.y | ((unsigned long long)pb.z << 34));
                    }
                    return false;
                };
                for (unsigned int i0 = 0; i0 < rn_a; i0 += kBktThreads) {
                    const unsigned int e = i0 + threadIdx.x;
                    const unsigned long long key = key_n;
                    const u32x4 pa = pa_n, pb = pb_n;
                    {   // the next batch is requested before this one meets the table
                        const unsigned int e2 = e + kBktThreads;
                        if (e2 < rn_a) {
                            key_n = K[r0 + e2];
                            const u32x4* p4 = reinterpret_cast<const u32x4*>(P + r0 + e2);
                            pa_n = p4[0]; pb_n = p4[1];
                        }
                    }
                    if (feed(e < rn_a, key, pa, pb)) { restart = true; break; }
                }
                // the second stretch (a region that overflowed: rare)
                for (unsigned int i0 = 0; !restart && i0 < ovr.n; i0 += kBktThreads) {
                    const unsigned int e = i0 + threadIdx.x;
                    const bool live = e < ovr.n;
                    unsigned long long key = 0ull;
                    u32x4 pa = u32x4{0u, 0u, 0u, 0u}, pb = pa;
                    if (live) {
                        key = ovr.keys[ovr.r0 + e];
                        const u32x4* p4 = reinterpret_cast<const u32x4*>(ovr.part + ovr.r0 + e);
                        pa = p4[0]; pb = p4[1];
                    }
                    if (feed(live, key, pa, pb)) restart = true;
                }
                // did anybody meet a key at or above T? (also the barrier behind the last batch's adds)
                const bool beyond = __syncthreads_or(over_any ? 1 : 0) != 0;
                BKT_STAMP(2);      // all partials in the table
                // the occupied slots, dense: every lane owns slots 2t, 2t + 1
                const unsigned long long k0 = tkey[2u * threadIdx.x], k1 = tkey[2u * threadIdx.x + 1u];
                const unsigned int c2 = (k0 != kEmptyKey) + (k1 != kEmptyKey);
                const unsigned int inc = wave_incl_scan(c2);
                if (lane == 63) wcnt[wave] = inc;
                __syncthreads();
                unsigned int pos = inc - c2, cnt = 0;
                for (unsigned int w = 0; w < kBktThreads / 64; w++) { const unsigned int t = wcnt[w]; pos += w < wave ? t : 0u; cnt += t; }
                if (k0 != kEmptyKey) dl[pos++] = (k0 << 10) | (2u * threadIdx.x);
                if (k1 != kEmptyKey) dl[pos] = (k1 << 10) | (2u * threadIdx.x + 1u);
                // A bucket that is done in ONE pass (almost all are) writes its voxels straight to their final place, behind those of
                // all earlier buckets; it tells the later buckets its voxel count NOW, before it sorts. A bucket that takes several
                // passes parks its records at its own offset and moves them when it is through — if it waited for the earlier
                // buckets between its passes, the crowded buckets of a call would run one after the other.
                const bool direct = whole_bucket && !restart && !beyond && emitted == 0u;
                if (direct) {
                    if (threadIdx.x == 0) __hip_atomic_store(pub + b, (gen << 26) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = true;
                }
                __syncthreads();
                // Key order without a barrier-per-stage sort: a wavefront sorts its 64 entries in registers, parks the sorted run
                // in LDS, and every entry then finds its rank by binary search in the other runs (6 + 1 LDS reads each, side by
                // side). Entries are distinct words (the slot rides in the low bits). A lane holds entry t and, in a pass with
                // more than 512 voxels, entry t + 512 (runs 8..15). (A bitonic sort of the 1024 slots through LDS was 55 barrier
                // stages and half of this kernel; ranking every entry by counting all others was worse.)
                const unsigned int n_runs = cnt > kBktThreads ? 16u : 8u;
                unsigned long long v0 = threadIdx.x < cnt ? dl[threadIdx.x] : kBktInf;
                unsigned long long v1 = threadIdx.x + kBktThreads < cnt ? dl[threadIdx.x + kBktThreads] : kBktInf;
                __syncthreads();                                    // dl is read; it becomes the runs' home
                BKT_STAMP(3);      // dense list, count published
                v0 = bkt_wave_sort(v0, lane);
                if (n_runs == 16u) v1 = bkt_wave_sort(v1, lane);    // workgroup-uniform
                BKT_STAMP(7);      // runs sorted in registers
                dl[wave * 64u + lane] = v0;
                if (n_runs == 16u) dl[kBktThreads + wave * 64u + lane] = v1;
                __syncthreads();
                BKT_STAMP(8);      // runs parked
                if (v0 != kBktInf) srt[n_runs == 16u ? bkt_rank<16>(dl, v0, wave, lane) : bkt_rank<8>(dl, v0, wave, lane)] = v0;
                if (n_runs == 16u && v1 != kBktInf) srt[bkt_rank<16>(dl, v1, 8u + wave, lane)] = v1;
                __syncthreads();
                if (restart) {
                    // more distinct keys in [L, T) than the table takes gracefully: lower T to the median of those seen so far and
                    // start the pass again. The keys are distinct and cnt >= 2 (a lone key always finds its first slot), so the
                    // median is above the smallest of them and below T.
                    const unsigned long long t_new = srt[cnt / 2u] >> 10;
                    __syncthreads();
                    if (t_new >= T || t_new <= L) {                // cannot happen (distinct keys, cnt >= 2): never loop on it
                        if (threadIdx.x == 0) atomicOr(ctl + 3, 1u);
                        more = false;
                        continue;
                    }
                    T = t_new;
                    continue;
                }
                BKT_STAMP(4);      // ranked: key order known
                if (direct) { base = bkt_base(pub, b, gen, ctl, wcnt, wait_bound); have_base = true; }
                BKT_STAMP(5);      // base known
                int16_t* const rec = direct ? out : tmp_rec;
                const unsigned int first = direct ? base : rank0 + emitted;
                for (unsigned int i = threadIdx.x; i < cnt; i += kBktThreads) {
                    const unsigned int slot = (unsigned int)(srt[i] & (kBktSlots - 1u));
                    const unsigned long long bn = tbn[slot];
                    write_voxel(rec, first + i, (long long)tx[slot], (long long)ty[slot], (long long)tz[slot], tr[slot], tg[slot],
                                bn & ((1ull << 34) - 1ull), (unsigned int)(bn >> 34));
                }
                // Next call's splitters: the quantile positions j * m / kBkt that fall into THIS PASS's share of the partials — the
                // passes of a bucket (and the key ranges of a split bucket) come in ascending key order, so the pass holds the
                // partials of sorted rank [o0 + fed, o0 + fed + n_in) — read off the pass's sorted keys. Ascending within the pass,
                // across the passes and across the buckets. (Taking all of a bucket's positions from its LAST pass squeezed its
                // splitters into the top of its key range: the bucket grew from call to call.)
                // (a pass that took the whole range — the common case — fed all of it: no need to count)
                const unsigned int n_in = (L == 0ull && !beyond) ? rn : bkt_block_sum(my_in, wcnt);
                if (m != 0u && n_in != 0u) {
                    for (unsigned int j = threadIdx.x + 1u; j < kBkt; j += kBktThreads) {
                        const unsigned int q = (unsigned int)(((unsigned long long)j * m) / kBkt);
                        if (q >= rank0 + fed && q < rank0 + fed + n_in) {
                            const unsigned int i = (unsigned int)(((unsigned long long)(q - rank0 - fed) * cnt) / n_in);
                            spl[j - 1u] = srt[i < cnt ? i : cnt - 1u] >> 10;
                        }
                    }
                }
                BKT_STAMP(6);      // records written, splitters refreshed
                fed += n_in;
                emitted += cnt;
                // keys at or above T were left out: they are the next pass
                if (beyond) { L = T; T = kBktInf; } else more = false;
                __syncthreads();
            }
        };

        const bool giant = n > kBktGiant;
        const unsigned int n_sub = giant ? bkt_presplit(Kb, Pb, kscr, pscr, o0, n_a, ov, rank0, n, dl, srt, scur, soff, wcnt) : 1u;
        const BktOverflow ov_run{ov.keys, ov.part, ov.r0, giant ? 0u : ov.n};      // (a split bucket's ranges hold everything already)
        for (unsigned int s = 0; s < n_sub; s++) {                 // (soff lives in LDS of its own: run() reuses dl / srt)
            const unsigned int s0 = giant ? soff[s] : 0u, s1 = giant ? soff[s + 1u] : n;
            run(giant ? kscr : Kb, giant ? pscr : Pb, giant ? rank0 + s0 : o0, giant ? s1 - s0 : n_a, ov_run, s1 - s0, !giant);
        }

        if (!published && threadIdx.x == 0)                    // empty buckets, and buckets that took several passes
            __hip_atomic_store(pub + b, (gen << 26) | emitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!have_base && emitted != 0u) {                     // several passes: the parked records to their final place
            base = bkt_base(pub, b, gen, ctl, wcnt, wait_bound);
            have_base = true;
            __threadfence_block();
            const int16_t* __restrict__ src = tmp_rec + (size_t)rank0 * PCS_POINT_SHORTS;
            int16_t* __restrict__ dst = out + (size_t)base * PCS_POINT_SHORTS;
            for (unsigned int i = threadIdx.x; i < emitted * PCS_POINT_SHORTS; i += kBktThreads) dst[i] = src[i];
        }
        if (threadIdx.x == 0) cursor_next[b] = 0u;             // the next bucket call's cursors (nobody reads them in this launch)
        if (b == kBkt - 1u) {
            // the last bucket knows the total; it also clears the next call's control block (plan_for)
            if (!have_base) base = bkt_base(pub, b, gen, ctl, wcnt, wait_bound);
            if (threadIdx.x == 0) {
                const unsigned int total = base + emitted;
                const bool failed = __hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                ctl[1] = total;
                if (out_points) *out_points = failed ? -1 : (int32_t)total;
                // what a warm next call partitions by: as many buckets as P1 would take for this many partials, regions of twice
                // a bucket's share (the splitters are quantiles; sensor noise and motion move the shares by far less)
                unsigned int bn = kBkt;
                while (bn > 32u && (unsigned long long)bn * kBktMinPer > m) bn >>= 1;
                reg_next[0] = bn;
                reg_next[1] = 2u * (m / bn) + 64u;
            } else if (threadIdx.x >= 64 && threadIdx.x < 64 + kCtlWords) {
                if (zero_next) zero_next[threadIdx.x - 64] = 0u;
            }
        }
        __syncthreads();
    }
}

struct Workspace {
    unsigned long long *keys_a, *keys_b;
    unsigned int *idx_a, *idx_b;
    VoxelPartial* part;
    unsigned int *table, *digit_total, *heads, *super;
    BlockPiece *lead, *trail;
    unsigned int *ctl, *ctl_next;       // this call's control words (ctl[0] = m partials, [1] = voxels, [32..35] key
                                        // bits) and the block the NEXT call on this workspace will use: see plan_for
    // bucket tail: splitters (kBkt words, persistent across calls), per-chunk bucket counts, bucket totals / offsets / voxel
    // counts, the bucket-sorted partials and the parked records. keys_s aliases keys_b, bucket_of aliases idx_a (the LSD
    // path's buffers; a call takes one tail or the other)
    unsigned long long* spl;
    unsigned int *btable, *btotal, *boff, *dcount;
    unsigned long long* keys_s;
    unsigned short* bucket_of;
    VoxelPartial* part_s;
    VoxelPartial* part_ws;              // the workspace's own partial array (w.part may be redirected to a caller's)
    int16_t* tmp_rec;
    // warm calls ("regions"): per-bucket cursors and {buckets, slots per region}, two of each (bucket call k uses k & 1 and
    // prepares the other for k + 1), and the regions themselves
    unsigned int *cursor, *reg;
    unsigned long long* keys_r;
    VoxelPartial* part_r;
    size_t region_slots, region_tail;
    size_t bytes;
};

// level: what the call's tail needs. 0 = LSD sort + segmented mean (~56 B per point); 1 = + the bucket tail's cold chain (chunk
// table, scattered partials, parked records: ~98 B); 2 = + the regions a warm bucket call's pre-aggregation fills (~219 B). The
// arrays of a higher level lie BEHIND those of the lower ones, so a workspace sized for a level holds every lower one unchanged.
inline Workspace carve(uint8_t* base, size_t n, int level = 2)
{
    Workspace w{};
    uint8_t* p = base;
    auto take = [&](size_t bytes) { uint8_t* q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    w.ctl = (unsigned int*)take(2 * kCtlWords * sizeof(unsigned int));      // first: where they are must not depend on n
    w.ctl_next = w.ctl + kCtlWords;
    w.dcount = (unsigned int*)take(kBkt * 4);                                // bucket tail: published voxel counts, cleared WITH the control blocks
    w.cursor = (unsigned int*)take(2 * kBkt * 4);                            // ... and so are the region cursors and sizes
    w.reg = (unsigned int*)take(2 * 64 * 4);
    w.spl = (unsigned long long*)take(kBkt * 8);                             // (its place must not depend on n either)
    w.btotal = (unsigned int*)take(kBkt * 4);
    w.boff = (unsigned int*)take((kBkt + 1) * 4);
    w.keys_a = (unsigned long long*)take(n * 8);
    w.keys_b = (unsigned long long*)take(n * 8);
    w.idx_a = (unsigned int*)take(n * 4);
    w.idx_b = (unsigned int*)take(n * 4);
    w.part = (VoxelPartial*)take(n * sizeof(VoxelPartial));
    {   // rows of the histogram table: m / sort_chunk_of(m) is at most kSortMinRows below 4 M partials, m / 8192 above
        const size_t rows = std::max<size_t>((n + kSortChunk - 1) / kSortChunk, std::min<size_t>(kSortMinRows, (n + kSortMinChunk - 1) / kSortMinChunk));
        w.table = (unsigned int*)take(rows * (size_t)kRadix * 4);
    }
    w.digit_total = (unsigned int*)take((size_t)kRadix * 4);
    w.heads = (unsigned int*)take(((n + kSegThreads - 1) / kSegThreads) * 4);
    w.super = (unsigned int*)take(((((n + kSegThreads - 1) / kSegThreads) >> kSuperShift) + 2) * (size_t)kSuperStride * 4);
    w.lead = (BlockPiece*)take(((n + kSegThreads - 1) / kSegThreads) * sizeof(BlockPiece));
    w.trail = (BlockPiece*)take(((n + kSegThreads - 1) / kSegThreads) * sizeof(BlockPiece));
    const size_t bytes0 = (size_t)(p - base);
    w.btable = (unsigned int*)take(((n + kBktChunk - 1) / kBktChunk + 1) * (size_t)kBkt * 4);
    w.part_s = (VoxelPartial*)take(n * sizeof(VoxelPartial));
    w.tmp_rec = (int16_t*)take(n * (size_t)PCS_POINT_BYTES + 16);      // records of buckets that take several passes
    // any cloud's regions fit: B x (2 (m / B) + 64) slots with m <= n partials in B <= kBkt buckets; behind them (region_tail) a
    // stretch of n slots: a warm call's scratch for the ranges of split buckets (the cold chain's is keys_a / part, where a warm
    // call keeps the partials that found their region full)
    const size_t bytes1 = (size_t)(p - base);
    w.region_slots = 2 * n + 64 * (size_t)kBkt + 64;
    w.region_tail = w.region_slots;
    w.keys_r = (unsigned long long*)take((w.region_slots + n) * 8);
    w.part_r = (VoxelPartial*)take((w.region_slots + n) * sizeof(VoxelPartial));
    w.part_ws = w.part;
    w.keys_s = w.keys_b;
    w.bucket_of = (unsigned short*)w.idx_a;
    w.bytes = level >= 2 ? (size_t)(p - base) : level == 1 ? bytes1 : bytes0;
    return w;
}

}  // namespace

// worst case: every point its own partial
size_t voxel_workspace_bytes(uint32_t n_points, int level) { return carve(nullptr, n_points, level).bytes + 256; }

namespace {

struct Plan {
    Workspace w;
    VoxelDiv dv;
    unsigned int bits, idx_bits;
    bool track_bits;      // have the pre-aggregation record which key bits vary, so that the sort can skip passes
    bool bucket = false;       // the bucket tail (5 launches) instead of the LSD sort + segmented mean (12)
    bool need_sample = false;  // bucket tail: the workspace holds no splitters for this leaf yet
    unsigned int gen = 1;      // bucket tail: tag of this call's published bucket counts (1 .. 63, never the previous call's)
    bool regions = false;      // bucket tail, warm: the pre-aggregation fills the buckets' regions, the tail is ONE launch
    unsigned int bpar = 0;     // bucket tail: which of the two cursor / region-size blocks this call uses
    RawKeys raw{nullptr, nullptr, 0u};      // exchange format: the first pass reads caller-held raw keys
};

// Which tail a call takes. Both give the same bytes for every input; they differ in what they cost. The bucket tail is built
// for the ~1 M partials of BASELINE configs[4] (leaves of a few centimetres and up on a room-sized scene): 1024 buckets of
// ~1000 partials. With many more partials its buckets outgrow the LDS table and are worked off in several passes — the LSD
// sort is the better tool there. The host does not know the number of partials (nothing is read back), so the choice goes by
// the leaf: PCS_VOXEL_TAIL=bucket / lsd overrides (read at every call: the tests run both).
bool choose_bucket_tail(int leaf_mm, int pref, bool stalled)
{
    if (stalled) return false;      // latched after a flagged call (PCS_VOXEL_TAIL_LSD_LATCHED): not for the environment to undo
    if (const char* v = getenv("PCS_VOXEL_TAIL")) {
        if (v[0] == 'b') return true;
        if (v[0] == 'l') return false;
    }
    if (pref == 1) return true;
    if (pref == 2) return false;
    // 16 x 1080p synthetic scene, ms per warm call bucket / LSD: 20 mm 0.601 / 0.567, 25 mm 0.443 / 0.458, 28 mm 0.391 / 0.430,
    // 32 mm 0.303 / 0.291, 36 mm 0.233 / 0.272, 40 mm 0.220 / 0.257, 50 mm 0.184 / 0.230, 100 mm 0.156 / 0.194
    return leaf_mm >= 34;
}

// (a bucket publishes its voxel count in 26 bits beside a 6-bit tag: clouds of 2^26 points and more take the LSD tail)
bool takes_bucket_tail(uint32_t n_points, int leaf_mm, const VoxelWsState& ws)
{
    return choose_bucket_tail(leaf_mm, ws.tail_pref, ws.stalled) && n_points < (1u << 26);
}

// The constants of floor(v / leaf) + bias for one leaf (pcs_voxel_agg.h: VoxelDiv) + the bits one axis takes.
hipError_t div_for(int leaf_mm, VoxelDiv& dv, unsigned int& bits)
{
    if (leaf_mm < 1 || leaf_mm > 32767) return hipErrorInvalidValue;
    bits = axis_bits(leaf_mm);
    const unsigned int bias_leaf = (32768u + (unsigned)leaf_mm - 1u) / (unsigned)leaf_mm * (unsigned)leaf_mm;
    dv = VoxelDiv{(float)(1.0 / leaf_mm), (float)(((double)bias_leaf + 0.5) / leaf_mm)};
    // (unsigned)fmaf(v, inv, c) == floor((v + bias * leaf) / leaf) for every int16 v: proven by the bound in pcs_voxel_agg.h,
    // tested over all leaves (tests/test_voxel_grid.py), and re-checked here with the host's fmaf — the same single rounding
    // as the device's v_fma_f32 — over the 65 536 values of the leaf in use (once per leaf per host thread). A failed check
    // (impossible by the bound) refuses the call.
    thread_local int cached_leaf = 0;
    thread_local bool cached_ok = false;
    if (cached_leaf != leaf_mm) {
        bool ok = true;
        for (int v = -32768; ok && v <= 32767; v++)
            ok = (unsigned int)std::fmaf((float)v, dv.inv, dv.c) == (unsigned int)(v + (int)bias_leaf) / (unsigned)leaf_mm;
        cached_leaf = leaf_mm;
        cached_ok = ok;
    }
    return cached_ok ? hipSuccess : hipErrorInvalidValue;
}

// Carves the workspace and derives the key layout for a cloud of at most n_points points.
// The control words: a call needs 64 zeroed words (counters, key bits). Instead of a memset launch per call (one more
// dependent dispatch) the workspace holds TWO blocks at its start; call k uses block k & 1 and its block-scan kernel
// clears the other one for call k + 1 (nothing of call k touches that block; call k - 1, which used it, is behind on the
// stream). `ws` remembers the parity; a workspace it has not seen, or one whose last call may not have been enqueued
// completely (ws.clean == false), gets both blocks cleared by a memset first.
hipError_t plan_for(uint32_t n_points, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState& ws, Plan& pl, hipStream_t st,
                    bool from_partials = false)
{
    if (ws_bytes < voxel_workspace_bytes(n_points, voxel_workspace_level(n_points, leaf_mm, ws, from_partials))) return hipErrorInvalidValue;
    uint8_t* base = static_cast<uint8_t*>(d_ws);
    base += (256 - ((uintptr_t)base & 255)) & 255;
    pl.w = carve(base, n_points);
#if PCS_BKT_TRACE
    { char buf[32]; snprintf(buf, sizeof buf, "%llu", (unsigned long long)(uintptr_t)base); setenv("PCS_BKT_WS_BASE", buf, 1); }      // lab: tools/lab/bkt_regions.py
#endif
    if (!ws.clean || ws.base != d_ws) {
        ws.clean = false;
        // (the control blocks and, right behind them, the bucket tail's published counts: their tags must not be garbage)
        const hipError_t e = hipMemsetAsync(pl.w.ctl, 0, (size_t)((uint8_t*)(pl.w.reg + 2 * 64) - (uint8_t*)pl.w.ctl), st);
        if (e != hipSuccess) return e;
        if (ws.base != d_ws) ws.spl_leaf = 0;                    // another workspace: whatever splitters it holds are not ours
        ws.base = d_ws; ws.phase = 0; ws.bkt_calls = 0; ws.clean = true;
    }
    if (ws.phase & 1u) std::swap(pl.w.ctl, pl.w.ctl_next);
    {
        const hipError_t e = div_for(leaf_mm, pl.dv, pl.bits);
        if (e != hipSuccess) return e;
    }
    // one 64-bit word per element when the packed key and the partial's index fit together
    pl.idx_bits = 1;
    while ((1ull << pl.idx_bits) < (unsigned long long)n_points) pl.idx_bits++;
    static const int pack_ok = [] { const char* v = getenv("PCS_VOXEL_PACKED"); return v ? atoi(v) : 1; }();
    if (!pack_ok || 3u * pl.bits + pl.idx_bits > 64u) pl.idx_bits = 0;
    // A pass is skipped when the varying bits fit one digit less than the key's 3 * bits. A scene some 8 m across has
    // 3 * log2(8 m / leaf) varying bits: 30 of 39 at 10 mm (3 passes instead of 4: 16 x 1080p from the rasters 1.36 -> 1.22
    // ms), 27 of 36 at 20 mm (0.65 -> 0.61 ms), 24 of 33 at 50 mm (3 of 3: nothing to gain), 21 of 30 at 100 mm (2 of 3).
    // Recording the bits costs the raster reader ~9 us per 16 x 1080p frame-set and a skipped pass is still three (empty)
    // launches, so it only pays where a pass is long: it is asked for when the key needs four or more passes (leaves below
    // 33 mm) — at 100 / 200 mm, where the sort handles ~0.1 M partials, it measured +5 us. A decision about speed only:
    // without the record every bit counts as varying. PCS_VOXEL_TRACK=0/1 overrides.
    static const int track_env = [] { const char* v = getenv("PCS_VOXEL_TRACK"); return v ? atoi(v) : -1; }();
    pl.track_bits = 3u * pl.bits > 3u * kRadixBits;
    if (track_env >= 0) pl.track_bits = track_env != 0;
    pl.bucket = takes_bucket_tail(n_points, leaf_mm, ws);
    if (pl.bucket) {
        pl.idx_bits = 0;              // raw keys, as in the exchange format: the bucket tail moves the partials themselves
        pl.track_bits = false;
        pl.need_sample = ws.spl_leaf != leaf_mm;
        pl.gen = 1u + ws.bkt_calls % 63u;                 // differs from the previous bucket call's; a cleared workspace holds tag 0
        pl.bpar = ws.bkt_calls & 1u;
        // warm: the previous bucket call on this workspace left splitters for this leaf, region sizes and zeroed cursors behind
        // (PCS_VOXEL_REGIONS=0: every bucket call takes the cold chain — read at every call, the tests run both)
        const char* regions_env = getenv("PCS_VOXEL_REGIONS");
        pl.regions = !(regions_env && regions_env[0] == '0') && !pl.need_sample && ws.bkt_calls > 0;
    }
    return hipSuccess;
}

// Fault injection: the next `launches` bucket-tail launches of this process run with a stalled first workgroup and end flagged
// (*out_points = -1). PCS_BKT_INJECT_STALL=<launches> sets the counter at its first use (the CLIs have no other way in).
std::atomic<int> g_stall_left{-1};
bool take_injected_stall()
{
    int v = g_stall_left.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("PCS_BKT_INJECT_STALL");
        int want = e ? atoi(e) : 0;
        if (want < 0) want = 0;
        if (g_stall_left.compare_exchange_strong(v, want)) v = want; else v = g_stall_left.load();
    }
    while (v > 0)
        if (g_stall_left.compare_exchange_weak(v, v - 1)) return true;
    return false;
}

// The bucket tail on the partials a pre-aggregation left in the workspace (or a caller handed over: pl.raw).
hipError_t bucket_tail(const Plan& pl, uint32_t n_points, int16_t* d_out, int32_t* d_out_points, hipStream_t st)
{
    const Workspace& w = pl.w;
    const unsigned int max_chunks = (n_points + kBktChunk - 1u) / kBktChunk;
    const unsigned int grid = std::max(1u, std::min(max_chunks, kBktGrid));
    if (pl.need_sample)
        hipLaunchKernelGGL(pcs_vox_bkt_sample_kernel, dim3(1), dim3(1024), 0, st, w.keys_a, w.ctl, pl.raw, w.spl);
    if (!pl.regions) {
    hipLaunchKernelGGL(pcs_vox_bkt_hist_kernel, dim3(grid), dim3(kBktThreads), 0, st, w.keys_a, w.ctl, pl.raw, w.spl, w.btable, w.bucket_of);
    hipLaunchKernelGGL(pcs_vox_bkt_colscan_kernel, dim3(kBkt / 4), dim3(256), 0, st, w.btable, w.ctl, w.btotal);
    hipLaunchKernelGGL(pcs_vox_bkt_scatter_kernel, dim3(grid), dim3(kBktThreads), 0, st, w.keys_a, w.part, w.bucket_of, w.ctl, pl.raw,
                       w.btable, w.btotal, w.keys_s, w.part_s, w.boff);
    }
    hipLaunchKernelGGL(pcs_vox_bkt_reduce_kernel, dim3(kBkt), dim3(kBktThreads), 0, st, w.keys_s, w.part_s, w.boff, w.ctl, d_out,
                       w.tmp_rec, w.dcount, pl.gen, w.spl, d_out_points, w.ctl_next, pl.regions ? w.keys_r + w.region_tail : w.keys_a,
                       pl.regions ? w.part_r + w.region_tail : w.part_ws,
                       (long long*)(PCS_BKT_TRACE && getenv("PCS_BKT_TRACE_PTR") ? strtoull(getenv("PCS_BKT_TRACE_PTR"), nullptr, 0) : 0ull),
                       pl.regions ? 1u : 0u, w.keys_r, w.part_r, (unsigned int)std::min<size_t>(w.region_slots, 0xFFFFFFFFu),
                       w.reg + 64 * pl.bpar, w.reg + 64 * (pl.bpar ^ 1u), w.cursor + kBkt * pl.bpar, w.cursor + kBkt * (pl.bpar ^ 1u),
                       w.keys_a, w.part, w.bucket_of, w.keys_s, w.part_s, take_injected_stall() ? 1u : 0u);
    return hipGetLastError();
}

// Steps 2 and 3 on the partials a pre-aggregation left in the workspace.
hipError_t sort_and_reduce(const Plan& pl, uint32_t n_points, int16_t* d_out, int32_t* d_out_points, hipStream_t st)
{
    if (pl.bucket) return bucket_tail(pl, n_points, d_out, d_out_points, st);
    const Workspace& w = pl.w;
    const unsigned int idx_bits = pl.idx_bits;
    const unsigned int n_passes = (3u * pl.bits + kRadixBits - 1u) / kRadixBits;  // the device may skip the first few (SortPass)
    const unsigned int bits = pl.bits | (pl.track_bits ? kTrackFlag : 0u);         // what the kernels get: width + tracking flag

    // grids sized for what the launch can need at most, capped: the kernels loop over chunks / blocks
    const unsigned int max_chunks = std::max((n_points + kSortChunk - 1) / kSortChunk,
                                             std::min(kSortMinRows, (n_points + kSortMinChunk - 1) / kSortMinChunk));
    const unsigned int sort_grid = max_chunks < kSortGrid ? max_chunks : kSortGrid;
    for (unsigned int pass = 0; pass < n_passes; pass++) {
        hipLaunchKernelGGL(pcs_voxel_hist_kernel, dim3(sort_grid), dim3(kSortThreads), 0, st, w.keys_a, w.keys_b, w.ctl, bits, idx_bits,
                           pass, n_passes, w.table, w.super, pl.raw);
        hipLaunchKernelGGL(pcs_voxel_colscan_kernel, dim3(kRadix / 4), dim3(256), 0, st, w.table, w.ctl, w.digit_total, bits, pass, n_passes);
        if (idx_bits)
            hipLaunchKernelGGL(pcs_voxel_scatter_kernel<true>, dim3(sort_grid), dim3(kSortThreads), 0, st, w.keys_a, w.idx_a, w.keys_b,
                               w.idx_b, w.ctl, bits, idx_bits, pass, n_passes, w.table, w.digit_total, pl.raw);
        else
            hipLaunchKernelGGL(pcs_voxel_scatter_kernel<false>, dim3(sort_grid), dim3(kSortThreads), 0, st, w.keys_a, w.idx_a, w.keys_b,
                               w.idx_b, w.ctl, bits, idx_bits, pass, n_passes, w.table, w.digit_total, pl.raw);
    }
    const unsigned int max_blocks = (n_points + kSegThreads - 1) / kSegThreads;
    const unsigned int seg_grid = max_blocks < kSegGrid ? max_blocks : kSegGrid;
    hipLaunchKernelGGL(pcs_voxel_heads_kernel, dim3(seg_grid), dim3(kSegThreads), 0, st, w.keys_a, w.keys_b, w.ctl, bits, idx_bits, w.heads,
                       w.super);
    hipLaunchKernelGGL(pcs_voxel_reduce_kernel, dim3(seg_grid), dim3(kSegThreads), 0, st, w.keys_a, w.idx_a, w.keys_b, w.idx_b, w.part,
                       w.ctl, bits, idx_bits, w.heads, w.super, d_out, w.lead, w.trail);
    const unsigned int fix_grid = (max_blocks + 255) / 256 < 64 ? (max_blocks + 255) / 256 : 64;
    hipLaunchKernelGGL(pcs_voxel_fixup_kernel, dim3(fix_grid), dim3(256), 0, st, w.ctl, w.lead, w.trail, d_out, w.super, d_out_points,
                       w.ctl_next);
    return hipGetLastError();
}

// What a pre-aggregation needs to fill the buckets' regions itself (warm bucket tail).
void stage_regions(const Plan& pl, VoxelStage& vs)
{
    const Workspace& w = pl.w;
    vs.regions = pl.regions ? 1u : 0u;
    vs.region_slots = (uint32_t)std::min<size_t>(w.region_slots, 0xFFFFFFFFu);
    vs.spl = w.spl; vs.reg = w.reg + 64 * pl.bpar; vs.cursor = w.cursor + kBkt * pl.bpar;
    vs.keys_r = w.keys_r; vs.part_r = w.part_r; vs.bucket_of = w.bucket_of;
}

// A call that was enqueued completely hands the other control block to the next one; anything else leaves the workspace
// to be cleared again.
hipError_t finish_call(VoxelWsState& ws, hipError_t e, int bucket_leaf = 0)
{
    if (e == hipSuccess) {
        ws.phase++;
        if (bucket_leaf) { ws.spl_leaf = bucket_leaf; ws.bkt_calls++; }      // the bucket tail leaves splitters for this leaf behind
    } else {
        ws.clean = false;
        ws.spl_leaf = 0;
    }
    return e;
}

}  // namespace

// What a call on `ws` for this leaf needs of the workspace (carve): the owner sizes it with voxel_workspace_bytes(n, level).
// (Caller-held partials are placed into regions too from a workspace's second bucket call on — pcs_vox_bkt_place_kernel — so both
// forms size alike; `from_partials` is kept for the day they do not.)
int voxel_workspace_level(uint32_t n_points, int leaf_mm, const VoxelWsState& ws, bool from_partials)
{
    (void)from_partials;
    return takes_bucket_tail(n_points, leaf_mm, ws) ? 2 : 0;
}

void inject_voxel_stall(int launches) { g_stall_left.store(launches < 0 ? 0 : launches); }

// d_n_points != nullptr: the number of points is read from device memory (<= n_points, which then is the capacity that
// sizes the workspace and the grids)
hipError_t launch_voxel_grid(const int16_t* d_payload, uint32_t n_points, const int32_t* d_n_points, int leaf_mm, void* d_ws,
                             size_t ws_bytes, VoxelWsState* ws, int16_t* d_out, int32_t* d_out_points, hipStream_t st)
{
    if (n_points == 0) {
        if (d_out_points) return hipMemsetAsync(d_out_points, 0, sizeof(int32_t), st);
        return hipSuccess;
    }
    Plan pl;
    hipError_t e = plan_for(n_points, leaf_mm, d_ws, ws_bytes, *ws, pl, st);
    if (e != hipSuccess) return e;
    const Workspace& w = pl.w;
    const unsigned int per_block = (unsigned)kAggThreads * (unsigned)kAggPerLane;
    const dim3 agg_grid((n_points + per_block - 1) / per_block);
    static const int wide_ok = [] { const char* v = getenv("PCS_VOXEL_WIDE"); return v ? atoi(v) : 1; }();
    static const int lane8_ok = [] { const char* v = getenv("PCS_VOXEL_LANE8"); return v ? atoi(v) : 1; }();
    if (lane8_ok && ((uintptr_t)d_payload & 15u) == 0u) {
        // 16-byte aligned payload: the reader that shares its table code with the raster reader (pcs_kernels.hip)
        VoxelStage vs{};
        vs.keys = w.keys_a; vs.idx = pl.bucket ? nullptr : w.idx_a; vs.part = w.part; vs.n_runs = w.ctl;
        vs.leaf = (uint32_t)leaf_mm; vs.div_inv = pl.dv.inv; vs.div_c = pl.dv.c; vs.bits = pl.bits; vs.idx_bits = pl.idx_bits;
        vs.track_bits = pl.track_bits ? 1u : 0u;
        stage_regions(pl, vs);
        e = launch_payload_voxel_partials(d_payload, n_points, d_n_points, vs, st);
        if (e != hipSuccess) return finish_call(*ws, e);
    } else {
        // the 1024-lane readers (payloads that are only 2- or 4-byte aligned) do not record which key bits vary: every bit
        // counts, no pass is skipped; nor do they fill regions
        pl.track_bits = false;
        pl.regions = false;
        if (wide_ok && n_points >= 2 && ((uintptr_t)d_payload & 3u) == 0u)
            hipLaunchKernelGGL(pcs_voxel_partials_kernel<true>, agg_grid, dim3(kAggThreads), 0, st, d_payload, n_points, d_n_points, pl.dv,
                               pl.bits, pl.idx_bits, w.keys_a, w.idx_a, w.part, w.ctl);
        else
            hipLaunchKernelGGL(pcs_voxel_partials_kernel<false>, agg_grid, dim3(kAggThreads), 0, st, d_payload, n_points, d_n_points, pl.dv,
                               pl.bits, pl.idx_bits, w.keys_a, w.idx_a, w.part, w.ctl);
    }
    return finish_call(*ws, sort_and_reduce(pl, n_points, d_out, d_out_points, st), pl.bucket ? leaf_mm : 0);
}

// Raster source (pcs_kernels.hip: launch_fused_voxel_partials fills the stage between these two calls).
hipError_t voxel_begin(uint32_t capacity_points, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws, VoxelStage* stage,
                       hipStream_t st)
{
    Plan pl;
    hipError_t e = plan_for(capacity_points, leaf_mm, d_ws, ws_bytes, *ws, pl, st);
    if (e != hipSuccess) return e;
    ws->clean = false;                    // until voxel_finish has enqueued the kernel that clears the other block
    stage->keys = pl.w.keys_a; stage->idx = pl.bucket ? nullptr : pl.w.idx_a; stage->part = pl.w.part; stage->n_runs = pl.w.ctl;
    stage->leaf = (uint32_t)leaf_mm; stage->div_inv = pl.dv.inv; stage->div_c = pl.dv.c;
    stage->bits = pl.bits; stage->idx_bits = pl.idx_bits;
    stage->track_bits = pl.track_bits ? 1u : 0u;
    stage_regions(pl, *stage);
    return hipSuccess;
}

hipError_t voxel_finish(uint32_t capacity_points, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws, int16_t* d_out,
                        int32_t* d_out_points, hipStream_t st)
{
    if (ws->base != d_ws) return hipErrorInvalidValue;          // not the workspace voxel_begin prepared
    ws->clean = true;                                           // (voxel_begin's plan, again: same parity, no memset)
    Plan pl;
    hipError_t e = plan_for(capacity_points, leaf_mm, d_ws, ws_bytes, *ws, pl, st);
    if (e != hipSuccess) return finish_call(*ws, e);
    return finish_call(*ws, sort_and_reduce(pl, capacity_points, d_out, d_out_points, st), pl.bucket ? leaf_mm : 0);
}

// ---- partials as an exchange format (multi-GPU config 5) -----------------------------------------------------------------
// A stage whose partials land in CALLER arrays as (raw key, sums): nothing of the sort's layout (index bits) leaks into them,
// so partials of several pre-aggregations — other GPUs' — can be concatenated and handed to launch_voxel_from_partials.
hipError_t voxel_partials_stage(int leaf_mm, unsigned long long* d_keys, void* d_partials, unsigned int* d_count, VoxelStage* stage,
                                hipStream_t st)
{
    VoxelDiv dv;
    unsigned int bits;
    hipError_t e = div_for(leaf_mm, dv, bits);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_count, 0, sizeof(unsigned int), st);      // one word: without track_bits the readers touch nothing else
    if (e != hipSuccess) return e;
    stage->keys = d_keys; stage->idx = nullptr; stage->part = d_partials; stage->n_runs = d_count;
    stage->leaf = (uint32_t)leaf_mm; stage->div_inv = dv.inv; stage->div_c = dv.c;
    stage->bits = bits; stage->idx_bits = 0u; stage->track_bits = 0u;
    stage->regions = 0u;
    return hipSuccess;
}

// Sort + segmented mean over n_partials (or *d_n_partials, at most n_partials) caller-held partials with raw keys.
hipError_t launch_voxel_from_partials(const unsigned long long* d_keys, const void* d_partials, uint32_t n_partials,
                                      const int32_t* d_n_partials, int leaf_mm, void* d_ws, size_t ws_bytes, VoxelWsState* ws,
                                      int16_t* d_out, int32_t* d_out_points, hipStream_t st)
{
    if (n_partials == 0) {
        if (d_out_points) return hipMemsetAsync(d_out_points, 0, sizeof(int32_t), st);
        return hipSuccess;
    }
    // The workspace is carved for as many partials as it HOLDS, not for this call's count: a root's count moves a little from frame-set
    // to frame-set, and regions sized by the previous call's (2 m' + 64 Ki slots) must fit the capacity this call carves (the owner
    // sizes the workspace with headroom: pcs_capi_voxel.cpp). Nothing below reads more than n_partials elements.
    uint32_t n_carve = n_partials;
    {
        const int level = voxel_workspace_level(n_partials, leaf_mm, *ws, true);
        uint32_t lo = n_partials, hi = (uint32_t)std::min<uint64_t>(4ull * n_partials + (1u << 20), level ? (1u << 26) - 1u : 0xFFFFFFF0u);
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1u) / 2u;
            if (voxel_workspace_bytes(mid, level) <= ws_bytes) lo = mid; else hi = mid - 1u;
        }
        n_carve = lo;
    }
    Plan pl;
    hipError_t e = plan_for(n_carve, leaf_mm, d_ws, ws_bytes, *ws, pl, st, true);
    if (e != hipSuccess) return e;
    pl.track_bits = false;                 // nobody recorded which key bits vary across the sources: every bit counts
    if (pl.bucket && pl.regions) {
        // WARM: the previous bucket call on this workspace left splitters for this leaf, region sizes and zeroed cursors — one launch
        // places the caller's partials into the regions (what the raster reader does for its own), the tail is G1 alone: 2 launches
        // instead of 4. The caller's arrays are only read; partials that find their region full land in the workspace's own list.
        const Workspace& w = pl.w;
        const unsigned int max_chunks = (n_partials + kBktChunk - 1u) / kBktChunk;
        const unsigned int grid = std::max(1u, std::min(max_chunks, kBktGrid));
        hipLaunchKernelGGL(pcs_vox_bkt_place_kernel, dim3(grid), dim3(kBktThreads), 0, st, RawKeys{d_keys, d_n_partials, n_partials},
                           static_cast<const VoxelPartial*>(d_partials), w.spl, w.reg + 64 * pl.bpar, w.cursor + kBkt * pl.bpar, w.keys_r, w.part_r,
                           (unsigned int)std::min<size_t>(w.region_slots, 0xFFFFFFFFu), w.keys_a, w.part, w.bucket_of, w.ctl);
        e = hipGetLastError();
        if (e != hipSuccess) return finish_call(*ws, e);
        return finish_call(*ws, sort_and_reduce(pl, n_partials, d_out, d_out_points, st), leaf_mm);
    }
    pl.w.part = const_cast<VoxelPartial*>(static_cast<const VoxelPartial*>(d_partials));      // read in place
    pl.regions = false;                    // cold: the chain partitions the caller's list itself
    pl.raw = RawKeys{d_keys, d_n_partials, n_partials};
    return finish_call(*ws, sort_and_reduce(pl, n_partials, d_out, d_out_points, st), pl.bucket ? leaf_mm : 0);
}

}  // namespace pcs
