// pcs_node.cpp — libpcs_node.so: several GPUs, one process, one grouped RCCL exchange to the root per frame-set.
// See include/pcs_node.h for what it replaces in the reference. Built on the public C ABI of libpcs_hip.so
// (it uses nothing from it that an outside caller could not).
//
// Vocabulary: a PEER is one entry of device_ids (the node's "rank": it owns streams_per_device cameras, a context of
// libpcs_hip with its kernel stream, two payload slots). A GPU is a distinct physical device: it owns the RCCL
// communicator rank and the communication stream. Normally peers and GPUs are the same list; a device id that repeats
// makes VIRTUAL peers that share a GPU (their transfers to the root become RCCL self send/recv pairs on that GPU's
// communicator) — how the one-GPU development boxes run every N > 1 code path on real RCCL.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/pcs_node.h"

namespace {

struct Peer {
    int dev = 0, gpu = 0;                 // HIP ordinal; index into pcs_node::gpus
    pcs_ctx* ctx = nullptr;
    size_t payload_shorts = 0;            // worst-case payload of this peer's cameras
    size_t vcap = 0;                      // worst-case voxel partials of this peer's cameras
    void* d_payload[2] = {nullptr, nullptr};      // peers > 0: the packed payload of slot 0 / 1 (the root packs into the stitched buffer)
    void* d_counts = nullptr;             // streams_per_device + 1 int32 (pcs_process_frames_device's counts)
    void* d_vkeys[2] = {nullptr, nullptr};        // voxel route: keys / partials / append counter per slot
    void* d_vparts[2] = {nullptr, nullptr};       //   (the root's arrays take everybody's partials)
    void* d_vcount[2] = {nullptr, nullptr};
    hipEvent_t packed[2] = {nullptr, nullptr};    // kernel stream: this slot's kernels and its counts read-back are done
    std::vector<void*> d_depth, d_color;  // host forms: raster staging
};

struct Gpu {
    int dev = 0;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;    // the exchange runs here, not on a kernel stream
    hipEvent_t drained[2] = {nullptr, nullptr};   // comm stream: the exchange that read / filled this slot has completed
};

enum TicketKind { kStitch = 0, kVoxel = 1 };

struct Ticket {
    bool busy = false, exchanged = false;
    bool one_call = false;                // voxel ticket of a one-peer node: the whole pipeline was enqueued at submit (nothing to exchange)
    bool timing = false;                  // pcs_node_set_timing as of the SUBMIT: the events of this slot were recorded (or not) then
    int id = -1, kind = kStitch, rc = PCS_OK;
    std::string err;
    // stitch
    int16_t* d_stitched = nullptr;
    std::vector<int32_t> cnt;             // [peer * (S + 1) + k], k == S: the peer's total
    size_t total = 0;                     // points (stitch) / partials (voxel) on the root
    // voxel
    int leaf = 0;
    int16_t* d_voxels = nullptr;
    size_t voxels_shorts = 0;
    pcs_ctx* vox_ctx = nullptr;           // one-call voxel ticket: the context it was enqueued on
    bool sink = false;                    // voxel ticket of several peers on ONE GPU: every peer pre-aggregated into the slot's sink (no exchange)
    pcs_voxel_sink sk{};
    std::vector<const uint16_t*> d_depth; // one-call / sink voxel ticket: the rasters, should the call have to be run again (flagged bucket tail)
    std::vector<const uint8_t*> d_color;
    int64_t exchanged_bytes = 0;          // moved by the grouped RCCL exchange
    int64_t direct_bytes = 0;             // stored into the root's buffer by the peers' own kernels (PCS_NODE_DIRECT_STORE)
    float submit_host_ms = 0.0f, exchange_host_ms = 0.0f;     // host time spent enqueueing (submit without / the exchange itself)
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct pcs_node {
    int n_peers = 0, per_dev = 0, n_streams = 0;
    uint32_t flags = 0, node_flags = 0;
    int downsample = 1;
    std::vector<Peer> peers;
    std::vector<Gpu> gpus;
    bool have_comm = false;
    bool broken = false;                  // an RCCL call failed: the communicators were aborted
    bool inject_fail = false;             // pcs_node_inject_exchange_failure: the next exchange fails as if RCCL had
    bool pred = false;                    // CUTOFF / DROP_INVALID: the exchange is sized by data-dependent counts
    bool timing = false;
    bool direct = false;                  // PCS_NODE_DIRECT_STORE and peer access granted: dense tickets need no exchange
    std::vector<int32_t> static_cnt;      // [peer * (S + 1) + k]: the counts of a frame-set without a predicate
    Ticket inflight[2];
    int next_ticket = 0;
    int32_t* h_counts[2] = {nullptr, nullptr};    // page-locked: [slot][peer * (S + 1) + k]
    int32_t* h_vcount[2] = {nullptr, nullptr};    // page-locked: [slot][peer] partial counts; [n_peers] = the root's voxel count
    std::vector<pcs_stream_config> cfg;
    void* d_stitched = nullptr; size_t stitched_cap_shorts = 0;      // host forms
    void* d_vox_out = nullptr;
    void* d_vox_n[2] = {nullptr, nullptr};        // root: voxel count per slot
    bool voxel_ready = false, voxel_counts_ready = false;
    int one_call = 2;                     // a one-peer node enqueues rasters -> voxels at submit: 0 no (partials pipeline), 1 on the peer's context,
                                          // 2 on two contexts used in turn (PCS_NODE_ONE_CALL, latched at create; pcs_node_set_one_call)
    pcs_ctx* alt_ctx = nullptr;           // one-call tickets of slot 1: a second context of the peer (own stream, workspace, splitters, regions),
    hipStream_t alt_stream = nullptr;     // so that the bucket tail of frame-set k runs beside the pre-aggregation of k+1
    // Several peers on ONE GPU (a device id that repeats throughout): nothing needs to travel — every peer's pre-aggregation writes into
    // the workspace of a SINK context of that GPU (pcs_voxel_sink_*: its buckets' regions on a warm call), whose tail follows on the sink's
    // own stream. Two sinks used in turn (slot 0 / 1), each with its stream, workspace, splitters and regions, as the one-peer node's two
    // contexts: the tail of frame-set k runs beside the pre-aggregations of k+1. PCS_NODE_VOXEL_SINK=0 at create / pcs_node_set_voxel_sink
    // keep the partials exchange (RCCL self send/recv), which is what such a node exists to exercise.
    int vox_sink = 1;
    int sink_streams = 0;                 // kernel streams the peers' pre-aggregations are dealt onto (peer r: stream r % sink_streams); 0: not yet
    bool sink_shared = false;             // peers >= sink_streams currently run on an earlier peer's stream
    pcs_ctx* sink_ctx[2] = {nullptr, nullptr};
    hipStream_t sink_stream[2] = {nullptr, nullptr};
    hipEvent_t sink_open[2] = {nullptr, nullptr};   // sink stream: the sink of this slot is open (its control blocks are clear)
    size_t vcap_total = 0;
    // root: a second context of libpcs_hip (own stream, own sort workspace) that runs the sort + segmented mean of frame-set k
    // while the root's kernel stream pre-aggregates frame-set k+1: the tail is a dozen latency-bound launches that leave the GPU
    // almost empty, the pre-aggregation is VALU-bound — side by side they cost what the longer one costs
    pcs_ctx* reduce_ctx = nullptr;
    hipStream_t reduce_stream = nullptr;
    // root: kernel stream events (timing enabled): start of the submit, own kernels done, reduce start, reduce done
    hipEvent_t ev_k0[2] = {nullptr, nullptr}, ev_k1[2] = {nullptr, nullptr}, ev_r0[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    hipEvent_t ev_x0[2] = {nullptr, nullptr};     // root: comm stream, the group is about to be enqueued
    pcs_node_stats last{};
    int voxel_reruns = 0;                 // voxel frame-sets whose bucket tail ended flagged (-1) and were run again on the LSD tail
    int rccl_version = 0;                 // ncclGetVersion of the library that answered (0: no communicator was asked for)
    std::string rccl_library;             // its path (dladdr)
    std::string err;
};

namespace {
thread_local std::string g_err;

int nfail(pcs_node* n, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n) n->err = buf; else g_err = buf;
    return status;
}
#define HIPCHK(n, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return nfail((n), PCS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define PCSCHK(n, c, expr) do { int rc_ = (expr); if (rc_ != PCS_OK) \
    return nfail((n), rc_, "%s: %s", #expr, pcs_last_error(c)); } while (0)

hipStream_t kstream(const Peer& p) { return static_cast<hipStream_t>(pcs_get_stream(p.ctx)); }

// One transfer of a grouped exchange: `bytes` from src on peer `from` to dst on the root.
struct Xfer { int from; const void* src; void* dst; size_t bytes; };

// The grouped exchange. Whatever happens between ncclGroupStart and ncclGroupEnd, the group is CLOSED before this
// returns; on any RCCL failure the communicators are aborted (a half-issued group can leave sends without their receives
// on the communication streams) and the node is marked unusable.
int run_exchange(pcs_node* n, const std::vector<Xfer>& xs)
{
    if (!n->have_comm || xs.empty()) return PCS_OK;
    Gpu& root = n->gpus[n->peers[0].gpu];
    ncclResult_t first = n->inject_fail ? ncclInternalError : ncclGroupStart();
    n->inject_fail = false;
    if (first == ncclSuccess) {
        for (const Xfer& x : xs) {
            if (!x.bytes) continue;
            Gpu& g = n->gpus[n->peers[x.from].gpu];
            // rank of a GPU in the communicator = its index in n->gpus; a virtual peer on the root's GPU sends to itself
            ncclResult_t r = ncclSend(x.src, x.bytes, ncclInt8, n->peers[0].gpu, g.comm, g.comm_stream);
            if (r == ncclSuccess) r = ncclRecv(x.dst, x.bytes, ncclInt8, n->peers[x.from].gpu, root.comm, root.comm_stream);
            if (r != ncclSuccess) { first = r; break; }
        }
        const ncclResult_t end = ncclGroupEnd();          // always: never leave a group open
        if (first == ncclSuccess) first = end;
    }
    if (first == ncclSuccess) return PCS_OK;
    for (Gpu& g : n->gpus) if (g.comm) { (void)ncclCommAbort(g.comm); g.comm = nullptr; }
    n->broken = true; n->have_comm = false;
    return nfail(n, PCS_ERR_HIP, "RCCL exchange failed: %s (communicators aborted; destroy the node)", ncclGetErrorString(first));
}

// Host rasters -> the owning GPUs' staging buffers (allocated on first use).
int upload_rasters(pcs_node* n, const uint16_t* const* depth, const uint8_t* const* color, std::vector<const uint16_t*>& dd,
                   std::vector<const uint8_t*>& dc)
{
    const int S = n->per_dev;
    dd.assign(n->n_streams, nullptr); dc.assign(n->n_streams, nullptr);
    for (int r = 0; r < n->n_peers; r++) {
        Peer& p = n->peers[r];
        HIPCHK(n, hipSetDevice(p.dev));
        for (int k = 0; k < S; k++) {
            const int g = r * S + k;
            const pcs_stream_config& sc = n->cfg[g];
            const size_t db = (size_t)sc.depth.width * sc.depth.height * sizeof(uint16_t);
            const size_t cb = (size_t)sc.color_stride * sc.color.height;
            if (!depth[g] || !color[g]) return nfail(n, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", g);
            if (!p.d_depth[k]) PCSCHK(n, p.ctx, pcs_device_malloc(p.ctx, &p.d_depth[k], db + 16));
            if (!p.d_color[k]) PCSCHK(n, p.ctx, pcs_device_malloc(p.ctx, &p.d_color[k], cb + 16));
            PCSCHK(n, p.ctx, pcs_memcpy_h2d(p.ctx, p.d_depth[k], depth[g], db));
            PCSCHK(n, p.ctx, pcs_memcpy_h2d(p.ctx, p.d_color[k], color[g], cb));
            dd[g] = static_cast<const uint16_t*>(p.d_depth[k]);
            dc[g] = static_cast<const uint8_t*>(p.d_color[k]);
        }
    }
    return PCS_OK;
}

// ---- a stream that really runs beside another one: pcs_pick_concurrent_stream / pcs_use_stream_beside (libpcs_hip) ---------------------
// *out = a new stream of the current device on which a launch completes while a spin kernel still occupies `busy`, or nullptr if none of
// six candidates did (the caller then keeps what it has).
int pick_concurrent_stream(pcs_node* n, hipStream_t busy, hipStream_t* out)
{
    void* found = nullptr;
    const int rc = pcs_pick_concurrent_stream(busy, &found);
    if (rc != PCS_OK) return nfail(n, rc, "could not probe for a concurrent stream");
    *out = static_cast<hipStream_t>(found);
    return PCS_OK;
}

// What every voxel ticket needs, on first use: the page-locked count words and the root's device count per slot.
int ensure_voxel_counts(pcs_node* n)
{
    if (n->voxel_counts_ready) return PCS_OK;
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    for (int sl = 0; sl < 2; sl++) {
        if (!n->h_vcount[sl]) HIPCHK(n, hipHostMalloc((void**)&n->h_vcount[sl], sizeof(int32_t) * (size_t)(n->n_peers + 1), hipHostMallocPortable));
        if (!n->d_vox_n[sl]) PCSCHK(n, root.ctx, pcs_device_malloc(root.ctx, &n->d_vox_n[sl], 64));
    }
    n->voxel_counts_ready = true;
    return PCS_OK;
}

// One-call tickets of slot 1 (one-peer node, mode 2): a second context of the peer on a stream that is seen to run beside its kernel stream.
int ensure_alt_context(pcs_node* n)
{
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    pcs_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.device = root.dev; cfg.n_streams = n->per_dev; cfg.streams = n->cfg.data(); cfg.flags = n->flags; cfg.downsample = n->downsample;
    const int rc = pcs_create(&n->alt_ctx, &cfg);
    if (rc != PCS_OK) return nfail(n, rc, "second context of the peer: %s", pcs_last_error(nullptr));
    const int rc2 = pick_concurrent_stream(n, kstream(root), &n->alt_stream);
    if (rc2 != PCS_OK) return rc2;
    if (n->alt_stream) PCSCHK(n, n->alt_ctx, pcs_set_stream(n->alt_ctx, n->alt_stream));
    return PCS_OK;
}

// What the partials pipeline needs on top, on ITS first use (a one-peer node on the one-call route, or route PAYLOADS, never
// pays for it): two slots of key / partial arrays per peer (the root's take everybody's partials), the root's second context and
// a stream that is seen to run beside its kernel stream.
int ensure_voxel_buffers(pcs_node* n)
{
    int rc0 = ensure_voxel_counts(n);
    if (rc0 != PCS_OK) return rc0;
    if (n->voxel_ready) return PCS_OK;
    for (int r = 0; r < n->n_peers; r++) {
        Peer& p = n->peers[r];
        HIPCHK(n, hipSetDevice(p.dev));
        const size_t cap = r == 0 ? n->vcap_total : p.vcap;
        for (int sl = 0; sl < 2; sl++) {
            if ((!p.d_vkeys[sl] && pcs_device_malloc(p.ctx, &p.d_vkeys[sl], cap * sizeof(uint64_t) + 64) != PCS_OK) ||
                (!p.d_vparts[sl] && pcs_device_malloc(p.ctx, &p.d_vparts[sl], cap * sizeof(pcs_voxel_partial) + 64) != PCS_OK) ||
                (!p.d_vcount[sl] && pcs_device_malloc(p.ctx, &p.d_vcount[sl], 64) != PCS_OK))
                return nfail(n, PCS_ERR_NOMEM, "device %d: %s", p.dev, pcs_last_error(p.ctx));     // (pcs_node_destroy frees what exists)
        }
    }
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    if (!n->reduce_ctx) {
        pcs_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.device = root.dev; cfg.n_streams = n->per_dev; cfg.streams = n->cfg.data(); cfg.flags = n->flags; cfg.downsample = n->downsample;
        const int rc = pcs_create(&n->reduce_ctx, &cfg);
        if (rc != PCS_OK) return nfail(n, rc, "second root context: %s", pcs_last_error(nullptr));
        // (its own default-priority stream. A high-priority stream was tried: 0.330 instead of 0.198 ms per 16 x 1080p frame-set —
        // every short launch then pre-empts the pre-aggregation's workgroups)
        // Which stream: the HIP runtime maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), round
        // robin as they are created, and two streams on one queue do not overlap at all — in the C++ CLI the context's own stream
        // happened to share the kernel stream's queue: 0.248 ms per frame-set instead of 0.195. Streams are therefore tried until
        // one is SEEN to run beside the root's kernel stream (a no-op launched behind a 300 us spin on the kernel stream must
        // finish first); stream priorities proved erratic (0.20 - 0.33 ms depending on the queue count) and are not used.
        // one GPU holds every camera: the tail of frame-set k runs beside the WHOLE pre-aggregation of k+1, and the lighter
        // neighbour wins there (16 x 1080p, 50 mm, two in flight: 0.196 ms per frame-set with the LSD tail, 0.213 with the bucket
        // tail, whose 72 KiB workgroups take table slots from the pre-aggregation). With the cameras spread over GPUs the root's own
        // pre-aggregation is short and the tail is the critical path: the default (bucket: 0.088 vs 0.113 ms root phase at 8 peers).
        if (n->gpus.size() == 1 && n->n_peers == 1) (void)pcs_set_voxel_tail(n->reduce_ctx, PCS_VOXEL_TAIL_LSD);
        int rc2 = pick_concurrent_stream(n, kstream(root), &n->reduce_stream);
        if (rc2 != PCS_OK) return rc2;
        if (n->reduce_stream) PCSCHK(n, n->reduce_ctx, pcs_set_stream(n->reduce_ctx, n->reduce_stream));
    }
    n->voxel_ready = true;
    return PCS_OK;
}

// The two sinks of a node whose peers all share the root's GPU, on first use.
int ensure_sinks(pcs_node* n)
{
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    for (int sl = 0; sl < 2; sl++) {
        if (n->sink_ctx[sl]) continue;
        pcs_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        // (a sink pre-aggregates nothing itself: it owns the workspace and the tail; any valid configuration will do)
        cfg.device = root.dev; cfg.n_streams = n->per_dev; cfg.streams = n->cfg.data(); cfg.flags = n->flags; cfg.downsample = n->downsample;
        const int rc = pcs_create(&n->sink_ctx[sl], &cfg);
        if (rc != PCS_OK) return nfail(n, rc, "voxel sink context: %s", pcs_last_error(nullptr));
        const int rc2 = pick_concurrent_stream(n, kstream(root), &n->sink_stream[sl]);
        if (rc2 != PCS_OK) return rc2;
        if (n->sink_stream[sl]) PCSCHK(n, n->sink_ctx[sl], pcs_set_stream(n->sink_ctx[sl], n->sink_stream[sl]));
        HIPCHK(n, hipEventCreateWithFlags(&n->sink_open[sl], hipEventDisableTiming));
    }
    // Eight peers of one GPU on eight streams cost the one host thread an event record and two stream waits per peer and frame-set, on top
    // of the launch (0.10 ms per 8 peers: as long as the GPU needs for their kernels), and the runtime folds the streams onto four
    // hardware queues anyway. The peers are therefore dealt onto FEWER kernel streams — peer r runs on peer (r % K)'s — and only the last
    // peer of each stream records an event for the sink (PCS_NODE_SINK_STREAMS=<K>, read when the first sink ticket is submitted).
    // 16 x 1080p at 50 mm, 8 peers, ms per frame-set / host ms per submit: K = 1 0.264 / 0.047 (eight launches of 510 workgroups, one
    // after the other: each ends on its slowest workgroup), 2 0.227 / 0.054, 3 0.231, 4 0.248 / 0.068, 8 0.243 / 0.114.
    if (!n->sink_streams) {
        const char* e = getenv("PCS_NODE_SINK_STREAMS");
        const int want = e ? atoi(e) : 2;
        n->sink_streams = std::max(1, std::min(want, n->n_peers));
    }
    if (!n->sink_shared) {
        for (int r = n->sink_streams; r < n->n_peers; r++)
            PCSCHK(n, n->peers[r].ctx, pcs_set_stream(n->peers[r].ctx, kstream(n->peers[r % n->sink_streams])));
        n->sink_shared = true;
    }
    return PCS_OK;
}

// The peers back on their own streams (the partials exchange, as a node of several GPUs runs it).
int unshare_streams(pcs_node* n)
{
    if (!n->sink_shared) return PCS_OK;
    HIPCHK(n, hipSetDevice(n->peers[0].dev));
    for (int r = n->sink_streams; r < n->n_peers; r++) PCSCHK(n, n->peers[r].ctx, pcs_set_stream(n->peers[r].ctx, nullptr));
    n->sink_shared = false;
    return PCS_OK;
}

// A sink ticket, start to end: the sink opened on its stream, every peer's pre-aggregation behind that on the peer's own stream, the
// tail behind all of them on the sink's stream, the voxel count on its way to page-locked memory. (The sink's previous tail — two
// tickets ago — was waited for by the host before this slot could be submitted again.)
int enqueue_sink_ticket(pcs_node* n, Ticket& tk, int slot)
{
    const int S = n->per_dev, P = n->n_peers;
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    pcs_ctx* sc = n->sink_ctx[slot];
    hipStream_t ss = static_cast<hipStream_t>(pcs_get_stream(sc));
    const int K = n->sink_streams;
    PCSCHK(n, sc, pcs_voxel_sink_begin(sc, n->vcap_total, tk.leaf, &tk.sk));
    const bool order_begin = tk.sk.work_enqueued != 0;       // (a clear of the control blocks: the sink's first ticket, or after a failed one)
    if (order_begin) HIPCHK(n, hipEventRecord(n->sink_open[slot], ss));
    for (int r = 0; r < P; r++) {
        Peer& p = n->peers[r];
        hipStream_t ks = kstream(p);
        if (order_begin && r < K) HIPCHK(n, hipStreamWaitEvent(ks, n->sink_open[slot], 0));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k0[slot], ks));
        PCSCHK(n, p.ctx, pcs_process_frames_voxel_into_sink_device(p.ctx, tk.d_depth.data() + (size_t)r * S, tk.d_color.data() + (size_t)r * S, &tk.sk));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k1[slot], ks));
        if (r + K >= P) {                                    // the last peer on its stream
            HIPCHK(n, hipEventRecord(p.packed[slot], ks));
            HIPCHK(n, hipStreamWaitEvent(ss, p.packed[slot], 0));
        }
    }
    if (tk.timing) HIPCHK(n, hipEventRecord(n->ev_r0[slot], ss));
    PCSCHK(n, sc, pcs_voxel_sink_finish(sc, &tk.sk, tk.d_voxels, tk.voxels_shorts, static_cast<int32_t*>(n->d_vox_n[slot])));
    HIPCHK(n, hipMemcpyAsync(n->h_vcount[slot] + P, n->d_vox_n[slot], sizeof(int32_t), hipMemcpyDeviceToHost, ss));
    HIPCHK(n, hipEventRecord(n->ev_done[slot], ss));
    return PCS_OK;
}

// Everything the exchange of ticket `tk` needs from the host, then the exchange itself, then (voxel) the root's reduce.
// Called from submit (no predicate: immediately), from the NEXT submit (after that frame-set's kernels are enqueued) or from
// wait, whichever comes first. Always leaves drained[slot] recorded on every GPU's communication stream and tk.exchanged
// set, so that neither a later submit's kernels nor a wait can block on this slot; a failure is parked in tk.rc.
void issue_exchange(pcs_node* n, Ticket& tk)
{
    const int slot = tk.id & 1, S = n->per_dev, P = n->n_peers;
    Peer& root = n->peers[0];
    Gpu& rootg = n->gpus[root.gpu];
    const double t_host0 = now_ms();
    auto body = [&]() -> int {
        // an earlier exchange failed and the communicators are gone: this frame-set was never gathered — say so instead of
        // returning counts for a buffer that holds the root's slice only (include/pcs_node.h: every later call fails)
        if (n->broken) return nfail(n, PCS_ERR_HIP, "the node's communicators were aborted after an RCCL failure: this frame-set was not gathered");
        const bool counts_on_device = tk.kind == kVoxel || n->pred;
        if (counts_on_device) {
            // the copies were enqueued behind the kernels in submit: by now (a whole submit later in a pipelined loop) they
            // have usually landed and these waits return at once
            for (int r = 0; r < P; r++) {
                HIPCHK(n, hipSetDevice(n->peers[r].dev));
                HIPCHK(n, hipEventSynchronize(n->peers[r].packed[slot]));
            }
        }
        std::vector<Xfer> xs;
        size_t off = 0;
        if (tk.kind == kStitch) {
            if (n->pred) for (int i = 0; i < P * (S + 1); i++) tk.cnt[i] = n->h_counts[slot][i];
            for (int r = 0; r < P; r++) {
                const int32_t c = tk.cnt[(size_t)r * (S + 1) + S];
                if (c < 0 || (size_t)c * PCS_POINT_SHORTS > n->peers[r].payload_shorts)
                    return nfail(n, PCS_ERR_HIP, "device %d reported %d points (capacity %zu)", n->peers[r].dev, c, n->peers[r].payload_shorts / PCS_POINT_SHORTS);
                if (r > 0)
                    xs.push_back(Xfer{r, n->peers[r].d_payload[slot], reinterpret_cast<int8_t*>(tk.d_stitched) + off * PCS_POINT_BYTES,
                                      (size_t)c * PCS_POINT_BYTES});
                off += (size_t)c;
            }
        } else {
            for (int r = 0; r < P; r++) {
                const int32_t m = n->h_vcount[slot][r];
                if (m < 0 || (size_t)m > n->peers[r].vcap)
                    return nfail(n, PCS_ERR_HIP, "device %d reported %d partials (capacity %zu)", n->peers[r].dev, m, n->peers[r].vcap);
                if (r > 0) {
                    xs.push_back(Xfer{r, n->peers[r].d_vkeys[slot], static_cast<uint64_t*>(root.d_vkeys[slot]) + off, (size_t)m * sizeof(uint64_t)});
                    xs.push_back(Xfer{r, n->peers[r].d_vparts[slot], static_cast<pcs_voxel_partial*>(root.d_vparts[slot]) + off,
                                      (size_t)m * sizeof(pcs_voxel_partial)});
                    tk.exchanged_bytes += (int64_t)m * PCS_VOXEL_PARTIAL_WIRE_BYTES;
                }
                off += (size_t)m;
            }
        }
        tk.total = off;
        if (!n->have_comm) {
            // one peer, or PCS_NODE_NO_EXCHANGE: nothing travels and nothing but the kernels themselves has to finish — no hop
            // through the communication streams (two barrier packets per frame-set that the single-GPU loop does not need)
            tk.exchanged_bytes = 0;
            if (tk.kind == kVoxel) tk.total = (size_t)n->h_vcount[slot][0];      // only the root's own partials are on the root
            return PCS_OK;
        }
        if (tk.kind == kStitch) for (const Xfer& x : xs) tk.exchanged_bytes += (int64_t)x.bytes;
        // every GPU's communication stream runs behind the kernels of the peers it hosts
        for (int r = 0; r < P; r++) {
            Gpu& g = n->gpus[n->peers[r].gpu];
            HIPCHK(n, hipSetDevice(g.dev));
            HIPCHK(n, hipStreamWaitEvent(g.comm_stream, n->peers[r].packed[slot], 0));
        }
        if (tk.timing) { HIPCHK(n, hipSetDevice(rootg.dev)); HIPCHK(n, hipEventRecord(n->ev_x0[slot], rootg.comm_stream)); }
        if (tk.kind == kStitch && n->direct && !n->pred) {
            // the peers' kernels stored into the root themselves: nothing is exchanged, and the stats say so
            tk.direct_bytes = tk.exchanged_bytes; tk.exchanged_bytes = 0;
            xs.clear();
        }
        if (!(n->node_flags & PCS_NODE_NO_EXCHANGE)) {
            const int rc = run_exchange(n, xs);
            if (rc != PCS_OK) return rc;
        }
        return PCS_OK;
    };
    int rc = body();
    if (n->have_comm || n->broken)
        for (Gpu& g : n->gpus)
            if (hipSetDevice(g.dev) == hipSuccess) (void)hipEventRecord(g.drained[slot], g.comm_stream);
    tk.exchanged = true;
    if (rc == PCS_OK && tk.kind == kVoxel) {
        // the root reduces everybody's partials behind the exchange; its voxel count travels to page-locked memory
        auto reduce = [&]() -> int {
            HIPCHK(n, hipSetDevice(root.dev));
            hipStream_t ks = static_cast<hipStream_t>(pcs_get_stream(n->reduce_ctx));      // not the root's kernel stream: see reduce_ctx
            HIPCHK(n, hipStreamWaitEvent(ks, root.packed[slot], 0));                        // the root's own partials
            if (n->have_comm) HIPCHK(n, hipStreamWaitEvent(ks, rootg.drained[slot], 0));    // everybody else's
            if (tk.timing) HIPCHK(n, hipEventRecord(n->ev_r0[slot], ks));
            PCSCHK(n, n->reduce_ctx, pcs_voxel_grid_from_partials_device(n->reduce_ctx, static_cast<const uint64_t*>(root.d_vkeys[slot]),
                                                                    static_cast<const pcs_voxel_partial*>(root.d_vparts[slot]), (int)tk.total,
                                                                    nullptr, tk.leaf, tk.d_voxels, tk.voxels_shorts,
                                                                    static_cast<int32_t*>(n->d_vox_n[slot])));
            HIPCHK(n, hipMemcpyAsync(n->h_vcount[slot] + P, n->d_vox_n[slot], sizeof(int32_t), hipMemcpyDeviceToHost, ks));
            HIPCHK(n, hipEventRecord(n->ev_done[slot], ks));
            return PCS_OK;
        };
        rc = reduce();
    }
    if (rc != PCS_OK) { tk.rc = rc; tk.err = n->err; }
    tk.exchange_host_ms = (float)(now_ms() - t_host0);
}

// The other slot's exchange, if a submit left it pending (predicate / voxel: it waits for device counts).
void flush_other(pcs_node* n, int slot)
{
    Ticket& o = n->inflight[slot ^ 1];
    if (o.busy && !o.exchanged) issue_exchange(n, o);
}

int check_submit(pcs_node* n, Ticket*& tk, int& slot)
{
    if (n->broken) return nfail(n, PCS_ERR_HIP, "the node's communicators were aborted after an RCCL failure: destroy it");
    slot = n->next_ticket & 1;
    tk = &n->inflight[slot];
    if (tk->busy) return nfail(n, PCS_ERR_CAPACITY, "two frame-sets are in flight: wait for the older one first");
    return PCS_OK;
}

void fill_stats(pcs_node* n, const Ticket& tk)
{
    pcs_node_stats& st = n->last;
    std::memset(&st, 0, sizeof st);
    st.ticket = tk.id;
    st.exchanged_bytes = tk.exchanged_bytes;
    st.reduced = (int64_t)tk.total;
    st.direct_bytes = tk.direct_bytes;
    st.submit_host_ms = tk.submit_host_ms;
    st.exchange_host_ms = tk.exchange_host_ms;
    if (!tk.timing) return;          // latched at submit: a set_timing while the ticket was in flight must not read events never recorded
    const int slot = tk.id & 1;
    Gpu& rootg = n->gpus[n->peers[0].gpu];
    if (hipSetDevice(rootg.dev) != hipSuccess) return;
    (void)hipEventElapsedTime(&st.kernels_ms, n->ev_k0[slot], n->ev_k1[slot]);
    if (n->have_comm && !tk.direct_bytes) (void)hipEventElapsedTime(&st.exchange_ms, n->ev_x0[slot], rootg.drained[slot]);
    if (tk.kind == kVoxel) (void)hipEventElapsedTime(&st.root_ms, n->ev_r0[slot], n->ev_done[slot]);
    (void)hipGetLastError();
}
}  // namespace

extern "C" {

const char* pcs_node_last_error(const pcs_node* n) { return n ? n->err.c_str() : g_err.c_str(); }
int pcs_node_devices(const pcs_node* n) { return n ? n->n_peers : 0; }
int pcs_node_rccl_ranks(const pcs_node* n) { return (n && n->have_comm) ? (int)n->gpus.size() : 0; }

size_t pcs_node_max_payload_shorts(const pcs_node* n)
{
    if (!n) return 0;
    size_t s = 0;
    for (const Peer& p : n->peers) s += p.payload_shorts;
    return s;
}

void pcs_node_destroy(pcs_node* n)
{
    if (!n) return;
    for (Gpu& g : n->gpus) {
        (void)hipSetDevice(g.dev);
        if (g.comm_stream) (void)hipStreamSynchronize(g.comm_stream);
    }
    if (n->reduce_ctx && !n->peers.empty()) { (void)hipSetDevice(n->peers[0].dev); (void)pcs_synchronize(n->reduce_ctx); }
    for (int sl = 0; sl < 2; sl++)
        if (n->sink_ctx[sl] && !n->peers.empty()) { (void)hipSetDevice(n->peers[0].dev); (void)pcs_synchronize(n->sink_ctx[sl]); }
    for (size_t r = 0; r < n->peers.size(); r++) {
        Peer& p = n->peers[r];
        if (!p.ctx) continue;
        (void)hipSetDevice(p.dev);
        (void)pcs_synchronize(p.ctx);
        for (int sl = 0; sl < 2; sl++) {
            if (p.d_payload[sl]) pcs_device_free(p.ctx, p.d_payload[sl]);
            if (p.d_vkeys[sl]) pcs_device_free(p.ctx, p.d_vkeys[sl]);
            if (p.d_vparts[sl]) pcs_device_free(p.ctx, p.d_vparts[sl]);
            if (p.d_vcount[sl]) pcs_device_free(p.ctx, p.d_vcount[sl]);
            if (p.packed[sl]) (void)hipEventDestroy(p.packed[sl]);
        }
        if (p.d_counts) pcs_device_free(p.ctx, p.d_counts);
        for (void* q : p.d_depth) if (q) pcs_device_free(p.ctx, q);
        for (void* q : p.d_color) if (q) pcs_device_free(p.ctx, q);
        if (r == 0) {
            if (n->d_stitched) pcs_device_free(p.ctx, n->d_stitched);
            if (n->d_vox_out) pcs_device_free(p.ctx, n->d_vox_out);
            for (int sl = 0; sl < 2; sl++) {
                if (n->d_vox_n[sl]) pcs_device_free(p.ctx, n->d_vox_n[sl]);
                for (hipEvent_t e : {n->ev_k0[sl], n->ev_k1[sl], n->ev_r0[sl], n->ev_done[sl], n->ev_x0[sl]}) if (e) (void)hipEventDestroy(e);
            }
        }
    }
    for (Gpu& g : n->gpus) {
        (void)hipSetDevice(g.dev);
        for (int sl = 0; sl < 2; sl++) if (g.drained[sl]) (void)hipEventDestroy(g.drained[sl]);
        if (g.comm_stream) (void)hipStreamDestroy(g.comm_stream);
        if (g.comm) (void)ncclCommDestroy(g.comm);
    }
    for (int sl = 0; sl < 2; sl++) {
        if (n->h_counts[sl]) (void)hipHostFree(n->h_counts[sl]);
        if (n->h_vcount[sl]) (void)hipHostFree(n->h_vcount[sl]);
    }
    if (n->reduce_ctx) { if (!n->peers.empty()) (void)hipSetDevice(n->peers[0].dev); (void)pcs_synchronize(n->reduce_ctx); pcs_destroy(n->reduce_ctx);
                         if (n->reduce_stream) (void)hipStreamDestroy(n->reduce_stream); }
    for (int sl = 0; sl < 2; sl++) {
        if (!n->sink_ctx[sl]) continue;
        if (!n->peers.empty()) (void)hipSetDevice(n->peers[0].dev);
        (void)pcs_synchronize(n->sink_ctx[sl]); pcs_destroy(n->sink_ctx[sl]);
        if (n->sink_stream[sl]) (void)hipStreamDestroy(n->sink_stream[sl]);
        if (n->sink_open[sl]) (void)hipEventDestroy(n->sink_open[sl]);
    }
    if (n->alt_ctx) { if (!n->peers.empty()) (void)hipSetDevice(n->peers[0].dev); (void)pcs_synchronize(n->alt_ctx); pcs_destroy(n->alt_ctx);
                      if (n->alt_stream) (void)hipStreamDestroy(n->alt_stream); }
    (void)unshare_streams(n);              // (a context must not be destroyed while it runs on the stream of one destroyed before it)
    for (Peer& p : n->peers) if (p.ctx) pcs_destroy(p.ctx);
    delete n;
}

int pcs_node_create_ex(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                       const pcs_stream_config* streams, uint32_t flags, int downsample, uint32_t node_flags)
{
    g_err.clear();
    if (!out) return nfail(nullptr, PCS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices < 1 || !device_ids || streams_per_device < 1 || !streams || downsample < 1)
        return nfail(nullptr, PCS_ERR_INVALID_ARG, "bad arguments");
    {   // the stitched payload's byte count travels as an int32 (src/pcs-camera-optimized.cpp:697, 718;
        // src/pcs-multicamera-client.cpp:394): bound the WHOLE node's cloud, in 64 bits, before any device is touched
        uint64_t all = 0;
        for (int g = 0; g < n_devices * streams_per_device; g++) {
            const pcs_stream_config& sc = streams[g];
            if (sc.depth.width <= 0 || sc.depth.height <= 0)
                return nfail(nullptr, PCS_ERR_INVALID_ARG, "stream %d: depth width/height must be positive", g);
            all += ((uint64_t)sc.depth.width * (uint64_t)sc.depth.height + (uint64_t)downsample - 1) / (uint64_t)downsample;
        }
        if (all * PCS_POINT_BYTES > 0x7FFFFFFFull)
            return nfail(nullptr, PCS_ERR_INVALID_ARG, "stitched payload of %llu points exceeds the int32 byte-count header "
                         "(214 748 364 points)", (unsigned long long)all);
    }
    const int avail = pcs_device_count();
    if (avail < 1) return nfail(nullptr, PCS_ERR_NO_DEVICE, "no HIP device (there is no CPU fallback)");
    for (int r = 0; r < n_devices; r++)
        if (device_ids[r] < 0 || device_ids[r] >= avail)
            return nfail(nullptr, PCS_ERR_NO_DEVICE, "device %d requested, %d available", device_ids[r], avail);
    pcs_node* n = new pcs_node;
    n->n_peers = n_devices; n->per_dev = streams_per_device; n->n_streams = n_devices * streams_per_device;
    n->flags = flags; n->downsample = downsample; n->node_flags = node_flags;
    { const char* e = getenv("PCS_NODE_ONE_CALL"); n->one_call = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }      // latched here, not read in the frame loop
    { const char* e = getenv("PCS_NODE_VOXEL_SINK"); n->vox_sink = (e && e[0] == '0') ? 0 : 1; }
    n->cfg.assign(streams, streams + n->n_streams);
    n->pred = (flags & (PCS_FLAG_CUTOFF | PCS_FLAG_DROP_INVALID)) != 0;
    n->peers.resize(n_devices);
    for (int r = 0; r < n_devices; r++) {
        Peer& p = n->peers[r];
        p.dev = device_ids[r];
        p.gpu = -1;
        for (size_t g = 0; g < n->gpus.size(); g++) if (n->gpus[g].dev == p.dev) p.gpu = (int)g;
        if (p.gpu < 0) { p.gpu = (int)n->gpus.size(); Gpu g; g.dev = p.dev; n->gpus.push_back(g); }
        p.d_depth.assign(streams_per_device, nullptr); p.d_color.assign(streams_per_device, nullptr);
    }
    auto bail = [&](int rc, const char* what, const char* detail) {
        const int e = nfail(nullptr, rc, "%s: %s", what, detail);
        pcs_node_destroy(n);
        return e;
    };
    for (int r = 0; r < n_devices; r++) {
        Peer& p = n->peers[r];
        pcs_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.device = p.dev; cfg.n_streams = streams_per_device; cfg.streams = streams + (size_t)r * streams_per_device;
        cfg.flags = flags; cfg.downsample = downsample;
        const int rc = pcs_create(&p.ctx, &cfg);
        if (rc != PCS_OK) return bail(rc, "pcs_create", pcs_last_error(nullptr));
        p.payload_shorts = pcs_max_payload_shorts(p.ctx);
        p.vcap = p.payload_shorts / PCS_POINT_SHORTS;
        n->vcap_total += p.vcap;
        if (pcs_device_malloc(p.ctx, &p.d_counts, sizeof(int32_t) * (streams_per_device + 1)) != PCS_OK ||
            (r > 0 && (pcs_device_malloc(p.ctx, &p.d_payload[0], p.payload_shorts * sizeof(int16_t) + 64) != PCS_OK ||
                       pcs_device_malloc(p.ctx, &p.d_payload[1], p.payload_shorts * sizeof(int16_t) + 64) != PCS_OK)))
            return bail(PCS_ERR_NOMEM, "device memory", pcs_last_error(p.ctx));
        hipError_t he = hipSetDevice(p.dev);
        for (int sl = 0; sl < 2 && he == hipSuccess; sl++) he = hipEventCreateWithFlags(&p.packed[sl], hipEventDisableTiming);
        if (he != hipSuccess) return bail(PCS_ERR_HIP, "event", hipGetErrorString(he));
    }
    for (Gpu& g : n->gpus) {
        hipError_t he = hipSetDevice(g.dev);
        if (he == hipSuccess && n_devices > 1) {
            // the exchange of frame-set k is meant to run beside the kernels of k+1: a communication stream that is SEEN to run
            // beside the kernel stream of the GPU's first peer (hardware-queue collisions: pick_concurrent_stream)
            for (const Peer& p : n->peers)
                if (n->gpus[p.gpu].dev == g.dev) { (void)pick_concurrent_stream(n, kstream(p), &g.comm_stream); break; }
        }
        if (he == hipSuccess && !g.comm_stream) he = hipStreamCreateWithFlags(&g.comm_stream, hipStreamNonBlocking);
        for (int sl = 0; sl < 2 && he == hipSuccess; sl++) he = hipEventCreate(&g.drained[sl]);
        if (he != hipSuccess) return bail(PCS_ERR_HIP, "communication stream", hipGetErrorString(he));
    }
    {
        hipError_t he = hipSetDevice(n->peers[0].dev);
        for (int sl = 0; sl < 2 && he == hipSuccess; sl++) {
            he = hipEventCreate(&n->ev_k0[sl]);
            if (he == hipSuccess) he = hipEventCreate(&n->ev_k1[sl]);
            if (he == hipSuccess) he = hipEventCreate(&n->ev_r0[sl]);
            if (he == hipSuccess) he = hipEventCreate(&n->ev_done[sl]);
            if (he == hipSuccess) he = hipEventCreate(&n->ev_x0[sl]);
            if (he == hipSuccess) he = hipHostMalloc((void**)&n->h_counts[sl], sizeof(int32_t) * (size_t)n_devices * (streams_per_device + 1),
                                                     hipHostMallocPortable);
        }
        if (he != hipSuccess) return bail(PCS_ERR_HIP, "root events / page-locked counts", hipGetErrorString(he));
    }
    n->static_cnt.assign((size_t)n_devices * (streams_per_device + 1), 0);
    for (int r = 0; r < n_devices; r++) {
        int64_t tot = 0;
        for (int k = 0; k < streams_per_device; k++) {
            const int32_t c = (pcs_stream_points(n->peers[r].ctx, k) + downsample - 1) / downsample;
            n->static_cnt[(size_t)r * (streams_per_device + 1) + k] = c;
            tot += c;
        }
        n->static_cnt[(size_t)r * (streams_per_device + 1) + streams_per_device] = (int32_t)tot;
    }
    if ((node_flags & PCS_NODE_DIRECT_STORE) && !(node_flags & PCS_NODE_NO_EXCHANGE) && n_devices > 1) {
        // every peer's kernels will store into the root GPU's memory themselves: that needs peer access from each other GPU
        bool ok = true;
        const int root_dev = n->peers[0].dev;
        for (const Gpu& g : n->gpus) {
            if (g.dev == root_dev) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g.dev, root_dev) != hipSuccess || !can) { ok = false; break; }
            if (hipSetDevice(g.dev) != hipSuccess) { ok = false; break; }
            const hipError_t e = hipDeviceEnablePeerAccess(root_dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ok = false; break; }
        }
        (void)hipGetLastError();
        if (!ok) return bail(PCS_ERR_UNSUPPORTED, "PCS_NODE_DIRECT_STORE", "a GPU of the node cannot address the root GPU's memory (no peer access)");
        n->direct = true;
    }
    if (n_devices > 1 && !(node_flags & PCS_NODE_NO_EXCHANGE)) {
        // Which RCCL answers: this library is compiled against /opt/rocm's rccl.h, but the dynamic loader binds whatever
        // librccl.so.1 the process loaded first (under Python: the one bundled with torch). Record version and path, and refuse
        // a different MAJOR version — its ABI is not the one these calls were compiled for.
        int ver = 0;
        if (ncclGetVersion(&ver) != ncclSuccess) return bail(PCS_ERR_HIP, "ncclGetVersion", "failed");
        n->rccl_version = ver;
        Dl_info di;
        if (dladdr(reinterpret_cast<void*>(&ncclGetVersion), &di) && di.dli_fname) n->rccl_library = di.dli_fname;
        const int major_rt = ver >= 10000 ? ver / 10000 : ver / 1000, major_hdr = NCCL_MAJOR;
        if (major_rt != major_hdr) {
            char msg[256];
            snprintf(msg, sizeof msg, "runtime version %d (%s) has major %d, compiled against %d.%d.%d", ver,
                     n->rccl_library.c_str(), major_rt, NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
            return bail(PCS_ERR_UNSUPPORTED, "RCCL", msg);
        }
        // one communicator rank per GPU, all in this process; virtual peers share their GPU's rank
        std::vector<int> ids;
        std::vector<ncclComm_t> comms(n->gpus.size(), nullptr);
        for (const Gpu& g : n->gpus) ids.push_back(g.dev);
        const ncclResult_t r = ncclCommInitAll(comms.data(), (int)ids.size(), ids.data());
        if (r != ncclSuccess) return bail(PCS_ERR_HIP, "ncclCommInitAll", ncclGetErrorString(r));
        for (size_t g = 0; g < n->gpus.size(); g++) n->gpus[g].comm = comms[g];
        n->have_comm = true;
    }
    *out = n;
    return PCS_OK;
}

int pcs_node_create(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                    const pcs_stream_config* streams, uint32_t flags, int downsample)
{
    return pcs_node_create_ex(out, n_devices, device_ids, streams_per_device, streams, flags, downsample, 0u);
}

int pcs_node_inject_exchange_failure(pcs_node* n)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    n->inject_fail = true;
    return PCS_OK;
}

int pcs_node_rccl_version(const pcs_node* n) { return n ? n->rccl_version : 0; }
int pcs_node_rccl_header_version(void) { return NCCL_VERSION_CODE; }
const char* pcs_node_rccl_library(const pcs_node* n) { return n ? n->rccl_library.c_str() : ""; }

int pcs_node_link_info(pcs_node* n, int peer, pcs_node_link* out)
{
    if (!n || !out || peer < 0 || peer >= n->n_peers) return nfail(n, PCS_ERR_INVALID_ARG, "bad peer index");
    std::memset(out, 0, sizeof *out);
    const int root = n->peers[0].dev, dev = n->peers[peer].dev;
    out->device = dev; out->root_device = root;
    out->same_device = dev == root;
    out->link_type = -1; out->hops = -1; out->performance_rank = -1;
    if (dev == root) { out->can_access_root = 1; out->native_atomics = 1; return PCS_OK; }
    int v = 0;
    if (hipDeviceCanAccessPeer(&v, dev, root) == hipSuccess) out->can_access_root = v;
    if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrPerformanceRank, dev, root) == hipSuccess) out->performance_rank = v;
    if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrNativeAtomicSupported, dev, root) == hipSuccess) out->native_atomics = v;
    uint32_t lt = 0, hc = 0;
    if (hipExtGetLinkTypeAndHopCount(dev, root, &lt, &hc) == hipSuccess) { out->link_type = (int32_t)lt; out->hops = (int32_t)hc; }
    (void)hipGetLastError();
    return PCS_OK;
}

// One peer at a time: `bytes` from the peer's payload slot 0 into the root's scratch, as its OWN group of one send/recv pair,
// bracketed by an event pair on the root's communication stream; `repeats` transfers per peer, the mean is reported. The
// grouped exchange of a frame-set launches all pairs as one RCCL operation, so per-peer figures cannot be read off it.
int pcs_node_probe_links(pcs_node* n, size_t bytes, int repeats, float* ms_per_peer)
{
    if (!n || !ms_per_peer) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (n->broken) return nfail(n, PCS_ERR_HIP, "the node's communicators were aborted after an RCCL failure: destroy it");
    if (n->inflight[0].busy || n->inflight[1].busy) return nfail(n, PCS_ERR_INVALID_ARG, "wait for the frame-sets in flight before probing the links");
    const int P = n->n_peers;
    for (int r = 0; r < P; r++) ms_per_peer[r] = 0.0f;
    if (!n->have_comm || P < 2) return PCS_OK;
    if (repeats < 1) repeats = 1;
    Peer& root = n->peers[0];
    Gpu& rootg = n->gpus[root.gpu];
    HIPCHK(n, hipSetDevice(root.dev));
    size_t cap = 0;
    for (int r = 1; r < P; r++) cap = cap ? std::min(cap, n->peers[r].payload_shorts * sizeof(int16_t)) : n->peers[r].payload_shorts * sizeof(int16_t);
    if (!bytes || bytes > cap) bytes = cap;
    void* scratch = nullptr;
    PCSCHK(n, root.ctx, pcs_device_malloc(root.ctx, &scratch, bytes + 64));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PCS_OK;
    auto body = [&]() -> int {
        HIPCHK(n, hipEventCreate(&e0));
        HIPCHK(n, hipEventCreate(&e1));
        for (int r = 1; r < P; r++) {
            float sum = 0.0f;
            for (int it = -1; it < repeats; it++) {             // it == -1: untimed (connection set-up on first use)
                HIPCHK(n, hipSetDevice(rootg.dev));
                HIPCHK(n, hipEventRecord(e0, rootg.comm_stream));
                std::vector<Xfer> one{Xfer{r, n->peers[r].d_payload[0], scratch, bytes}};
                const int xr = run_exchange(n, one);
                if (xr != PCS_OK) return xr;
                HIPCHK(n, hipSetDevice(rootg.dev));
                HIPCHK(n, hipEventRecord(e1, rootg.comm_stream));
                HIPCHK(n, hipEventSynchronize(e1));
                Gpu& g = n->gpus[n->peers[r].gpu];
                HIPCHK(n, hipSetDevice(g.dev));
                HIPCHK(n, hipStreamSynchronize(g.comm_stream));
                float ms = 0.0f;
                HIPCHK(n, hipSetDevice(rootg.dev));
                HIPCHK(n, hipEventElapsedTime(&ms, e0, e1));
                if (it >= 0) sum += ms;
            }
            ms_per_peer[r] = sum / (float)repeats;
        }
        return PCS_OK;
    };
    rc = body();
    (void)hipSetDevice(root.dev);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    pcs_device_free(root.ctx, scratch);
    return rc;
}

int pcs_node_set_timing(pcs_node* n, int enable)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    n->timing = enable != 0;
    return PCS_OK;
}

int pcs_node_last_stats(const pcs_node* n, pcs_node_stats* out)
{
    if (!n || !out) return PCS_ERR_INVALID_ARG;
    *out = n->last;
    return PCS_OK;
}

// Pipelined device form. Per submit:
//   1. every peer r: its kernel stream waits for drained[slot] of its GPU (the exchange that last read this payload slot),
//      pcs_process_frames_device packs its cameras (the root straight into the head of the stitched buffer); with a
//      predicate the counts follow on the same stream into page-locked memory; packed[slot][r] is recorded;
//   2. the OTHER slot's exchange, if it was left pending, is issued now — its kernels were enqueued a whole submit ago, so
//      the wait for its counts is (normally) over before it starts, and this frame-set's kernels are already queued behind;
//   3. without a predicate the counts follow from the configuration and this slot's exchange is issued at once; with one it
//      stays pending until the next submit or pcs_node_wait.
// The exchange itself (issue_exchange): every GPU's communication stream waits for packed[slot] of its peers, ONE group —
// peer r ncclSend()s its payload, the root ncclRecv()s it at its camera-order offset — and drained[slot] is recorded on
// every communication stream on every path.
int pcs_node_submit_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                           int16_t* d_stitched, size_t stitched_shorts, int* ticket)
{
    if (!n || !d_depth || !d_color || !d_stitched || !ticket) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (stitched_shorts < pcs_node_max_payload_shorts(n))
        return nfail(n, PCS_ERR_CAPACITY, "stitched payload holds %zu shorts, %zu needed", stitched_shorts, pcs_node_max_payload_shorts(n));
    Ticket* tkp = nullptr; int slot = 0;
    int rc = check_submit(n, tkp, slot);
    if (rc != PCS_OK) return rc;
    Ticket& tk = *tkp;
    const int S = n->per_dev, P = n->n_peers;
    const double t_host0 = now_ms();
    tk = Ticket{};
    tk.cnt.assign((size_t)P * (S + 1), 0);
    tk.kind = kStitch; tk.d_stitched = d_stitched; tk.id = n->next_ticket; tk.timing = n->timing;
    // 1. kernels. A failure here leaves the ticket free and the slot's drained events as they were (complete): the kernels
    //    already enqueued write payload slots nothing will read, and the next submit simply reuses the slot.
    for (int r = 0; r < P; r++) {
        Peer& p = n->peers[r];
        HIPCHK(n, hipSetDevice(p.dev));
        hipStream_t ks = kstream(p);
        if (n->have_comm) HIPCHK(n, hipStreamWaitEvent(ks, n->gpus[p.gpu].drained[slot], 0));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k0[slot], ks));
        // where this peer packs: the root into the head of the stitched buffer; a peer into its payload slot — or, with direct
        // stores and no predicate (its camera-order offset follows from the configuration), straight into the ROOT GPU's stitched
        // buffer over xGMI: the pack kernel's own stores are the gather
        const bool into_root = r == 0 || (n->direct && !n->pred);
        size_t before = 0;
        if (into_root) for (int q = 0; q < r; q++) before += (size_t)n->static_cnt[(size_t)q * (S + 1) + S];
        int16_t* dst = into_root ? d_stitched + before * PCS_POINT_SHORTS : static_cast<int16_t*>(p.d_payload[slot]);
        PCSCHK(n, p.ctx, pcs_process_frames_device(p.ctx, d_depth + (size_t)r * S, d_color + (size_t)r * S, dst,
                                                   into_root ? stitched_shorts - before * PCS_POINT_SHORTS : p.payload_shorts,
                                                   n->pred ? static_cast<int32_t*>(p.d_counts) : nullptr));
        if (n->pred)
            HIPCHK(n, hipMemcpyAsync(n->h_counts[slot] + (size_t)r * (S + 1), p.d_counts, sizeof(int32_t) * (S + 1), hipMemcpyDeviceToHost, ks));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k1[slot], ks));
        HIPCHK(n, hipEventRecord(p.packed[slot], ks));
    }
    tk.busy = true;
    tk.submit_host_ms = (float)(now_ms() - t_host0);        // every peer's kernels enqueued; the exchange's own enqueue is exchange_host_ms
    *ticket = n->next_ticket++;
    // 2. the older frame-set's exchange, if it was waiting for its counts
    flush_other(n, slot);
    // 3. this frame-set's, when nothing on the device sizes it
    if (!n->pred) {
        tk.cnt = n->static_cnt;
        issue_exchange(n, tk);
        if (tk.rc != PCS_OK) {          // report it here; the ticket is gone (its drained events are recorded)
            tk.busy = false;
            n->err = tk.err;
            return tk.rc;
        }
    }
    return PCS_OK;
}

static int wait_common(pcs_node* n, int ticket, int kind, Ticket*& out)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    Ticket& tk = n->inflight[ticket & 1];
    if (ticket < 0 || ticket >= n->next_ticket || !tk.busy || tk.id != ticket || tk.kind != kind)
        return nfail(n, PCS_ERR_INVALID_ARG, "ticket %d is not in flight", ticket);
    if (!tk.exchanged) issue_exchange(n, tk);
    tk.busy = false;                                    // whatever happens below, the slot is free again
    out = &tk;
    const int slot = ticket & 1;
    if (n->have_comm || n->broken) {
        for (Gpu& g : n->gpus) {
            HIPCHK(n, hipSetDevice(g.dev));
            HIPCHK(n, hipEventSynchronize(g.drained[slot]));
        }
    } else {
        for (Peer& p : n->peers) {                      // no exchange: the frame-set is complete when every peer's kernels are
            HIPCHK(n, hipSetDevice(p.dev));
            HIPCHK(n, hipEventSynchronize(p.packed[slot]));
        }
    }
    if (tk.rc != PCS_OK) { n->err = tk.err; return tk.rc; }
    return PCS_OK;
}

int pcs_node_wait(pcs_node* n, int ticket, int* points_per_stream, int* total_points)
{
    Ticket* tk = nullptr;
    const int rc = wait_common(n, ticket, kStitch, tk);
    if (rc != PCS_OK) return rc;
    const int S = n->per_dev;
    if (tk->timing) {           // the root's own kernels may still be running when the peers' payloads have landed
        HIPCHK(n, hipSetDevice(n->peers[0].dev));
        HIPCHK(n, hipEventSynchronize(n->peers[0].packed[ticket & 1]));
    }
    if (points_per_stream)
        for (int r = 0; r < n->n_peers; r++) for (int k = 0; k < S; k++) points_per_stream[r * S + k] = tk->cnt[(size_t)r * (S + 1) + k];
    if (total_points) *total_points = (int)tk->total;
    fill_stats(n, *tk);
    return PCS_OK;
}

int pcs_node_process_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                            int16_t* d_stitched, size_t stitched_shorts, int* points_per_stream, int* total_points)
{
    int ticket = -1;
    const int rc = pcs_node_submit_device(n, d_depth, d_color, d_stitched, stitched_shorts, &ticket);
    if (rc != PCS_OK) return rc;
    return pcs_node_wait(n, ticket, points_per_stream, total_points);
}

int pcs_node_process(pcs_node* n, const uint16_t* const* depth, const uint8_t* const* color, int16_t* stitched,
                     size_t stitched_shorts, int write_header, int* points_per_stream, int* out_size_bytes)
{
    if (!n || !depth || !color || !stitched) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    const size_t max_sh = pcs_node_max_payload_shorts(n);
    if (stitched_shorts < PCS_HEADER_SHORTS + max_sh)
        return nfail(n, PCS_ERR_CAPACITY, "stitched buffer holds %zu shorts, %zu needed", stitched_shorts, PCS_HEADER_SHORTS + max_sh);
    std::vector<const uint16_t*> dd;
    std::vector<const uint8_t*> dc;
    int rc = upload_rasters(n, depth, color, dd, dc);
    if (rc != PCS_OK) return rc;
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    if (!n->d_stitched) {
        PCSCHK(n, root.ctx, pcs_device_malloc(root.ctx, &n->d_stitched, max_sh * sizeof(int16_t) + 64));
        n->stitched_cap_shorts = max_sh;
    }
    int total = 0;
    rc = pcs_node_process_device(n, dd.data(), dc.data(), static_cast<int16_t*>(n->d_stitched), n->stitched_cap_shorts,
                                 points_per_stream, &total);
    if (rc != PCS_OK) return rc;
    HIPCHK(n, hipSetDevice(root.dev));
    const int32_t size = (int32_t)((size_t)total * PCS_POINT_BYTES);
    if (size) PCSCHK(n, root.ctx, pcs_memcpy_d2h(root.ctx, stitched + PCS_HEADER_SHORTS, n->d_stitched, (size_t)size));
    if (write_header) std::memcpy(stitched, &size, sizeof size);
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

// ---- config 5: voxel grid of the node's stitched cloud -----------------------------------------------------------------
// Route PARTIALS, pipelined. Per submit:
//   1. every peer r: kernel stream waits for drained[slot] of its GPU, pcs_process_frames_voxel_partials_device into slot's
//      key / partial arrays (the root appends to the head of its merged arrays), one asynchronous 4-byte read-back of the
//      partial count into page-locked memory, packed[slot][r];
//   2. the other slot's pending exchange + reduce are issued (its counts have landed by now);
// and the exchange of THIS slot stays pending until the next submit or pcs_node_wait_voxel: ONE group on the communication
// streams (peer r ncclSend()s m_r keys and m_r partials, the root ncclRecv()s them behind its own and the earlier peers'),
// then, on the root's kernel stream behind it, pcs_voxel_grid_from_partials_device over all M = sum m_r partials and the
// read-back of the voxel count. So the pre-aggregation of frame-set k+1 overlaps the exchange and the root's sort of k.
int pcs_node_submit_voxel_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm,
                                 int16_t* d_voxels, size_t voxels_shorts, int* ticket)
{
    if (!n || !d_depth || !d_color || !d_voxels || !ticket) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (leaf_mm < 1 || leaf_mm > 32767) return nfail(n, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (voxels_shorts < pcs_node_max_payload_shorts(n))
        return nfail(n, PCS_ERR_CAPACITY, "voxel buffer holds %zu shorts; the worst case (every point its own voxel) needs %zu",
                     voxels_shorts, pcs_node_max_payload_shorts(n));
    Ticket* tkp = nullptr; int slot = 0;
    int rc = check_submit(n, tkp, slot);
    if (rc != PCS_OK) return rc;
    const int S = n->per_dev, P = n->n_peers;
    const bool one_call = P == 1 && n->one_call != 0;
    const bool sink = P > 1 && n->gpus.size() == 1 && n->vox_sink != 0;
    rc = (one_call || sink) ? ensure_voxel_counts(n) : ensure_voxel_buffers(n);
    if (rc != PCS_OK) return rc;
    if (sink) {
        rc = ensure_sinks(n);
        if (rc != PCS_OK) return rc;
    }
    if (one_call && n->one_call == 2 && slot == 1 && !n->alt_ctx) {
        rc = ensure_alt_context(n);
        if (rc != PCS_OK) return rc;
    }
    Ticket& tk = *tkp;
    const double t_host0 = now_ms();
    tk = Ticket{};
    tk.kind = kVoxel; tk.id = n->next_ticket; tk.leaf = leaf_mm; tk.d_voxels = d_voxels; tk.voxels_shorts = voxels_shorts;
    tk.timing = n->timing;
    // ONE peer holds every camera: there is nothing to exchange, and the partials need not leave the library's workspace — the
    // call that goes from the rasters to the voxel cloud (warm bucket tail: two launches) on the peer's own stream. Two frame-sets
    // in flight then simply queue behind each other: 16 x 1080p at 50 mm 0.174 ms per frame-set, against 0.205 for the
    // partials / reduce-on-a-second-context pipeline below (its tail beside the next pre-aggregation; PCS_NODE_ONE_CALL=0 keeps it).
    if (one_call) {
        // Two frame-sets in flight then run on two contexts in turn (slot 0: the peer's own, slot 1: alt_ctx), each with its own stream,
        // workspace, splitters and regions: the bucket tail of frame-set k — a latency chain that leaves the chip almost empty — runs
        // beside the pre-aggregation of k+1 (16 x 1080p, 50 mm: 0.155 ms per frame-set instead of 0.174 with one context; the
        // partials / reduce-on-a-second-context pipeline below: 0.205; PCS_NODE_ONE_CALL=1 / 0 keep those).
        Peer& p = n->peers[0];
        HIPCHK(n, hipSetDevice(p.dev));
        pcs_ctx* oc = (n->one_call == 2 && slot == 1 && n->alt_ctx) ? n->alt_ctx : p.ctx;
        hipStream_t ks = static_cast<hipStream_t>(pcs_get_stream(oc));
        tk.vox_ctx = oc;
        if (tk.timing) HIPCHK(n, hipEventRecord(n->ev_k0[slot], ks));
        PCSCHK(n, oc, pcs_process_frames_voxel_device(oc, d_depth, d_color, leaf_mm, d_voxels, voxels_shorts,
                                                      static_cast<int32_t*>(n->d_vox_n[slot])));
        HIPCHK(n, hipMemcpyAsync(n->h_vcount[slot] + P, n->d_vox_n[slot], sizeof(int32_t), hipMemcpyDeviceToHost, ks));
        if (tk.timing) { HIPCHK(n, hipEventRecord(n->ev_k1[slot], ks)); HIPCHK(n, hipEventRecord(n->ev_r0[slot], ks)); }
        HIPCHK(n, hipEventRecord(p.packed[slot], ks));
        HIPCHK(n, hipEventRecord(n->ev_done[slot], ks));
        tk.one_call = true; tk.exchanged = true; tk.busy = true;
        tk.d_depth.assign(d_depth, d_depth + (size_t)P * S); tk.d_color.assign(d_color, d_color + (size_t)P * S);
        tk.submit_host_ms = (float)(now_ms() - t_host0);
        *ticket = n->next_ticket++;
        flush_other(n, slot);
        return PCS_OK;
    }
    if (sink) {
        // every peer shares the root's GPU: its partials go straight into the slot's sink, nothing is exchanged (see pcs_node::vox_sink)
        tk.sink = true; tk.vox_ctx = n->sink_ctx[slot];
        tk.d_depth.assign(d_depth, d_depth + (size_t)P * S); tk.d_color.assign(d_color, d_color + (size_t)P * S);
        rc = enqueue_sink_ticket(n, tk, slot);
        if (rc != PCS_OK) return rc;
        tk.exchanged = true; tk.busy = true;
        tk.submit_host_ms = (float)(now_ms() - t_host0);
        *ticket = n->next_ticket++;
        flush_other(n, slot);
        return PCS_OK;
    }
    for (int r = 0; r < P; r++) {
        Peer& p = n->peers[r];
        HIPCHK(n, hipSetDevice(p.dev));
        hipStream_t ks = kstream(p);
        if (n->have_comm) HIPCHK(n, hipStreamWaitEvent(ks, n->gpus[p.gpu].drained[slot], 0));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k0[slot], ks));
        PCSCHK(n, p.ctx, pcs_process_frames_voxel_partials_device(p.ctx, d_depth + (size_t)r * S, d_color + (size_t)r * S, leaf_mm,
                                                                  static_cast<uint64_t*>(p.d_vkeys[slot]),
                                                                  static_cast<pcs_voxel_partial*>(p.d_vparts[slot]),
                                                                  r == 0 ? n->vcap_total : p.vcap, static_cast<int32_t*>(p.d_vcount[slot])));
        HIPCHK(n, hipMemcpyAsync(n->h_vcount[slot] + r, p.d_vcount[slot], sizeof(int32_t), hipMemcpyDeviceToHost, ks));
        if (r == 0 && tk.timing) HIPCHK(n, hipEventRecord(n->ev_k1[slot], ks));
        HIPCHK(n, hipEventRecord(p.packed[slot], ks));
    }
    tk.busy = true;
    tk.submit_host_ms = (float)(now_ms() - t_host0);
    *ticket = n->next_ticket++;
    flush_other(n, slot);
    return PCS_OK;
}

int pcs_node_wait_voxel(pcs_node* n, int ticket, int* n_voxels)
{
    Ticket* tk = nullptr;
    const int rc = wait_common(n, ticket, kVoxel, tk);
    if (rc != PCS_OK) return rc;
    const int slot = ticket & 1, P = n->n_peers;
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    HIPCHK(n, hipEventSynchronize(n->ev_done[slot]));
    fill_stats(n, *tk);
    if (n->h_vcount[slot][P] < 0) {
        // The bucket tail gave up waiting for one of its own workgroups (include/pcs_hip.h): the bytes in d_voxels are not valid.
        // Once more on the LSD tail, which waits for nobody, latched for that context; what the tail read is still where it was
        // (the rasters of a one-call ticket are the caller's until this wait returns; the merged partials sit in this slot's
        // arrays until the slot's next submit). It queues behind whatever the next submit already enqueued on that stream.
        pcs_ctx* vc = (tk->one_call || tk->sink) ? tk->vox_ctx : n->reduce_ctx;
        PCSCHK(n, vc, pcs_set_voxel_tail(vc, PCS_VOXEL_TAIL_LSD_LATCHED));
        n->voxel_reruns++;
        hipStream_t vs = static_cast<hipStream_t>(pcs_get_stream(vc));
        if (tk->sink) {
            const int rr = enqueue_sink_ticket(n, *tk, slot);          // every peer again, into the same sink, its tail now LSD
            if (rr != PCS_OK) return rr;
        } else if (tk->one_call)
            PCSCHK(n, vc, pcs_process_frames_voxel_device(vc, tk->d_depth.data(), tk->d_color.data(), tk->leaf, tk->d_voxels, tk->voxels_shorts,
                                                          static_cast<int32_t*>(n->d_vox_n[slot])));
        else
            PCSCHK(n, vc, pcs_voxel_grid_from_partials_device(vc, static_cast<const uint64_t*>(root.d_vkeys[slot]),
                                                              static_cast<const pcs_voxel_partial*>(root.d_vparts[slot]), (int)tk->total, nullptr,
                                                              tk->leaf, tk->d_voxels, tk->voxels_shorts, static_cast<int32_t*>(n->d_vox_n[slot])));
        if (!tk->sink) HIPCHK(n, hipMemcpyAsync(n->h_vcount[slot] + P, n->d_vox_n[slot], sizeof(int32_t), hipMemcpyDeviceToHost, vs));
        HIPCHK(n, hipStreamSynchronize(vs));
        if (n->h_vcount[slot][P] < 0)
            return nfail(n, PCS_ERR_HIP, "the voxel pipeline reported a negative count on the LSD tail too (device stalled?)");
    }
    if (n_voxels) *n_voxels = n->h_vcount[slot][P];
    return PCS_OK;
}

int pcs_node_voxel_reruns(const pcs_node* n) { return n ? n->voxel_reruns : 0; }

int pcs_node_set_one_call(pcs_node* n, int mode)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    if (mode < 0 || mode > 2) return nfail(n, PCS_ERR_INVALID_ARG, "one-call mode %d outside 0..2", mode);
    if (n->inflight[0].busy || n->inflight[1].busy) return nfail(n, PCS_ERR_INVALID_ARG, "wait for the frame-sets in flight first");
    n->one_call = mode;
    return PCS_OK;
}

int pcs_node_set_voxel_sink(pcs_node* n, int on)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    if (n->inflight[0].busy || n->inflight[1].busy) return nfail(n, PCS_ERR_INVALID_ARG, "wait for the frame-sets in flight first");
    n->vox_sink = on ? 1 : 0;
    if (!on) return unshare_streams(n);
    return PCS_OK;
}

int pcs_node_voxel_sink(const pcs_node* n) { return (n && n->n_peers > 1 && n->gpus.size() == 1 && n->vox_sink) ? 1 : 0; }

int pcs_node_process_voxel_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm,
                                  int route, int16_t* d_voxels, size_t voxels_shorts, int* n_voxels, pcs_node_voxel_stats* stats)
{
    if (!n || !d_depth || !d_color || !d_voxels || !n_voxels) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (leaf_mm < 1 || leaf_mm > 32767) return nfail(n, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (route != PCS_NODE_VOXEL_PARTIALS && route != PCS_NODE_VOXEL_PAYLOADS) return nfail(n, PCS_ERR_INVALID_ARG, "unknown route %d", route);
    if (voxels_shorts < pcs_node_max_payload_shorts(n))
        return nfail(n, PCS_ERR_CAPACITY, "voxel buffer holds %zu shorts; the worst case (every point its own voxel) needs %zu",
                     voxels_shorts, pcs_node_max_payload_shorts(n));
    if (n->inflight[0].busy || n->inflight[1].busy)
        return nfail(n, PCS_ERR_INVALID_ARG, "wait for the frame-sets in flight before a synchronous voxel call");
    if (route == PCS_NODE_VOXEL_PAYLOADS && n->n_peers > 1 && !n->have_comm)
        // nothing is gathered without a communicator: the stitched buffer holds the root's slice only, and a voxel grid over
        // the whole node's point count would read memory no kernel wrote
        return nfail(n, PCS_ERR_UNSUPPORTED, "route PAYLOADS needs the exchange: the node was created with PCS_NODE_NO_EXCHANGE "
                     "(or its communicators were aborted)");
    const bool was_timing = n->timing;
    n->timing = n->timing || stats != nullptr;
    struct Restore { pcs_node* n; bool t; ~Restore() { n->timing = t; } } restore{n, was_timing};
    int rc, ticket = -1, nv = 0;
    pcs_node_stats st{};
    if (route == PCS_NODE_VOXEL_PARTIALS) {
        rc = pcs_node_submit_voxel_device(n, d_depth, d_color, leaf_mm, d_voxels, voxels_shorts, &ticket);
        if (rc != PCS_OK) return rc;
        rc = pcs_node_wait_voxel(n, ticket, &nv);
        if (rc != PCS_OK) return rc;
        st = n->last;
    } else {
        // the reference's shape: camera-order concatenation of the (compacted) payloads on the root, downsample there
        rc = ensure_voxel_counts(n);
        if (rc != PCS_OK) return rc;
        Peer& root = n->peers[0];
        const size_t max_sh = pcs_node_max_payload_shorts(n);
        HIPCHK(n, hipSetDevice(root.dev));
        if (!n->d_stitched) {
            PCSCHK(n, root.ctx, pcs_device_malloc(root.ctx, &n->d_stitched, max_sh * sizeof(int16_t) + 64));
            n->stitched_cap_shorts = max_sh;
        }
        int total = 0;
        rc = pcs_node_submit_device(n, d_depth, d_color, static_cast<int16_t*>(n->d_stitched), n->stitched_cap_shorts, &ticket);
        if (rc != PCS_OK) return rc;
        rc = pcs_node_wait(n, ticket, nullptr, &total);
        if (rc != PCS_OK) return rc;
        st = n->last;
        HIPCHK(n, hipSetDevice(root.dev));
        hipStream_t ks = kstream(root);
        HIPCHK(n, hipEventRecord(n->ev_r0[0], ks));
        PCSCHK(n, root.ctx, pcs_voxel_grid_device(root.ctx, static_cast<const int16_t*>(n->d_stitched), total, leaf_mm, d_voxels,
                                                  voxels_shorts, static_cast<int32_t*>(n->d_vox_n[0])));
        HIPCHK(n, hipEventRecord(n->ev_done[0], ks));
        int32_t v = 0;
        PCSCHK(n, root.ctx, pcs_memcpy_d2h(root.ctx, &v, n->d_vox_n[0], sizeof v));        // synchronises the root's kernel stream
        if (v < 0) {        // flagged bucket tail: the stitched cloud is still there — again on the LSD tail (pcs_node_wait_voxel)
            PCSCHK(n, root.ctx, pcs_set_voxel_tail(root.ctx, PCS_VOXEL_TAIL_LSD_LATCHED));
            n->voxel_reruns++;
            PCSCHK(n, root.ctx, pcs_voxel_grid_device(root.ctx, static_cast<const int16_t*>(n->d_stitched), total, leaf_mm, d_voxels,
                                                      voxels_shorts, static_cast<int32_t*>(n->d_vox_n[0])));
            PCSCHK(n, root.ctx, pcs_memcpy_d2h(root.ctx, &v, n->d_vox_n[0], sizeof v));
            if (v < 0) return nfail(n, PCS_ERR_HIP, "the voxel pipeline reported a negative count on the LSD tail too (device stalled?)");
        }
        nv = v;
        (void)hipEventElapsedTime(&st.root_ms, n->ev_r0[0], n->ev_done[0]);
        st.reduced = total;
    }
    *n_voxels = nv;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->kernels_ms = st.kernels_ms; stats->exchange_ms = st.exchange_ms; stats->root_voxel_ms = st.root_ms;
        stats->exchanged_bytes = st.exchanged_bytes; stats->partials = (int32_t)st.reduced; stats->voxels = nv;
    }
    return PCS_OK;
}

int pcs_node_process_voxel(pcs_node* n, const uint16_t* const* depth, const uint8_t* const* color, int leaf_mm, int route,
                           int16_t* out, size_t out_shorts, int write_header, int* out_size_bytes, pcs_node_voxel_stats* stats)
{
    if (!n || !depth || !color || !out) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    std::vector<const uint16_t*> dd;
    std::vector<const uint8_t*> dc;
    int rc = upload_rasters(n, depth, color, dd, dc);
    if (rc != PCS_OK) return rc;
    Peer& root = n->peers[0];
    HIPCHK(n, hipSetDevice(root.dev));
    const size_t max_sh = pcs_node_max_payload_shorts(n);
    if (!n->d_vox_out) PCSCHK(n, root.ctx, pcs_device_malloc(root.ctx, &n->d_vox_out, max_sh * sizeof(int16_t) + 64));
    int nv = 0;
    rc = pcs_node_process_voxel_device(n, dd.data(), dc.data(), leaf_mm, route, static_cast<int16_t*>(n->d_vox_out), max_sh, &nv, stats);
    if (rc != PCS_OK) return rc;
    if (out_shorts < PCS_HEADER_SHORTS + (size_t)nv * PCS_POINT_SHORTS)
        return nfail(n, PCS_ERR_CAPACITY, "output holds %zu shorts, %zu needed", out_shorts, PCS_HEADER_SHORTS + (size_t)nv * PCS_POINT_SHORTS);
    HIPCHK(n, hipSetDevice(root.dev));
    const int32_t size = (int32_t)((size_t)nv * PCS_POINT_BYTES);
    if (size) PCSCHK(n, root.ctx, pcs_memcpy_d2h(root.ctx, out + PCS_HEADER_SHORTS, n->d_vox_out, (size_t)size));
    if (write_header) std::memcpy(out, &size, sizeof size);
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

}  // extern "C"
