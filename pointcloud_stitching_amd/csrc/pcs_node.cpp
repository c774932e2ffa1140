// pcs_node.cpp — libpcs_node.so: several GPUs, one process, one grouped RCCL exchange to the root.
// See include/pcs_node.h for what it replaces in the reference. Built on the public C ABI of libpcs_hip.so
// (it uses nothing from it that an outside caller could not).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/pcs_node.h"

struct pcs_node {
    int n_dev = 0, per_dev = 0, n_streams = 0;
    uint32_t flags = 0;
    int downsample = 1;
    std::vector<int> dev;
    std::vector<pcs_ctx*> ctx;
    std::vector<ncclComm_t> comm;
    bool broken = false;                     // an RCCL call failed: the communicators were aborted
    // per device, two slots (index 0 unused: the root packs into the stitched buffer): the kernel of frame-set k+1
    // fills one payload buffer while the exchange of frame-set k drains the other
    std::vector<void*> d_payload[2];
    std::vector<size_t> payload_shorts;      // per device capacity
    std::vector<hipStream_t> comm_stream;    // per device: the exchange runs here, not on the kernel stream
    std::vector<hipEvent_t> packed[2];       // per device and slot: payload packed (kernel stream -> comm stream)
    std::vector<hipEvent_t> drained[2];      // per device and slot: exchange done (comm stream -> kernel stream / host)
    struct Ticket { bool busy = false; int slot = 0; std::vector<std::vector<int32_t>> cnt; size_t total = 0; };
    Ticket inflight[2];
    int next_ticket = 0;
    bool pred = false;
    std::vector<void*> d_counts;             // per device: per_dev + 1 int32
    int32_t* h_counts[2] = {nullptr, nullptr};   // page-locked: [slot][device * (per_dev + 1) + k]  (asynchronous read-back)
    std::vector<std::vector<void*>> d_depth, d_color;   // staging for the host form, per global stream
    std::vector<pcs_stream_config> cfg;
    void* d_stitched = nullptr; size_t stitched_cap_shorts = 0;
    // voxel route (config 5): per device key / partial arrays (the root's are the merged arrays, sized for all devices)
    std::vector<void*> d_vkeys, d_vparts, d_vcount;
    std::vector<size_t> vcap;                // per device: worst-case partials of its own cameras
    size_t vcap_total = 0;
    int32_t* h_vcount = nullptr;             // page-locked, n_dev
    void* d_vox_out = nullptr;               // root, host form: the voxel cloud
    void* d_vox_n = nullptr;                 // root: voxel count
    hipEvent_t ev_v[4] = {nullptr, nullptr, nullptr, nullptr};     // root: start, own kernel done, exchange done, voxels done
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

hipStream_t kstream(pcs_node* n, int r) { return static_cast<hipStream_t>(pcs_get_stream(n->ctx[r])); }

// One transfer of a grouped exchange: `bytes` from src on device index `from` to dst on the root.
struct Xfer { int from; const void* src; void* dst; size_t bytes; };

// The grouped exchange. Whatever happens between ncclGroupStart and ncclGroupEnd, the group is CLOSED before this
// returns; on any RCCL failure the communicators are aborted (a half-issued group can leave sends without their receives
// on the communication streams) and the node is marked unusable.
int run_exchange(pcs_node* n, const std::vector<Xfer>& xs)
{
    if (n->n_dev < 2 || xs.empty()) return PCS_OK;
    ncclResult_t first = ncclGroupStart();
    if (first == ncclSuccess) {
        for (const Xfer& x : xs) {
            if (!x.bytes) continue;
            ncclResult_t r = ncclSend(x.src, x.bytes, ncclInt8, 0, n->comm[x.from], n->comm_stream[x.from]);
            if (r == ncclSuccess) r = ncclRecv(x.dst, x.bytes, ncclInt8, x.from, n->comm[0], n->comm_stream[0]);
            if (r != ncclSuccess) { first = r; break; }
        }
        const ncclResult_t end = ncclGroupEnd();          // always: never leave a group open
        if (first == ncclSuccess) first = end;
    }
    if (first == ncclSuccess) return PCS_OK;
    for (ncclComm_t& c : n->comm) if (c) { (void)ncclCommAbort(c); c = nullptr; }
    n->broken = true;
    return nfail(n, PCS_ERR_HIP, "RCCL exchange failed: %s (communicators aborted; destroy the node)", ncclGetErrorString(first));
}

// Host rasters -> the owning GPUs' staging buffers (allocated on first use).
int upload_rasters(pcs_node* n, const uint16_t* const* depth, const uint8_t* const* color, std::vector<const uint16_t*>& dd,
                   std::vector<const uint8_t*>& dc)
{
    const int S = n->per_dev;
    dd.assign(n->n_streams, nullptr); dc.assign(n->n_streams, nullptr);
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        for (int k = 0; k < S; k++) {
            const int g = r * S + k;
            const pcs_stream_config& sc = n->cfg[g];
            const size_t db = (size_t)sc.depth.width * sc.depth.height * sizeof(uint16_t);
            const size_t cb = (size_t)sc.color_stride * sc.color.height;
            if (!depth[g] || !color[g]) return nfail(n, PCS_ERR_INVALID_ARG, "stream %d: NULL raster pointer", g);
            if (!n->d_depth[r][k]) PCSCHK(n, n->ctx[r], pcs_device_malloc(n->ctx[r], &n->d_depth[r][k], db + 16));
            if (!n->d_color[r][k]) PCSCHK(n, n->ctx[r], pcs_device_malloc(n->ctx[r], &n->d_color[r][k], cb + 16));
            PCSCHK(n, n->ctx[r], pcs_memcpy_h2d(n->ctx[r], n->d_depth[r][k], depth[g], db));
            PCSCHK(n, n->ctx[r], pcs_memcpy_h2d(n->ctx[r], n->d_color[r][k], color[g], cb));
            dd[g] = static_cast<const uint16_t*>(n->d_depth[r][k]);
            dc[g] = static_cast<const uint8_t*>(n->d_color[r][k]);
        }
    }
    return PCS_OK;
}

int ensure_voxel_buffers(pcs_node* n)
{
    if (!n->d_vkeys.empty()) return PCS_OK;
    std::vector<void*> keys(n->n_dev, nullptr), parts(n->n_dev, nullptr), cnt(n->n_dev, nullptr);
    auto undo = [&]() {
        for (int r = 0; r < n->n_dev; r++) {
            (void)hipSetDevice(n->dev[r]);
            if (keys[r]) pcs_device_free(n->ctx[r], keys[r]);
            if (parts[r]) pcs_device_free(n->ctx[r], parts[r]);
            if (cnt[r]) pcs_device_free(n->ctx[r], cnt[r]);
        }
    };
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        const size_t cap = r == 0 ? n->vcap_total : n->vcap[r];        // the root's arrays take everybody's partials
        if (pcs_device_malloc(n->ctx[r], &keys[r], cap * sizeof(uint64_t) + 64) != PCS_OK ||
            pcs_device_malloc(n->ctx[r], &parts[r], cap * sizeof(pcs_voxel_partial) + 64) != PCS_OK ||
            pcs_device_malloc(n->ctx[r], &cnt[r], 64) != PCS_OK) {
            const int rc = nfail(n, PCS_ERR_NOMEM, "device %d: %s", n->dev[r], pcs_last_error(n->ctx[r]));
            undo();
            return rc;
        }
    }
    hipError_t e = hipSetDevice(n->dev[0]);
    if (e == hipSuccess && !n->h_vcount) e = hipHostMalloc((void**)&n->h_vcount, sizeof(int32_t) * (size_t)n->n_dev, hipHostMallocPortable);
    if (e == hipSuccess && !n->d_vox_n) e = pcs_device_malloc(n->ctx[0], &n->d_vox_n, 64) == PCS_OK ? hipSuccess : hipErrorOutOfMemory;
    for (int k = 0; k < 4 && e == hipSuccess; k++) if (!n->ev_v[k]) e = hipEventCreate(&n->ev_v[k]);
    if (e != hipSuccess) { undo(); return nfail(n, PCS_ERR_HIP, "voxel route set-up: %s", hipGetErrorString(e)); }
    n->d_vkeys = keys; n->d_vparts = parts; n->d_vcount = cnt;
    return PCS_OK;
}
}  // namespace

extern "C" {

const char* pcs_node_last_error(const pcs_node* n) { return n ? n->err.c_str() : g_err.c_str(); }
int pcs_node_devices(const pcs_node* n) { return n ? n->n_dev : 0; }

size_t pcs_node_max_payload_shorts(const pcs_node* n)
{
    if (!n) return 0;
    size_t s = 0;
    for (pcs_ctx* c : n->ctx) s += pcs_max_payload_shorts(c);
    return s;
}

void pcs_node_destroy(pcs_node* n)
{
    if (!n) return;
    for (size_t r = 0; r < n->ctx.size(); r++) {
        if (!n->ctx[r]) continue;
        (void)hipSetDevice(n->dev[r]);
        if (r < n->comm_stream.size() && n->comm_stream[r]) (void)hipStreamSynchronize(n->comm_stream[r]);
        for (int sl = 0; sl < 2; sl++) {
            if (r < n->d_payload[sl].size() && n->d_payload[sl][r]) pcs_device_free(n->ctx[r], n->d_payload[sl][r]);
            if (r < n->packed[sl].size() && n->packed[sl][r]) (void)hipEventDestroy(n->packed[sl][r]);
            if (r < n->drained[sl].size() && n->drained[sl][r]) (void)hipEventDestroy(n->drained[sl][r]);
        }
        if (r < n->comm_stream.size() && n->comm_stream[r]) (void)hipStreamDestroy(n->comm_stream[r]);
        if (r < n->d_counts.size() && n->d_counts[r]) pcs_device_free(n->ctx[r], n->d_counts[r]);
        for (int k = 0; k < n->per_dev && r < n->d_depth.size(); k++) {
            if (k < (int)n->d_depth[r].size() && n->d_depth[r][k]) pcs_device_free(n->ctx[r], n->d_depth[r][k]);
            if (k < (int)n->d_color[r].size() && n->d_color[r][k]) pcs_device_free(n->ctx[r], n->d_color[r][k]);
        }
        if (r < n->d_vkeys.size() && n->d_vkeys[r]) pcs_device_free(n->ctx[r], n->d_vkeys[r]);
        if (r < n->d_vparts.size() && n->d_vparts[r]) pcs_device_free(n->ctx[r], n->d_vparts[r]);
        if (r < n->d_vcount.size() && n->d_vcount[r]) pcs_device_free(n->ctx[r], n->d_vcount[r]);
        if (r == 0) {
            if (n->d_stitched) pcs_device_free(n->ctx[0], n->d_stitched);
            if (n->d_vox_out) pcs_device_free(n->ctx[0], n->d_vox_out);
            if (n->d_vox_n) pcs_device_free(n->ctx[0], n->d_vox_n);
            for (hipEvent_t e : n->ev_v) if (e) (void)hipEventDestroy(e);
        }
    }
    for (int sl = 0; sl < 2; sl++) if (n->h_counts[sl]) (void)hipHostFree(n->h_counts[sl]);
    if (n->h_vcount) (void)hipHostFree(n->h_vcount);
    for (ncclComm_t c : n->comm) if (c) (void)ncclCommDestroy(c);
    for (pcs_ctx* c : n->ctx) if (c) pcs_destroy(c);
    delete n;
}

int pcs_node_create(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                    const pcs_stream_config* streams, uint32_t flags, int downsample)
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
    n->n_dev = n_devices; n->per_dev = streams_per_device; n->n_streams = n_devices * streams_per_device;
    n->flags = flags; n->downsample = downsample;
    n->dev.assign(device_ids, device_ids + n_devices);
    n->cfg.assign(streams, streams + n->n_streams);
    n->ctx.assign(n_devices, nullptr);
    n->payload_shorts.assign(n_devices, 0); n->d_counts.assign(n_devices, nullptr);
    n->comm_stream.assign(n_devices, nullptr);
    n->vcap.assign(n_devices, 0);
    for (int sl = 0; sl < 2; sl++) {
        n->d_payload[sl].assign(n_devices, nullptr);
        n->packed[sl].assign(n_devices, nullptr); n->drained[sl].assign(n_devices, nullptr);
    }
    n->pred = (flags & (PCS_FLAG_CUTOFF | PCS_FLAG_DROP_INVALID)) != 0;
    n->d_depth.assign(n_devices, std::vector<void*>(streams_per_device, nullptr));
    n->d_color.assign(n_devices, std::vector<void*>(streams_per_device, nullptr));
    for (int r = 0; r < n_devices; r++) {
        pcs_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.device = device_ids[r]; cfg.n_streams = streams_per_device; cfg.streams = streams + (size_t)r * streams_per_device;
        cfg.flags = flags; cfg.downsample = downsample;
        int rc = pcs_create(&n->ctx[r], &cfg);
        if (rc != PCS_OK) { int e = nfail(nullptr, rc, "device %d: %s", device_ids[r], pcs_last_error(nullptr)); pcs_node_destroy(n); return e; }
        n->payload_shorts[r] = pcs_max_payload_shorts(n->ctx[r]);
        n->vcap[r] = n->payload_shorts[r] / PCS_POINT_SHORTS;
        n->vcap_total += n->vcap[r];
        if (pcs_device_malloc(n->ctx[r], &n->d_counts[r], sizeof(int32_t) * (streams_per_device + 1)) != PCS_OK ||
            (r > 0 && (pcs_device_malloc(n->ctx[r], &n->d_payload[0][r], n->payload_shorts[r] * sizeof(int16_t) + 64) != PCS_OK ||
                       pcs_device_malloc(n->ctx[r], &n->d_payload[1][r], n->payload_shorts[r] * sizeof(int16_t) + 64) != PCS_OK))) {
            int e = nfail(nullptr, PCS_ERR_NOMEM, "device %d: %s", device_ids[r], pcs_last_error(n->ctx[r])); pcs_node_destroy(n); return e;
        }
        hipError_t he = hipSetDevice(device_ids[r]);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&n->comm_stream[r], hipStreamNonBlocking);
        for (int sl = 0; sl < 2 && he == hipSuccess; sl++) {
            he = hipEventCreateWithFlags(&n->packed[sl][r], hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&n->drained[sl][r], hipEventDisableTiming);
        }
        if (he != hipSuccess) { int e = nfail(nullptr, PCS_ERR_HIP, "device %d: %s", device_ids[r], hipGetErrorString(he)); pcs_node_destroy(n); return e; }
    }
    for (int sl = 0; sl < 2; sl++) {
        const hipError_t he = hipHostMalloc((void**)&n->h_counts[sl], sizeof(int32_t) * (size_t)n_devices * (streams_per_device + 1),
                                            hipHostMallocPortable);
        if (he != hipSuccess) { int e = nfail(nullptr, PCS_ERR_NOMEM, "hipHostMalloc: %s", hipGetErrorString(he)); pcs_node_destroy(n); return e; }
    }
    if (n_devices > 1) {       // one communicator per GPU, all in this process
        n->comm.assign(n_devices, nullptr);
        ncclResult_t r = ncclCommInitAll(n->comm.data(), n_devices, device_ids);
        if (r != ncclSuccess) { int e = nfail(nullptr, PCS_ERR_HIP, "ncclCommInitAll: %s", ncclGetErrorString(r)); pcs_node_destroy(n); return e; }
    }
    *out = n;
    return PCS_OK;
}

// Pipelined device form. Per submit, in this order:
//   1. every GPU r: kernel stream waits for drained[slot][r] (the exchange that last read this payload slot), then
//      pcs_process_frames_device packs its cameras (the root straight into the head of the stitched buffer);
//   2. counts: from the configuration, or — with a predicate — one hipMemcpyAsync per GPU into page-locked memory, all in
//      flight together, then one hipStreamSynchronize per GPU;
//   3. every GPU r: packed[slot][r] recorded on the kernel stream, its communication stream waits for it;
//   4. ONE group: rank r ncclSend()s its payload, the root ncclRecv()s it at its camera-order offset, on the
//      communication streams — so the kernels of the NEXT frame-set (other payload slot) overlap it; the group is closed
//      on every path;
//   5. drained[slot][r] recorded on every communication stream — also when step 4 failed, so pcs_node_wait never blocks.
int pcs_node_submit_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                           int16_t* d_stitched, size_t stitched_shorts, int* ticket)
{
    if (!n || !d_depth || !d_color || !d_stitched || !ticket) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (n->broken) return nfail(n, PCS_ERR_HIP, "the node's communicators were aborted after an RCCL failure: destroy it");
    if (stitched_shorts < pcs_node_max_payload_shorts(n))
        return nfail(n, PCS_ERR_CAPACITY, "stitched payload holds %zu shorts, %zu needed", stitched_shorts, pcs_node_max_payload_shorts(n));
    const int slot = n->next_ticket & 1;
    pcs_node::Ticket& tk = n->inflight[slot];
    if (tk.busy) return nfail(n, PCS_ERR_CAPACITY, "two frame-sets are in flight: pcs_node_wait() the older one first");
    const int S = n->per_dev;
    tk.cnt.assign(n->n_dev, std::vector<int32_t>(S + 1, 0));
    // 1. every GPU packs its cameras (its payload slot was drained by the exchange two submits ago)
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        hipStream_t ks = kstream(n, r);
        HIPCHK(n, hipStreamWaitEvent(ks, n->drained[slot][r], 0));
        int16_t* dst = r == 0 ? d_stitched : static_cast<int16_t*>(n->d_payload[slot][r]);
        PCSCHK(n, n->ctx[r], pcs_process_frames_device(n->ctx[r], d_depth + (size_t)r * S, d_color + (size_t)r * S, dst,
                                                       r == 0 ? stitched_shorts : n->payload_shorts[r],
                                                       n->pred ? static_cast<int32_t*>(n->d_counts[r]) : nullptr));
    }
    // 2. counts: known from the configuration unless a predicate makes them data dependent
    if (n->pred) {
        int32_t* hc = n->h_counts[slot];
        for (int r = 0; r < n->n_dev; r++) {
            HIPCHK(n, hipSetDevice(n->dev[r]));
            HIPCHK(n, hipMemcpyAsync(hc + (size_t)r * (S + 1), n->d_counts[r], sizeof(int32_t) * (S + 1), hipMemcpyDeviceToHost, kstream(n, r)));
        }
        for (int r = 0; r < n->n_dev; r++) {
            HIPCHK(n, hipSetDevice(n->dev[r]));
            HIPCHK(n, hipStreamSynchronize(kstream(n, r)));
            for (int k = 0; k <= S; k++) tk.cnt[r][k] = hc[(size_t)r * (S + 1) + k];
        }
    } else {
        for (int r = 0; r < n->n_dev; r++) {
            int64_t tot = 0;
            for (int k = 0; k < S; k++) {
                tk.cnt[r][k] = (pcs_stream_points(n->ctx[r], k) + n->downsample - 1) / n->downsample;
                tot += tk.cnt[r][k];
            }
            tk.cnt[r][S] = (int32_t)tot;
        }
    }
    // 3. the exchange runs on the communication streams, behind each GPU's kernel
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        HIPCHK(n, hipEventRecord(n->packed[slot][r], kstream(n, r)));
        HIPCHK(n, hipStreamWaitEvent(n->comm_stream[r], n->packed[slot][r], 0));
    }
    // 4. one group
    size_t off = (size_t)tk.cnt[0][S];          // points
    std::vector<Xfer> xs;
    for (int r = 1; r < n->n_dev; r++) {
        xs.push_back(Xfer{r, n->d_payload[slot][r], reinterpret_cast<int8_t*>(d_stitched) + off * PCS_POINT_BYTES,
                          (size_t)tk.cnt[r][S] * PCS_POINT_BYTES});
        off += (size_t)tk.cnt[r][S];
    }
    const int xrc = run_exchange(n, xs);
    // 5. drained events: always, so that neither the next submit's kernels nor pcs_node_wait can block on this slot
    for (int r = 0; r < n->n_dev; r++) {
        if (hipSetDevice(n->dev[r]) == hipSuccess) (void)hipEventRecord(n->drained[slot][r], n->comm_stream[r]);
    }
    if (xrc != PCS_OK) return xrc;
    tk.total = off; tk.slot = slot; tk.busy = true;
    *ticket = n->next_ticket++;
    return PCS_OK;
}

int pcs_node_wait(pcs_node* n, int ticket, int* points_per_stream, int* total_points)
{
    if (!n) return PCS_ERR_INVALID_ARG;
    pcs_node::Ticket& tk = n->inflight[ticket & 1];
    if (ticket < 0 || ticket >= n->next_ticket || ticket < n->next_ticket - 2 || !tk.busy)
        return nfail(n, PCS_ERR_INVALID_ARG, "ticket %d is not in flight", ticket);
    const int S = n->per_dev;
    tk.busy = false;                                    // whatever happens below, the slot is free again
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        HIPCHK(n, hipEventSynchronize(n->drained[tk.slot][r]));
    }
    if (points_per_stream)
        for (int r = 0; r < n->n_dev; r++) for (int k = 0; k < S; k++) points_per_stream[r * S + k] = tk.cnt[r][k];
    if (total_points) *total_points = (int)tk.total;
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
    HIPCHK(n, hipSetDevice(n->dev[0]));
    if (!n->d_stitched) {
        PCSCHK(n, n->ctx[0], pcs_device_malloc(n->ctx[0], &n->d_stitched, max_sh * sizeof(int16_t) + 64));
        n->stitched_cap_shorts = max_sh;
    }
    int total = 0;
    rc = pcs_node_process_device(n, dd.data(), dc.data(), static_cast<int16_t*>(n->d_stitched), n->stitched_cap_shorts,
                                 points_per_stream, &total);
    if (rc != PCS_OK) return rc;
    HIPCHK(n, hipSetDevice(n->dev[0]));
    const int32_t size = (int32_t)((size_t)total * PCS_POINT_BYTES);
    if (size) PCSCHK(n, n->ctx[0], pcs_memcpy_d2h(n->ctx[0], stitched + PCS_HEADER_SHORTS, n->d_stitched, (size_t)size));
    if (write_header) std::memcpy(stitched, &size, sizeof size);
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

// ---- config 5: voxel grid of the node's stitched cloud -----------------------------------------------------------------
// Route PARTIALS, per call:
//   1. every GPU r: pcs_process_frames_voxel_partials_device on its kernel stream (the root appends to the head of the
//      merged key / partial arrays), then one asynchronous 4-byte read-back of its partial count;
//   2. one wait per GPU for the counts (they size the exchange), exclusive scan on the host;
//   3. ONE group on the communication streams: rank r ncclSend()s m_r keys and m_r partials, the root ncclRecv()s them
//      behind its own (and the earlier ranks') partials;
//   4. the root's kernel stream waits for its communication stream, then pcs_voxel_grid_from_partials_device over all
//      M = sum m_r partials; the voxel count is read back.
int pcs_node_process_voxel_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm,
                                  int route, int16_t* d_voxels, size_t voxels_shorts, int* n_voxels, pcs_node_voxel_stats* stats)
{
    if (!n || !d_depth || !d_color || !d_voxels || !n_voxels) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
    if (n->broken) return nfail(n, PCS_ERR_HIP, "the node's communicators were aborted after an RCCL failure: destroy it");
    if (leaf_mm < 1 || leaf_mm > 32767) return nfail(n, PCS_ERR_INVALID_ARG, "leaf_mm %d outside 1..32767", leaf_mm);
    if (route != PCS_NODE_VOXEL_PARTIALS && route != PCS_NODE_VOXEL_PAYLOADS) return nfail(n, PCS_ERR_INVALID_ARG, "unknown route %d", route);
    if (voxels_shorts < pcs_node_max_payload_shorts(n))
        return nfail(n, PCS_ERR_CAPACITY, "voxel buffer holds %zu shorts; the worst case (every point its own voxel) needs %zu",
                     voxels_shorts, pcs_node_max_payload_shorts(n));
    if (n->inflight[0].busy || n->inflight[1].busy)
        return nfail(n, PCS_ERR_INVALID_ARG, "pcs_node_wait() the frame-sets in flight before a voxel call");
    int rc = ensure_voxel_buffers(n);
    if (rc != PCS_OK) return rc;
    const int S = n->per_dev;
    HIPCHK(n, hipSetDevice(n->dev[0]));
    HIPCHK(n, hipEventRecord(n->ev_v[0], kstream(n, 0)));
    int64_t reduced = 0, exchanged = 0;

    if (route == PCS_NODE_VOXEL_PAYLOADS) {
        // the reference's shape: camera-order concatenation of the (compacted) payloads on the root, downsample there
        const size_t max_sh = pcs_node_max_payload_shorts(n);
        if (!n->d_stitched) {
            PCSCHK(n, n->ctx[0], pcs_device_malloc(n->ctx[0], &n->d_stitched, max_sh * sizeof(int16_t) + 64));
            n->stitched_cap_shorts = max_sh;
        }
        int ticket = -1, total = 0;
        rc = pcs_node_submit_device(n, d_depth, d_color, static_cast<int16_t*>(n->d_stitched), n->stitched_cap_shorts, &ticket);
        if (rc != PCS_OK) return rc;
        HIPCHK(n, hipSetDevice(n->dev[0]));
        HIPCHK(n, hipEventRecord(n->ev_v[1], kstream(n, 0)));
        rc = pcs_node_wait(n, ticket, nullptr, &total);
        if (rc != PCS_OK) return rc;
        HIPCHK(n, hipSetDevice(n->dev[0]));
        HIPCHK(n, hipEventRecord(n->ev_v[2], n->comm_stream[0]));
        HIPCHK(n, hipStreamWaitEvent(kstream(n, 0), n->ev_v[2], 0));
        PCSCHK(n, n->ctx[0], pcs_voxel_grid_device(n->ctx[0], static_cast<const int16_t*>(n->d_stitched), total, leaf_mm, d_voxels,
                                                   voxels_shorts, static_cast<int32_t*>(n->d_vox_n)));
        reduced = total;
        exchanged = ((int64_t)total - (int64_t)n->inflight[ticket & 1].cnt[0][S]) * PCS_POINT_BYTES;
    } else {
        // 1. pre-aggregation on every GPU
        for (int r = 0; r < n->n_dev; r++) {
            HIPCHK(n, hipSetDevice(n->dev[r]));
            PCSCHK(n, n->ctx[r], pcs_process_frames_voxel_partials_device(
                                     n->ctx[r], d_depth + (size_t)r * S, d_color + (size_t)r * S, leaf_mm,
                                     static_cast<uint64_t*>(n->d_vkeys[r]), static_cast<pcs_voxel_partial*>(n->d_vparts[r]),
                                     r == 0 ? n->vcap_total : n->vcap[r], static_cast<int32_t*>(n->d_vcount[r])));
            HIPCHK(n, hipMemcpyAsync(n->h_vcount + r, n->d_vcount[r], sizeof(int32_t), hipMemcpyDeviceToHost, kstream(n, r)));
            if (r == 0) HIPCHK(n, hipEventRecord(n->ev_v[1], kstream(n, 0)));
        }
        // 2. counts (one wait per GPU; the copies were all in flight)
        for (int r = 0; r < n->n_dev; r++) {
            HIPCHK(n, hipSetDevice(n->dev[r]));
            HIPCHK(n, hipStreamSynchronize(kstream(n, r)));
            if (n->h_vcount[r] < 0 || (size_t)n->h_vcount[r] > n->vcap[r])
                return nfail(n, PCS_ERR_HIP, "device %d reported %d partials (capacity %zu)", n->dev[r], n->h_vcount[r], n->vcap[r]);
        }
        // 3. one group: keys behind keys, partials behind partials
        size_t off = (size_t)n->h_vcount[0];
        std::vector<Xfer> xs;
        for (int r = 1; r < n->n_dev; r++) {
            const size_t m = (size_t)n->h_vcount[r];
            xs.push_back(Xfer{r, n->d_vkeys[r], static_cast<uint64_t*>(n->d_vkeys[0]) + off, m * sizeof(uint64_t)});
            xs.push_back(Xfer{r, n->d_vparts[r], static_cast<pcs_voxel_partial*>(n->d_vparts[0]) + off, m * sizeof(pcs_voxel_partial)});
            off += m;
            exchanged += (int64_t)m * PCS_VOXEL_PARTIAL_WIRE_BYTES;
        }
        rc = run_exchange(n, xs);           // (the kernel streams are idle: step 2 synchronised them)
        if (rc != PCS_OK) return rc;
        // 4. the root reduces everybody's partials
        HIPCHK(n, hipSetDevice(n->dev[0]));
        HIPCHK(n, hipEventRecord(n->ev_v[2], n->comm_stream[0]));
        HIPCHK(n, hipStreamWaitEvent(kstream(n, 0), n->ev_v[2], 0));
        PCSCHK(n, n->ctx[0], pcs_voxel_grid_from_partials_device(n->ctx[0], static_cast<const uint64_t*>(n->d_vkeys[0]),
                                                                 static_cast<const pcs_voxel_partial*>(n->d_vparts[0]), (int)off, nullptr,
                                                                 leaf_mm, d_voxels, voxels_shorts, static_cast<int32_t*>(n->d_vox_n)));
        reduced = (int64_t)off;
    }
    HIPCHK(n, hipSetDevice(n->dev[0]));
    HIPCHK(n, hipEventRecord(n->ev_v[3], kstream(n, 0)));
    int32_t nv = 0;
    PCSCHK(n, n->ctx[0], pcs_memcpy_d2h(n->ctx[0], &nv, n->d_vox_n, sizeof nv));         // synchronises the root's kernel stream
    for (int r = 1; r < n->n_dev; r++) {                                                  // the peers' sends have completed too
        HIPCHK(n, hipSetDevice(n->dev[r]));
        HIPCHK(n, hipStreamSynchronize(n->comm_stream[r]));
    }
    HIPCHK(n, hipSetDevice(n->dev[0]));
    *n_voxels = nv;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        (void)hipEventElapsedTime(&stats->kernels_ms, n->ev_v[0], n->ev_v[1]);
        (void)hipEventElapsedTime(&stats->exchange_ms, n->ev_v[1], n->ev_v[2]);
        (void)hipEventElapsedTime(&stats->root_voxel_ms, n->ev_v[2], n->ev_v[3]);
        stats->exchanged_bytes = exchanged; stats->partials = (int32_t)reduced; stats->voxels = nv;
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
    HIPCHK(n, hipSetDevice(n->dev[0]));
    const size_t max_sh = pcs_node_max_payload_shorts(n);
    if (!n->d_vox_out) PCSCHK(n, n->ctx[0], pcs_device_malloc(n->ctx[0], &n->d_vox_out, max_sh * sizeof(int16_t) + 64));
    int nv = 0;
    rc = pcs_node_process_voxel_device(n, dd.data(), dc.data(), leaf_mm, route, static_cast<int16_t*>(n->d_vox_out), max_sh, &nv, stats);
    if (rc != PCS_OK) return rc;
    if (out_shorts < PCS_HEADER_SHORTS + (size_t)nv * PCS_POINT_SHORTS)
        return nfail(n, PCS_ERR_CAPACITY, "output holds %zu shorts, %zu needed", out_shorts, PCS_HEADER_SHORTS + (size_t)nv * PCS_POINT_SHORTS);
    HIPCHK(n, hipSetDevice(n->dev[0]));
    const int32_t size = (int32_t)((size_t)nv * PCS_POINT_BYTES);
    if (size) PCSCHK(n, n->ctx[0], pcs_memcpy_d2h(n->ctx[0], out + PCS_HEADER_SHORTS, n->d_vox_out, (size_t)size));
    if (write_header) std::memcpy(out, &size, sizeof size);
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

}  // extern "C"
