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
    std::vector<std::vector<void*>> d_depth, d_color;   // staging for the host form, per global stream
    std::vector<pcs_stream_config> cfg;
    void* d_stitched = nullptr; size_t stitched_cap_shorts = 0;
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
#define NCCLCHK(n, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return nfail((n), PCS_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); } while (0)
#define HIPCHK(n, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return nfail((n), PCS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define PCSCHK(n, c, expr) do { int rc_ = (expr); if (rc_ != PCS_OK) \
    return nfail((n), rc_, "%s: %s", #expr, pcs_last_error(c)); } while (0)
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
        if (r == 0 && n->d_stitched) pcs_device_free(n->ctx[0], n->d_stitched);
    }
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
    if (n_devices < 1 || !device_ids || streams_per_device < 1 || !streams)
        return nfail(nullptr, PCS_ERR_INVALID_ARG, "bad arguments");
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
    if (n_devices > 1) {       // one communicator per GPU, all in this process
        n->comm.assign(n_devices, nullptr);
        ncclResult_t r = ncclCommInitAll(n->comm.data(), n_devices, device_ids);
        if (r != ncclSuccess) { int e = nfail(nullptr, PCS_ERR_HIP, "ncclCommInitAll: %s", ncclGetErrorString(r)); pcs_node_destroy(n); return e; }
    }
    *out = n;
    return PCS_OK;
}

// Pipelined device form. submit: every GPU packs its cameras on its kernel stream (the root straight into the head of
// the stitched buffer), then ONE grouped exchange — rank r ncclSend()s its payload, the root ncclRecv()s it at its
// camera-order offset — runs on the GPUs' communication streams, so the kernels of the NEXT frame-set (other payload
// slot) overlap it. Without a predicate the counts are the configuration's and nothing is read back; with one the
// per-GPU counts must reach the host before the exchange can be sized (one synchronisation per GPU inside submit).
int pcs_node_submit_device(pcs_node* n, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                           int16_t* d_stitched, size_t stitched_shorts, int* ticket)
{
    if (!n || !d_depth || !d_color || !d_stitched || !ticket) return nfail(n, PCS_ERR_INVALID_ARG, "NULL pointer");
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
        hipStream_t ks = static_cast<hipStream_t>(pcs_get_stream(n->ctx[r]));
        HIPCHK(n, hipStreamWaitEvent(ks, n->drained[slot][r], 0));
        int16_t* dst = r == 0 ? d_stitched : static_cast<int16_t*>(n->d_payload[slot][r]);
        PCSCHK(n, n->ctx[r], pcs_process_frames_device(n->ctx[r], d_depth + (size_t)r * S, d_color + (size_t)r * S, dst,
                                                       r == 0 ? stitched_shorts : n->payload_shorts[r],
                                                       n->pred ? static_cast<int32_t*>(n->d_counts[r]) : nullptr));
    }
    // 2. counts: known from the configuration unless a predicate makes them data dependent
    for (int r = 0; r < n->n_dev; r++) {
        if (n->pred) {
            HIPCHK(n, hipSetDevice(n->dev[r]));
            PCSCHK(n, n->ctx[r], pcs_memcpy_d2h(n->ctx[r], tk.cnt[r].data(), n->d_counts[r], sizeof(int32_t) * (S + 1)));   // synchronises ctx r
        } else {
            int64_t tot = 0;
            for (int k = 0; k < S; k++) {
                tk.cnt[r][k] = (pcs_stream_points(n->ctx[r], k) + n->downsample - 1) / n->downsample;
                tot += tk.cnt[r][k];
            }
            tk.cnt[r][S] = (int32_t)tot;
        }
    }
    // 3. the exchange, on the communication streams, behind each GPU's kernel
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        HIPCHK(n, hipEventRecord(n->packed[slot][r], static_cast<hipStream_t>(pcs_get_stream(n->ctx[r]))));
        HIPCHK(n, hipStreamWaitEvent(n->comm_stream[r], n->packed[slot][r], 0));
    }
    size_t off = (size_t)tk.cnt[0][S];          // points
    if (n->n_dev > 1) {
        NCCLCHK(n, ncclGroupStart());
        for (int r = 1; r < n->n_dev; r++) {
            const size_t bytes = (size_t)tk.cnt[r][S] * PCS_POINT_BYTES;
            if (bytes) {
                NCCLCHK(n, ncclSend(n->d_payload[slot][r], bytes, ncclInt8, 0, n->comm[r], n->comm_stream[r]));
                NCCLCHK(n, ncclRecv(reinterpret_cast<int8_t*>(d_stitched) + off * PCS_POINT_BYTES, bytes, ncclInt8, r, n->comm[0],
                                    n->comm_stream[0]));
            }
            off += (size_t)tk.cnt[r][S];
        }
        NCCLCHK(n, ncclGroupEnd());
    }
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        HIPCHK(n, hipEventRecord(n->drained[slot][r], n->comm_stream[r]));
    }
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
    const int S = n->per_dev;
    const size_t max_sh = pcs_node_max_payload_shorts(n);
    if (stitched_shorts < PCS_HEADER_SHORTS + max_sh)
        return nfail(n, PCS_ERR_CAPACITY, "stitched buffer holds %zu shorts, %zu needed", stitched_shorts, PCS_HEADER_SHORTS + max_sh);
    std::vector<const uint16_t*> dd(n->n_streams);
    std::vector<const uint8_t*> dc(n->n_streams);
    for (int r = 0; r < n->n_dev; r++) {
        HIPCHK(n, hipSetDevice(n->dev[r]));
        for (int k = 0; k < S; k++) {
            const int g = r * S + k;
            const pcs_stream_config& sc = n->cfg[g];
            const size_t db = (size_t)sc.depth.width * sc.depth.height * sizeof(uint16_t);
            const size_t cb = (size_t)sc.color_stride * sc.color.height;
            if (!n->d_depth[r][k]) PCSCHK(n, n->ctx[r], pcs_device_malloc(n->ctx[r], &n->d_depth[r][k], db + 16));
            if (!n->d_color[r][k]) PCSCHK(n, n->ctx[r], pcs_device_malloc(n->ctx[r], &n->d_color[r][k], cb + 16));
            PCSCHK(n, n->ctx[r], pcs_memcpy_h2d(n->ctx[r], n->d_depth[r][k], depth[g], db));
            PCSCHK(n, n->ctx[r], pcs_memcpy_h2d(n->ctx[r], n->d_color[r][k], color[g], cb));
            dd[g] = static_cast<const uint16_t*>(n->d_depth[r][k]);
            dc[g] = static_cast<const uint8_t*>(n->d_color[r][k]);
        }
    }
    HIPCHK(n, hipSetDevice(n->dev[0]));
    if (!n->d_stitched) {
        PCSCHK(n, n->ctx[0], pcs_device_malloc(n->ctx[0], &n->d_stitched, max_sh * sizeof(int16_t) + 64));
        n->stitched_cap_shorts = max_sh;
    }
    int total = 0;
    int rc = pcs_node_process_device(n, dd.data(), dc.data(), static_cast<int16_t*>(n->d_stitched), n->stitched_cap_shorts,
                                     points_per_stream, &total);
    if (rc != PCS_OK) return rc;
    HIPCHK(n, hipSetDevice(n->dev[0]));
    const int32_t size = (int32_t)((size_t)total * PCS_POINT_BYTES);
    if (size) PCSCHK(n, n->ctx[0], pcs_memcpy_d2h(n->ctx[0], stitched + PCS_HEADER_SHORTS, n->d_stitched, (size_t)size));
    if (write_header) std::memcpy(stitched, &size, sizeof size);
    if (out_size_bytes) *out_size_bytes = size;
    return PCS_OK;
}

}  // extern "C"
