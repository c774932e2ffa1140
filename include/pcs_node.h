/*
 * pcs_node.h — C ABI of libpcs_node.so: ONE process driving several MI355X GPUs of a node.
 *
 * The reference spreads cameras over edge machines and pulls their packed payloads to a central box over
 * TCP (src/pcs-camera-optimized.cpp:715-720 -> src/pcs-multicamera-client.cpp:363-409). On an 8-GPU node
 * the same star becomes: camera streams sharded over the GPUs in camera order, the fused kernel
 * (libpcs_hip) on every GPU, then ONE grouped RCCL exchange over xGMI — every non-root GPU ncclSend()s its
 * payload, the root ncclRecv()s each straight into its slice of the stitched buffer (sendStitchToUnity's
 * camera-order concatenation, :385-395). The root's own slice is written in place by its kernel.
 *
 * This is the single-process form (ncclCommInitAll), the one `bench.py --gpus N` measures by default (--route node).
 * The one-process-per-GPU form lives in pointcloud_stitching_amd/stitch.py (torch.distributed / RCCL; --route ranks).
 *
 * Peers and GPUs. An entry of device_ids is a PEER: it owns streams_per_device cameras, a libpcs_hip context and two
 * payload slots. Distinct device ids are the GPUs: each owns one RCCL communicator rank and one communication stream.
 * A device id may REPEAT: the peers that share it are virtual peers of one GPU, and their transfers to the root become
 * RCCL self send/recv pairs on that GPU's communicator. That is how a one-GPU box runs every N > 1 code path — offsets,
 * slots, events, the grouped ncclSend/ncclRecv — on real RCCL (tests/test_node.py, bench.py --node-devices 0,0).
 *
 * Threading: a node is driven by ONE host thread at a time (it switches the thread's current device as it goes and keeps
 * per-ticket state without locks); different nodes may be driven by different threads.
 *
 * Status: every path runs in the test-suite on ONE GPU (virtual peers, RCCL self exchange). A run over several physical
 * GPUs is what bench.py --gpus N produces; none is on record yet (the development boxes have one GPU).
 */
#ifndef PCS_NODE_H
#define PCS_NODE_H

#include "pcs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcs_node pcs_node;

/* Streams [r*streams_per_device, (r+1)*streams_per_device) belong to device_ids[r]; device_ids[0] is the root.
 * `streams` has n_devices*streams_per_device entries in global camera order. flags/downsample as pcs_config.
 * Like pcs_create, refuses (PCS_ERR_INVALID_ARG, before any device is touched) a configuration whose stitched payload
 * would not fit the wire format's int32 byte count (more than 214 748 364 points in all). */
int  pcs_node_create(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                     const pcs_stream_config* streams, uint32_t flags, int downsample);
/* The same with node flags. PCS_NODE_NO_EXCHANGE: no communicator is created and nothing is gathered — every peer packs its
 * own cameras and only the root's slice of the stitched buffer is written (the counts are still the whole node's). A
 * diagnostic ("what do the kernels alone sustain") and bench.py's fall-back when RCCL refuses to come up.              */
#define PCS_NODE_NO_EXCHANGE 0x1u
/* PCS_NODE_DIRECT_STORE: frame-sets whose counts follow from the configuration (no CUTOFF / DROP_INVALID) are gathered by the
 * pack kernels themselves — every peer's kernel writes its records straight into its camera-order slice of the ROOT GPU's
 * stitched buffer over xGMI (peer access), so there is no exchange step, no RCCL kernel and no second copy of the payload.
 * Same bytes. Tickets under a predicate, and the voxel route, still use the grouped RCCL exchange. pcs_node_create_ex fails
 * with PCS_ERR_UNSUPPORTED if a GPU of the node cannot address the root's memory. Opt-in: xGMI favours RCCL's large
 * transfers over a kernel's 16-byte stores on some topologies — measure both (bench.py reports the pair at N > 1).        */
#define PCS_NODE_DIRECT_STORE 0x2u
int  pcs_node_create_ex(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                        const pcs_stream_config* streams, uint32_t flags, int downsample, uint32_t node_flags);
void pcs_node_destroy(pcs_node* node);
const char* pcs_node_last_error(const pcs_node* node);      /* NULL node: error of the last failed create */

int  pcs_node_devices(const pcs_node* node);                 /* peers (entries of device_ids)                              */
int  pcs_node_rccl_ranks(const pcs_node* node);              /* size of the RCCL communicator (distinct GPUs); 0 = none    */
size_t pcs_node_max_payload_shorts(const pcs_node* node);

/* Host rasters in (uploaded to the owning GPUs), stitched buffer out on the host, header like pcs_process_frames. */
int  pcs_node_process(pcs_node* node, const uint16_t* const* depth, const uint8_t* const* color,
                      int16_t* stitched, size_t stitched_shorts, int write_header,
                      int* points_per_stream, int* out_size_bytes);

/* Device form: rasters already on their owning GPUs; the stitched PAYLOAD (no header) is left on the root GPU.
 * Synchronous on return (the exchange has completed). */
int  pcs_node_process_device(pcs_node* node, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                             int16_t* d_stitched_payload_root, size_t stitched_shorts,
                             int* points_per_stream, int* total_points);

/* Pipelined device form: up to two frame-sets in flight. pcs_node_submit_device enqueues the kernels of one frame-set on
 * every GPU's kernel stream and its exchange on per-GPU communication streams, then returns a ticket; pcs_node_wait
 * blocks until that frame-set's stitched payload is complete on the root. Writing the loop as  submit(k+1); wait(k);
 * overlaps the xGMI exchange of frame-set k with the kernels of k+1 (each GPU keeps two payload buffers). The two
 * frame-sets need different stitched buffers. Without CUTOFF / DROP_INVALID nothing is read back from the GPUs and the
 * exchange is enqueued by submit itself. With a predicate the exchange is sized by data-dependent counts: submit enqueues
 * the kernels and an asynchronous read-back of the counts and RETURNS; the exchange is enqueued by the next submit (after
 * that frame-set's kernels are queued, when the counts have long landed) or by pcs_node_wait, whichever comes first — the
 * host never waits for a kernel it has just enqueued.
 * PCS_ERR_CAPACITY from submit = two frame-sets already in flight. pcs_node_process_device = submit + wait.
 * A submit that fails after some GPUs' kernels were enqueued leaves no ticket behind and the node usable: submit again.
 * An RCCL failure inside the exchange closes the group, aborts the communicators and leaves the node unusable
 * (every later call fails with PCS_ERR_HIP); pcs_node_wait never blocks on a failed exchange.                    */
int  pcs_node_submit_device(pcs_node* node, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                            int16_t* d_stitched_payload_root, size_t stitched_shorts, int* ticket);
int  pcs_node_wait(pcs_node* node, int ticket, int* points_per_stream, int* total_points);
/* Fault injection for the test-suite: the NEXT grouped exchange fails as if ncclGroupStart had (communicators aborted, node
 * unusable) — the only way to walk the abort path on a healthy box. */
int  pcs_node_inject_exchange_failure(pcs_node* node);

/* Where a frame-set's time went, from HIP events on the ROOT GPU (pcs_node_set_timing(node, 1) before the submit; the
 * events cost a few host microseconds per submit, hence opt-in). pcs_node_last_stats returns the figures of the ticket most
 * recently waited for; exchanged_bytes / reduced are filled with or without timing.                                     */
typedef struct pcs_node_stats {
    int32_t  ticket;
    float    kernels_ms;         /* root kernel stream: this submit's first enqueue -> the root's own kernels done            */
    float    exchange_ms;        /* root communication stream: group enqueued (all its peers' kernels done) -> payloads landed */
    float    root_ms;            /* voxel tickets: the root's sort + segmented mean                                            */
    int64_t  exchanged_bytes;    /* bytes the grouped RCCL exchange moved to the root (0 for a direct-store ticket)            */
    int64_t  reduced;            /* points in the stitched cloud (stitch tickets) / partials the root reduced (voxel tickets)  */
    int64_t  direct_bytes;       /* PCS_NODE_DIRECT_STORE tickets: bytes the peers' own kernels stored into the root's buffer  */
    float    submit_host_ms;     /* HOST time the submit spent enqueueing every peer's kernels (filled with or without timing) */
    float    exchange_host_ms;   /* HOST time spent enqueueing this ticket's exchange (+ the root's reduce for a voxel ticket) */
} pcs_node_stats;
/* The setting is latched into a ticket when it is SUBMITTED: changing it while a frame-set is in flight affects later ones only. */
int  pcs_node_set_timing(pcs_node* node, int enable);
int  pcs_node_last_stats(const pcs_node* node, pcs_node_stats* out);

/* Which RCCL answered. libpcs_node is compiled against /opt/rocm's rccl.h; the dynamic loader binds the librccl.so.1 the
 * process loaded first (under Python that is the one bundled with torch). pcs_node_rccl_version = ncclGetVersion() of the
 * bound library (0: the node has no communicator), pcs_node_rccl_header_version = NCCL_VERSION_CODE it was compiled for,
 * pcs_node_rccl_library = the bound library's path. pcs_node_create_ex refuses (PCS_ERR_UNSUPPORTED) a different MAJOR version. */
int  pcs_node_rccl_version(const pcs_node* node);
int  pcs_node_rccl_header_version(void);
const char* pcs_node_rccl_library(const pcs_node* node);

/* What connects peer `peer`'s GPU to the root GPU (hipDeviceCanAccessPeer, hipDeviceGetP2PAttribute,
 * hipExtGetLinkTypeAndHopCount); -1 where the runtime gave no answer. link_type follows HSA_AMD_LINK_INFO_TYPE_*
 * (1 = QPI, 2 = PCIe, 3 = InfiniBand, 4 = xGMI). */
typedef struct pcs_node_link {
    int32_t device, root_device;
    int32_t same_device;         /* a virtual peer of the root's GPU: no link involved                         */
    int32_t can_access_root;     /* hipDeviceCanAccessPeer(device -> root): what PCS_NODE_DIRECT_STORE needs   */
    int32_t link_type, hops;     /* hipExtGetLinkTypeAndHopCount                                               */
    int32_t performance_rank;    /* hipDevP2PAttrPerformanceRank                                               */
    int32_t native_atomics;      /* hipDevP2PAttrNativeAtomicSupported                                         */
} pcs_node_link;
int  pcs_node_link_info(pcs_node* node, int peer, pcs_node_link* out);
/* Per-peer transfer time into the root, one peer at a time: `bytes` (0 or too large = the smallest peer payload capacity) from
 * the peer's payload slot into a scratch buffer on the root as a group of ONE ncclSend/ncclRecv pair, between an event pair on
 * the root's communication stream; mean of `repeats` after one untimed transfer. ms_per_peer[n_peers], entry 0 (the root) = 0.
 * The node must be idle (no ticket in flight). Without a communicator every entry is 0. A frame-set's grouped exchange is one
 * RCCL operation for all pairs, so this is the only way to see a single link. */
int  pcs_node_probe_links(pcs_node* node, size_t bytes, int repeats, float* ms_per_peer);

/* ---- BASELINE configs[4]: voxel-grid downsample of the cloud the node's cameras stitch to ---------------------------- *
 * 16 x 1920x1080 streams, 2 per GPU, invalid-depth compaction, voxel grid of the stitched cloud on the root. Two routes, the
 * SAME bytes (the voxel sums are integers, so neither the order of the points nor where they were pre-summed matters):
 *   PCS_NODE_VOXEL_PARTIALS  every GPU pre-aggregates its own cameras into voxel partials
 *                            (pcs_process_frames_voxel_partials_device), ONE grouped exchange moves the partials to the
 *                            root (40 B per occupied voxel and pixel patch — 16 x 1080p at 50 mm: ~40 MB instead of the
 *                            298 MB of packed points), one sort + segmented mean there
 *                            (pcs_voxel_grid_from_partials_device). The default.
 *   PCS_NODE_VOXEL_PAYLOADS  the literal shape of src/pcs-multicamera-client.cpp:373-409 + a downsample on the centre
 *                            (src/pcs-multicamera-optimized.cpp:226-248): the (compacted) payloads are gathered to the root
 *                            in camera order as pcs_node_process_device does, then pcs_voxel_grid_device on the stitched cloud.
 * Device form: rasters on their owning GPUs, the voxel cloud (records, no header) left on the root GPU; *n_voxels on the
 * host. Synchronous on return. d_voxels_root needs room for every pixel of the node in the worst case
 * (pcs_node_max_payload_shorts with downsample 1). `stats` (optional) receives how the call spent its time.            */
#define PCS_NODE_VOXEL_PARTIALS 0
#define PCS_NODE_VOXEL_PAYLOADS 1
typedef struct pcs_node_voxel_stats {
    float    kernels_ms;         /* root GPU: start of the call -> its own pre-aggregation (or pack) kernel done           */
    float    exchange_ms;        /* root GPU: -> every peer's partials (payload) received (includes waiting for the peers' kernels) */
    float    root_voxel_ms;      /* root GPU: -> sort + segmented mean done                                                  */
    int64_t  exchanged_bytes;    /* bytes the peers sent to the root                                                          */
    int32_t  partials;           /* partials (route PARTIALS) or points (route PAYLOADS) the root reduced                    */
    int32_t  voxels;
} pcs_node_voxel_stats;
int  pcs_node_process_voxel_device(pcs_node* node, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                                   int leaf_mm, int route, int16_t* d_voxels_root, size_t voxels_shorts, int* n_voxels,
                                   pcs_node_voxel_stats* stats);
/* Pipelined form of route PARTIALS, two frame-sets in flight like pcs_node_submit_device (the two kinds of ticket share the
 * two slots): submit enqueues every GPU's pre-aggregation and the read-back of its partial count and returns; the exchange
 * (sized by those counts) and, behind it on the root, the sort + segmented mean are enqueued by the next submit or by
 * pcs_node_wait_voxel. Written as  submit(k+1); wait(k);  the pre-aggregation of k+1 overlaps the exchange and the root's
 * sort of k. The two frame-sets need different d_voxels_root buffers. pcs_node_process_voxel_device(PARTIALS) = submit + wait.
 * A node of ONE peer has nothing to exchange: submit enqueues pcs_process_frames_voxel_device (rasters -> voxel cloud, two launches
 * on a warm context) and wait only waits; stats report partials = 0 (they never leave the library's workspace), and such a node
 * never allocates the partials pipeline's arrays. Its two slots run on TWO contexts of the peer used in turn — each with its own
 * stream, workspace, splitters and regions (the splitters a frame-set partitions by are then two frame-sets old) — so that the bucket
 * tail of frame-set k, a latency chain that leaves the chip almost empty, runs beside the pre-aggregation of k+1: 16 x 1080p at 50 mm
 * 0.155 ms per frame-set instead of 0.174. pcs_node_set_one_call(node, mode) with nothing in flight, or PCS_NODE_ONE_CALL=<mode> in
 * the environment WHEN THE NODE IS CREATED: 2 (default) as described, 1 one context (frame-sets queue behind each other), 0 the
 * partials pipeline of a node of several peers (its stats then read like theirs: partials / reduced > 0, root_ms > 0). A/B; tests.  */
int  pcs_node_submit_voxel_device(pcs_node* node, const uint16_t* const* d_depth, const uint8_t* const* d_color, int leaf_mm,
                                  int16_t* d_voxels_root, size_t voxels_shorts, int* ticket);
int  pcs_node_wait_voxel(pcs_node* node, int ticket, int* n_voxels);
int  pcs_node_set_one_call(pcs_node* node, int mode);
/* Several peers that ALL share one GPU (a device id that repeats throughout — the development boxes' way to run the N > 1 flow, or more
 * cameras than one context is configured for): nothing needs to travel, so by default a voxel ticket of such a node goes through a
 * SINK (include/pcs_hip.h: pcs_voxel_sink_*): every peer's pre-aggregation, on the peer's own context and stream, writes its partials
 * straight into the workspace of a sink context of that GPU — its buckets' regions on a warm call — and the one-launch tail follows on
 * the sink's stream behind all of them; two sinks used in turn (slot 0 / 1) let the tail of frame-set k run beside the pre-aggregations
 * of k+1. No exchange, no concatenation, no placement: 16 x 1080p at 50 mm, 8 peers of one GPU, 0.40 -> 0.23 ms per frame-set (the peers
 * dealt onto PCS_NODE_SINK_STREAMS = 2 kernel streams: 1 -> 0.26, 8 -> 0.24 ms, where the host thread pays an event per peer). Stats
 * read as a one-call ticket's (exchanged_bytes = 0, partials = 0). Peers on DIFFERENT GPUs always exchange partials (the pre-aggregation's
 * atomics are device-scope). pcs_node_set_voxel_sink(node, 0) with nothing in flight, or PCS_NODE_VOXEL_SINK=0 in the environment when the
 * node is created, keeps the partials exchange (RCCL self send/recv) on such a node — the route it exists to exercise; the tests run both.
 * pcs_node_voxel_sink: 1 when the next voxel ticket takes the sink.                                                                       */
int  pcs_node_set_voxel_sink(pcs_node* node, int on);
int  pcs_node_voxel_sink(const pcs_node* node);
/* A voxel frame-set whose bucket tail ended flagged (device count -1: include/pcs_hip.h, pcs_voxel_grid_device) is run again by
 * the wait that finds it, on the LSD tail, which is then latched for that context; *n_voxels is never negative, and PCS_ERR_HIP is
 * returned if the second run is flagged too. For that the rasters handed to pcs_node_submit_voxel_device stay the caller's to
 * keep valid until the ticket's wait has returned. This counts the frame-sets that were run again (0 on a healthy device).    */
int  pcs_node_voxel_reruns(const pcs_node* node);
/* Host form: rasters uploaded to their owning GPUs, the voxel cloud downloaded into `out` with the wire header like
 * pcs_node_process ([int32 bytes][records] when write_header; records always start at out + 2 shorts).              */
int  pcs_node_process_voxel(pcs_node* node, const uint16_t* const* depth, const uint8_t* const* color, int leaf_mm, int route,
                            int16_t* out, size_t out_shorts, int write_header, int* out_size_bytes,
                            pcs_node_voxel_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
