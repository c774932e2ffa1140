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
 * This is the single-process form (ncclCommInitAll). The one-process-per-GPU form used by bench.py lives
 * in pointcloud_stitching_amd/stitch.py (torch.distributed / RCCL).
 */
#ifndef PCS_NODE_H
#define PCS_NODE_H

#include "pcs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcs_node pcs_node;

/* Streams [r*streams_per_device, (r+1)*streams_per_device) belong to device_ids[r]; device_ids[0] is the root.
 * `streams` has n_devices*streams_per_device entries in global camera order. flags/downsample as pcs_config. */
int  pcs_node_create(pcs_node** out, int n_devices, const int* device_ids, int streams_per_device,
                     const pcs_stream_config* streams, uint32_t flags, int downsample);
void pcs_node_destroy(pcs_node* node);
const char* pcs_node_last_error(const pcs_node* node);      /* NULL node: error of the last failed create */

int  pcs_node_devices(const pcs_node* node);
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
 * frame-sets need different stitched buffers. Without CUTOFF / DROP_INVALID nothing is read back from the GPUs;
 * with a predicate submit synchronises each GPU once (the exchange is sized by the data dependent counts).
 * PCS_ERR_CAPACITY from submit = two frame-sets already in flight. pcs_node_process_device = submit + wait.     */
int  pcs_node_submit_device(pcs_node* node, const uint16_t* const* d_depth, const uint8_t* const* d_color,
                            int16_t* d_stitched_payload_root, size_t stitched_shorts, int* ticket);
int  pcs_node_wait(pcs_node* node, int ticket, int* points_per_stream, int* total_points);

#ifdef __cplusplus
}
#endif
#endif
