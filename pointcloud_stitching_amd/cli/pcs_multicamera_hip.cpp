// pcs-multicamera-optimized — the central stitcher (src/pcs-multicamera-optimized.cpp / pcs-multicamera-client.cpp,
// non-visual path) on MI355X. Installed under the reference's program name; `pcs-multicamera-hip` is a link to it.
//
// Two ways to feed it:
//   -i synth:<W>x<H> | frames.pcsraw   all cameras' rasters are on this node: ONE fused launch produces the
//                                       stitched buffer (edge + central collapsed)
//   -c host:port[,host:port...]         existing edge servers (the reference's, or pcs-camera-optimized -s):
//                                       one reader thread per camera like readCloud (:363-371); payloads are
//                                       concatenated on the GPU with the stride -d (sendStitchToUnity :373-395)
// The stitched cloud is served exactly like the reference: wait for 'Z' on port 9000, write
// [int32 bytes][points] (:397-403). -t prints the running average like runStitching (:417-430).
//
//   reference flags (getopt "hftsvd:n", src/pcs-multicamera-optimized.cpp:90): -h -f -t -s -v -d <n> -n
//     -f is the reference's boolean "fast" switch (:96-98; it only dropped colour in the PCL viewer): accepted, inert,
//     with a one-line notice — an existing `-f -t -d2` invocation keeps working. -s (save PLY), -v (PCL visualiser)
//     and -n belong to the PCL viewer, which this build does not have: refused with a reason.
//   additions: -i <src>  -c <list>  -N <streams>  -g <gpu>  -p <serve port>  -r <frame-sets>  -o <file>  -q (no server)
//              -G <n>  shard the cameras of -i over n GPUs: libpcs_node (ncclCommInitAll + one grouped send/recv to GPU 0)
//              -G <id,id,...>  the same with explicit device ids, one per peer; an id that repeats makes virtual peers of one GPU
//                      (their transfers are RCCL self send/recv pairs): `-G 0,0,0,0` runs the 4-GPU flow on a one-GPU box
//              -V <mm> serve the voxel-grid downsample (leaf in mm, BASELINE config 5) of the stitched cloud instead of the
//                      cloud itself: with -i one device call from the rasters (pcs_process_frames_voxel_device), with -c the
//                      voxel grid of the concatenated payloads;  -Z  drop invalid-depth pixels (PCS_FLAG_DROP_INVALID, -i only)
//              -G <n> -V <mm>  BASELINE configs[4]: cameras sharded over n GPUs, per-GPU voxel partials, one RCCL exchange, sort +
//                      segmented mean on GPU 0 (libpcs_node: pcs_node_process_voxel); -R payloads gathers the packed payloads instead
//              -P      (with -G and -i synth:) the frame loop of a node whose rasters are already on their GPUs: a ring of synthetic
//                      frame-sets is uploaded once, then pcs_node_submit_device(k+1); pcs_node_wait(k) (with -V: the voxel
//                      tickets) keeps two frame-sets in flight for -r iterations; prints the period per frame-set and where a
//                      frame-set's time went on GPU 0 (pcs_node_last_stats). -o dumps the last frame-set like the other modes.
//              -T <file>  (with -c) what the reference's pcs-multicamera-optimized does and pcs-multicamera-client does not: every
//                      camera's payload is decoded to metres, moved by transform[i] (line i of the file: 16 values, row-major,
//                      the format of pcs-camera-optimized -e) and re-encoded before the concatenation
//                      (src/pcs-multicamera-optimized.cpp:226-265, 289; pcs_transform_payloads_device). Lossy, like the
//                      reference's round trip; default off = pcs-multicamera-client's lossless concatenation.
//     with neither -i nor -c the cameras are 8 synthetic 1280x720 streams on this node (there are no live cameras here).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <functional>
#include <vector>

#include <getopt.h>
#include <signal.h>

#include "pcs_synth.h"
#include "pcs_wire.h"
#include "../../include/pcs_node.h"

typedef std::chrono::high_resolution_clock clockTime;
typedef std::chrono::duration<double, std::milli> timeMilli;

static bool timer = false, serve = true;
static int downsample = 1, n_streams = 8, device = 0, serve_port = 9000, max_sets = 30, n_gpus = 0, voxel_leaf = 0;
static bool drop_invalid = false, pipelined = false;
static std::vector<int> gpu_ids;
static int voxel_route = PCS_NODE_VOXEL_PARTIALS;
static const char* source = nullptr;
static const char* cameras = nullptr;
static const char* dump_path = nullptr;
static const char* transform_path = nullptr;

static void usage()
{
    std::cout << "\nMulticamera pointcloud stitching (MI355X)\nUsage: pcs-multicamera-optimized [options]\n\nOptions:\n"
              << " -h (help)        Display command line options\n"
              << " -f (fast)        Accepted for compatibility (it thinned the reference's PCL viewer); no effect\n"
              << " -t (timer)       Displays the runtime of certain functions\n"
              << " -d (downsample)  Downsamples the stitched pointcloud by the specified integer\n"
              << " -i <src>         cameras on this node: synth:<W>x<H> or frames.pcsraw (default synth:1280x720)\n"
              << " -c <list>        edge servers host:port,... (pull 'Z' protocol)\n"
              << " -N <n> streams   -g <gpu>   -p <port> (default 9000)   -r <frame-sets>   -o <file>   -q no server\n"
              << " -G <n>           shard the -i cameras over n GPUs of this node (one process, RCCL gather to GPU 0)\n"
              << " -G <id,id,...>   the same with explicit device ids per peer (a repeated id = virtual peers of one GPU)\n"
              << " -V <mm>          serve the voxel-grid downsample (leaf in millimetres) of the stitched cloud;  -Z drop invalid depth\n"
              << "                  with -G: every GPU pre-aggregates its cameras, ONE exchange of the voxel partials, reduced on GPU 0\n"
              << "                  (-R payloads: gather the packed payloads instead and downsample the stitched cloud on GPU 0)\n"
              << " -T <file>        with -c: re-transform every camera's payload by transform[i] (16 values per line) before stitching,\n"
              << "                  as the reference's pcs-multicamera-optimized does (decode, pcl::transformPointCloud, re-encode)\n"
              << " -P               with -G and -i synth:<W>x<H>: device-resident frame loop, two frame-sets in flight (submit / wait)\n"
              << " -s / -v / -n     PCL viewer features of the reference; not available in this build\n";
}

int main(int argc, char** argv)
{
    signal(SIGPIPE, SIG_IGN);
    int c;
    while ((c = getopt(argc, argv, "hftsvd:nc:N:g:p:r:o:qG:i:V:ZR:PT:")) != -1) {
        switch (c) {
            case 't': timer = true; break;
            case 'd': downsample = atoi(optarg); break;
            case 'f':      // src/pcs-multicamera-optimized.cpp:96-98
                std::cerr << "-f (fast) only thinned the reference's PCL viewer; accepted, no effect on the stitched cloud" << std::endl;
                break;
            case 'i': source = optarg; break;
            case 'c': cameras = optarg; break;
            case 'N': n_streams = atoi(optarg); break;
            case 'g': device = atoi(optarg); break;
            case 'p': serve_port = atoi(optarg); break;
            case 'r': max_sets = atoi(optarg); break;
            case 'o': dump_path = optarg; break;
            case 'q': serve = false; break;
            case 'G':
                if (strchr(optarg, ',')) {
                    for (const char* q = optarg; *q;) { gpu_ids.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; q++; }
                    n_gpus = (int)gpu_ids.size();
                } else {
                    n_gpus = atoi(optarg);
                }
                break;
            case 'V': voxel_leaf = atoi(optarg); break;
            case 'R': voxel_route = (optarg[0] == 'p' && optarg[1] == 'a' && optarg[2] == 'y') ? PCS_NODE_VOXEL_PAYLOADS : PCS_NODE_VOXEL_PARTIALS; break;
            case 'Z': drop_invalid = true; break;
            case 'P': pipelined = true; break;
            case 'T': transform_path = optarg; break;
            case 's': case 'v': case 'n':
                std::cerr << "-" << (char)c << " drives the reference's PCL viewer / PLY writer, which this build does not include" << std::endl;
                return 2;
            default: usage(); return c == 'h' ? 0 : 2;
        }
    }
    if (downsample < 1) { std::cerr << "downsample must be >= 1" << std::endl; return 2; }
    if (source && cameras) { std::cerr << "give at most one of -i <src> or -c <edge list>" << std::endl; usage(); return 2; }
    if (!source && !cameras) source = "synth:1280x720";      // no live cameras on this node: the synthetic generator
    if (voxel_leaf < 0 || voxel_leaf > 32767) { std::cerr << "-V leaf must be 1..32767 mm" << std::endl; return 2; }
    if (optind < argc) {      // a leftover positional argument: most likely the pre-round-2 `-f <src>` spelling
        std::cerr << "unexpected argument '" << argv[optind] << "': the frame source is given with -i <src> (-f is the reference's "
                     "boolean switch)" << std::endl;
        usage();
        return 2;
    }
    if (voxel_leaf && n_gpus > 0 && cameras) { std::cerr << "-G shards cameras of this node (-i), not edge servers (-c)" << std::endl; return 2; }
    if (transform_path && !cameras) { std::cerr << "-T re-transforms payloads received from edge servers (-c); cameras of this node (-i) "
                                                   "get their extrinsic in the fused kernel" << std::endl; return 2; }
    if (drop_invalid && !source) { std::cerr << "-Z applies to cameras on this node (-i); edge servers drop with their own -c" << std::endl; return 2; }

    // ---- frame source / edge connections -------------------------------------------------------
    std::vector<pcs_stream_config> cfgs;
    std::vector<int> cam_fd;
    int W = 0, H = 0;
    FILE* raw = nullptr; int raw_frames = 0;
    if (source) {
        if (strncmp(source, "synth:", 6) == 0) {
            if (sscanf(source + 6, "%dx%d", &W, &H) != 2) { std::cerr << "bad synth spec" << std::endl; return 2; }
            for (int s = 0; s < n_streams; s++) cfgs.push_back(pcs_synth::stream_config(W, H, s, false));
        } else {
            raw = fopen(source, "rb");
            char magic[8]; int32_t ns = 0;
            if (!raw || fread(magic, 1, 8, raw) != 8 || memcmp(magic, "PCSRAW1", 8) != 0 || fread(&ns, 4, 1, raw) != 1 ||
                fread(&raw_frames, 4, 1, raw) != 1 || ns < 1 || ns > PCS_MAX_STREAMS) { std::cerr << "cannot read " << source << std::endl; return 2; }
            n_streams = ns; cfgs.resize(ns);
            if (fread(cfgs.data(), sizeof(pcs_stream_config), ns, raw) != (size_t)ns) return 2;
        }
    } else {
        std::string list = cameras;
        size_t pos = 0;
        while (pos < list.size()) {
            size_t comma = list.find(',', pos);
            std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            pos = comma == std::string::npos ? list.size() : comma + 1;
            size_t colon = item.rfind(':');
            if (colon == std::string::npos) { std::cerr << "edge entry needs host:port: " << item << std::endl; return 2; }
            int fd = pcs_wire::connect_to(item.substr(0, colon).c_str(), atoi(item.c_str() + colon + 1));
            if (fd < 0) { std::cerr << "Connection failed at " << item << std::endl; return 1; }       // :201-204
            cam_fd.push_back(fd);
        }
        n_streams = (int)cam_fd.size();
        // the stitch-only context needs a config; geometry is irrelevant for pcs_stitch_device
        cfgs.push_back(pcs_synth::stream_config(64, 48, 0, false));
    }

    std::vector<pcs_payload_desc> xf;           // -T: transform[i] per camera (src/pcs-multicamera-optimized.cpp:417-455 as a file)
    if (transform_path) {
        FILE* f = fopen(transform_path, "r");
        if (!f) { std::cerr << "cannot open transform file " << transform_path << std::endl; return 2; }
        char line[1024];
        while ((int)xf.size() < n_streams && fgets(line, sizeof line, f)) {
            char* hash = strchr(line, '#');
            if (hash) *hash = 0;
            pcs_payload_desc d; memset(&d, 0, sizeof d);
            int got = 0, pos = 0, adv = 0;
            while (got < 16 && sscanf(line + pos, " %f%n", &d.transform[got], &adv) == 1) { got++; pos += adv; if (line[pos] == ',') pos++; }
            if (got == 0) continue;
            if (got != 16) { std::cerr << transform_path << ": expected 16 values per line" << std::endl; fclose(f); return 2; }
            xf.push_back(d);
        }
        fclose(f);
        if ((int)xf.size() < n_streams) { std::cerr << transform_path << ": only " << xf.size() << " matrices for " << n_streams << " cameras" << std::endl; return 2; }
    }

    pcs_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = device; cfg.n_streams = (int)cfgs.size(); cfg.streams = cfgs.data();
    cfg.downsample = source ? downsample : 1;
    cfg.flags = (source && drop_invalid) ? PCS_FLAG_DROP_INVALID : 0u;
    pcs_ctx* ctx = nullptr;
    int rc = pcs_create(&ctx, &cfg);
    if (rc != PCS_OK) { std::cerr << "pcs_create: " << pcs_strerror(rc) << ": " << pcs_last_error(nullptr) << std::endl; return 1; }

    pcs_node* node = nullptr;
    if (n_gpus > 0) {
        if (!source) { std::cerr << "-G applies to cameras on this node (-i)" << std::endl; return 2; }
        if (n_streams % n_gpus) { std::cerr << "-N " << n_streams << " streams do not divide over -G " << n_gpus << " GPUs" << std::endl; return 2; }
        std::vector<int> ids(n_gpus);
        for (int g = 0; g < n_gpus; g++) ids[g] = gpu_ids.empty() ? device + g : gpu_ids[g];
        rc = pcs_node_create(&node, n_gpus, ids.data(), n_streams / n_gpus, cfgs.data(), cfg.flags, downsample);
        if (rc != PCS_OK) { std::cerr << "pcs_node_create: " << pcs_strerror(rc) << ": " << pcs_node_last_error(nullptr) << std::endl; return 1; }
        std::cout << "Sharding " << n_streams << " cameras over " << n_gpus << " GPU(s), RCCL gather to GPU " << ids[0]
                  << " (communicator of " << pcs_node_rccl_ranks(node) << " rank(s))" << std::endl;
    }

    // ---- buffers -------------------------------------------------------------------------------
    int size_bytes = 0;
    const size_t cam_cap_bytes = (size_t)10 * 4u * 1000 * 1000;           // per-camera receive buffer (reference: 10 MB, :554)
    size_t stitched_shorts = source ? PCS_HEADER_SHORTS + pcs_max_payload_shorts(ctx)
                                    : PCS_HEADER_SHORTS + (size_t)n_streams * cam_cap_bytes / 2;
    // page-locked, like the rasters handed to pcs_process_frames below: the call then runs zero copy (the kernels read
    // the rasters and write the stitched payload over PCIe themselves)
    struct Pinned {
        pcs_ctx* ctx; int16_t* p = nullptr; size_t n = 0;
        explicit Pinned(pcs_ctx* c) : ctx(c) {}
        void release() { if (p) pcs_host_free(ctx, p); p = nullptr; n = 0; }     // before pcs_destroy
        bool resize(size_t shorts)
        {
            if (shorts <= n) return true;
            if (p) pcs_host_free(ctx, p);
            p = nullptr; n = 0;
            if (pcs_host_malloc(ctx, (void**)&p, shorts * sizeof(int16_t)) != PCS_OK) return false;
            n = shorts;
            return true;
        }
        int16_t* data() { return p; }
        size_t size() const { return n; }
    } stitched(ctx);
    if (!stitched.resize(stitched_shorts)) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
    std::vector<uint16_t*> pin_d(source ? n_streams : 0, nullptr);
    std::vector<uint8_t*> pin_c(source ? n_streams : 0, nullptr);
    std::vector<size_t> pin_db(pin_d.size(), 0), pin_cb(pin_c.size(), 0);
    std::vector<std::vector<uint16_t>> depth(source ? n_streams : 0);
    std::vector<std::vector<uint8_t>> color(source ? n_streams : 0);
    std::vector<std::vector<uint8_t>> cam_buf(source ? 0 : n_streams, std::vector<uint8_t>(source ? 0 : cam_cap_bytes));
    std::vector<void*> d_cam(source ? 0 : n_streams, nullptr);
    void* d_stitched = nullptr;
    if (!source) {
        for (int i = 0; i < n_streams; i++) if (pcs_device_malloc(ctx, &d_cam[i], cam_cap_bytes) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
        if (pcs_device_malloc(ctx, &d_stitched, (size_t)n_streams * cam_cap_bytes + 64) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
    }

    // -V: device-resident rasters (with -i) and the voxel cloud
    std::vector<void*> d_depth, d_color;
    void *d_vox = nullptr, *d_nvox = nullptr;
    size_t vox_cap_points = 0;
    if (voxel_leaf) {
        vox_cap_points = source ? pcs_max_payload_shorts(ctx) / PCS_POINT_SHORTS : (size_t)n_streams * cam_cap_bytes / PCS_POINT_BYTES;
        if (source && cfg.downsample != 1) {
            vox_cap_points = 0;
            for (auto& sc : cfgs) vox_cap_points += (size_t)sc.depth.width * sc.depth.height;
        }
        if (!node && (pcs_device_malloc(ctx, &d_vox, vox_cap_points * PCS_POINT_BYTES + 64) != PCS_OK ||
                      pcs_device_malloc(ctx, &d_nvox, 64) != PCS_OK)) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
        if (source && !node) {      // (with -G the node library stages the rasters on their owning GPUs itself)
            d_depth.resize(n_streams, nullptr); d_color.resize(n_streams, nullptr);
            for (int s = 0; s < n_streams; s++) {
                const size_t db = (size_t)cfgs[s].depth.width * cfgs[s].depth.height * 2;
                const size_t cb = (size_t)cfgs[s].color_stride * cfgs[s].color.height;
                if (pcs_device_malloc(ctx, &d_depth[s], db + 64) != PCS_OK || pcs_device_malloc(ctx, &d_color[s], cb + 64) != PCS_OK) {
                    std::cerr << pcs_last_error(ctx) << std::endl; return 1;
                }
            }
        }
        if (!stitched.resize(PCS_HEADER_SHORTS + vox_cap_points * PCS_POINT_SHORTS)) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
    }
    // the voxel cloud of whatever d_vox / d_nvox hold -> stitched (header + records). `again` repeats the device call that filled
    // them: a count of -1 (the bucket tail gave up on a stalled device, include/pcs_hip.h) is answered by one more run on the LSD
    // tail; a negative length never reaches the size arithmetic below or the wire (the reference sends `size` as it is, :394-403)
    auto fetch_voxels = [&](const std::function<int()>& again) -> bool {
        int32_t nv = 0;
        if (pcs_memcpy_d2h(ctx, &nv, d_nvox, sizeof nv) != PCS_OK || pcs_synchronize(ctx) != PCS_OK) return false;
        if (nv < 0) {
            std::cerr << "voxel grid: the bucket tail gave up waiting for a workgroup; running the frame-set again on the LSD tail" << std::endl;
            if (pcs_set_voxel_tail(ctx, PCS_VOXEL_TAIL_LSD_LATCHED) != PCS_OK || again() != PCS_OK) return false;
            if (pcs_memcpy_d2h(ctx, &nv, d_nvox, sizeof nv) != PCS_OK || pcs_synchronize(ctx) != PCS_OK) return false;
            if (nv < 0) { std::cerr << "voxel grid: negative count on the LSD tail too" << std::endl; return false; }
        }
        size_bytes = nv * PCS_POINT_BYTES;
        if (size_bytes && pcs_memcpy_d2h(ctx, stitched.data() + PCS_HEADER_SHORTS, d_vox, (size_t)size_bytes) != PCS_OK) return false;
        if (pcs_synchronize(ctx) != PCS_OK) return false;
        memcpy(stitched.data(), &size_bytes, sizeof(int));
        return true;
    };

    if (pipelined) {
        // ---- the frame loop of a node with device-resident rasters: submit(k+1); wait(k) ------------------------------------
        if (!node || !source || raw) { std::cerr << "-P needs -G <n> and -i synth:<W>x<H>" << std::endl; return 2; }
        const int per = n_streams / n_gpus, ring = 3;
        std::vector<int> ids(n_gpus);
        for (int g = 0; g < n_gpus; g++) ids[g] = gpu_ids.empty() ? device + g : gpu_ids[g];
        // one small context per peer for its device memory (the node's own contexts are private to it)
        std::vector<pcs_ctx*> mem(n_gpus, nullptr);
        for (int g = 0; g < n_gpus; g++) {
            pcs_config mc; memset(&mc, 0, sizeof mc);
            mc.device = ids[g]; mc.n_streams = 1; mc.streams = cfgs.data(); mc.downsample = 1;
            if (pcs_create(&mem[g], &mc) != PCS_OK) { std::cerr << "pcs_create: " << pcs_last_error(nullptr) << std::endl; return 1; }
        }
        std::vector<std::vector<const uint16_t*>> rd(ring, std::vector<const uint16_t*>(n_streams));
        std::vector<std::vector<const uint8_t*>> rc_(ring, std::vector<const uint8_t*>(n_streams));
        std::vector<uint16_t> hd; std::vector<uint8_t> hc;
        for (int k = 0; k < ring; k++)
            for (int s = 0; s < n_streams; s++) {
                pcs_ctx* m = mem[s / per];
                pcs_synth::depth(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)k, hd);
                pcs_synth::color(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)k, hc);
                void *dd = nullptr, *dc = nullptr;
                if (pcs_device_malloc(m, &dd, hd.size() * 2 + 64) != PCS_OK || pcs_device_malloc(m, &dc, hc.size() + 64) != PCS_OK ||
                    pcs_memcpy_h2d(m, dd, hd.data(), hd.size() * 2) != PCS_OK || pcs_memcpy_h2d(m, dc, hc.data(), hc.size()) != PCS_OK) {
                    std::cerr << pcs_last_error(m) << std::endl; return 1;
                }
                rd[k][s] = static_cast<const uint16_t*>(dd); rc_[k][s] = static_cast<const uint8_t*>(dc);
            }
        const size_t cap = pcs_node_max_payload_shorts(node);
        void* d_out[2] = {nullptr, nullptr};
        for (int k = 0; k < 2; k++)
            if (pcs_device_malloc(mem[0], &d_out[k], cap * sizeof(int16_t) + 64) != PCS_OK) { std::cerr << pcs_last_error(mem[0]) << std::endl; return 1; }
        pcs_node_set_timing(node, timer ? 1 : 0);
        auto submit = [&](int k, int* t) {
            return voxel_leaf ? pcs_node_submit_voxel_device(node, rd[k % ring].data(), rc_[k % ring].data(), voxel_leaf,
                                                             static_cast<int16_t*>(d_out[k & 1]), cap, t)
                              : pcs_node_submit_device(node, rd[k % ring].data(), rc_[k % ring].data(), static_cast<int16_t*>(d_out[k & 1]), cap, t);
        };
        int last_n = 0;
        auto wait = [&](int t) {
            return voxel_leaf ? pcs_node_wait_voxel(node, t, &last_n) : pcs_node_wait(node, t, nullptr, &last_n);
        };
        const int warm = 5, iters = max_sets < 1 ? 1 : max_sets;
        int t_prev = -1, t_cur = -1;
        clockTime::time_point t0;
        pcs_node_stats st; memset(&st, 0, sizeof st);
        double k_ms = 0, x_ms = 0, r_ms = 0;
        for (int k = 0; k < warm + iters; k++) {
            if (k == warm) {                                    // drain, then time `iters` steady-state periods
                if (t_prev >= 0 && wait(t_prev) != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
                t_prev = -1;
                t0 = clockTime::now();
            }
            if (submit(k, &t_cur) != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
            if (t_prev >= 0) {
                if (wait(t_prev) != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
                if (timer && k > warm) { pcs_node_last_stats(node, &st); k_ms += st.kernels_ms; x_ms += st.exchange_ms; r_ms += st.root_ms; }
            }
            t_prev = t_cur;
        }
        if (wait(t_prev) != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
        const double ms = timeMilli(clockTime::now() - t0).count() / iters;
        const double mpix = (double)n_streams * W * H / 1e6;
        std::cout << "Pipelined " << (voxel_leaf ? "voxel grid" : "stitch") << " over " << n_gpus << " peer(s): " << ms << " ms per frame-set, "
                  << mpix / ms * 1e3 << " Mpoints/s in, " << last_n << (voxel_leaf ? " voxels" : " points") << std::endl;
        if (voxel_leaf && pcs_node_voxel_reruns(node) > 0)
            std::cout << "Voxel frame-sets run again on the LSD tail after a flagged bucket tail: " << pcs_node_voxel_reruns(node) << std::endl;
        if (timer && iters > 1)
            std::cout << "GPU " << ids[0] << " per frame-set: kernels " << k_ms / (iters - 1) << " ms, exchange " << x_ms / (iters - 1)
                      << " ms, root " << r_ms / (iters - 1) << " ms (" << st.exchanged_bytes << " B into the root)" << std::endl;
        if (dump_path) {       // the last frame-set, in the wire format of the other modes
            size_bytes = last_n * PCS_POINT_BYTES;
            if (!stitched.resize(PCS_HEADER_SHORTS + (size_t)last_n * PCS_POINT_SHORTS + 8)) return 1;
            if (size_bytes && pcs_memcpy_d2h(mem[0], stitched.data() + PCS_HEADER_SHORTS, d_out[(warm + iters - 1) & 1], (size_t)size_bytes) != PCS_OK) return 1;
            memcpy(stitched.data(), &size_bytes, sizeof(int));
            FILE* f = fopen(dump_path, "wb");
            if (f) { fwrite(stitched.data(), 1, (size_t)size_bytes + 4, f); fclose(f); }
        }
        pcs_node_destroy(node);
        for (int k = 0; k < ring; k++)
            for (int s = 0; s < n_streams; s++) { pcs_device_free(mem[s / per], const_cast<uint16_t*>(rd[k][s])); pcs_device_free(mem[s / per], const_cast<uint8_t*>(rc_[k][s])); }
        for (int k = 0; k < 2; k++) pcs_device_free(mem[0], d_out[k]);
        for (pcs_ctx* m : mem) pcs_destroy(m);
        stitched.release();
        pcs_destroy(ctx);
        return 0;
    }

    int listen_fd = -1, client_fd = -1;
    if (serve) {
        listen_fd = pcs_wire::listen_on(serve_port);
        if (listen_fd < 0) { std::cerr << "Couldn't bind server sockfd" << std::endl; return 1; }            // :226-229
        std::cout << "Waiting for client..." << std::endl;
        client_fd = ::accept(listen_fd, NULL, NULL);
        if (client_fd < 0) { std::cerr << "Connection failed" << std::endl; return 1; }
        std::cout << "Established connection with client_sock: " << client_fd << std::endl;
    }

    // the reference primes each camera with one pull before its loop (:557 sendPullRequest)
    for (int fd : cam_fd) pcs_wire::send_pull(fd);

    double total = 0;
    int loop_count = 1;
    for (int set = 0; set < max_sets; set++) {
        auto stitch_start = clockTime::now();
        if (source) {
            if (raw) {
                if (set >= raw_frames) break;
                for (int s = 0; s < n_streams; s++) {
                    depth[s].resize((size_t)cfgs[s].depth.width * cfgs[s].depth.height);
                    color[s].resize((size_t)cfgs[s].color_stride * cfgs[s].color.height);
                    if (fread(depth[s].data(), 2, depth[s].size(), raw) != depth[s].size() ||
                        fread(color[s].data(), 1, color[s].size(), raw) != color[s].size()) { std::cerr << "short read" << std::endl; return 1; }
                }
            } else {
                for (int s = 0; s < n_streams; s++) {
                    pcs_synth::depth(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)set, depth[s]);
                    pcs_synth::color(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)set, color[s]);
                }
                stitch_start = clockTime::now();                   // generation is not part of the stitch
            }
            std::vector<const uint16_t*> dp(n_streams); std::vector<const uint8_t*> cp(n_streams);
            for (int s = 0; s < n_streams; s++) {
                const size_t db = depth[s].size() * sizeof(uint16_t), cb = color[s].size();
                if (db > pin_db[s]) {
                    if (pin_d[s]) pcs_host_free(ctx, pin_d[s]);
                    if (pcs_host_malloc(ctx, (void**)&pin_d[s], db) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
                    pin_db[s] = db;
                }
                if (cb > pin_cb[s]) {
                    if (pin_c[s]) pcs_host_free(ctx, pin_c[s]);
                    if (pcs_host_malloc(ctx, (void**)&pin_c[s], cb) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
                    pin_cb[s] = cb;
                }
                memcpy(pin_d[s], depth[s].data(), db);
                memcpy(pin_c[s], color[s].data(), cb);
                dp[s] = pin_d[s]; cp[s] = pin_c[s];
            }
            if (!raw) stitch_start = clockTime::now();             // handing the frames over is the source's part, not the stitch
            if (voxel_leaf && node) {
                // BASELINE configs[4]: the cameras' voxel partials are pre-aggregated on their GPUs, exchanged once, reduced on GPU 0
                pcs_node_voxel_stats vstats;
                rc = pcs_node_process_voxel(node, dp.data(), cp.data(), voxel_leaf, voxel_route, stitched.data(), stitched.size(), 1,
                                            &size_bytes, &vstats);
                if (rc != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
                if (timer)
                    std::cout << "Voxel grid over " << n_gpus << " GPU(s): kernels " << vstats.kernels_ms << " ms, exchange " << vstats.exchange_ms
                              << " ms (" << vstats.exchanged_bytes << " B), root " << vstats.root_voxel_ms << " ms, " << vstats.partials
                              << (voxel_route == PCS_NODE_VOXEL_PARTIALS ? " partials -> " : " points -> ") << vstats.voxels << " voxels" << std::endl;
            } else if (voxel_leaf) {
                for (int s = 0; s < n_streams; s++) {
                    if (pcs_memcpy_h2d(ctx, d_depth[s], depth[s].data(), depth[s].size() * 2) != PCS_OK ||
                        pcs_memcpy_h2d(ctx, d_color[s], color[s].data(), color[s].size()) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
                }
                auto run_voxel = [&]() {
                    return pcs_process_frames_voxel_device(ctx, reinterpret_cast<const uint16_t* const*>(d_depth.data()),
                                                           reinterpret_cast<const uint8_t* const*>(d_color.data()), voxel_leaf,
                                                           static_cast<int16_t*>(d_vox), vox_cap_points * PCS_POINT_SHORTS, static_cast<int32_t*>(d_nvox));
                };
                rc = run_voxel();
                if (rc != PCS_OK || !fetch_voxels(run_voxel)) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
            } else if (node) {
                rc = pcs_node_process(node, dp.data(), cp.data(), stitched.data(), stitched.size(), 1, nullptr, &size_bytes);
                if (rc != PCS_OK) { std::cerr << pcs_node_last_error(node) << std::endl; return 1; }
            } else {
                rc = pcs_process_frames(ctx, dp.data(), cp.data(), stitched.data(), stitched.size(), 1, nullptr, &size_bytes);
                if (rc != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
            }
        } else {
            // one reader thread per camera, joined in camera order (:381-386)
            std::vector<int32_t> got(n_streams, -1);
            std::vector<std::thread> th;
            for (int i = 0; i < n_streams; i++)
                th.emplace_back([&, i]() {
                    got[i] = pcs_wire::recv_frame(cam_fd[i], cam_buf[i].data(), cam_buf[i].size());
                    if (got[i] >= 0) pcs_wire::send_pull(cam_fd[i]);                    // :370 next pull right away
                });
            for (auto& t : th) t.join();
            std::vector<int> pts(n_streams);
            std::vector<const int16_t*> dptr(n_streams);
            bool ok = true;
            for (int i = 0; i < n_streams; i++) {
                if (got[i] < 0) { ok = false; break; }
                pts[i] = got[i] / PCS_POINT_BYTES;
                if (got[i] && pcs_memcpy_h2d(ctx, d_cam[i], cam_buf[i].data(), (size_t)got[i]) != PCS_OK) ok = false;
                dptr[i] = static_cast<const int16_t*>(d_cam[i]);
            }
            if (!ok) { std::cout << "camera stream ended" << std::endl; break; }
            int total_pts = 0;
            if (transform_path) {       // the reference program's own semantics: decode, transform[i], re-encode, concatenate
                for (int i = 0; i < n_streams; i++) { xf[i].d_payload = dptr[i]; xf[i].n_points = pts[i]; }
                rc = pcs_transform_payloads_device(ctx, n_streams, xf.data(), downsample, static_cast<int16_t*>(d_stitched),
                                                   (size_t)n_streams * cam_cap_bytes / 2, nullptr, &total_pts);
            } else
            rc = pcs_stitch_device(ctx, dptr.data(), pts.data(), n_streams, downsample, static_cast<int16_t*>(d_stitched),
                                   (size_t)n_streams * cam_cap_bytes / 2, &total_pts);
            if (rc != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
            if (voxel_leaf) {
                auto run_voxel = [&]() {
                    return pcs_voxel_grid_device(ctx, static_cast<const int16_t*>(d_stitched), total_pts, voxel_leaf, static_cast<int16_t*>(d_vox),
                                                 vox_cap_points * PCS_POINT_SHORTS, static_cast<int32_t*>(d_nvox));
                };
                rc = run_voxel();
                if (rc != PCS_OK || !fetch_voxels(run_voxel)) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
            } else {
                size_bytes = total_pts * PCS_POINT_BYTES;
                if (size_bytes && pcs_memcpy_d2h(ctx, stitched.data() + PCS_HEADER_SHORTS, d_stitched, (size_t)size_bytes) != PCS_OK) return 1;
                pcs_synchronize(ctx);
                memcpy(stitched.data(), &size_bytes, sizeof(int));                                         // :394-395
            }
        }
        if (serve) {
            int req = pcs_wire::recv_pull(client_fd);                                                      // :398
            if (req < 0) { std::cout << "Client disconnected" << std::endl; break; }
            if (req != pcs_wire::kPullXYZRGB) { std::cerr << "Faulty pull request" << std::endl; return 1; }   // :405-408
            if (!pcs_wire::send_frame(client_fd, stitched.data(), size_bytes)) { std::cout << "Client disconnected" << std::endl; break; }
        }
        if (timer) {
            total += timeMilli(clockTime::now() - stitch_start).count();
            std::cout << "Stitching: " << total / loop_count << " ms, " << "Frame: " << loop_count << std::endl;   // :426-428
            loop_count++;
        }
    }
    if (dump_path) {
        FILE* f = fopen(dump_path, "wb");
        if (f) { fwrite(stitched.data(), 1, (size_t)size_bytes + 4, f); fclose(f); }
    }
    for (int fd : cam_fd) ::close(fd);
    if (client_fd >= 0) ::close(client_fd);
    if (listen_fd >= 0) ::close(listen_fd);
    for (void* p : d_cam) if (p) pcs_device_free(ctx, p);
    for (void* p : pin_d) if (p) pcs_host_free(ctx, p);
    for (void* p : pin_c) if (p) pcs_host_free(ctx, p);
    for (void* p : d_depth) if (p) pcs_device_free(ctx, p);
    for (void* p : d_color) if (p) pcs_device_free(ctx, p);
    if (d_vox) pcs_device_free(ctx, d_vox);
    if (d_nvox) pcs_device_free(ctx, d_nvox);
    if (d_stitched) pcs_device_free(ctx, d_stitched);
    if (node) pcs_node_destroy(node);
    stitched.release();
    pcs_destroy(ctx);
    return 0;
}
