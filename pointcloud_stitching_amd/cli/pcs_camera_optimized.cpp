// pcs-camera-optimized (MI355X) — work-alike of the reference's edge benchmark/server
// (src/pcs-camera-optimized.cpp) on top of libpcs_hip.so. Host code only: getopt surface, frame loop,
// timing, stdout lines and the TCP push are kept; the per-frame work goes through the C ABI.
//
//   reference flags (getopt "hf:vst:cmz", :122):  -h  -f <file>  -v  -s  -t <n>  -c  -m  -z
//   additions:  -g <dev>  -n <streams>  -d <stride>  -i (drop invalid depth)  -C (-c without the reference's lane quirk)
//               -r <frames>  -o <file> (dump last stitched buffer)  -p <port>
//               -e <file> (camera-to-world matrices instead of the ones pasted into the reference's sources)
//               -H (texture coordinates as older librealsense releases computed them: (pixel + 0.5) / size)
//
//   -f takes "synth:<W>x<H>" (deterministic synthetic frames; the reference's bags are LFS stubs and
//   need librealsense), a .pcsraw dump (see pointcloud_stitching_amd/synthetic.py: write_pcsraw) or a
//   librealsense recording (.bag, read by pcs_bag.h — validated only against this repo's own writer).
//   Without -f the reference grabs from a live RealSense camera; that needs librealsense and a camera,
//   neither of which this build has: it says so and exits non-zero.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <getopt.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include "pcs_bag.h"
#include "pcs_synth.h"
#include "pcs_wire.h"

typedef std::chrono::high_resolution_clock clockTime;
typedef std::chrono::duration<double, std::milli> timeMilli;

static const char* filename = nullptr;
static bool display_updates = false, send_buffer = false, cutoff = false, use_hip = false, compress = false;
static bool cutoff_compat = false, drop_invalid = false, pull_mode = false, half_pixel = false, pageable = false;
static int num_of_threads = 1, device = 0, n_streams = 1, downsample = 1, max_frames = 60, port = 8000;
static const char* dump_path = nullptr;
static const char* extrinsics_path = nullptr;
static int client_sock = 0, sockfd = 0;

static void print_usage()
{
    printf("\nUsage: pcs-camera-optimized -f <synth:WxH | frames.pcsraw | recording.bag> [-m] [-t n] [-c] [-s] [-n streams] [-g gpu]\n"
           "  -f <src>  frame source (synthetic generator or raw dump)\n"
           "  -s        send data to central camera server if available (TCP push on port 8000)\n"
           "  -m        use the MI355X HIP path (the reference's SIMD switch)\n"
           "  -t <n>    OpenMP threads of the reference path; accepted, inert on the HIP path\n"
           "  -c        cutoff 0<z<=1.5, -2<x<=2 with compaction, exactly as the reference's -c -m computes it (point k of an\n"
           "            aligned group of four is gated by point 3-k's test)   -C  every point by its own test\n"
           "  -i        drop invalid-depth pixels   -d <n> keep every n-th point   -n <N> camera streams\n"
           "  -g <dev>  GPU ordinal   -r <frames>   -o <file> dump last stitched buffer   -p <port>\n"
           "  -P        serve frames on 'Z' pull requests (the live server's protocol) instead of pushing them\n"
           "  -e <file> camera-to-world matrices, 16 row-major floats per line (python -m pointcloud_stitching_amd.calibration)\n"
           "  -H        texture coordinates as older librealsense releases: (pixel + 0.5) / size\n"
           "  -M        hand the frames over in ordinary pageable memory, as librealsense owns them in the reference's timed region\n"
           "            (:291-293): uploads are staged then. Default: frames copied to page-locked rasters BEFORE the timer starts\n"
           "            (zero copy) - the printed times then belong to a capture pipeline that delivers page-locked frames\n\n");
}

static void parseArgs(int argc, char** argv)
{
    int c;
    while ((c = getopt(argc, argv, "hf:vst:cmzg:n:d:iCr:o:p:e:PHM")) != -1) {
        switch (c) {
            case 'h': print_usage(); exit(0);
            case 'f': filename = optarg; break;
            case 'v': display_updates = true; break;
            case 's': send_buffer = true; break;
            case 't': num_of_threads = atoi(optarg); break;
            case 'c': cutoff = true; cutoff_compat = true; break;     // the reference's -c -m payload, lane quirk included (:501-502, 519)
            case 'C': cutoff = true; cutoff_compat = false; break;    // every point by its own range test
            case 'H': half_pixel = true; break;
            case 'm': use_hip = true; break;
            case 'z': compress = true; break;
            case 'g': device = atoi(optarg); break;
            case 'n': n_streams = atoi(optarg); break;
            case 'd': downsample = atoi(optarg); break;
            case 'i': drop_invalid = true; break;
            case 'r': max_frames = atoi(optarg); break;
            case 'o': dump_path = optarg; break;
            case 'p': port = atoi(optarg); break;
            case 'e': extrinsics_path = optarg; break;
            case 'P': pull_mode = true; send_buffer = true; break;
            case 'M': pageable = true; break;
            default: print_usage(); exit(2);
        }
    }
}

// One consumer per edge server, as in the reference (:75-105): listen on the port, take the first client.
static bool openServer(int p)
{
    sockfd = pcs_wire::listen_on(p);
    if (sockfd < 0) { std::cerr << "\ncannot listen on port " << p << std::endl; return false; }
    std::cout << "Waiting for client..." << std::endl;
    client_sock = ::accept(sockfd, nullptr, nullptr);
    if (client_sock < 0) { std::cerr << "\nConnection failed" << std::endl; return false; }
    std::cout << "Established connection with client_sock: " << client_sock << std::endl;
    return true;
}

static void sigintHandler(int) { std::cout << "\n Exiting \n " << std::endl; exit(0); }

struct FrameSource {
    std::vector<pcs_stream_config> cfg;
    std::vector<std::vector<uint16_t>> depth;     // per stream, current frame
    std::vector<std::vector<uint8_t>> color;
    FILE* fp = nullptr;
    int frames_in_file = 0, W = 0, H = 0;
    bool synth = false, bag = false;
    pcs_bag::Recording rec;            // -f <recording.bag>: librealsense ROS bag (what the reference plays, :168-176)

    bool open(const char* spec)
    {
        if (strncmp(spec, "synth:", 6) == 0) {
            if (sscanf(spec + 6, "%dx%d", &W, &H) != 2 || W <= 0 || H <= 0) return false;
            synth = true;
            for (int s = 0; s < n_streams; s++) cfg.push_back(pcs_synth::stream_config(W, H, s, n_streams == 1));
        } else if (strlen(spec) > 4 && strcmp(spec + strlen(spec) - 4, ".bag") == 0) {
            std::string err;
            if (!rec.open(spec, err)) { std::cerr << spec << ": " << err << std::endl; return false; }
            bag = true; n_streams = 1; frames_in_file = rec.frames;
            cfg.push_back(rec.config);
            memcpy(cfg[0].cam_to_world, pcs_synth::stream_config(8, 8, 0, true).cam_to_world, sizeof cfg[0].cam_to_world);   // tf_mat :64-67
            W = cfg[0].depth.width; H = cfg[0].depth.height;
        } else {
            fp = fopen(spec, "rb");
            if (!fp) return false;
            char magic[8]; int32_t ns = 0, nf = 0;
            if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "PCSRAW1", 8) != 0) return false;
            if (fread(&ns, 4, 1, fp) != 1 || fread(&nf, 4, 1, fp) != 1 || ns < 1 || ns > PCS_MAX_STREAMS) return false;
            n_streams = ns; frames_in_file = nf;
            cfg.resize(ns);
            if (fread(cfg.data(), sizeof(pcs_stream_config), ns, fp) != (size_t)ns) return false;
            W = cfg[0].depth.width; H = cfg[0].depth.height;
        }
        depth.resize(n_streams); color.resize(n_streams);
        return true;
    }
    bool next(int frame)
    {
        if (synth) {
            for (int s = 0; s < n_streams; s++) {
                pcs_synth::depth(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)frame, depth[s]);
                pcs_synth::color(W, H, s, pcs_synth::kSeed + 7919u * (uint32_t)frame, color[s]);
            }
            return true;
        }
        if (frame >= frames_in_file) return false;          // the reference stops when the bag loops (:276)
        if (bag) {
            std::string err;
            if (!rec.read(frame, depth[0], color[0], err)) { std::cerr << "frame " << frame << ": " << err << std::endl; return false; }
            return true;
        }
        for (int s = 0; s < n_streams; s++) {
            depth[s].resize((size_t)cfg[s].depth.width * cfg[s].depth.height);
            color[s].resize((size_t)cfg[s].color_stride * cfg[s].color.height);
            if (fread(depth[s].data(), 2, depth[s].size(), fp) != depth[s].size()) return false;
            if (fread(color[s].data(), 1, color[s].size(), fp) != color[s].size()) return false;
        }
        return true;
    }
};

int main(int argc, char** argv)
{
    parseArgs(argc, argv);
    signal(SIGINT, sigintHandler);
    if (filename == NULL) {
        std::cerr << "Live capture needs librealsense2 and a RealSense camera, which this build does not link.\n"
                     "Use -f synth:<W>x<H>, -f <frames.pcsraw> or -f <recording.bag>." << std::endl;
        return 2;
    }
    std::cout << "Reading Frames from File: " << filename << std::endl;
    FrameSource src;
    if (!src.open(filename)) { std::cerr << "cannot open frame source " << filename << std::endl; return 2; }
    std::cout << "Camera Info: synthetic D400-like stream x" << n_streams << " FW ver:n/a" << std::endl;
    if (num_of_threads) std::cout << "OpenMP Threads: " << num_of_threads << std::endl;
    if (!use_hip)
        std::cout << "note: without -m the reference runs its scalar loop; this build has no CPU path and uses the HIP path "
                     "(the scalar variant differs from -m by +-1 LSB and is not reproduced)" << std::endl;

    if (extrinsics_path) {      // replaces editing tf_mat / transform[i] in source (:64-67)
        FILE* f = fopen(extrinsics_path, "r");
        if (!f) { std::cerr << "cannot open extrinsics file " << extrinsics_path << std::endl; return 2; }
        char line[1024];
        int cam = 0;
        while (cam < n_streams && fgets(line, sizeof line, f)) {
            char* hash = strchr(line, '#');
            if (hash) *hash = 0;
            float m[16];
            int got = 0, pos = 0, adv = 0;
            while (got < 16 && sscanf(line + pos, " %f%n", &m[got], &adv) == 1) { got++; pos += adv; if (line[pos] == ',') pos++; }
            if (got == 0) continue;
            if (got != 16) { std::cerr << extrinsics_path << ": expected 16 values per line" << std::endl; return 2; }
            memcpy(src.cfg[cam++].cam_to_world, m, sizeof m);
        }
        fclose(f);
        if (cam < n_streams) { std::cerr << extrinsics_path << ": only " << cam << " matrices for " << n_streams << " streams" << std::endl; return 2; }
    }

    pcs_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = device; cfg.n_streams = n_streams; cfg.streams = src.cfg.data(); cfg.downsample = downsample;
    cfg.flags = (cutoff ? PCS_FLAG_CUTOFF : 0u) | (cutoff_compat ? PCS_FLAG_CUTOFF_COMPAT : 0u) | (drop_invalid ? PCS_FLAG_DROP_INVALID : 0u) |
                (half_pixel ? PCS_FLAG_TEXCOORD_HALF_PIXEL : 0u);
    pcs_ctx* ctx = nullptr;
    int rc = pcs_create(&ctx, &cfg);
    if (rc != PCS_OK) { std::cerr << "pcs_create: " << pcs_strerror(rc) << ": " << pcs_last_error(nullptr) << std::endl; return 1; }

    const size_t buf_shorts = PCS_HEADER_SHORTS + pcs_max_payload_shorts(ctx);
    short* buffer = nullptr;                                               // the reference mallocs BUF_SIZE shorts (:157);
    if (pcs_host_malloc(ctx, (void**)&buffer, sizeof(short) * buf_shorts) != PCS_OK) {   // page-locked: D2H at link speed
        std::cerr << pcs_last_error(ctx) << std::endl; return 1;
    }
    std::vector<const uint16_t*> dptr(n_streams);
    std::vector<const uint8_t*> cptr(n_streams);
    std::vector<int> counts(n_streams);
    pcs_kernel_timing(ctx, 1);
    // The frames are handed over in page-locked rasters (as a capture pipeline that owns its buffers would deliver them):
    // with the buffer above page-locked too, pcs_process_frames runs zero copy. Filling them is the frame source's job
    // and, like the reference's wait_for_frames / pointcloud::calculate, outside the timed region (:291-293).
    std::vector<uint16_t*> pin_d(n_streams, nullptr);
    std::vector<uint8_t*> pin_c(n_streams, nullptr);
    std::vector<size_t> pin_db(n_streams, 0), pin_cb(n_streams, 0);

    int i = 0, buff_size = 0;
    double duration_sum = 0, buff_size_sum = 0;
    size_t points_in = 0;
    if (send_buffer && !openServer(port)) return 1;

    while (i < max_frames && src.next(i)) {
        i++;
        for (int s = 0; s < n_streams; s++) {
            const size_t db = src.depth[s].size() * sizeof(uint16_t), cb = src.color[s].size();
            if (!pageable && db > pin_db[s]) {
                if (pin_d[s]) pcs_host_free(ctx, pin_d[s]);
                if (pcs_host_malloc(ctx, (void**)&pin_d[s], db) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
                pin_db[s] = db;
            }
            if (!pageable && cb > pin_cb[s]) {
                if (pin_c[s]) pcs_host_free(ctx, pin_c[s]);
                if (pcs_host_malloc(ctx, (void**)&pin_c[s], cb) != PCS_OK) { std::cerr << pcs_last_error(ctx) << std::endl; return 1; }
                pin_cb[s] = cb;
            }
            if (pageable) {       // -M: the frame source's own (pageable) memory goes straight in, like librealsense's frames do
                dptr[s] = src.depth[s].data(); cptr[s] = src.color[s].data();
                continue;
            }
            memcpy(pin_d[s], src.depth[s].data(), db);
            memcpy(pin_c[s], src.color[s].data(), cb);
            dptr[s] = pin_d[s]; cptr[s] = pin_c[s];
        }
        auto time_start = clockTime::now();                                   // :291
        rc = pcs_process_frames(ctx, dptr.data(), cptr.data(), buffer, buf_shorts, send_buffer ? 1 : 0, counts.data(), &buff_size);
        auto time_end = clockTime::now();                                     // :293
        if (rc != PCS_OK) { std::cerr << "pcs_process_frames: " << pcs_last_error(ctx) << std::endl; return 1; }
        if (pull_mode) {      // the live server's protocol (:180-184): one 'Z' per frame, anything else is fatal
            int req = pcs_wire::recv_pull(client_sock);
            if (req < 0) { std::cout << "Client disconnected" << std::endl; break; }
            if (req != pcs_wire::kPullXYZRGB) { std::cerr << "Faulty pull request" << std::endl; return 1; }
        }
        if (send_buffer && !pcs_wire::send_frame(client_sock, buffer, buff_size)) { std::cout << "Client disconnected" << std::endl; break; }   // :719
        const double ms = timeMilli(time_end - time_start).count();
        std::cout << "Frame Time: " << ms << " ms " << "FPS: " << 1000.0 / ms
                  << "\t Buffer size: " << float(buff_size) / (1 << 20) << " MBytes" << std::endl;       // :295-297
        duration_sum += ms;
        buff_size_sum += buff_size;
        for (int s = 0; s < n_streams; s++) points_in += (size_t)pcs_stream_points(ctx, s);
    }
    if (send_buffer) { close(client_sock); close(sockfd); }
    if (i == 0) { std::cerr << "no frames" << std::endl; return 1; }

    std::vector<float> kms(i);
    int nk = 0;
    pcs_kernel_times_ms(ctx, kms.data(), i, &nk);
    double ksum = 0; for (int k = 0; k < nk && k < i; k++) ksum += kms[k];
    const double kavg = nk ? ksum / (nk < i ? nk : i) : 0.0;
    const size_t pts = points_in / i;

    // summary block — same lines as :317-342
    std::cout << "\n### Video Frames H x W : " << src.cfg[0].color.height << " x " << src.cfg[0].color.width << std::endl;
    std::cout << "### Depth Frames H x W : " << src.cfg[0].depth.height << " x " << src.cfg[0].depth.width << std::endl;
    std::cout << "### # Points : " << pts << std::endl;
    std::cout << "\n### Total Frames = " << i << std::endl;
    std::cout << "### AVG Frame Time: " << duration_sum / i << " ms" << std::endl;
    std::cout << "### AVG FPS: " << 1000.0 / (duration_sum / i) << std::endl;
    if (num_of_threads) std::cout << "### OpenMP Threads : " << num_of_threads << std::endl;
    else std::cout << "### Running Serialized" << std::endl;
    if (compress) {
        std::cout << "\n### Sending Compressed Stream" << std::endl;
        std::cout << "### AVG Bytes/Frame: " << float(buff_size_sum) / (i * 1000000.0) << " MBytes" << std::endl;
        std::cout << "### AVG Compression Ratio " << float(buff_size_sum) / ((pts / 100.0) * 5 * sizeof(short) * i) << " %" << std::endl;
    } else {
        std::cout << "\n### AVG Bytes/Frame: " << float(buff_size_sum) / (i * 1000000.0) << " MBytes" << std::endl;
        std::cout << "### AVG Filter Compress Ratio " << float(buff_size_sum) / ((pts / 100.0) * 5 * sizeof(short) * i) << " %" << std::endl;
    }
    // additions
    std::cout << "\n### HIP streams : " << n_streams << " on GPU " << device << " (arithmetic policy " << pcs_stream_math(ctx, 0) << ")" << std::endl;
    // every buffer handed to pcs_process_frames is page-locked, so unless PCS_ZERO_COPY=0 the kernels read the rasters and
    // write the payload over PCIe themselves: their time then is a link figure, not an HBM one
    const char* zc_env = getenv("PCS_ZERO_COPY");
    const bool zero_copy = !(zc_env && atoi(zc_env) == 0) && !pageable;
    std::cout << "### Frame hand-over : " << (pageable ? "pageable rasters (-M): staged uploads inside the timed region, as the reference's frames"
                                                       : "page-locked rasters filled before the timer starts (zero copy)") << std::endl;
    if (zero_copy) {
        std::cout << "### AVG Kernel Time: " << kavg << " ms  (hipEvent; zero copy: rasters read from and payload written to host memory over PCIe)" << std::endl;
        if (kavg > 0)
            std::cout << "### Kernel PCIe GB/s (15 B/point, both directions): " << pts * 15.0 / kavg / 1e6 << std::endl;
    } else {
        std::cout << "### AVG Kernel Time: " << kavg << " ms  (hipEvent, deproject+transform+pack only)" << std::endl;
        if (kavg > 0) {
            std::cout << "### Kernel Mpoints/s: " << pts / kavg / 1e3 << std::endl;
            std::cout << "### Kernel HBM GB/s (15 B/point algorithmic): " << pts * 15.0 / kavg / 1e6
                      << "  = " << pts * 15.0 / kavg / 1e6 / 80.0 << " % of 8 TB/s" << std::endl;
        }
    }
    std::cout << "### Host-API Mpoints/s (PCIe both ways included): " << pts / (duration_sum / i) / 1e3 << std::endl;

    if (dump_path) {
        FILE* f = fopen(dump_path, "wb");
        if (f) { fwrite(buffer, 1, (size_t)buff_size + 4, f); fclose(f); }
    }
    pcs_host_free(ctx, buffer);
    for (int s = 0; s < n_streams; s++) { if (pin_d[s]) pcs_host_free(ctx, pin_d[s]); if (pin_c[s]) pcs_host_free(ctx, pin_c[s]); }
    pcs_destroy(ctx);
    return 0;
}
