// pcs_bag.h — reader for librealsense recordings (`rs-record`, `rs2::recorder`): ROS bag v2.0 container
// carrying raw Z16 / RGB8 images, CameraInfo, stream extrinsics and the depth-units option.
//
// Replaces what the reference gets from librealsense when started as `pcs-camera-optimized -f file.bag`
// (src/pcs-camera-optimized.cpp:168-176: cfg.enable_device_from_file(filename); :186-191 depth scale;
// :266-289 frames). Host-only, no dependencies: container per the published "ROS bag format 2.0"
// (records = <hlen><name=value fields><dlen><data>; op 0x03 bag header, 0x05 chunk, 0x07 connection,
// 0x02 message, 0x04 index, 0x06 chunk info), chunk compression "none" or "lz4" (LZ4 frame format, decoded
// here), messages per the ROS 1 serialisation of sensor_msgs/Image, sensor_msgs/CameraInfo,
// geometry_msgs/Transform and std_msgs/Float32, topics per librealsense's ros file format:
//     /device_0/sensor_<s>/<Depth|Color>_<i>/image/data            sensor_msgs/Image
//     /device_0/sensor_<s>/<Depth|Color>_<i>/info/camera_info      sensor_msgs/CameraInfo
//     /device_0/sensor_<s>/<Depth|Color>_<i>/tf/<ref>              geometry_msgs/Transform (stream -> reference)
//     /device_0/sensor_<s>/option/Depth Units/value                std_msgs/Float32
//
// VALIDATION STATUS: the reference's samples/*.bag are git-LFS stubs and no librealsense exists in this
// image, so this reader has only been exercised against bags written by tests/bag_writer.py from the
// same published layouts ("bz2" chunks are rejected, not decoded). Treat real-bag compatibility as unproven.
#ifndef PCS_BAG_H
#define PCS_BAG_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pcs_hip.h"

namespace pcs_bag {

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline double   rdf64(const uint8_t* p) { double v; memcpy(&v, p, 8); return v; }

// ---- LZ4 frame format (what rosbag's "lz4" chunks hold) ------------------------------------------
// Decodes into `out` (resized to `expect` bytes, the chunk header's uncompressed size). Checksums are skipped,
// not verified. Returns false on any malformed input; never reads or writes out of bounds.
inline bool lz4_frame_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
    out.assign(expect, 0);
    size_t ip = 0, op = 0;
    if (n < 7 || rd32(src) != 0x184D2204u) return false;
    const uint8_t flg = src[4];
    if ((flg >> 6) != 1) return false;                        // version
    const bool block_checksum = (flg >> 4) & 1, content_size = (flg >> 3) & 1, content_checksum = (flg >> 2) & 1;
    const bool dict_id = flg & 1;
    ip = 6;
    if (content_size) ip += 8;
    if (dict_id) ip += 4;
    ip += 1;                                                  // header checksum byte
    if (ip > n) return false;
    for (;;) {
        if (ip + 4 > n) return false;
        const uint32_t word = rd32(src + ip); ip += 4;
        if (word == 0) break;                                 // EndMark
        const bool raw = (word >> 31) != 0;
        const size_t bs = word & 0x7FFFFFFFu;
        if (bs > n - ip) return false;
        if (raw) {
            if (bs > expect - op) return false;
            memcpy(out.data() + op, src + ip, bs); op += bs;
        } else {
            const uint8_t* b = src + ip; const uint8_t* const be = b + bs;
            while (b < be) {
                const uint8_t token = *b++;
                size_t lit = token >> 4;
                if (lit == 15) { uint8_t x; do { if (b >= be) return false; x = *b++; lit += x; } while (x == 255); }
                if (lit > (size_t)(be - b) || lit > expect - op) return false;
                memcpy(out.data() + op, b, lit); b += lit; op += lit;
                if (b >= be) break;                           // last sequence: literals only
                if (be - b < 2) return false;
                const size_t offset = (size_t)b[0] | ((size_t)b[1] << 8); b += 2;
                if (offset == 0 || offset > op) return false;
                size_t ml = token & 15;
                if (ml == 15) { uint8_t x; do { if (b >= be) return false; x = *b++; ml += x; } while (x == 255); }
                ml += 4;
                if (ml > expect - op) return false;
                for (size_t k = 0; k < ml; k++) out[op + k] = out[op + k - offset];   // may overlap: bytewise
                op += ml;
            }
        }
        ip += bs;
        if (block_checksum) ip += 4;
    }
    (void)content_checksum;
    return op == expect;
}

// ---- record layer ----------------------------------------------------------------------------------
struct Fields {
    std::map<std::string, std::pair<const uint8_t*, uint32_t>> f;
    bool parse(const uint8_t* p, uint32_t len)
    {
        uint32_t i = 0;
        while (i < len) {
            if (len - i < 4) return false;
            const uint32_t fl = rd32(p + i); i += 4;
            if (fl > len - i) return false;
            const uint8_t* eq = (const uint8_t*)memchr(p + i, '=', fl);
            if (!eq) return false;
            f[std::string((const char*)p + i, eq - (p + i))] = {eq + 1, (uint32_t)(fl - (eq + 1 - (p + i)))};
            i += fl;
        }
        return true;
    }
    bool u8(const char* k, uint8_t& v) const { auto it = f.find(k); if (it == f.end() || it->second.second != 1) return false; v = *it->second.first; return true; }
    bool u32(const char* k, uint32_t& v) const { auto it = f.find(k); if (it == f.end() || it->second.second != 4) return false; v = rd32(it->second.first); return true; }
    bool u64(const char* k, uint64_t& v) const { auto it = f.find(k); if (it == f.end() || it->second.second != 8) return false; v = rd64(it->second.first); return true; }
    bool str(const char* k, std::string& v) const { auto it = f.find(k); if (it == f.end()) return false; v.assign((const char*)it->second.first, it->second.second); return true; }
};

// Bounded cursor over a serialised ROS message.
struct Cursor {
    const uint8_t* p; size_t n, i = 0; bool ok = true;
    Cursor(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    bool need(size_t k) { if (!ok || k > n - i) { ok = false; return false; } return true; }
    uint8_t  u8()  { if (!need(1)) return 0; return p[i++]; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v = rd32(p + i); i += 4; return v; }
    double   f64() { if (!need(8)) return 0; double v = rdf64(p + i); i += 8; return v; }
    float    f32() { if (!need(4)) return 0; float v; memcpy(&v, p + i, 4); i += 4; return v; }
    std::string str() { const uint32_t l = u32(); if (!need(l)) return {}; std::string s((const char*)p + i, l); i += l; return s; }
    const uint8_t* bytes(size_t k) { if (!need(k)) return nullptr; const uint8_t* q = p + i; i += k; return q; }
    void header() { u32(); u32(); u32(); str(); }              // std_msgs/Header: seq, stamp, frame_id
};

struct FrameRef {           // where one image message lives
    uint64_t t_ns;
    size_t   chunk;         // index into Reader::chunks
    size_t   off, len;      // message body inside the decompressed chunk
};

struct StreamMeta {
    bool            have_info = false, have_tf = false;
    pcs_intrinsics  intr{};
    float           q[4] = {0, 0, 0, 1};    // x y z w, stream -> reference
    float           t[3] = {0, 0, 0};
    std::vector<FrameRef> frames;
};

struct Chunk { size_t data_pos, data_len, usize; bool lz4; };

class Reader {
public:
    ~Reader() { if (fp_) fclose(fp_); }

    bool open(const char* path, std::string& err)
    {
        fp_ = fopen(path, "rb");
        if (!fp_) { err = "cannot open file"; return false; }
        char magic[40] = {0};
        const size_t got = fread(magic, 1, sizeof magic - 1, fp_);
        if (got >= 13 && memcmp(magic, "#ROSBAG V2.0\n", 13) == 0) {
            /* a recording */
        } else if (got >= 27 && memcmp(magic, "version https://git-lfs", 23) == 0) {
            // what a checkout without `git lfs pull` holds in place of the sample recordings (.gitattributes:1)
            err = "a Git-LFS pointer file, not the recording itself (run `git lfs pull` to fetch the .bag)";
            return false;
        } else { err = "not a ROS bag v2.0 file"; return false; }
        fseek(fp_, 0, SEEK_END);
        const size_t fsize = (size_t)ftell(fp_);
        size_t pos = 13;
        std::vector<uint8_t> hdr, data;
        while (pos + 8 <= fsize) {
            uint32_t hlen;
            if (!read_at(pos, 4, hdr)) { err = "short read"; return false; }
            hlen = rd32(hdr.data());
            if (hlen > fsize - pos - 4) { err = "record header exceeds file"; return false; }
            if (!read_at(pos + 4, hlen + 4, hdr)) { err = "short read"; return false; }
            const uint32_t dlen = rd32(hdr.data() + hlen);
            const size_t dpos = pos + 4 + hlen + 4;
            if (dlen > fsize - dpos) { err = "record data exceeds file"; return false; }
            Fields f;
            uint8_t op = 0;
            if (!f.parse(hdr.data(), hlen) || !f.u8("op", op)) { err = "malformed record header"; return false; }
            if (op == 0x05) {           // chunk
                std::string comp; uint32_t usize = 0;
                if (!f.str("compression", comp) || !f.u32("size", usize)) { err = "chunk without compression/size"; return false; }
                if (comp != "none" && comp != "lz4") { err = "chunk compression '" + comp + "' is not supported (none, lz4)"; return false; }
                chunks.push_back({dpos, dlen, usize, comp == "lz4"});
                if (!load_chunk(chunks.size() - 1, err)) return false;
                if (!scan_chunk(chunks.size() - 1, err)) return false;
            } else if (op == 0x07) {    // connection outside a chunk (the index section repeats them)
                if (!read_at(dpos, dlen, data)) { err = "short read"; return false; }
                if (!connection(f, data.data(), dlen)) { err = "malformed connection record"; return false; }
            }                            // 0x03 bag header, 0x04 index, 0x06 chunk info: not needed for a linear scan
            pos = dpos + dlen;
        }
        return true;
    }

    // Message body of a frame (valid until the next body() call that needs another chunk).
    const uint8_t* body(const FrameRef& r, std::string& err)
    {
        if (!load_chunk(r.chunk, err)) return nullptr;
        return cur_.data() + r.off;
    }

    std::vector<Chunk> chunks;
    StreamMeta depth, color;
    float depth_units = 0.0f;        // 0 = option not recorded

private:
    FILE* fp_ = nullptr;
    std::vector<uint8_t> cur_, raw_;
    size_t cur_chunk_ = (size_t)-1;
    enum Kind { kNone, kDepthImage, kColorImage, kDepthInfo, kColorInfo, kDepthTf, kColorTf, kDepthUnits };
    std::map<uint32_t, Kind> conn_kind_;

    bool read_at(size_t pos, size_t len, std::vector<uint8_t>& buf)
    {
        buf.resize(len);
        if (fseek(fp_, (long)pos, SEEK_SET) != 0) return false;
        return len == 0 || fread(buf.data(), 1, len, fp_) == len;
    }

    bool load_chunk(size_t k, std::string& err)
    {
        if (k == cur_chunk_) return true;
        const Chunk& c = chunks[k];
        cur_chunk_ = (size_t)-1;
        if (!c.lz4) {
            if (c.usize != c.data_len) { err = "uncompressed chunk size mismatch"; return false; }
            if (!read_at(c.data_pos, c.data_len, cur_)) { err = "short read"; return false; }
        } else {
            if (!read_at(c.data_pos, c.data_len, raw_)) { err = "short read"; return false; }
            if (!lz4_frame_decode(raw_.data(), raw_.size(), cur_, c.usize)) { err = "corrupt lz4 chunk"; return false; }
        }
        cur_chunk_ = k;
        return true;
    }

    static bool ends_with(const std::string& s, const char* suf)
    {
        const size_t n = strlen(suf);
        return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
    }

    bool connection(const Fields& f, const uint8_t* data, uint32_t dlen)
    {
        uint32_t id; std::string topic;
        if (!f.u32("conn", id) || !f.str("topic", topic)) return false;
        Fields ch;
        if (!ch.parse(data, dlen)) return false;
        const bool is_depth = topic.find("/Depth_") != std::string::npos;
        const bool is_color = topic.find("/Color_") != std::string::npos;
        Kind k = kNone;
        if (ends_with(topic, "/image/data")) k = is_depth ? kDepthImage : is_color ? kColorImage : kNone;
        else if (ends_with(topic, "/info/camera_info")) k = is_depth ? kDepthInfo : is_color ? kColorInfo : kNone;
        else if (topic.find("/tf/") != std::string::npos) k = is_depth ? kDepthTf : is_color ? kColorTf : kNone;
        else if (ends_with(topic, "/option/Depth Units/value")) k = kDepthUnits;
        conn_kind_[id] = k;
        return true;
    }

    static int model_of(const std::string& s)
    {
        if (s == "Brown Conrady") return PCS_DISTORTION_BROWN_CONRADY;
        if (s == "Modified Brown Conrady") return PCS_DISTORTION_MODIFIED_BROWN_CONRADY;
        if (s == "Inverse Brown Conrady") return PCS_DISTORTION_INVERSE_BROWN_CONRADY;
        if (s == "Ftheta") return PCS_DISTORTION_FTHETA;
        if (s == "Kannala Brandt4") return 5;               // rs2 numbering; pcs_create rejects it unless coeffs are zero
        return PCS_DISTORTION_NONE;
    }

    static bool camera_info(const uint8_t* p, size_t n, pcs_intrinsics& in)
    {
        Cursor c(p, n);
        c.header();
        const uint32_t h = c.u32(), w = c.u32();
        const std::string model = c.str();
        const uint32_t nd = c.u32();
        double D[5] = {0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < nd; k++) { const double v = c.f64(); if (k < 5) D[k] = v; }
        double K[9];
        for (int k = 0; k < 9; k++) K[k] = c.f64();
        if (!c.ok) return false;
        memset(&in, 0, sizeof in);
        in.width = (int)w; in.height = (int)h;
        in.fx = (float)K[0]; in.ppx = (float)K[2]; in.fy = (float)K[4]; in.ppy = (float)K[5];
        in.model = model_of(model);
        for (int k = 0; k < 5; k++) in.coeffs[k] = (float)D[k];
        return true;
    }

    bool scan_chunk(size_t k, std::string& err)
    {
        const uint8_t* p = cur_.data();
        const size_t n = cur_.size();
        size_t i = 0;
        while (i + 8 <= n) {
            const uint32_t hlen = rd32(p + i);
            if (hlen > n - i - 8) { err = "record header exceeds chunk"; return false; }
            const uint32_t dlen = rd32(p + i + 4 + hlen);
            const size_t dpos = i + 4 + hlen + 4;
            if (dlen > n - dpos) { err = "record data exceeds chunk"; return false; }
            Fields f; uint8_t op = 0;
            if (!f.parse(p + i + 4, hlen) || !f.u8("op", op)) { err = "malformed record in chunk"; return false; }
            if (op == 0x07) {
                if (!connection(f, p + dpos, dlen)) { err = "malformed connection record"; return false; }
            } else if (op == 0x02) {
                uint32_t conn; uint64_t tm;
                if (!f.u32("conn", conn) || !f.u64("time", tm)) { err = "message without conn/time"; return false; }
                const uint64_t t_ns = (tm & 0xFFFFFFFFull) * 1000000000ull + (tm >> 32);
                auto it = conn_kind_.find(conn);
                const Kind kind = it == conn_kind_.end() ? kNone : it->second;
                switch (kind) {
                    case kDepthImage: depth.frames.push_back({t_ns, k, dpos, dlen}); break;
                    case kColorImage: color.frames.push_back({t_ns, k, dpos, dlen}); break;
                    case kDepthInfo:  depth.have_info = camera_info(p + dpos, dlen, depth.intr) || depth.have_info; break;
                    case kColorInfo:  color.have_info = camera_info(p + dpos, dlen, color.intr) || color.have_info; break;
                    case kDepthTf: case kColorTf: {
                        Cursor c(p + dpos, dlen);
                        double v[7];
                        for (int j = 0; j < 7; j++) v[j] = c.f64();
                        if (c.ok) {
                            StreamMeta& m = kind == kDepthTf ? depth : color;
                            for (int j = 0; j < 3; j++) m.t[j] = (float)v[j];
                            for (int j = 0; j < 4; j++) m.q[j] = (float)v[3 + j];
                            m.have_tf = true;
                        }
                        break;
                    }
                    case kDepthUnits: { Cursor c(p + dpos, dlen); const float u = c.f32(); if (c.ok) depth_units = u; break; }
                    default: break;
                }
            }
            i = dpos + dlen;
        }
        return true;
    }
};

// One decoded image (points into the reader's current chunk).
struct Image { uint32_t width = 0, height = 0, step = 0; std::string encoding; const uint8_t* data = nullptr; size_t size = 0; };

inline bool parse_image(const uint8_t* p, size_t n, Image& im)
{
    Cursor c(p, n);
    c.header();
    im.height = c.u32(); im.width = c.u32();
    im.encoding = c.str();
    c.u8();                       // is_bigendian
    im.step = c.u32();
    im.size = c.u32();
    im.data = c.bytes(im.size);
    return c.ok && im.data;
}

// row-major 3x3 from a unit quaternion (x y z w)
inline void quat_to_rot(const float q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// A recording as the hot path wants it: one depth+colour stream pair, its configuration and its frames.
class Recording {
public:
    pcs_stream_config config{};
    int frames = 0;

    bool open(const char* path, std::string& err)
    {
        if (!rd_.open(path, err)) return false;
        if (rd_.depth.frames.empty() || rd_.color.frames.empty()) { err = "bag holds no Depth_*/Color_* image topics"; return false; }
        if (!rd_.depth.have_info || !rd_.color.have_info) { err = "bag holds no camera_info for depth/colour"; return false; }
        memset(&config, 0, sizeof config);
        config.depth = rd_.depth.intr;
        config.color = rd_.color.intr;
        config.depth_scale = rd_.depth_units > 0.0f ? rd_.depth_units : 0.001f;    // D400 default
        // depth -> colour = (colour -> ref)^-1 o (depth -> ref); column-major rotation as rs2_extrinsics
        double Rd[9], Rc[9];
        quat_to_rot(rd_.depth.q, Rd); quat_to_rot(rd_.color.q, Rc);
        double R[9], t[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double a = 0;
                for (int k = 0; k < 3; k++) a += Rc[3 * k + i] * Rd[3 * k + j];     // Rc^T * Rd
                R[3 * i + j] = a;
            }
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += Rc[3 * k + i] * ((double)rd_.depth.t[k] - (double)rd_.color.t[k]);
            t[i] = a;
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) config.depth_to_color.rotation[3 * j + i] = (float)R[3 * i + j];
        for (int i = 0; i < 3; i++) config.depth_to_color.translation[i] = (float)t[i];
        for (int i = 0; i < 4; i++) config.cam_to_world[5 * i] = 1.0f;
        // first frame fixes encoding / stride
        Image d, c;
        if (!image(rd_.depth.frames[0], d, err) ) return false;
        if (d.encoding != "mono16" && d.encoding != "16UC1") { err = "depth encoding '" + d.encoding + "' is not Z16 (mono16 / 16UC1)"; return false; }
        if ((int)d.width != config.depth.width || (int)d.height != config.depth.height || d.step != d.width * 2) { err = "depth image geometry differs from camera_info"; return false; }
        if (!image(rd_.color.frames[0], c, err)) return false;
        int bpp = 0;
        if (c.encoding == "rgb8" || c.encoding == "bgr8") bpp = 3;
        else if (c.encoding == "rgba8" || c.encoding == "bgra8") bpp = 4;
        else { err = "colour encoding '" + c.encoding + "' is not covered (rgb8, bgr8, rgba8, bgra8)"; return false; }
        if ((int)c.width != config.color.width || (int)c.height != config.color.height || c.step < c.width * (uint32_t)bpp) { err = "colour image geometry differs from camera_info"; return false; }
        config.color_bpp = bpp; config.color_stride = (int)c.step;
        frames = (int)rd_.depth.frames.size();
        return true;
    }

    // Frame k: the k-th depth image with the colour image nearest in time (the frameset librealsense's syncer
    // would hand out is not reproducible without it; nearest-timestamp is this build's definition).
    bool read(int k, std::vector<uint16_t>& depth, std::vector<uint8_t>& color, std::string& err)
    {
        if (k < 0 || k >= frames) { err = "frame index out of range"; return false; }
        Image im;
        if (!image(rd_.depth.frames[k], im, err)) return false;
        if (im.size != (size_t)config.depth.width * config.depth.height * 2) { err = "depth frame size changed"; return false; }
        depth.resize(im.size / 2);
        memcpy(depth.data(), im.data, im.size);
        const uint64_t t = rd_.depth.frames[k].t_ns;
        const auto& cf = rd_.color.frames;
        size_t best = 0; uint64_t bd = ~0ull;
        for (size_t j = 0; j < cf.size(); j++) {
            const uint64_t dlt = cf[j].t_ns > t ? cf[j].t_ns - t : t - cf[j].t_ns;
            if (dlt < bd) { bd = dlt; best = j; }
        }
        if (!image(cf[best], im, err)) return false;
        if (im.size != (size_t)config.color_stride * config.color.height) { err = "colour frame size changed"; return false; }
        color.assign(im.data, im.data + im.size);
        return true;
    }

private:
    Reader rd_;
    bool image(const FrameRef& r, Image& im, std::string& err)
    {
        const uint8_t* b = rd_.body(r, err);
        if (!b) return false;
        if (!parse_image(b, r.len, im)) { err = "malformed sensor_msgs/Image"; return false; }
        return true;
    }
};

}  // namespace pcs_bag
#endif
