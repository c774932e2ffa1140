// pcs-bag-info — prints what pcs_bag.h extracts from a librealsense recording as one JSON object:
// stream configuration and, per frame, FNV-1a hashes of the depth and colour rasters. `-x <out.pcsraw>`
// additionally converts the recording to this build's raw frame-dump format.
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pcs_bag.h"

static uint64_t fnv1a(const void* p, size_t n)
{
    const uint8_t* b = static_cast<const uint8_t*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static void intr_json(const char* name, const pcs_intrinsics& in)
{
    printf("\"%s\": {\"width\": %d, \"height\": %d, \"ppx\": %.9g, \"ppy\": %.9g, \"fx\": %.9g, \"fy\": %.9g, \"model\": %d, "
           "\"coeffs\": [%.9g, %.9g, %.9g, %.9g, %.9g]}",
           name, in.width, in.height, in.ppx, in.ppy, in.fx, in.fy, in.model, in.coeffs[0], in.coeffs[1], in.coeffs[2],
           in.coeffs[3], in.coeffs[4]);
}

int main(int argc, char** argv)
{
    const char* path = nullptr; const char* dump = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-x") && i + 1 < argc) dump = argv[++i];
        else path = argv[i];
    }
    if (!path) { fprintf(stderr, "usage: pcs-bag-info <recording.bag> [-x out.pcsraw]\n"); return 2; }
    pcs_bag::Recording rec;
    std::string err;
    if (!rec.open(path, err)) { fprintf(stderr, "%s: %s\n", path, err.c_str()); return 1; }
    const pcs_stream_config& c = rec.config;
    printf("{");
    intr_json("depth", c.depth); printf(", ");
    intr_json("color", c.color);
    printf(", \"depth_scale\": %.9g, \"color_bpp\": %d, \"color_stride\": %d, \"rotation\": [", c.depth_scale, c.color_bpp, c.color_stride);
    for (int k = 0; k < 9; k++) printf("%s%.9g", k ? ", " : "", c.depth_to_color.rotation[k]);
    printf("], \"translation\": [%.9g, %.9g, %.9g], \"frames\": [", c.depth_to_color.translation[0],
           c.depth_to_color.translation[1], c.depth_to_color.translation[2]);
    FILE* out = nullptr;
    if (dump) {
        out = fopen(dump, "wb");
        if (!out) { fprintf(stderr, "cannot write %s\n", dump); return 1; }
        const int32_t ns = 1, nf = rec.frames;
        fwrite("PCSRAW1\0", 1, 8, out); fwrite(&ns, 4, 1, out); fwrite(&nf, 4, 1, out);
        fwrite(&c, sizeof c, 1, out);
    }
    std::vector<uint16_t> d; std::vector<uint8_t> col;
    for (int k = 0; k < rec.frames; k++) {
        if (!rec.read(k, d, col, err)) { fprintf(stderr, "frame %d: %s\n", k, err.c_str()); return 1; }
        printf("%s[\"%016" PRIx64 "\", \"%016" PRIx64 "\"]", k ? ", " : "", fnv1a(d.data(), d.size() * 2), fnv1a(col.data(), col.size()));
        if (out) { fwrite(d.data(), 2, d.size(), out); fwrite(col.data(), 1, col.size(), out); }
    }
    printf("]}\n");
    if (out) fclose(out);
    return 0;
}
