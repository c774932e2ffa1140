// pcs_synth.h — deterministic synthetic camera frames for the C++ host programs.
// Same integer-only generator as pointcloud_stitching_amd/synthetic.py; the GPU test
// tests/test_cli.py checks that both produce identical stitched output.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pcs_hip.h"

namespace pcs_synth {

constexpr uint32_t kSeed = 0xC0FFEEu;

inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

inline void depth(int W, int H, int stream, uint32_t seed, std::vector<uint16_t>& out)
{
    out.resize((size_t)W * H);
    const uint32_t key = seed + 0x9E3779B9u * (uint32_t)(stream + 1);
    const int bx = (W / 3) & ~7, by = H / 4;
    for (uint32_t i = 0; i < (uint32_t)W * (uint32_t)H; i++) {
        const uint32_t h = hash32(i * 2654435761u + key);
        const int64_t r = i / (uint32_t)W, c = i % (uint32_t)W;
        int64_t ph = (3 * c * 1024) / W + (2 * r * 1024) / H + 128 * (int64_t)stream;
        ph &= 1023;
        const int64_t tri = ph < 512 ? ph : 1024 - ph;
        int64_t d = 500 + (tri * 4000) / 512;
        d += (int64_t)(h & 0xF) - 8 + (int64_t)((h >> 4) & 1);
        const bool hole = ((h >> 8) % 10u) == 0;
        const bool block = c >= bx && c < bx + 64 && r >= by && r < by + 64;
        out[i] = (hole || block) ? 0 : (uint16_t)d;
    }
}

inline void color(int W, int H, int stream, uint32_t seed, std::vector<uint8_t>& out)
{
    const size_t nbytes = (size_t)3 * W * H, nwords = (nbytes + 3) / 4;
    std::vector<uint32_t> w(nwords);
    const uint32_t key = (uint32_t)(((uint64_t)(seed ^ 0x5BD1E995u) + 0x7F4A7C15ull * (uint64_t)(stream + 1)) & 0xFFFFFFFFull);
    for (size_t i = 0; i < nwords; i++) w[i] = hash32((uint32_t)i * 0x9E3779B1u + key);
    out.resize(nbytes);
    std::memcpy(out.data(), w.data(), nbytes);
}

// src/pcs-multicamera-optimized.cpp:417-455 (transform[0..7]) and src/pcs-camera-optimized.cpp:64-67 (tf_mat)
inline const float* reference_extrinsic(int index /* -1 = tf_mat */)
{
    static const float tf_mat[16] = {-0.99977970f, 0.00926272f, 0.01883480f, 0.0f, -0.01638983f, 0.21604544f, -0.97624574f, 3.416f,
                                     -0.01311186f, -0.97633937f, -0.21584603f, 1.802f, 0, 0, 0, 1};
    static const float tr[8][16] = {
        {-0.69888007f, -0.32213748f, 0.63858757f, -2.229f, -0.71520905f, 0.32290986f, -0.61984291f, 2.918f, -0.00653159f, -0.88991947f, -0.45607091f, 0.364f, 0, 0, 0, 1},
        {-0.96127595f, 0.09045863f, -0.26031862f, 0.317f, 0.27558764f, 0.31552831f, -0.90801615f, 2.833f, 0.0f, -0.94459469f, -0.32823906f, 0.381f, 0, 0, 0, 1},
        {-0.63305575f, 0.28270490f, -0.72063747f, 2.803f, 0.77409926f, 0.22724638f, -0.59087175f, 2.055f, -0.00328008f, -0.93189968f, -0.36270128f, 0.421f, 0, 0, 0, 1},
        {0.17021299f, 0.28598815f, -0.94299433f, 2.51f, 0.98527137f, -0.03349883f, 0.16768470f, -0.273f, 0.01636663f, -0.95764743f, -0.28747787f, 0.359f, 0, 0, 0, 1},
        {0.72625904f, 0.26139935f, -0.63578155f, 1.909f, 0.68735231f, -0.26305364f, 0.67701520f, -2.817f, 0.00972668f, -0.92869433f, -0.37071853f, 0.379f, 0, 0, 0, 1},
        {0.98744750f, 0.00686296f, 0.15779838f, -0.574f, -0.14665062f, -0.33120318f, 0.93209337f, -2.697f, 0.05866025f, -0.94353450f, -0.32603930f, 0.309f, 0, 0, 0, 1},
        {0.67295609f, 0.40193638f, 0.62094867f, -2.973f, -0.35777412f, -0.55787451f, 0.74884826f, -0.417f, 0.64740079f, -0.72610136f, -0.23162261f, 0.434f, 0, 0, 0, 1},
        {0.08929624f, -0.21535297f, 0.97244500f, -2.957f, -0.67610010f, -0.73004840f, -0.09958907f, -0.339f, 0.73137872f, -0.64857723f, -0.21079074f, 0.338f, 0, 0, 0, 1}};
    return index < 0 ? tf_mat : tr[index & 7];
}

inline pcs_stream_config stream_config(int W, int H, int stream, bool single)
{
    pcs_stream_config sc;
    std::memset(&sc, 0, sizeof sc);
    sc.depth.width = W; sc.depth.height = H;
    sc.depth.fx = sc.depth.fy = (float)(0.7 * W);
    sc.depth.ppx = (float)(W / 2.0 - 0.5 + 3.7);
    sc.depth.ppy = (float)(H / 2.0 - 0.5 - 2.1);
    sc.color = sc.depth;
    sc.depth_to_color.rotation[0] = sc.depth_to_color.rotation[4] = sc.depth_to_color.rotation[8] = 1.0f;
    sc.depth_to_color.translation[0] = 0.015f;
    sc.depth_scale = 0.001f;
    sc.color_bpp = 3; sc.color_stride = 3 * W;
    std::memcpy(sc.cam_to_world, reference_extrinsic(single ? -1 : stream), sizeof sc.cam_to_world);
    return sc;
}

}  // namespace pcs_synth
