// ImageWriter.hpp -- PPM and PNG output of a float framebuffer with the reference's quantisation
// (ImageSaver.h:18-75): byte = (uint8_t)clamp(c * 255.0f, 0.0f, 255.0f) (truncation), pixels written in
// framebuffer order.  The PNG is a valid stored-deflate (uncompressed) stream; JPEG is not provided
// (image encoding is I/O outside the hot path) and a name with any other extension (".jpg") is written as PPM under that name with ".ppm" in its place.
#pragma once

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "rt_types.hpp"

namespace cg_datastructures {

inline std::vector<uint8_t> quantise(const std::vector<Vec3>& px) {
    std::vector<uint8_t> out(px.size() * 3);
    for (size_t i = 0; i < px.size(); ++i) {
        out[3 * i + 0] = static_cast<uint8_t>(std::clamp(px[i].x * 255.0f, 0.0f, 255.0f));
        out[3 * i + 1] = static_cast<uint8_t>(std::clamp(px[i].y * 255.0f, 0.0f, 255.0f));
        out[3 * i + 2] = static_cast<uint8_t>(std::clamp(px[i].z * 255.0f, 0.0f, 255.0f));
    }
    return out;
}

// 8-bit RGB in, as hvrt_render_views_rgb8 / hvrt_quantise_rgb8_device deliver it (quantised on the GPU)
inline bool saveBytesToPPM(const std::string& filename, const uint8_t* q, int width, int height) {
    std::ofstream f(filename, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << width << " " << height << "\n255\n";
    f.write(reinterpret_cast<const char*>(q), (std::streamsize)((size_t)width * height * 3));
    return (bool)f;
}
inline bool saveToPPM(const std::string& filename, const std::vector<Vec3>& pixels, int width, int height) {
    return saveBytesToPPM(filename, quantise(pixels).data(), width, height);
}

namespace png_detail {
inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    // slicing by 8: eight table look-ups per eight bytes instead of a dependent look-up per byte (a 4K frame is 25 MB)
    static uint32_t table[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = table[0][table[t - 1][i] & 0xFF] ^ (table[t - 1][i] >> 8);
        init = true;
    }
    crc = ~crc;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint32_t lo = crc ^ ((uint32_t)p[i] | (uint32_t)p[i + 1] << 8 | (uint32_t)p[i + 2] << 16 | (uint32_t)p[i + 3] << 24);
        const uint32_t hi = (uint32_t)p[i + 4] | (uint32_t)p[i + 5] << 8 | (uint32_t)p[i + 6] << 16 | (uint32_t)p[i + 7] << 24;
        crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
              table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    }
    for (; i < n; ++i) crc = table[0][(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
inline void chunk(std::ofstream& f, const char* type, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> buf;
    be32(buf, (uint32_t)data.size());
    buf.insert(buf.end(), type, type + 4);
    buf.insert(buf.end(), data.begin(), data.end());
    uint32_t c = crc32(buf.data() + 4, buf.size() - 4);
    be32(buf, c);
    f.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)buf.size());
}
}  // namespace png_detail

inline bool saveBytesToPNG(const std::string& filename, const uint8_t* q, int width, int height) {
    using namespace png_detail;
    std::ofstream f(filename, std::ios::binary);
    if (!f) return false;
    const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    f.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width);
    be32(ihdr, (uint32_t)height);
    ihdr.insert(ihdr.end(), { 8, 2, 0, 0, 0 });   // 8-bit RGB
    chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);   // filter: none
        raw.insert(raw.end(), q + (size_t)y * width * 3, q + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z = { 0x78, 0x01 };
    uint32_t a = 1, b = 0;   // adler32
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8);
        z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n;) {       // adler32 with the modulo taken every 5552 bytes (the most that cannot overflow 32 bits)
            const size_t m = std::min<size_t>(n - i, 5552);
            for (size_t k = 0; k < m; ++k) { a += raw[pos + i + k]; b += a; }
            a %= 65521; b %= 65521;
            i += m;
        }
        pos += n;
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    return (bool)f;
}
inline bool saveToPNG(const std::string& filename, const std::vector<Vec3>& pixels, int width, int height) {
    return saveBytesToPNG(filename, quantise(pixels).data(), width, height);
}

}  // namespace cg_datastructures
