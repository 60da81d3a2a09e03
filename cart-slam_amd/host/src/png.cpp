#include "cartslam_amd/png.hpp"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace cart::util {
namespace {
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

bool readPng(const std::string &path, HostImage &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char *type = reinterpret_cast<const char *>(&file[pos + 4]);
        if (pos + 12 + len > file.size()) throw std::runtime_error("truncated PNG chunk: " + path);
        const uint8_t *body = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error("short PNG header chunk: " + path);
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    int sch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (w > 16384 || h > 16384) throw std::runtime_error("PNG larger than 16384 x 16384 (the engine's limit): " + path);   // before any allocation sized by the header
    if (w <= 0 || h <= 0 || depth != 8 || sch == 0 || interlace != 0) throw std::runtime_error("unsupported PNG (need 8-bit, non-interlaced, gray/RGB[A]): " + path);
    const size_t stride = (size_t)w * sch;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) throw std::runtime_error("PNG inflate failed: " + path);
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {  // undo the per-scanline filters (PNG spec section 9)
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)sch ? cur[i - sch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)sch) ? up[i - sch] : 0;
            int v = src[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error("bad PNG filter type: " + path);
            }
            cur[i] = (uint8_t)v;
        }
    }
    // cv::imread(path) == IMREAD_COLOR: always 3-channel BGR, gray replicated, alpha dropped
    out.w = w; out.h = h; out.channels = 3;
    out.data.resize((size_t)w * h * 3);
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        const uint8_t *s = &img[p * sch];
        uint8_t *d = &out.data[p * 3];
        if (sch <= 2) { d[0] = d[1] = d[2] = s[0]; } else { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; }
    }
    return true;
}
}  // namespace cart::util
