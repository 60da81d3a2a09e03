// png.hpp -- minimal PNG reader (8-bit gray / gray+alpha / RGB / RGBA, non-interlaced) on top of zlib's inflate.
// The reference reads KITTI frames with cv::imread (src/sources/kitti.cpp:131,152-153); OpenCV / libpng are not
// available here, zlib is.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cart::util {
struct HostImage {
    int w = 0, h = 0, channels = 0;  // channels: 1 = gray, 3 = BGR (cv::imread order)
    std::vector<uint8_t> data;
};
// returns false if the file does not exist; throws std::runtime_error on a malformed / unsupported file
bool readPng(const std::string &path, HostImage &out);
}  // namespace cart::util
