#include "cartslam_amd/datasource.hpp"

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <vector>

namespace cart::sources {
namespace {
struct HostImage {
    int w = 0, h = 0, channels = 0;
    std::vector<uint8_t> data;
};

// binary PGM (P5) / PPM (P6), maxval 255. PPM data is RGB on disk and is swapped to BGR like cv::imread delivers it.
bool readPnm(const std::string &path, HostImage &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5" && magic != "P6") throw std::runtime_error("unsupported image file " + path);
    auto nextInt = [&]() {
        int c;
        while ((c = f.peek()) == '#' || std::isspace(c)) {
            if (c == '#') { std::string line; std::getline(f, line); } else f.get();
        }
        int v; f >> v; return v;
    };
    out.w = nextInt(); out.h = nextInt();
    const int maxv = nextInt();
    if (maxv != 255) throw std::runtime_error("only 8-bit PGM/PPM supported: " + path);
    f.get();
    out.channels = magic == "P6" ? 3 : 1;
    out.data.resize((size_t)out.w * out.h * out.channels);
    f.read(reinterpret_cast<char *>(out.data.data()), (std::streamsize)out.data.size());
    if (!f) throw std::runtime_error("truncated image file " + path);
    if (out.channels == 3)
        for (size_t i = 0; i < out.data.size(); i += 3) std::swap(out.data[i], out.data[i + 2]);
    return true;
}

std::string framePath(const std::string &dir, int cam, int frame, const char *ext) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "/image_%d/%06d.%s", cam, frame, ext);
    return dir + buf;
}

bool readFrame(const std::string &dir, int cam, int frame, HostImage &img) {
    return readPnm(framePath(dir, cam, frame, "pgm"), img) || readPnm(framePath(dir, cam, frame, "ppm"), img);
}
}  // namespace

RawSequenceDataSource::RawSequenceDataSource(const std::string &basePath, int sequence) : DataSource(Size{}) {
    char seq[16];
    std::snprintf(seq, sizeof(seq), "%02d", sequence);
    dir = basePath + "/sequences/" + seq;  // kitti.cpp:93-96
    HostImage first;
    if (!readFrame(dir, 2, 0, first)) throw std::runtime_error("Could not read first frame under " + dir);
    imageSize.width = first.w; imageSize.height = first.h;
}

bool RawSequenceDataSource::isFinished() {
    HostImage probe;
    std::ifstream a(framePath(dir, 2, currentFrame, "pgm")), b(framePath(dir, 2, currentFrame, "ppm"));
    return !a.is_open() && !b.is_open();
}

std::shared_ptr<DataElement> RawSequenceDataSource::getNextInternal() {
    HostImage l, r;
    if (!readFrame(dir, 2, currentFrame, l) || !readFrame(dir, 3, currentFrame, r)) throw std::runtime_error("Could not read frame " + std::to_string(currentFrame));
    ++currentFrame;
    const int type = l.channels == 3 ? CV_8UC3 : CV_8UC1;
    image_t dl(l.h, l.w, type), dr(r.h, r.w, type);
    dl.upload(l.data.data(), (size_t)l.w * l.channels);  // kitti.cpp:163-164
    dr.upload(r.data.data(), (size_t)r.w * r.channels);
    return std::make_shared<StereoDataElement>(dl, dr);
}
}  // namespace cart::sources
