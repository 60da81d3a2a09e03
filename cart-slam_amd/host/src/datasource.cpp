#include "cartslam_amd/datasource.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>

#include "cartslam_amd/png.hpp"
#include "cart_engine.h"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <vector>

namespace cart::sources {
namespace {
using cart::util::HostImage;

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
    return cart::util::readPng(framePath(dir, cam, frame, "png"), img) || readPnm(framePath(dir, cam, frame, "pgm"), img) ||
           readPnm(framePath(dir, cam, frame, "ppm"), img);
}

bool frameExists(const std::string &dir, int cam, int frame) {
    for (const char *ext : {"png", "pgm", "ppm"})
        if (std::ifstream(framePath(dir, cam, frame, ext)).is_open()) return true;
    return false;
}

// A projection row of KITTI's calib.txt: "P<camera>: m00 m01 ... m23" (3x4, row-major).  Rows that are not a
// projection matrix ("Tr: ...") or do not hold exactly twelve numbers are skipped, as kitti.cpp:28-86 skips them.
struct ProjectionRow {
    int camera = -1;
    float m[12] = {};
    float fx() const { return m[0]; }
    float cx() const { return m[2]; }
    float cy() const { return m[6]; }
    float baseline() const { return -m[3] / m[0]; }   // m[3] = -fx * baseline (kitti.cpp:80)
};

bool parseProjectionRow(const std::string &line, ProjectionRow &row) {
    std::istringstream in(line);
    std::string label;
    if (!(in >> label) || label.size() < 3 || label.front() != 'P' || label.back() != ':') return false;
    ProjectionRow parsed;
    parsed.camera = std::atoi(label.c_str() + 1);
    int count = 0;
    for (float v; in >> v; ++count)
        if (count < 12) parsed.m[count] = v;
    if (count != 12 || !in.eof()) return false;   // too few, too many, or something that is not a number
    row = parsed;
    return true;
}
}  // namespace

// Read-ahead: `workers` threads read and decode the next frames (a 1242x375 KITTI PNG costs milliseconds of inflate, the
// reference's cv::imread pays it inside getNext, kitti.cpp:155-161) while the frame loop is busy with earlier ones;
// getNextInternal then only uploads.  At most `depth` decoded frames wait in host memory.  Errors travel with the frame
// they belong to and are thrown by the getNext call that asks for it, exactly where the synchronous reader throws them.
class KITTIDataSource::ReadAhead {
   public:
    struct Frame { HostImage left, right; std::string error; };
    ReadAhead(std::string dir, int workers, int depth) : dir(std::move(dir)), depth(depth) {
        for (int i = 0; i < workers; ++i) threads.emplace_back([this] { work(); });
    }
    ~ReadAhead() {
        { std::lock_guard<std::mutex> lock(mutex); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
    Frame take(int frame) {
        std::unique_lock<std::mutex> lock(mutex);
        consumer = frame;   // frames before this one are never asked for again
        cv.notify_all();
        cv.wait(lock, [&] { return ready.count(frame) != 0 || frame > lastFrame; });
        if (!ready.count(frame)) { Frame none; none.error = "Could not read frame " + std::to_string(frame); return none; }  // past the end of the sequence
        Frame out = std::move(ready[frame]);
        ready.erase(frame);
        consumer = frame + 1;
        cv.notify_all();
        return out;
    }

   private:
    void work() {
        for (;;) {
            int frame;
            {
                std::unique_lock<std::mutex> lock(mutex);
                cv.wait(lock, [&] { return stop || (claim < consumer + depth && claim <= lastFrame); });
                if (stop) return;
                frame = claim++;
            }
            Frame f;
            bool exists = true;
            try {
                exists = readFrame(dir, 2, frame, f.left);
                if (exists && !readFrame(dir, 3, frame, f.right)) f.error = "Could not read frame " + std::to_string(frame);
            } catch (const std::exception &e) { f.error = e.what(); }
            std::lock_guard<std::mutex> lock(mutex);
            if (!exists) { f.error = "Could not read frame " + std::to_string(frame); lastFrame = std::min(lastFrame, frame); }  // end of the sequence: stop claiming
            ready[frame] = std::move(f);
            cv.notify_all();
        }
    }
    const std::string dir;
    const int depth;
    std::mutex mutex;
    std::condition_variable cv;
    std::map<int, Frame> ready;
    int claim = 0, consumer = 0, lastFrame = 1 << 30;
    bool stop = false;
    std::vector<std::thread> threads;
};

KITTIDataSource::~KITTIDataSource() = default;

KITTIDataSource::KITTIDataSource(const std::string &basePath, int sequence, Size requested) : DataSource(requested) {
    char seq[16];
    std::snprintf(seq, sizeof(seq), "%02d", sequence);
    dir = basePath + "/sequences/" + seq;  // kitti.cpp:89-90
    const std::string calibPath = dir + "/calib.txt";
    std::ifstream calib(calibPath);
    ProjectionRow cams[2];   // P2 = left colour camera, P3 = right
    bool haveCalib = false;
    if (calib.is_open()) {
        for (std::string line; std::getline(calib, line);) {
            ProjectionRow row;
            if (parseProjectionRow(line, row) && (row.camera == 2 || row.camera == 3)) cams[row.camera - 2] = row;
        }
        if (cams[0].camera != 2 || cams[1].camera != 3) throw std::runtime_error("Failed to read calibration file");  // kitti.cpp:126-128
        haveCalib = true;
    } else if (std::ifstream(framePath(dir, 2, 0, "png")).is_open()) {
        throw std::runtime_error("Failed to open calibration file at " + calibPath + ": " + std::strerror(errno));  // kitti.cpp:100-103
    }
    HostImage first;   // "a bit hacky, but we need to read the first image to get the image size" (kitti.cpp:129-135)
    if (!readFrame(dir, 2, 0, first)) throw std::runtime_error("Could not read first frame under " + dir);
    fileSize.width = first.w; fileSize.height = first.h;
    if (imageSize.width == 0 || imageSize.height == 0) imageSize = fileSize;
    if (haveCalib) {
        const float scaleWidth = static_cast<float>(imageSize.width) / fileSize.width;      // kitti.cpp:137-138
        const float scaleHeight = static_cast<float>(imageSize.height) / fileSize.height;
        const ProjectionRow &l = cams[0], &r = cams[1];
        const float base = l.baseline();
        float *Q = intrinsics.Q;  // kitti.cpp:140-148
        Q[0 * 4 + 3] = -l.cx() * scaleWidth;
        Q[1 * 4 + 3] = -l.cy() * scaleHeight;
        Q[2 * 4 + 2] = 0;
        Q[2 * 4 + 3] = l.fx() * scaleWidth;
        Q[3 * 4 + 2] = (float)(-1.0 / base);
        Q[3 * 4 + 3] = ((l.cx() - r.cx()) * scaleWidth / base);
    }
    // CARTSLAM_READAHEAD = decoder threads (default 4, 0 = read inside getNext like the reference)
    const char *env = std::getenv("CARTSLAM_READAHEAD");
    readAheadWorkers = env ? std::max(0, std::min(16, std::atoi(env))) : 4;
}

bool KITTIDataSource::isFinished() { return !frameExists(dir, 2, currentFrame); }

std::shared_ptr<DataElement> KITTIDataSource::getNextInternal() {
    HostImage l, r;
    if (readAheadWorkers > 0) {
        if (!readAhead) readAhead = std::make_unique<ReadAhead>(dir, readAheadWorkers, 2 * readAheadWorkers);
        ReadAhead::Frame f = readAhead->take(currentFrame);
        if (!f.error.empty()) throw std::runtime_error(f.error);
        l = std::move(f.left); r = std::move(f.right);
    } else if (!readFrame(dir, 2, currentFrame, l) || !readFrame(dir, 3, currentFrame, r)) {
        throw std::runtime_error("Could not read frame " + std::to_string(currentFrame));
    }
    if (l.w != r.w || l.h != r.h || l.channels != r.channels)
        throw std::runtime_error("Frame " + std::to_string(currentFrame) + ": left and right image differ in size or channel count");
    ++currentFrame;
    const int type = l.channels == 3 ? CV_8UC3 : CV_8UC1;
    image_t dl(l.h, l.w, type), dr(r.h, r.w, type);
    dl.upload(l.data.data(), (size_t)l.w * l.channels);  // kitti.cpp:163-164
    dr.upload(r.data.data(), (size_t)r.w * r.channels);
    if (imageSize.width != l.w || imageSize.height != l.h) {   // kitti.cpp:169-172: cv::cuda::resize(..., INTER_LINEAR), oracle S16
        image_t rl(imageSize.height, imageSize.width, type), rr(imageSize.height, imageSize.width, type);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (cart_resize_linear(dev, dl.ptr<uint8_t>(), dl.step, l.w, l.h, l.channels, rl.ptr<uint8_t>(), rl.step, imageSize.width, imageSize.height, nullptr) != 0 ||
            cart_resize_linear(dev, dr.ptr<uint8_t>(), dr.step, r.w, r.h, r.channels, rr.ptr<uint8_t>(), rr.step, imageSize.width, imageSize.height, nullptr) != 0)
            throw std::runtime_error(std::string("cart_resize_linear: ") + cart_last_error(nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) throw std::runtime_error("resize failed");
        dl = rl; dr = rr;
    }
    return std::make_shared<StereoDataElement>(dl, dr);
}
}  // namespace cart::sources
