// image.hpp -- the subset of cv::cuda::GpuMat the hot path touches (reference: `typedef cv::cuda::GpuMat image_t`,
// include/datasource.hpp:9): a pitched device image with OpenCV's type codes, owned by shared storage.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

namespace cart {

// OpenCV type codes (depth + ((channels-1) << 3)), so `type()` checks read like the reference's (planeseg.cu:255)
enum : int { CV_8UC1 = 0, CV_8UC3 = 16, CV_16UC1 = 2, CV_16SC1 = 3, CV_16SC2 = 11, CV_32SC1 = 4, CV_32SC2 = 12 };

inline size_t elemSize(int type) {
    static const size_t depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};  // CV_8U,8S,16U,16S,32S,32F,64F,16F
    return depth_bytes[type & 7] * (size_t)((type >> 3) + 1);
}

class DeviceImage {
   public:
    DeviceImage() = default;
    DeviceImage(int rows, int cols, int type) { create(rows, cols, type); }

    void create(int rows, int cols, int type);           // pitched device allocation (hipMallocPitch)
    void upload(const void *host, size_t host_step);     // blocking H2D
    void download(void *host, size_t host_step) const;   // blocking D2H
    std::vector<uint8_t> downloadTight() const;
    void setTo(int byte_value);

    bool empty() const { return !storage || rows == 0 || cols == 0; }
    int type() const { return type_; }
    template <typename T> T *ptr() const { return static_cast<T *>(data); }

    void *data = nullptr;
    size_t step = 0;  // bytes per row, like GpuMat::step
    int rows = 0, cols = 0;

   private:
    int type_ = 0;
    std::shared_ptr<void> storage;  // shared like a GpuMat header copy
};

typedef DeviceImage image_t;

struct Size {
    int width = 0, height = 0;
};

}  // namespace cart
