#include "cartslam_amd/image.hpp"

#include <hip/hip_runtime_api.h>

#include <string>

namespace cart {
namespace {
void check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace

void DeviceImage::create(int r, int c, int t) {
    void *p = nullptr;
    size_t pitch = 0;
    check(hipMallocPitch(&p, &pitch, (size_t)c * elemSize(t), (size_t)r), "hipMallocPitch");
    storage = std::shared_ptr<void>(p, [](void *q) { (void)hipFree(q); });
    data = p; step = pitch; rows = r; cols = c; type_ = t;
}

void DeviceImage::upload(const void *host, size_t host_step) {
    check(hipMemcpy2D(data, step, host, host_step, (size_t)cols * elemSize(type_), rows, hipMemcpyHostToDevice), "hipMemcpy2D H2D");
}

void DeviceImage::download(void *host, size_t host_step) const {
    check(hipMemcpy2D(host, host_step, data, step, (size_t)cols * elemSize(type_), rows, hipMemcpyDeviceToHost), "hipMemcpy2D D2H");
}

std::vector<uint8_t> DeviceImage::downloadTight() const {
    std::vector<uint8_t> out((size_t)rows * cols * elemSize(type_));
    if (!out.empty()) download(out.data(), (size_t)cols * elemSize(type_));
    return out;
}

void DeviceImage::setTo(int byte_value) { check(hipMemset2D(data, step, byte_value, (size_t)cols * elemSize(type_), rows), "hipMemset2D"); }
}  // namespace cart
