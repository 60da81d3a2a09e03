#include "cartslam_amd/image.hpp"

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace cart {
namespace {
void check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Caching allocator for frame-sized device images.  hipMalloc / hipFree cost milliseconds and synchronise the device;
// the reference allocates fresh GpuMats in every module call (disparity.cu:70, planeseg.cu:260,293), which on this
// runtime capped the frame loop at ~60 frames/s.  Blocks are recycled per exact byte size and freed at process exit.
class DevicePool {
   public:
    static DevicePool &instance() { static DevicePool p; return p; }
    void *acquire(size_t bytes) {
        {
            std::lock_guard<std::mutex> lock(mutex);
            auto it = free_.find(bytes);
            if (it != free_.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                return p;
            }
        }
        void *p = nullptr;
        check(hipMalloc(&p, bytes), "hipMalloc");
        return p;
    }
    void release(void *p, size_t bytes) {
        std::lock_guard<std::mutex> lock(mutex);
        free_[bytes].push_back(p);
    }
    ~DevicePool() {
        for (auto &kv : free_)
            for (void *p : kv.second) (void)hipFree(p);
    }

   private:
    std::mutex mutex;
    std::map<size_t, std::vector<void *>> free_;
};

// Pinned staging block for image transfers.  A pitched hipMemcpy2D from pageable memory is issued row by row by the
// runtime (375 rows -> ~2.7 ms per KITTI image, which capped the frame loop at ~180 frames/s); rows are repacked to the
// device pitch on the host and moved with ONE contiguous copy instead.
class PinnedPool {
   public:
    static PinnedPool &instance() { static PinnedPool p; return p; }
    void *acquire(size_t bytes) {
        {
            std::lock_guard<std::mutex> lock(mutex);
            auto it = free_.find(bytes);
            if (it != free_.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                return p;
            }
        }
        void *p = nullptr;
        check(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc");
        return p;
    }
    void release(void *p, size_t bytes) {
        std::lock_guard<std::mutex> lock(mutex);
        free_[bytes].push_back(p);
    }
    ~PinnedPool() {
        for (auto &kv : free_)
            for (void *p : kv.second) (void)hipHostFree(p);
    }

   private:
    std::mutex mutex;
    std::map<size_t, std::vector<void *>> free_;
};

struct PinnedLease {
    explicit PinnedLease(size_t n) : bytes(n), ptr(static_cast<uint8_t *>(PinnedPool::instance().acquire(n))) {}
    ~PinnedLease() { PinnedPool::instance().release(ptr, bytes); }
    size_t bytes;
    uint8_t *ptr;
};
}  // namespace

void DeviceImage::create(int r, int c, int t) {
    const size_t pitch = (((size_t)c * elemSize(t)) + 255) & ~(size_t)255;  // 256-byte aligned rows, like cudaMallocPitch
    const size_t bytes = pitch * (size_t)r;
    void *p = DevicePool::instance().acquire(bytes);
    storage = std::shared_ptr<void>(p, [bytes](void *q) { DevicePool::instance().release(q, bytes); });
    data = p; step = pitch; rows = r; cols = c; type_ = t;
}

void DeviceImage::upload(const void *host, size_t host_step) {
    if (empty()) return;
    const size_t row_bytes = (size_t)cols * elemSize(type_);
    PinnedLease stage(step * (size_t)rows);
    for (int y = 0; y < rows; ++y)
        std::memcpy(stage.ptr + (size_t)y * step, static_cast<const uint8_t *>(host) + (size_t)y * host_step, row_bytes);
    check(hipMemcpy(data, stage.ptr, stage.bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}

void DeviceImage::download(void *host, size_t host_step) const {
    if (empty()) return;
    const size_t row_bytes = (size_t)cols * elemSize(type_);
    PinnedLease stage(step * (size_t)rows);
    check(hipMemcpy(stage.ptr, data, stage.bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    for (int y = 0; y < rows; ++y)
        std::memcpy(static_cast<uint8_t *>(host) + (size_t)y * host_step, stage.ptr + (size_t)y * step, row_bytes);
}

std::vector<uint8_t> DeviceImage::downloadTight() const {
    std::vector<uint8_t> out((size_t)rows * cols * elemSize(type_));
    if (!out.empty()) download(out.data(), (size_t)cols * elemSize(type_));
    return out;
}

void DeviceImage::setTo(int byte_value) { check(hipMemset2D(data, step, byte_value, (size_t)cols * elemSize(type_), rows), "hipMemset2D"); }
}  // namespace cart
