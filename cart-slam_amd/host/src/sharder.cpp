#include "cartslam_amd/sharder.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdexcept>

namespace cart {
namespace {
void hipOk(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void ncclOk(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
void cartOk(int rc, cart_engine *e, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + cart_last_error(e));
}
template <typename T>
T *devAlloc(size_t n) {
    void *p = nullptr;
    hipOk(hipMalloc(&p, n * sizeof(T)), "hipMalloc");
    return static_cast<T *>(p);
}
}  // namespace

struct FrameSharder::Rank {
    int device = 0;
    cart_engine *engine = nullptr;
    cart_plane_schedule *schedule = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    uint8_t *left = nullptr, *right = nullptr, *planes = nullptr;   // this GPU's share, [framesPerGpu][h][w]
    int16_t *disparity = nullptr, *derivative = nullptr;
    int32_t *hist = nullptr;          // [framesPerGpu][256]
    int32_t *histByRank = nullptr;    // [gpus][framesPerGpu][256], as the all-gather delivers them
    int32_t *histById = nullptr;      // [capacity][256], frame-id order
    cart_plane_params *paramsAll = nullptr, *paramsMine = nullptr;
    ~Rank() {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (schedule) cart_plane_schedule_destroy(schedule);
        if (engine) cart_engine_destroy(engine);
        for (void *p : {(void *)left, (void *)right, (void *)planes, (void *)disparity, (void *)derivative, (void *)hist, (void *)histByRank, (void *)histById,
                        (void *)paramsAll, (void *)paramsMine})
            if (p) (void)hipFree(p);
        if (comm) (void)ncclCommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

FrameSharder::FrameSharder(const std::vector<int> &devices, cart_engine_params params, int framesPerGpu, int updateInterval, int resetInterval)
    : framesPerGpu(framesPerGpu), width(params.width), height(params.height) {
    if (devices.empty() || framesPerGpu < 1) throw std::invalid_argument("FrameSharder needs at least one GPU and one frame per GPU");
    const int n = (int)devices.size();
    const size_t npx = (size_t)width * height;
    std::vector<ncclComm_t> comms(n);
    ncclOk(ncclCommInitAll(comms.data(), n, devices.data()), "ncclCommInitAll");
    for (int r = 0; r < n; ++r) {
        auto rank = std::make_unique<Rank>();
        rank->device = devices[r];
        rank->comm = comms[r];
        hipOk(hipSetDevice(devices[r]), "hipSetDevice");
        hipOk(hipStreamCreateWithFlags(&rank->stream, hipStreamNonBlocking), "hipStreamCreate");
        params.device_id = devices[r];
        params.max_inflight = framesPerGpu;
        if (cart_engine_create(&params, &rank->engine) != 0) throw std::runtime_error(std::string("cart_engine_create: ") + cart_last_error(nullptr));
        cartOk(cart_plane_schedule_create(rank->engine, /*histogram_peak*/ 1, nullptr, updateInterval, resetInterval, &rank->schedule), rank->engine,
               "cart_plane_schedule_create");
        rank->left = devAlloc<uint8_t>(framesPerGpu * npx);
        rank->right = devAlloc<uint8_t>(framesPerGpu * npx);
        rank->planes = devAlloc<uint8_t>(framesPerGpu * npx);
        rank->disparity = devAlloc<int16_t>(framesPerGpu * npx);
        rank->derivative = devAlloc<int16_t>(framesPerGpu * npx);
        rank->hist = devAlloc<int32_t>((size_t)framesPerGpu * 256);
        rank->histByRank = devAlloc<int32_t>((size_t)n * framesPerGpu * 256);
        rank->histById = devAlloc<int32_t>((size_t)n * framesPerGpu * 256);
        rank->paramsAll = devAlloc<cart_plane_params>((size_t)n * framesPerGpu);
        rank->paramsMine = devAlloc<cart_plane_params>((size_t)framesPerGpu);
        ranks.push_back(std::move(rank));
    }
}

FrameSharder::~FrameSharder() = default;

void FrameSharder::processSequence(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes) {
    const int n = gpus();
    if (!left || !right || !disparity || !planes) throw std::invalid_argument("NULL image pointer");
    if (nFrames < 1 || nFrames % n || nFrames > capacity())
        throw std::invalid_argument("sequence length " + std::to_string(nFrames) + " is not a positive multiple of the " + std::to_string(n) +
                                    " GPUs within the capacity of " + std::to_string(capacity()) + " frames");
    const int local = nFrames / n;
    const size_t npx = (size_t)width * height;
    Rank &root = *ranks[0];

    // ---- scatter: frame k -> GPU k mod n, local index k / n.  One group: every send has its receive posted with it.
    ncclOk(ncclGroupStart(), "ncclGroupStart");
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < local; ++j) {
            const size_t k = (size_t)j * n + r;
            ncclOk(ncclSend(left + k * npx, npx, ncclUint8, r, root.comm, root.stream), "ncclSend");
            ncclOk(ncclRecv(ranks[r]->left + (size_t)j * npx, npx, ncclUint8, 0, ranks[r]->comm, ranks[r]->stream), "ncclRecv");
            ncclOk(ncclSend(right + k * npx, npx, ncclUint8, r, root.comm, root.stream), "ncclSend");
            ncclOk(ncclRecv(ranks[r]->right + (size_t)j * npx, npx, ncclUint8, 0, ranks[r]->comm, ranks[r]->stream), "ncclRecv");
        }
    ncclOk(ncclGroupEnd(), "ncclGroupEnd");

    // ---- every GPU: disparity and plane derivative + per-frame histograms of its share
    for (auto &rk : ranks) {
        hipOk(hipSetDevice(rk->device), "hipSetDevice");
        cartOk(cart_compute_disparity_batch(rk->engine, local, rk->left, (size_t)width, npx, rk->right, (size_t)width, npx, 1, rk->disparity, (size_t)width * 2,
                                            npx * 2, rk->stream),
               rk->engine, "cart_compute_disparity_batch");
        hipOk(hipMemsetAsync(rk->hist, 0, (size_t)local * 256 * sizeof(int32_t), rk->stream), "hipMemsetAsync");
        cartOk(cart_plane_derivative_hist(rk->engine, local, rk->disparity, (size_t)width * 2, npx * 2, rk->derivative, (size_t)width * 2, npx * 2, rk->hist, 256,
                                          rk->stream),
               rk->engine, "cart_plane_derivative_hist");
    }

    // ---- the path's only exchange step: all-gather of the per-frame histograms (1 KB per frame)
    ncclOk(ncclGroupStart(), "ncclGroupStart");
    for (auto &rk : ranks) ncclOk(ncclAllGather(rk->hist, rk->histByRank, (size_t)local * 256, ncclInt32, rk->comm, rk->stream), "ncclAllGather");
    ncclOk(ncclGroupEnd(), "ncclGroupEnd");

    // ---- every GPU: histograms into frame-id order ([rank][j] -> j * n + rank), schedule replay for the whole sequence,
    //      its own frames' parameters (every n-th row), classification
    for (int r = 0; r < n; ++r) {
        Rank &rk = *ranks[r];
        hipOk(hipSetDevice(rk.device), "hipSetDevice");
        for (int src = 0; src < n; ++src)
            hipOk(hipMemcpy2DAsync(rk.histById + (size_t)src * 256, (size_t)n * 1024, rk.histByRank + (size_t)src * local * 256, 1024, 1024, local,
                                   hipMemcpyDeviceToDevice, rk.stream),
                  "hipMemcpy2DAsync");
        cartOk(cart_plane_schedule_advance(rk.schedule, nextId, nFrames, rk.histById, rk.paramsAll, rk.stream), rk.engine, "cart_plane_schedule_advance");
        hipOk(hipMemcpy2DAsync(rk.paramsMine, sizeof(cart_plane_params), rk.paramsAll + r, (size_t)n * sizeof(cart_plane_params), sizeof(cart_plane_params), local,
                               hipMemcpyDeviceToDevice, rk.stream),
              "hipMemcpy2DAsync");
        cartOk(cart_plane_classify_dev(rk.engine, local, rk.derivative, (size_t)width * 2, npx * 2, rk.paramsMine, 1, rk.planes, (size_t)width, npx, rk.stream),
               rk.engine, "cart_plane_classify_dev");
    }

    // ---- gather: outputs back to GPU 0 in sequence order (RCCL has no 16-bit integer type: disparities travel as bytes)
    ncclOk(ncclGroupStart(), "ncclGroupStart");
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < local; ++j) {
            const size_t k = (size_t)j * n + r;
            ncclOk(ncclSend(ranks[r]->disparity + (size_t)j * npx, npx * 2, ncclUint8, 0, ranks[r]->comm, ranks[r]->stream), "ncclSend");
            ncclOk(ncclRecv(disparity + k * npx, npx * 2, ncclUint8, r, root.comm, root.stream), "ncclRecv");
            ncclOk(ncclSend(ranks[r]->planes + (size_t)j * npx, npx, ncclUint8, 0, ranks[r]->comm, ranks[r]->stream), "ncclSend");
            ncclOk(ncclRecv(planes + k * npx, npx, ncclUint8, r, root.comm, root.stream), "ncclRecv");
        }
    ncclOk(ncclGroupEnd(), "ncclGroupEnd");
    for (auto &rk : ranks) {
        hipOk(hipSetDevice(rk->device), "hipSetDevice");
        hipOk(hipStreamSynchronize(rk->stream), "hipStreamSynchronize");
    }
    nextId += nFrames;
}
}  // namespace cart
