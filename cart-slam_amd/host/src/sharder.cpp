#include "cartslam_amd/sharder.hpp"

#include <cstdio>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <stdexcept>
#include <thread>

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
    hipOk(hipMalloc(&p, (n ? n : 1) * sizeof(T)), "hipMalloc");
    return static_cast<T *>(p);
}
// communicators of one ncclCommInitAll: the ones no Rank has taken over yet are destroyed on the way out of a failed constructor
struct CommSet {
    std::vector<ncclComm_t> comms;
    explicit CommSet(const std::vector<int> &devices) : comms(devices.size(), nullptr) {
        ncclOk(ncclCommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
    }
    ncclComm_t take(int r) { ncclComm_t c = comms[r]; comms[r] = nullptr; return c; }
    ~CommSet() {
        for (ncclComm_t c : comms)
            if (c) (void)ncclCommDestroy(c);
    }
};
// an exception between ncclGroupStart and ncclGroupEnd must not leave the group open
struct NcclGroup {
    bool open = true;
    NcclGroup() { ncclOk(ncclGroupStart(), "ncclGroupStart"); }
    void end() { open = false; ncclOk(ncclGroupEnd(), "ncclGroupEnd"); }
    ~NcclGroup() { if (open) (void)ncclGroupEnd(); }
};
}  // namespace

struct FrameSharder::Rank {
    int device = 0;
    cart_engine *engine = nullptr;
    cart_plane_schedule *schedule = nullptr;
    hipStream_t compute = nullptr, copy = nullptr;   // kernels / every RCCL operation (scatter, histogram all-gather, gather) + the packing copies
    // ONE communicator per GPU, and every operation of it is enqueued on the copy stream: a communicator's operations execute in the order they
    // were enqueued, the same order on every GPU (one host thread enqueues them), so at no time are two collectives of different communicators in
    // flight on a device (rounds 3-4 ran the all-gather on a second communicator on the compute stream beside the send / recv traffic of the copy
    // stream: concurrent collectives of two communicators on one device are not guaranteed to make progress, and N > 1 had never run)
    ncclComm_t comm = nullptr;
    // two buffer sets (sequence i uses set i & 1): this GPU's share, [framesPerGpu][h][w]
    uint8_t *left[2] = {}, *right[2] = {}, *planes[2] = {};
    int16_t *disparity[2] = {}, *derivative[2] = {};
    int32_t *hist[2] = {};          // [framesPerGpu][256]
    int32_t *histByRank = nullptr;  // [gpus][framesPerGpu][256], as the all-gather delivers them
    int32_t *histById = nullptr;    // [capacity][256], frame-id order
    cart_plane_params *paramsAll = nullptr, *paramsMine = nullptr;
    // GPU 0 only: peers' frames packed per peer ([gpus][framesPerGpu][h][w]) on their way out / back
    uint8_t *stageLeft[2] = {}, *stageRight[2] = {}, *stagePlanes[2] = {};
    int16_t *stageDisparity[2] = {};
    hipEvent_t scattered[2] = {}, histReady[2] = {}, histGathered[2] = {}, computed[2] = {}, gathered[2] = {}, inputsReady = nullptr;
    ~Rank() {
        (void)hipSetDevice(device);
        for (hipStream_t s : {compute, copy})
            if (s) (void)hipStreamSynchronize(s);
        if (schedule) cart_plane_schedule_destroy(schedule);
        if (engine) cart_engine_destroy(engine);
        for (int b = 0; b < 2; ++b) {
            for (void *p : {(void *)left[b], (void *)right[b], (void *)planes[b], (void *)disparity[b], (void *)derivative[b], (void *)hist[b],
                            (void *)stageLeft[b], (void *)stageRight[b], (void *)stagePlanes[b], (void *)stageDisparity[b]})
                if (p) (void)hipFree(p);
            for (hipEvent_t e : {scattered[b], histReady[b], histGathered[b], computed[b], gathered[b]})
                if (e) (void)hipEventDestroy(e);
        }
        for (void *p : {(void *)histByRank, (void *)histById, (void *)paramsAll, (void *)paramsMine})
            if (p) (void)hipFree(p);
        if (inputsReady) (void)hipEventDestroy(inputsReady);
        if (comm) (void)ncclCommDestroy(comm);
        for (hipStream_t s : {compute, copy})
            if (s) (void)hipStreamDestroy(s);
    }
};

FrameSharder::FrameSharder(const std::vector<int> &devices, cart_engine_params params, int framesPerGpu, int updateInterval, int resetInterval, int placementTries)
    : framesPerGpu(framesPerGpu), width(params.width), height(params.height) {
    if (devices.empty() || framesPerGpu < 1) throw std::invalid_argument("FrameSharder needs at least one GPU and one frame per GPU");
    const int n = (int)devices.size();
    const size_t npx = (size_t)width * height, share = (size_t)framesPerGpu * npx;
    CommSet comms(devices);
    for (int r = 0; r < n; ++r) {
        auto rank = std::make_unique<Rank>();
        rank->device = devices[r];
        rank->comm = comms.take(r);
        hipOk(hipSetDevice(devices[r]), "hipSetDevice");
        hipOk(hipStreamCreateWithFlags(&rank->compute, hipStreamNonBlocking), "hipStreamCreate");
        hipOk(hipStreamCreateWithFlags(&rank->copy, hipStreamNonBlocking), "hipStreamCreate");
        params.device_id = devices[r];
        params.max_inflight = framesPerGpu;
        if (cart_engine_create(&params, &rank->engine) != 0) throw std::runtime_error(std::string("cart_engine_create: ") + cart_last_error(nullptr));
        if (placementTries > 1 && cart_engine_tune_placement(rank->engine, std::min(framesPerGpu, 16), placementTries, /*default cap*/ 0, nullptr) != 0)
            std::fprintf(stderr, "[cartslam_amd] GPU %d: placement tuning failed (%s); keeping the first placement\n", devices[r], cart_last_error(rank->engine));
        cartOk(cart_plane_schedule_create(rank->engine, /*histogram_peak*/ 1, nullptr, updateInterval, resetInterval, &rank->schedule), rank->engine,
               "cart_plane_schedule_create");
        for (int b = 0; b < 2; ++b) {
            rank->left[b] = devAlloc<uint8_t>(share);
            rank->right[b] = devAlloc<uint8_t>(share);
            rank->planes[b] = devAlloc<uint8_t>(share);
            rank->disparity[b] = devAlloc<int16_t>(share);
            rank->derivative[b] = devAlloc<int16_t>(share);
            rank->hist[b] = devAlloc<int32_t>((size_t)framesPerGpu * 256);
            if (r == 0 && n > 1) {
                rank->stageLeft[b] = devAlloc<uint8_t>(share * n);
                rank->stageRight[b] = devAlloc<uint8_t>(share * n);
                rank->stagePlanes[b] = devAlloc<uint8_t>(share * n);
                rank->stageDisparity[b] = devAlloc<int16_t>(share * n);
            }
            for (hipEvent_t *e : {&rank->scattered[b], &rank->histReady[b], &rank->histGathered[b], &rank->computed[b], &rank->gathered[b]})
                hipOk(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
        }
        hipOk(hipEventCreateWithFlags(&rank->inputsReady, hipEventDisableTiming), "hipEventCreate");
        rank->histByRank = devAlloc<int32_t>((size_t)n * framesPerGpu * 256);
        rank->histById = devAlloc<int32_t>((size_t)n * framesPerGpu * 256);
        rank->paramsAll = devAlloc<cart_plane_params>((size_t)n * framesPerGpu);
        rank->paramsMine = devAlloc<cart_plane_params>((size_t)framesPerGpu);
        ranks.push_back(std::move(rank));
    }
}

FrameSharder::~FrameSharder() = default;

// Order of one GPU's two streams for sequences i, i+1 (b = buffer set):
//   submit(i)      copy:    scatter(i)                                   compute: [wait scattered] disparity(i), derivative + histograms(i) -> histReady
//   submit(i+1)    copy:    scatter(i+1), then finish(i):                compute: ... then disparity(i+1) ...
//     finish(i)    copy:    [wait histReady(i)] all-gather(i) -> histGathered
//                  compute: [wait histGathered(i)] schedule replay + classification(i) -> computed
//                  copy:    [wait computed(i)] gather(i) -> gathered
// so the scatter of i+1 runs beside the disparity kernels of i, the gather of i beside those of i+1, and every RCCL operation of the one
// communicator sits on the copy stream in the order scatter(i+1), all-gather(i), gather(i) -- on every GPU alike.  wait() finishes the last one.
void FrameSharder::streamWait(void *stream, void *event) {
    hipOk(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0), "hipStreamWaitEvent");
    ++counters_.streamWaits;
}

void FrameSharder::submit(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes, void *callerStream) {
    const int n = gpus();
    if (!left || !right || !disparity || !planes) throw std::invalid_argument("NULL image pointer");
    if (nFrames < 1 || nFrames > capacity())
        throw std::invalid_argument("sequence length " + std::to_string(nFrames) + " is outside [1, " + std::to_string(capacity()) + "] (frames per GPU x GPUs)");
    const size_t npx = (size_t)width * height, share = (size_t)framesPerGpu * npx;
    const int b = (int)(submitted & 1);
    const int nMax = shareOf(nFrames, 0, n);   // the longest share: what the all-gather moves per GPU
    Rank &root = *ranks[0];
    hipOk(hipSetDevice(root.device), "hipSetDevice");
    if (submitted >= 2) hipOk(hipEventSynchronize(root.gathered[b]), "hipEventSynchronize");   // at most two sequences in flight

    // ---- scatter on the copy streams: frame k -> GPU k mod n, local index k / n.  GPU 0 packs every peer's frames (every n-th
    //      of the sequence) into one contiguous area, so a peer gets ONE send per image kind; its own share is a local copy.
    hipOk(hipEventRecord(root.inputsReady, static_cast<hipStream_t>(callerStream)), "hipEventRecord");
    streamWait(root.copy, root.inputsReady);
    for (int r = 0; r < n; ++r) {
        const int local = shareOf(nFrames, r, n);
        if (!local) continue;
        uint8_t *dl = r == 0 ? root.left[b] : root.stageLeft[b] + r * share, *dr = r == 0 ? root.right[b] : root.stageRight[b] + r * share;
        hipOk(hipMemcpy2DAsync(dl, npx, left + (size_t)r * npx, (size_t)n * npx, npx, local, hipMemcpyDeviceToDevice, root.copy), "hipMemcpy2DAsync");
        hipOk(hipMemcpy2DAsync(dr, npx, right + (size_t)r * npx, (size_t)n * npx, npx, local, hipMemcpyDeviceToDevice, root.copy), "hipMemcpy2DAsync");
    }
    {
        NcclGroup group;   // every send has its receive posted with it
        for (int r = 1; r < n; ++r) {
            const size_t count = (size_t)shareOf(nFrames, r, n) * npx;
            if (!count) continue;
            ncclOk(ncclSend(root.stageLeft[b] + r * share, count, ncclUint8, r, root.comm, root.copy), "ncclSend");
            ncclOk(ncclRecv(ranks[r]->left[b], count, ncclUint8, 0, ranks[r]->comm, ranks[r]->copy), "ncclRecv");
            ncclOk(ncclSend(root.stageRight[b] + r * share, count, ncclUint8, r, root.comm, root.copy), "ncclSend");
            ncclOk(ncclRecv(ranks[r]->right[b], count, ncclUint8, 0, ranks[r]->comm, ranks[r]->copy), "ncclRecv");
        }
        group.end();
        ++counters_.collectiveGroups;
    }
    for (auto &rk : ranks) {
        hipOk(hipSetDevice(rk->device), "hipSetDevice");
        hipOk(hipEventRecord(rk->scattered[b], rk->copy), "hipEventRecord");
    }

    // ---- the previous sequence's all-gather, classification and gather go BEHIND this scatter: the scatter is not held up by the
    //      wait for that sequence's kernels, and its gather runs beside the kernels enqueued below
    if (pending.live) {
        finish(pending);
        pending.live = false;
    }

    // ---- every GPU: disparity and plane derivative + per-frame histograms of its share
    for (int r = 0; r < n; ++r) {
        Rank &rk = *ranks[r];
        const int local = shareOf(nFrames, r, n);
        hipOk(hipSetDevice(rk.device), "hipSetDevice");
        streamWait(rk.compute, rk.scattered[b]);
        if (submitted >= 2) streamWait(rk.compute, rk.gathered[b]);   // the outputs of sequence i-2 have left this buffer set
        hipOk(hipMemsetAsync(rk.hist[b], 0, (size_t)nMax * 256 * sizeof(int32_t), rk.compute), "hipMemsetAsync");   // short shares travel zero-padded
        if (local) {
            cartOk(cart_compute_disparity_batch(rk.engine, local, rk.left[b], (size_t)width, npx, rk.right[b], (size_t)width, npx, 1, rk.disparity[b],
                                                (size_t)width * 2, npx * 2, rk.compute),
                   rk.engine, "cart_compute_disparity_batch");
            cartOk(cart_plane_derivative_hist(rk.engine, local, rk.disparity[b], (size_t)width * 2, npx * 2, rk.derivative[b], (size_t)width * 2, npx * 2,
                                              rk.hist[b], 256, rk.compute),
                   rk.engine, "cart_plane_derivative_hist");
        }
        hipOk(hipEventRecord(rk.histReady[b], rk.compute), "hipEventRecord");
    }
    pending = Pending{nFrames, nextId, disparity, planes, b, true};
    nextId += nFrames;
    ++submitted;
}

// all-gather of the histograms, schedule replay + classification, outputs back to GPU 0 in sequence order -- of a sequence whose
// disparity kernels are enqueued (RCCL has no 16-bit integer type: disparities travel as bytes)
void FrameSharder::finish(const Pending &p) {
    const int n = gpus(), b = p.buf, nMax = shareOf(p.nFrames, 0, n);
    const size_t npx = (size_t)width * height, share = (size_t)framesPerGpu * npx;
    Rank &root = *ranks[0];
    // ---- the path's only exchange step: all-gather of the per-frame histograms (1 KB per frame), on the copy stream like every other
    //      operation of the communicator
    for (auto &rk : ranks) {
        hipOk(hipSetDevice(rk->device), "hipSetDevice");
        streamWait(rk->copy, rk->histReady[b]);
    }
    {
        NcclGroup group;
        for (auto &rk : ranks)
            ncclOk(ncclAllGather(rk->hist[b], rk->histByRank, (size_t)nMax * 256, ncclInt32, rk->comm, rk->copy), "ncclAllGather");
        group.end();
        ++counters_.collectiveGroups;
    }
    // ---- every GPU: histograms into frame-id order ([rank][j] -> j * n + rank; the padded rows are exactly the ids >= nFrames),
    //      schedule replay for the whole sequence, its own frames' parameters (every n-th row), classification
    for (int r = 0; r < n; ++r) {
        Rank &rk = *ranks[r];
        const int local = shareOf(p.nFrames, r, n);
        hipOk(hipSetDevice(rk.device), "hipSetDevice");
        hipOk(hipEventRecord(rk.histGathered[b], rk.copy), "hipEventRecord");
        streamWait(rk.compute, rk.histGathered[b]);
        for (int src = 0; src < n; ++src)
            hipOk(hipMemcpy2DAsync(rk.histById + (size_t)src * 256, (size_t)n * 1024, rk.histByRank + (size_t)src * nMax * 256, 1024, 1024, nMax,
                                   hipMemcpyDeviceToDevice, rk.compute),
                  "hipMemcpy2DAsync");
        cartOk(cart_plane_schedule_advance(rk.schedule, p.firstId, p.nFrames, rk.histById, rk.paramsAll, rk.compute), rk.engine, "cart_plane_schedule_advance");
        if (local) {
            hipOk(hipMemcpy2DAsync(rk.paramsMine, sizeof(cart_plane_params), rk.paramsAll + r, (size_t)n * sizeof(cart_plane_params), sizeof(cart_plane_params),
                                   local, hipMemcpyDeviceToDevice, rk.compute),
                  "hipMemcpy2DAsync");
            cartOk(cart_plane_classify_dev(rk.engine, local, rk.derivative[b], (size_t)width * 2, npx * 2, rk.paramsMine, 1, rk.planes[b], (size_t)width, npx,
                                           rk.compute),
                   rk.engine, "cart_plane_classify_dev");
        }
        hipOk(hipEventRecord(rk.computed[b], rk.compute), "hipEventRecord");
        streamWait(rk.copy, rk.computed[b]);
    }
    {
        NcclGroup group;
        for (int r = 1; r < n; ++r) {
            const size_t count = (size_t)shareOf(p.nFrames, r, n) * npx;
            if (!count) continue;
            ncclOk(ncclSend(ranks[r]->disparity[b], count * 2, ncclUint8, 0, ranks[r]->comm, ranks[r]->copy), "ncclSend");
            ncclOk(ncclRecv(root.stageDisparity[b] + r * share, count * 2, ncclUint8, r, root.comm, root.copy), "ncclRecv");
            ncclOk(ncclSend(ranks[r]->planes[b], count, ncclUint8, 0, ranks[r]->comm, ranks[r]->copy), "ncclSend");
            ncclOk(ncclRecv(root.stagePlanes[b] + r * share, count, ncclUint8, r, root.comm, root.copy), "ncclRecv");
        }
        group.end();
        ++counters_.collectiveGroups;
    }
    hipOk(hipSetDevice(root.device), "hipSetDevice");
    for (int r = 0; r < n; ++r) {   // share of GPU r -> every n-th frame of the caller's arrays
        const int local = shareOf(p.nFrames, r, n);
        if (!local) continue;
        const int16_t *sd = r == 0 ? root.disparity[b] : root.stageDisparity[b] + r * share;
        const uint8_t *sp = r == 0 ? root.planes[b] : root.stagePlanes[b] + r * share;
        hipOk(hipMemcpy2DAsync(p.disparity + (size_t)r * npx, (size_t)n * npx * 2, sd, npx * 2, npx * 2, local, hipMemcpyDeviceToDevice, root.copy), "hipMemcpy2DAsync");
        hipOk(hipMemcpy2DAsync(p.planes + (size_t)r * npx, (size_t)n * npx, sp, npx, npx, local, hipMemcpyDeviceToDevice, root.copy), "hipMemcpy2DAsync");
    }
    for (auto &rk : ranks) {
        hipOk(hipSetDevice(rk->device), "hipSetDevice");
        hipOk(hipEventRecord(rk->gathered[b], rk->copy), "hipEventRecord");
    }
    ++counters_.sequences;
}

void FrameSharder::wait(double timeoutSeconds) {
    if (pending.live) {
        finish(pending);
        pending.live = false;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeoutSeconds);
    for (int r = 0; r < gpus(); ++r) {
        Rank &rk = *ranks[r];
        hipOk(hipSetDevice(rk.device), "hipSetDevice");
        for (hipStream_t s : {rk.compute, rk.copy}) {
            for (;;) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) throw std::runtime_error("GPU " + std::to_string(rk.device) + " (rank " + std::to_string(r) + "): " + hipGetErrorString(q));
                if (std::chrono::steady_clock::now() > deadline) {
                    // which side is stuck, and the asynchronous error state of this GPU's communicator
                    ncclResult_t ec = ncclSuccess;
                    (void)ncclCommGetAsyncError(rk.comm, &ec);
                    throw std::runtime_error("GPU " + std::to_string(rk.device) + " (rank " + std::to_string(r) + ") did not finish its " +
                                             (s == rk.compute ? "kernels" : "scatter / histogram all-gather / gather") + " within " +
                                             std::to_string(timeoutSeconds) + " s; RCCL state of its communicator: " + ncclGetErrorString(ec));
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
    }
}

void FrameSharder::processSequence(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes) {
    submit(left, right, nFrames, disparity, planes);
    wait();
}
}  // namespace cart
