// modules.cpp -- the three hot-path modules on top of the C ABI (include/cart_engine.h).
#include <exception>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "cartslam_amd/coalescer.hpp"
#include "cartslam_amd/modules/depth.hpp"
#include "cartslam_amd/modules/disparity.hpp"
#include "cartslam_amd/modules/planeseg.hpp"
#include "cartslam_amd/modules/superpixels.hpp"

namespace cart {
namespace {
void hipCheck(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// The reference creates and destroys one stream per invocation (disparity.cu:56, planeseg.cu:279-280,300-301); stream
// creation costs ~100 us here, so invocations borrow a stream from a pool instead (same concurrency, no churn).
// Two classes: the disparity module's launches fill the GPU for a millisecond at a time ("bulk", default priority);
// every other module enqueues short kernels, and on a default-priority stream their workgroups queue behind the whole
// remaining grid of whatever bulk kernel is resident (a 20 us classify kernel then takes 0.4-0.8 ms).  Those streams
// get the highest priority, so the dispatcher places their few workgroups as soon as any slot frees up.
// CARTSLAM_STREAM_PRIORITY=0 puts everything on default-priority streams.
class StreamPool {
   public:
    static StreamPool &instance() { static StreamPool p; return p; }
    hipStream_t acquire(bool bulk) {
        std::vector<hipStream_t> &idle = bulk ? idleBulk : idleShort;
        {
            std::lock_guard<std::mutex> lock(mutex);
            if (!idle.empty()) { hipStream_t s = idle.back(); idle.pop_back(); return s; }
        }
        static const bool usePriority = [] { const char *e = std::getenv("CARTSLAM_STREAM_PRIORITY"); return !e || std::atoi(e) != 0; }();
        int least = 0, greatest = 0;  // numerically: greatest priority = lowest number
        hipCheck(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        hipStream_t s = nullptr;
        hipCheck(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, (bulk || !usePriority) ? least : greatest), "hipStreamCreateWithPriority");
        return s;
    }
    void release(hipStream_t s, bool bulk) { std::lock_guard<std::mutex> lock(mutex); (bulk ? idleBulk : idleShort).push_back(s); }

   private:
    std::mutex mutex;
    std::vector<hipStream_t> idleBulk, idleShort;
};

struct ScopedStream {
    hipStream_t s = nullptr;
    const bool bulk;
    explicit ScopedStream(bool bulk = false) : s(StreamPool::instance().acquire(bulk)), bulk(bulk), exceptionsAtEntry(std::uncaught_exceptions()) {}
    ~ScopedStream() {
        if (!s) return;
        // leaving through an exception with kernels still queued: drain them before the stream goes back to the pool and
        // before the images they touch (declared earlier, destroyed later) go back to theirs
        if (std::uncaught_exceptions() > exceptionsAtEntry) (void)hipStreamSynchronize(s);
        StreamPool::instance().release(s, bulk);
    }
    const int exceptionsAtEntry;
    void wait() { hipCheck(hipStreamSynchronize(s), "hipStreamSynchronize"); }
};

// CARTSLAM_PLACEMENT_TRIES = placements of the slab workspace cart_engine_tune_placement may try.  Default 1 = keep the allocation the
// engine was created with: a module constructor does not go looking for device memory on its own.  A deployment that wants the 2-4 %
// (include/cart_engine.h) sets it to 2..10; the search then holds at most two units of slab memory beyond the workspace (the call's
// default cap) and logs what it found.
int placementTries() {
    const char *env = std::getenv("CARTSLAM_PLACEMENT_TRIES");
    return env ? std::max(1, std::atoi(env)) : 1;
}

cart_engine_params paramsFor(Size res, int minDisparity, int numDisparities, int radius, int iterations, int paths, int p1, int p2, int uniq) {
    cart_engine_params p;
    cart_engine_default_params(&p);
    p.width = res.width; p.height = res.height;
    p.min_disparity = minDisparity; p.num_disparities = numDisparities; p.paths = paths; p.p1 = p1; p.p2 = p2;
    p.uniqueness_ratio = uniq; p.smoothing_radius = radius; p.smoothing_iterations = iterations;
    p.max_inflight = (int)concurrentRunLimit();
    return p;
}
}  // namespace

EngineHandle::EngineHandle(Size, const cart_engine_params &params) {
    if (cart_engine_create(&params, &engine) != 0) throw std::runtime_error(std::string("cart_engine_create: ") + cart_last_error(nullptr));
    // Opt-in (CARTSLAM_PLACEMENT_TRIES > 1): pick the fastest of a few physical placements of the cost-slab workspace (include/cart_engine.h,
    // cart_engine_tune_placement: the aggregation launch runs 8-9 % faster on some).  Not fatal: a failed probe leaves the first placement.
    if (params.num_disparities > 0 && placementTries() > 1) {
        cart_placement_report rep;
        static const char *const modes[] = {"unknown", "fast", "mixed", "uniform"};
        if (cart_engine_tune_placement(engine, std::min(params.max_inflight, 16), placementTries(), /*max_extra_bytes: default cap*/ 0, &rep) != 0)
            std::fprintf(stderr, "[cartslam_amd] placement tuning failed (%s); keeping the first placement\n", cart_last_error(engine));
        else
            std::fprintf(stderr, "[cartslam_amd] placement tuning: launch pair %.3f -> %.3f ms, %d placements timed in %.2f s, mode %s\n", rep.ms_first, rep.ms_kept,
                         rep.candidates, rep.seconds, modes[rep.mode & 3]);
    }
}
EngineHandle::~EngineHandle() { cart_engine_destroy(engine); }
void EngineHandle::fail(const char *what) const { throw std::runtime_error(std::string(what) + ": " + cart_last_error(engine)); }

// ---------------------------------------------------------------- frame coalescing (cartslam_amd/coalescer.hpp)
// CARTSLAM_COALESCE = frame groups of one module allowed on the GPU at once; 0 = one launch sequence per frame.
// Default 1: while a group is on the GPU the next one collects every frame that arrives, so the groups are as large as
// the frames in flight allow (12 in flight: 5.6 frames per launch and 4.98 k pairs/s at D=128 / 8 paths, against 3.7 and
// 4.58 k with two groups and 2.5 / 4.56 k with three -- profiles/tools/r02_coalesce.sh; a launch of 3 frames is far from
// filling the chip, and two of them side by side do not make up for it).
static int coalesceGroups() {
    const char *env = std::getenv("CARTSLAM_COALESCE");
    return env ? std::atoi(env) : 1;
}
static int coalesceMaxGroup() { return (int)std::min<size_t>(concurrentRunLimit(), 16); }  // 16 = frames per launch sequence
// CARTSLAM_COALESCE_AHEAD = requests that must have gathered before a group is queued behind a running one (with CARTSLAM_COALESCE >= 2)
static int coalesceMinAhead() {
    const char *env = std::getenv("CARTSLAM_COALESCE_AHEAD");
    return env ? std::max(1, std::atoi(env)) : std::max(2, (int)std::min<size_t>(concurrentRunLimit(), 32) / 2);
}

// ---------------------------------------------------------------- disparity (disparity.cu:49-80)
struct DisparityRequest : CoalescedRequest {
    const uint8_t *left, *right; size_t leftStep, rightStep; int channels;
    int16_t *out; size_t outStep;
};
class DisparityCoalescer : public FrameCoalescer<DisparityRequest> {
   public:
    using FrameCoalescer<DisparityRequest>::FrameCoalescer;
};

ImageDisparityModule::ImageDisparityModule(const Size imageRes, int minDisparity, int numDisparities, int /*blockSize: ignored by the CUDA SGM too*/,
                                           int smoothingRadius, int smoothingIterations, int paths, int p1, int p2, int uniquenessRatio)
    : SyncWrapperSystemModule("ImageDisparity"), imageRes(imageRes) {
    this->providesData.push_back(CARTSLAM_KEY_DISPARITY);
    engine = std::make_shared<EngineHandle>(imageRes, paramsFor(imageRes, minDisparity, numDisparities, smoothingRadius, smoothingIterations, paths, p1, p2, uniquenessRatio));
    if (coalesceGroups() > 0) {
        auto eng = engine;
        coalescer = std::make_shared<DisparityCoalescer>(
            coalesceMaxGroup(), coalesceGroups(),
            [](const DisparityRequest &a, const DisparityRequest &b) {
                return a.channels == b.channels && a.leftStep == b.leftStep && a.rightStep == b.rightStep && a.outStep == b.outStep;
            },
            [eng](const std::vector<DisparityRequest *> &group) {
                std::vector<const uint8_t *> lefts, rights;
                std::vector<int16_t *> outs;
                for (const DisparityRequest *q : group) { lefts.push_back(q->left); rights.push_back(q->right); outs.push_back(q->out); }
                const DisparityRequest &rq = *group[0];
                ScopedStream stream(true);
                if (cart_compute_disparity_multi(eng->get(), (int)group.size(), lefts.data(), rq.leftStep, rights.data(), rq.rightStep, rq.channels,
                                                 outs.data(), rq.outStep, stream.s) != 0)
                    eng->fail("cart_compute_disparity_multi");
                stream.wait();  // stream.waitForCompletion(), disparity.cu:77
            },
            coalesceMinAhead());
    }
}

double ImageDisparityModule::meanFramesPerLaunch() const { return coalescer ? coalescer->meanGroup() : 1.0; }

system_data_t ImageDisparityModule::runInternal(System &, SystemRunData &data) {
    if (data.dataElement->type != DataElementType::STEREO) throw std::runtime_error("ImageDisparityModule requires StereoDataElement");
    auto stereo = std::static_pointer_cast<StereoDataElement>(data.dataElement);
    const image_t &l = stereo->left, &r = stereo->right;
    const int channels = l.type() == CV_8UC3 ? 3 : 1;
    if ((l.type() != CV_8UC3 && l.type() != CV_8UC1) || r.type() != l.type()) throw std::runtime_error("ImageDisparityModule requires CV_8UC1 or CV_8UC3 images");
    // the engine's workspaces are sized for the resolution given to the constructor (disparity.hpp:26): anything else would run past them
    if (l.cols != imageRes.width || l.rows != imageRes.height || r.cols != l.cols || r.rows != l.rows)
        throw std::runtime_error("ImageDisparityModule: image size " + std::to_string(l.cols) + "x" + std::to_string(l.rows) + " does not match the module's " +
                                 std::to_string(imageRes.width) + "x" + std::to_string(imageRes.height));
    auto disparity = std::make_shared<image_t>(l.rows, l.cols, CV_16SC1);
    if (coalescer) {
        DisparityRequest rq;
        rq.left = l.ptr<uint8_t>(); rq.right = r.ptr<uint8_t>(); rq.leftStep = l.step; rq.rightStep = r.step; rq.channels = channels;
        rq.out = disparity->ptr<int16_t>(); rq.outStep = disparity->step;
        coalescer->run(rq);
        return MODULE_RETURN(CARTSLAM_KEY_DISPARITY, disparity);
    }
    ScopedStream stream(true);
    if (cart_compute_disparity(engine->get(), l.ptr<uint8_t>(), l.step, r.ptr<uint8_t>(), r.step, channels, disparity->ptr<int16_t>(), disparity->step, stream.s) != 0)
        engine->fail("cart_compute_disparity");
    stream.wait();  // stream.waitForCompletion(), disparity.cu:77
    return MODULE_RETURN(CARTSLAM_KEY_DISPARITY, disparity);
}

// ---------------------------------------------------------------- derivative (derivative.cu:151-184)
ImageDisparityDerivativeModule::ImageDisparityDerivativeModule() : SyncWrapperSystemModule("ImageDisparityDerivative") {
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY));
    this->providesData.push_back(CARTSLAM_KEY_DISPARITY_DERIVATIVE);
    this->providesData.push_back(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM);
}

static std::shared_ptr<EngineHandle> postEngine(std::mutex &mu, std::shared_ptr<EngineHandle> &slot, const image_t &disp) {
    std::lock_guard<std::mutex> lk(mu);
    if (!slot) {  // the post stages only need the geometry: num_disparities = paths = 0 -> no SGM workspaces
        Size res; res.width = disp.cols; res.height = disp.rows;
        cart_engine_params p = paramsFor(res, 0, 0, -1, 0, 0, 10, 120, 12);
        slot = std::make_shared<EngineHandle>(res, p);
    }
    return slot;
}

system_data_t ImageDisparityDerivativeModule::runInternal(System &, SystemRunData &data) {
    auto disparity = data.getData<image_t>(CARTSLAM_KEY_DISPARITY);
    if (disparity->empty() || disparity->type() != CV_16SC1) throw std::runtime_error("Disparity must be of type CV_16SC1");
    auto eng = postEngine(engineMutex, engine, *disparity);
    auto derivatives = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_16SC2);
    auto histogram = std::make_shared<image_t>(1, 256, CV_32SC2);
    ScopedStream stream;
    if (cart_disparity_derivative(eng->get(), 1, disparity->ptr<int16_t>(), disparity->step, 0, derivatives->ptr<int16_t>(), derivatives->step, 0,
                                  histogram->ptr<int32_t>(), stream.s) != 0)
        eng->fail("cart_disparity_derivative");
    stream.wait();
    return MODULE_RETURN_ALL(std::make_pair(std::string(CARTSLAM_KEY_DISPARITY_DERIVATIVE), std::shared_ptr<void>(derivatives)),
                             std::make_pair(std::string(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM), std::shared_ptr<void>(histogram)));
}

// ---------------------------------------------------------------- depth (depth.cpp:9-25)
system_data_t DepthModule::runInternal(System &system, SystemRunData &data) {
    auto disparity = data.getData<image_t>(CARTSLAM_KEY_DISPARITY);
    if (disparity->empty() || disparity->type() != CV_16SC1) throw std::runtime_error("Disparity must be of type CV_16SC1");
    auto eng = postEngine(engineMutex, engine, *disparity);
    const CameraIntrinsics K = system.getDataSource()->getCameraIntrinsics();
    auto depth = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_32FC3);
    ScopedStream stream;
    if (cart_reproject_depth(eng->get(), 1, disparity->ptr<int16_t>(), disparity->step, 0, K.Q, depth->ptr<float>(), depth->step, 0, stream.s) != 0)
        eng->fail("cart_reproject_depth");
    stream.wait();
    return MODULE_RETURN(CARTSLAM_KEY_DEPTH, depth);
}

// ---------------------------------------------------------------- plane labels (planeseg.cu:246-458)
DisparityPlaneSegmentationModule::DisparityPlaneSegmentationModule(std::shared_ptr<PlaneParameterProvider> provider, const int updateInterval, const int resetInterval,
                                                                   const bool useTemporalSmoothing, const unsigned int temporalSmoothingDistance, const bool labelComponents)
    : SyncWrapperSystemModule("PlaneSegmentation"), useTemporalSmoothing(useTemporalSmoothing), temporalSmoothingDistance(temporalSmoothingDistance),
      updateInterval(updateInterval), resetInterval(resetInterval), labelComponents(labelComponents), planeParameterProvider(provider) {
    if (useTemporalSmoothing && (temporalSmoothingDistance < 1 || temporalSmoothingDistance > CART_MAX_TEMPORAL))
        throw std::runtime_error("temporal_smoothing_distance must be in [1, 8]");
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY));
    if (useTemporalSmoothing) {  // planeseg.hpp:128-137: optical flow and the earlier frames' unsmoothed planes
        this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_OPTFLOW));
        for (size_t i = 1; i <= this->temporalSmoothingDistance; i++) {
            this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_PLANES_UNSMOOTHED, -(int)i));
            if ((i + 1) <= this->temporalSmoothingDistance) this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_OPTFLOW, -(int)i));
        }
    }
    this->providesData.push_back(CARTSLAM_KEY_PLANES);
    if (useTemporalSmoothing) this->providesData.push_back(CARTSLAM_KEY_PLANES_UNSMOOTHED);
    if (labelComponents) {
        this->providesData.push_back(CARTSLAM_KEY_PLANE_COMPONENTS);
        this->providesData.push_back(CARTSLAM_KEY_PLANE_COMPONENT_TABLE);
        this->providesData.push_back(CARTSLAM_KEY_PLANE_COMPONENT_COUNT);
    }
}

DisparityPlaneSegmentationModule::~DisparityPlaneSegmentationModule() {
    if (derivativeHistogram) (void)hipFree(derivativeHistogram);
}

struct PlaneRequest : CoalescedRequest {
    const int16_t *disparity; size_t disparityStep;
    int16_t *derivatives; size_t derivativesStep;
    uint8_t *planes; size_t planesStep;
};
class PlaneCoalescer : public FrameCoalescer<PlaneRequest> {
   public:
    using FrameCoalescer<PlaneRequest>::FrameCoalescer;
};

void DisparityPlaneSegmentationModule::ensureHistogram() {
    static std::mutex createMutex;
    std::lock_guard<std::mutex> lk(createMutex);
    if (!derivativeHistogram) {
        hipCheck(hipMalloc(reinterpret_cast<void **>(&derivativeHistogram), 256 * sizeof(int32_t)), "hipMalloc");
        hipCheck(hipMemset(derivativeHistogram, 0, 256 * sizeof(int32_t)), "hipMemset");
    }
}

system_data_t DisparityPlaneSegmentationModule::runInternal(System &system, SystemRunData &data) {
    auto disparity = data.getData<image_t>(CARTSLAM_KEY_DISPARITY);
    if (disparity->empty()) return MODULE_NO_RETURN_VALUE;  // planeseg.cu:250-253
    if (disparity->type() != CV_16SC1) throw std::runtime_error("Disparity must be of type CV_16SC1");
    auto eng = postEngine(engineMutex, engine, *disparity);
    auto derivatives = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_16SC1);
    const bool updateFrame = (int)(data.id % (uint32_t)this->updateInterval) == 1;
    // Frames that neither refresh the parameters nor need per-frame extras go through the coalescer: the frames waiting
    // here together get one derivative launch (all adding to the cumulative histogram, like concurrent frames of the
    // reference do) and one classify launch with the parameters current at that moment.
    if (!updateFrame && !this->useTemporalSmoothing && !this->labelComponents && coalesceGroups() > 0) {
        {
            std::lock_guard<std::mutex> lk(engineMutex);
            if (!coalescer) {
                coalescer = std::make_shared<PlaneCoalescer>(
                    coalesceMaxGroup(), coalesceGroups(),
                    [](const PlaneRequest &a, const PlaneRequest &b) {
                        return a.disparityStep == b.disparityStep && a.derivativesStep == b.derivativesStep && a.planesStep == b.planesStep;
                    },
                    [this, eng](const std::vector<PlaneRequest *> &group) {
                        std::shared_lock<std::shared_mutex> histogramLock(derivativeHistogramMutex);  // until the group's kernels have finished
                        ensureHistogram();
                        std::vector<const int16_t *> disps, derivsIn;
                        std::vector<int16_t *> derivs;
                        std::vector<uint8_t *> labels;
                        for (const PlaneRequest *q : group) { disps.push_back(q->disparity); derivs.push_back(q->derivatives); derivsIn.push_back(q->derivatives); labels.push_back(q->planes); }
                        const PlaneRequest &rq = *group[0];
                        ScopedStream stream;
                        if (cart_plane_derivative_hist_multi(eng->get(), (int)group.size(), disps.data(), rq.disparityStep, derivs.data(), rq.derivativesStep,
                                                             derivativeHistogram, 0, stream.s) != 0)
                            eng->fail("cart_plane_derivative_hist_multi");
                        const PlaneParameters pp = planeParameterProvider->getPlaneParameters();
                        cart_plane_params cp{pp.horizontalRange.first, pp.horizontalRange.second, pp.verticalRange.first, pp.verticalRange.second, pp.horizontalCenter, pp.verticalCenter};
                        if (cart_plane_classify_multi(eng->get(), (int)group.size(), derivsIn.data(), rq.derivativesStep, &cp, 0, labels.data(), rq.planesStep, stream.s) != 0)
                            eng->fail("cart_plane_classify_multi");
                        stream.wait();
                    },
                    coalesceMinAhead());
            }
        }
        auto planes = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_8UC1);
        PlaneRequest rq;
        rq.disparity = disparity->ptr<int16_t>(); rq.disparityStep = disparity->step;
        rq.derivatives = derivatives->ptr<int16_t>(); rq.derivativesStep = derivatives->step;
        rq.planes = planes->ptr<uint8_t>(); rq.planesStep = planes->step;
        coalescer->run(rq);
        return MODULE_RETURN(CARTSLAM_KEY_PLANES, planes);
    }
    // Read-lock section, planeseg.cu:269-288: the histogram download of an update frame (unique lock) must not overlap a
    // derivative kernel that is still adding to it.  An update frame (id % updateInterval == 1) synchronises and
    // releases the lock right after its derivative kernel, like the reference; every other frame has nothing to do
    // between the two kernels, keeps the lock and enqueues the rest of the module behind the derivative kernel on the
    // same stream: one stream synchronisation per frame instead of two.
    std::shared_lock<std::shared_mutex> histogramLock(derivativeHistogramMutex);
    ensureHistogram();
    ScopedStream stream;
    if (cart_plane_derivative_hist(eng->get(), 1, disparity->ptr<int16_t>(), disparity->step, 0, derivatives->ptr<int16_t>(), derivatives->step, 0,
                                   derivativeHistogram, 0, stream.s) != 0)
        eng->fail("cart_plane_derivative_hist");
    if (updateFrame) {
        stream.wait();
        histogramLock.unlock();
        this->updatePlaneParameters(system, data);
    }

    auto planes = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_8UC1);
    const PlaneParameters pp = planeParameterProvider->getPlaneParameters();
    cart_plane_params cp{pp.horizontalRange.first, pp.horizontalRange.second, pp.verticalRange.first, pp.verticalRange.second, pp.horizontalCenter, pp.verticalCenter};
    if (cart_plane_classify(eng->get(), 1, derivatives->ptr<int16_t>(), derivatives->step, 0, &cp, 0, planes->ptr<uint8_t>(), planes->step, 0, stream.s) != 0)
        eng->fail("cart_plane_classify");
    std::shared_ptr<image_t> smoothed;
    std::vector<std::shared_ptr<image_t>> keepAlive;
    if (this->useTemporalSmoothing && data.id > 1) {  // planeseg.cu:303-347
        smoothed = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_8UC1);
        const uint8_t *prevPlanes[CART_MAX_TEMPORAL];
        size_t prevSteps[CART_MAX_TEMPORAL];
        const int16_t *flows[CART_MAX_TEMPORAL];
        size_t flowSteps[CART_MAX_TEMPORAL];
        int previousPlaneCount = 0;
        auto optFlowCurr = data.getData<image_t>(CARTSLAM_KEY_OPTFLOW);
        keepAlive.push_back(optFlowCurr);
        flows[0] = optFlowCurr->ptr<int16_t>(); flowSteps[0] = optFlowCurr->step;
        for (int i = 1; i <= (int)this->temporalSmoothingDistance; i++) {
            if ((int64_t)data.id - i <= 0) break;
            auto relativeRun = data.getRelativeRun((int8_t)-i);
            auto prev = relativeRun->getData<image_t>(CARTSLAM_KEY_PLANES_UNSMOOTHED);
            keepAlive.push_back(prev);
            prevPlanes[previousPlaneCount] = prev->ptr<uint8_t>(); prevSteps[previousPlaneCount] = prev->step;
            previousPlaneCount++;
            if (relativeRun->id > 1 && previousPlaneCount < (int)this->temporalSmoothingDistance) {
                auto optFlow = relativeRun->getData<image_t>(CARTSLAM_KEY_OPTFLOW);
                keepAlive.push_back(optFlow);
                flows[previousPlaneCount] = optFlow->ptr<int16_t>(); flowSteps[previousPlaneCount] = optFlow->step;
            }
        }
        if (cart_plane_temporal_vote(eng->get(), planes->ptr<uint8_t>(), planes->step, previousPlaneCount, prevPlanes, prevSteps, flows, flowSteps,
                                     smoothed->ptr<uint8_t>(), smoothed->step, stream.s) != 0)
            eng->fail("cart_plane_temporal_vote");
    }
    std::shared_ptr<image_t> components, componentTable, componentCount;
    if (labelComponents) {
        components = std::make_shared<image_t>(disparity->rows, disparity->cols, CV_32SC1);
        componentTable = std::make_shared<image_t>(CARTSLAM_PLANE_COMPONENT_TABLE_ROWS, 7, CV_32SC1);
        componentCount = std::make_shared<image_t>(1, 1, CV_32SC1);
        static_assert(sizeof(cart_component) == 7 * sizeof(int32_t), "table rows are 7 x int32");
        if (componentTable->step != 7 * sizeof(int32_t)) {  // DeviceImage pads rows to 256 B: the table wants tight rows
            componentTable = std::make_shared<image_t>(1, CARTSLAM_PLANE_COMPONENT_TABLE_ROWS * 7, CV_32SC1);
        }
        // ids, count and table in one call: the pass that writes the final ids also gathers the component statistics (four launches)
        if (cart_plane_ccl_table(eng->get(), 1, planes->ptr<uint8_t>(), planes->step, 0, components->ptr<int32_t>(), components->step, 0,
                                 componentTable->ptr<cart_component>(), CARTSLAM_PLANE_COMPONENT_TABLE_ROWS, componentCount->ptr<int32_t>(), stream.s) != 0)
            eng->fail("cart_plane_ccl_table");
    }
    stream.wait();
    system_data_t out;
    if (this->useTemporalSmoothing) {  // planeseg.cu:361-374: frame 1 returns the same image under both keys
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANES), std::shared_ptr<void>(data.id == 1 ? planes : smoothed)));
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANES_UNSMOOTHED), std::shared_ptr<void>(planes)));
    } else {
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANES), std::shared_ptr<void>(planes)));
    }
    if (labelComponents) {
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANE_COMPONENTS), std::shared_ptr<void>(components)));
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANE_COMPONENT_TABLE), std::shared_ptr<void>(componentTable)));
        out.push_back(std::make_pair(std::string(CARTSLAM_KEY_PLANE_COMPONENT_COUNT), std::shared_ptr<void>(componentCount)));
    }
    return out;
}

// ---------------------------------------------------------------- superpixels (superpixels.cu:19-118)
FrameOrder::Turn::Turn(FrameOrder &o, uint32_t id) : order(o), id(id) {
    std::unique_lock<std::mutex> lock(order.mutex);
    order.cv.wait(lock, [&] { return order.next >= id; });   // every earlier frame has finished, one way or the other
}
FrameOrder::Turn::~Turn() { order.finish(id); }

void FrameOrder::startAt(uint32_t id) {
    {
        std::lock_guard<std::mutex> lock(mutex);
        if (id <= next) return;
        next = id;
        while (!finishedAhead.empty() && *finishedAhead.begin() <= next) { if (*finishedAhead.begin() == next) ++next; finishedAhead.erase(finishedAhead.begin()); }
    }
    cv.notify_all();
}

void FrameOrder::finish(uint32_t id) {
    {
        std::lock_guard<std::mutex> lock(mutex);
        if (id < next) return;
        finishedAhead.insert(id);
        while (!finishedAhead.empty() && *finishedAhead.begin() == next) { finishedAhead.erase(finishedAhead.begin()); ++next; }
    }
    cv.notify_all();
}

SuperPixelModule::SuperPixelModule(const Size imageRes, const unsigned int initialIterations, const unsigned int iterations, const unsigned int blockSize,
                                   const unsigned int resetIterations, const double directCliqueCost, const double diagonalCliqueCost, const double compactnessWeight,
                                   const double progressiveCompactnessCost, const double imageWeight, const double disparityWeight)
    : SyncWrapperSystemModule("SuperPixelDetect"), initialIterations(initialIterations), iterations(iterations), resetIterations(resetIterations),
      blockSize(blockSize), requiresDisparityDerivative(disparityWeight > 0) {
    if (blockSize < 1) throw std::invalid_argument("blockSize must be more than 1");                                        // superpixels.cu:37-39
    if (directCliqueCost < 0) throw std::invalid_argument("directCliqueCost must be non-negative");                         // :41-43
    if (compactnessWeight < 0 || imageWeight < 0 || disparityWeight < 0) throw std::invalid_argument("weight must be non-negative");  // :45-47
    if (resetIterations < 1) throw std::invalid_argument("resetIterations must be at least 1");
    if (disparityWeight > 0) this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY_DERIVATIVE));
    this->providesData.push_back(CARTSLAM_KEY_SUPERPIXELS);
    this->providesData.push_back(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL);
    engine = std::make_shared<EngineHandle>(imageRes, paramsFor(imageRes, 0, 0, -1, 0, 0, 10, 120, 12));
    cart_superpixel_params p{directCliqueCost, diagonalCliqueCost, compactnessWeight, progressiveCompactnessCost, imageWeight, disparityWeight};
    if (cart_superpixels_create(engine->get(), &p, (int)blockSize, (int)blockSize, &contourRelaxation) != 0) engine->fail("cart_superpixels_create");
}

SuperPixelModule::~SuperPixelModule() { cart_superpixels_destroy(contourRelaxation); }

system_data_t SuperPixelModule::runInternal(System &, SystemRunData &data) {
    const image_t image = getReferenceImage(data.dataElement);  // the YCrCb conversion (superpixels.cu:81) happens inside cart_superpixels_relax
    if (image.type() != CV_8UC3 && image.type() != CV_8UC1) throw std::runtime_error("SuperPixelModule requires CV_8UC1 or CV_8UC3 images");
    std::shared_ptr<image_t> disparityDerivative;
    if (this->requiresDisparityDerivative) {
        disparityDerivative = data.getData<image_t>(CARTSLAM_KEY_DISPARITY_DERIVATIVE);
        if (disparityDerivative->type() != CV_16SC2) throw std::runtime_error("Disparity derivative must be of type CV_16SC2");
    }
    const unsigned int numIterations = (data.id == 1 || data.id % this->resetIterations == 0) ? this->initialIterations : this->iterations;  // :92
    auto relaxedLabelImage = std::make_shared<image_t>(image.rows, image.cols, CV_16UC1);
    int maxLabelId = 0;
    ScopedStream stream;
    {
        // The reference's mutex (:97-99), taken in frame order.  It covers the ENQUEUE only: cart_superpixels orders the
        // calls on the device (event of the previous call), so the next frame's sweeps queue up right behind this frame's
        // while this thread is still waiting for its own result -- the label state never leaves the GPU between frames.
        FrameOrder::Turn turn(order, data.id);
        if (data.id % this->resetIterations == 0)  // :104-112
            if (cart_superpixels_reset(contourRelaxation, stream.s) != 0) engine->fail("cart_superpixels_reset");
        if (cart_superpixels_relax(contourRelaxation, image.ptr<uint8_t>(), image.step, image.type() == CV_8UC3 ? 3 : 1,
                                   disparityDerivative ? disparityDerivative->ptr<int16_t>() : nullptr, disparityDerivative ? disparityDerivative->step : 0,
                                   (int)numIterations, relaxedLabelImage->ptr<uint16_t>(), relaxedLabelImage->step, stream.s) != 0)
            engine->fail("cart_superpixels_relax");
        maxLabelId = cart_superpixels_max_label(contourRelaxation);
    }
    stream.wait();
    return MODULE_RETURN_ALL(std::make_pair(std::string(CARTSLAM_KEY_SUPERPIXELS), std::shared_ptr<void>(relaxedLabelImage)),
                             std::make_pair(std::string(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL), std::shared_ptr<void>(std::make_shared<contour::label_t>((contour::label_t)maxLabelId))));
}

// ---------------------------------------------------------------- superpixel plane labels (sp_planeseg.cu:180-388)
SuperPixelDisparityPlaneSegmentationModule::SuperPixelDisparityPlaneSegmentationModule(std::shared_ptr<PlaneParameterProvider> provider, const int updateInterval,
                                                                                       const int resetInterval, const bool useTemporalSmoothing,
                                                                                       const unsigned int temporalSmoothingDistance)
    : SyncWrapperSystemModule("SPPlaneSegmentation"), useTemporalSmoothing(useTemporalSmoothing), temporalSmoothingDistance(temporalSmoothingDistance),
      updateInterval(updateInterval), resetInterval(resetInterval), planeParameterProvider(provider) {
    if (useTemporalSmoothing && (temporalSmoothingDistance < 1 || temporalSmoothingDistance > CART_MAX_TEMPORAL))
        throw std::runtime_error("temporal_smoothing_distance must be in [1, 8]");
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_SUPERPIXELS));  // sp_planeseg.cu:191-194
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL));
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY_DERIVATIVE));
    this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM));
    if (useTemporalSmoothing) {  // :196-205
        this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_OPTFLOW));
        for (size_t i = 1; i <= this->temporalSmoothingDistance; i++) {
            this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_PLANES_UNSMOOTHED, -(int)i));
            if ((i + 1) <= this->temporalSmoothingDistance) this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_OPTFLOW, -(int)i));
        }
    }
    this->providesData.push_back(CARTSLAM_KEY_PLANES);
    if (useTemporalSmoothing) this->providesData.push_back(CARTSLAM_KEY_PLANES_UNSMOOTHED);
}

system_data_t SuperPixelDisparityPlaneSegmentationModule::runInternal(System &system, SystemRunData &data) {
    auto derivatives = data.getData<image_t>(CARTSLAM_KEY_DISPARITY_DERIVATIVE);
    if (derivatives->empty()) return MODULE_NO_RETURN_VALUE;  // sp_planeseg.cu:223-226
    if (derivatives->type() != CV_16SC2) throw std::runtime_error("Disparity must be of type CV_16SC2");  // :228-231
    Size res; res.width = derivatives->cols; res.height = derivatives->rows;
    std::shared_ptr<EngineHandle> eng;
    {
        std::lock_guard<std::mutex> lk(engineMutex);
        if (!engine) engine = std::make_shared<EngineHandle>(res, paramsFor(res, 0, 0, -1, 0, 0, 10, 120, 12));
        eng = engine;
    }
    cart_plane_params cp;
    {
        FrameOrder::Turn turn(order, data.id);
        this->updatePlaneParameters(system, data);  // :237
        const PlaneParameters pp = planeParameterProvider->getPlaneParameters();
        cp = cart_plane_params{pp.horizontalRange.first, pp.horizontalRange.second, pp.verticalRange.first, pp.verticalRange.second, pp.horizontalCenter, pp.verticalCenter};
    }
    auto planes = std::make_shared<image_t>(derivatives->rows, derivatives->cols, CV_8UC1);
    auto smoothed = std::make_shared<image_t>(derivatives->rows, derivatives->cols, CV_8UC1);
    const uint8_t *prevPlanes[CART_MAX_TEMPORAL];
    size_t prevSteps[CART_MAX_TEMPORAL];
    const int16_t *flows[CART_MAX_TEMPORAL];
    size_t flowSteps[CART_MAX_TEMPORAL];
    int previousPlaneCount = 0;
    std::vector<std::shared_ptr<image_t>> keepAlive;
    if (this->useTemporalSmoothing && data.id > 1) {  // :250-300
        auto optFlowCurr = data.getData<image_t>(CARTSLAM_KEY_OPTFLOW);
        keepAlive.push_back(optFlowCurr);
        flows[0] = optFlowCurr->ptr<int16_t>(); flowSteps[0] = optFlowCurr->step;
        for (int i = 1; i <= (int)this->temporalSmoothingDistance; i++) {
            if ((int64_t)data.id - i <= 0) break;
            auto relativeRun = data.getRelativeRun((int8_t)-i);
            std::shared_ptr<image_t> prev;
            try { prev = relativeRun->getData<image_t>(CARTSLAM_KEY_PLANES_UNSMOOTHED); } catch (const std::exception &) { break; }  // :270-275
            keepAlive.push_back(prev);
            prevPlanes[previousPlaneCount] = prev->ptr<uint8_t>(); prevSteps[previousPlaneCount] = prev->step;
            previousPlaneCount++;
            if (relativeRun->id > 1 && previousPlaneCount < (int)this->temporalSmoothingDistance) {
                std::shared_ptr<image_t> optFlow;
                try { optFlow = relativeRun->getData<image_t>(CARTSLAM_KEY_OPTFLOW); } catch (const std::exception &) { break; }  // :289-294
                keepAlive.push_back(optFlow);
                flows[previousPlaneCount] = optFlow->ptr<int16_t>(); flowSteps[previousPlaneCount] = optFlow->step;
            }
        }
    }
    auto labels = data.getData<image_t>(CARTSLAM_KEY_SUPERPIXELS);
    const contour::label_t maxLabel = *data.getData<contour::label_t>(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL);
    if (labels->type() != CV_16UC1) throw std::runtime_error("Superpixels must be of type CV_16UC1");
    if (((size_t)maxLabel + 1) * 3 * sizeof(uint16_t) > 32768)  // the reference's shared-memory bound (:317-321), kept as the accepted range
        throw std::runtime_error("Shared memory size exceeds maximum. Reduce image size or increase block size.");
    ScopedStream stream;
    if (cart_superpixel_plane_classify(eng->get(), derivatives->ptr<int16_t>(), derivatives->step, labels->ptr<uint16_t>(), labels->step, (int)maxLabel, &cp,
                                       previousPlaneCount, prevPlanes, prevSteps, flows, flowSteps, planes->ptr<uint8_t>(), planes->step, smoothed->ptr<uint8_t>(),
                                       smoothed->step, stream.s) != 0)
        eng->fail("cart_superpixel_plane_classify");
    stream.wait();
    return MODULE_RETURN_ALL(std::make_pair(std::string(CARTSLAM_KEY_PLANES), std::shared_ptr<void>(smoothed)),  // :341-343: both keys, always
                             std::make_pair(std::string(CARTSLAM_KEY_PLANES_UNSMOOTHED), std::shared_ptr<void>(planes)));
}

void SuperPixelDisparityPlaneSegmentationModule::updatePlaneParameters(System &system, SystemRunData &data) {
    // channel 0 (vertical derivative) of the frame's CV_32SC2 1x256 histogram (sp_planeseg.cu:350-356)
    auto histImage = data.getData<image_t>(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM);
    std::vector<uint8_t> raw = histImage->downloadTight();
    const int32_t *two = reinterpret_cast<const int32_t *>(raw.data());
    std::vector<int32_t> histogram(256);
    for (int i = 0; i < 256; ++i) histogram[i] = two[2 * i];
    if (this->derivativeHistogram.empty()) {
        this->derivativeHistogram.assign(256, 0);  // :360-361: the first frame starts the running total at ZERO and is itself not added
    } else {
        for (int i = 0; i < 256; ++i) this->derivativeHistogram[i] += histogram[i];  // :363-365
        histogram = this->derivativeHistogram;
    }
    if ((int)(data.id % (uint32_t)(this->updateInterval * this->resetInterval)) == 1) this->derivativeHistogram.assign(256, 0);  // :368-371
    if ((int)(data.id % (uint32_t)this->updateInterval) != 1) return;  // :374-376
    this->planeParameterProvider->updatePlaneParameters(system, data, histogram);
    system.insertGlobalData(CARTSLAM_KEY_PLANE_PARAMETERS, std::make_shared<PlaneParameters>(this->planeParameterProvider->getPlaneParameters()));
    system.insertGlobalData(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HIST, std::make_shared<std::vector<int32_t>>(histogram));
}

// ---------------------------------------------------------------- optical flow (optflow.cpp:52-140)
ImageOpticalFlowModule::ImageOpticalFlowModule(const Size imageRes, int searchRadius, int blockRadius)
    : SyncWrapperSystemModule("ImageOpticalFlow"), searchRadius(searchRadius), blockRadius(blockRadius) {
    if (searchRadius < 1 || searchRadius > 16) throw std::invalid_argument("search_radius must be in [1, 16]");
    if (blockRadius < 1 || blockRadius > 3) throw std::invalid_argument("block_radius must be in [1, 3]");
    this->providesData.push_back(CARTSLAM_KEY_OPTFLOW);
    engine = std::make_shared<EngineHandle>(imageRes, paramsFor(imageRes, 0, 0, -1, 0, 0, 10, 120, 12));
}

system_data_t ImageOpticalFlowModule::runInternal(System &, SystemRunData &data) {
    if (data.id <= 1) return MODULE_RETURN(CARTSLAM_KEY_OPTFLOW, std::shared_ptr<void>());  // first run, no previous data (optflow.cpp:126-128)
    std::shared_ptr<SystemRunData> previousRun = data.getRelativeRun(-1);
    const image_t referenceCurrent = getReferenceImage(data.dataElement);
    const image_t referencePrevious = getReferenceImage(previousRun->dataElement);
    if ((referenceCurrent.type() != CV_8UC1 && referenceCurrent.type() != CV_8UC3) || referencePrevious.type() != referenceCurrent.type())
        throw std::runtime_error("ImageOpticalFlowModule requires CV_8UC1 or CV_8UC3 images");
    auto flow = std::make_shared<image_t>(referenceCurrent.rows, referenceCurrent.cols, CV_16SC2);
    ScopedStream stream;
    if (cart_optical_flow(engine->get(), referenceCurrent.ptr<uint8_t>(), referenceCurrent.step, referencePrevious.ptr<uint8_t>(), referencePrevious.step,
                          referenceCurrent.type() == CV_8UC3 ? 3 : 1, searchRadius, blockRadius, flow->ptr<int16_t>(), flow->step, stream.s) != 0)
        engine->fail("cart_optical_flow");
    stream.wait();
    return MODULE_RETURN(CARTSLAM_KEY_OPTFLOW, flow);
}

system_data_t OpticalFlowFileModule::runInternal(System &system, SystemRunData &data) {
    const std::string dir = system.getDataSource()->getPath();
    const Size size = system.getDataSource()->getImageSize();
    char name[64];
    std::snprintf(name, sizeof(name), "/flow/%06u.bin", data.id - 1);
    std::vector<int16_t> host((size_t)size.width * size.height * 2);
    FILE *f = std::fopen((dir + name).c_str(), "rb");
    if (!f) throw std::runtime_error("Could not open optical flow file " + dir + name);
    const size_t got = std::fread(host.data(), sizeof(int16_t), host.size(), f);
    std::fclose(f);
    if (got != host.size()) throw std::runtime_error("Truncated optical flow file " + dir + name);
    auto flow = std::make_shared<image_t>(size.height, size.width, CV_16SC2);
    flow->upload(host.data(), (size_t)size.width * 4);
    return MODULE_RETURN(CARTSLAM_KEY_OPTFLOW, flow);
}

void DisparityPlaneSegmentationModule::updatePlaneParameters(System &system, SystemRunData &data) {
    if ((int)(data.id % (uint32_t)this->updateInterval) != 1) return;  // planeseg.cu:381-383
    std::vector<int32_t> histogram(256);
    {
        std::unique_lock<std::shared_mutex> lock(derivativeHistogramMutex);
        hipCheck(hipMemcpy(histogram.data(), derivativeHistogram, 256 * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy");
        if ((int)(data.id % (uint32_t)(this->updateInterval * this->resetInterval)) == 1)
            hipCheck(hipMemset(derivativeHistogram, 0, 256 * sizeof(int32_t)), "hipMemset");  // reset to avoid overflow (:391-394)
    }
    this->planeParameterProvider->updatePlaneParameters(system, data, histogram);
    system.insertGlobalData(CARTSLAM_KEY_PLANE_PARAMETERS, std::make_shared<PlaneParameters>(this->planeParameterProvider->getPlaneParameters()));
    system.insertGlobalData(CARTSLAM_KEY_DISPARITY_DERIVATIVE_HIST, std::make_shared<std::vector<int32_t>>(histogram));
}

void HistogramPeakPlaneParameterProvider::updatePlaneParameters(System &, SystemRunData &, const std::vector<int32_t> &histogram) {
    cart_plane_params p{horizontalRange.first, horizontalRange.second, verticalRange.first, verticalRange.second, horizontalCenter, verticalCenter};
    if (cart_find_plane_params(histogram.data(), &p) < 0) throw std::runtime_error(std::string("cart_find_plane_params: ") + cart_last_error(nullptr));
    horizontalRange = std::make_pair(p.horizontal_min, p.horizontal_max);
    verticalRange = std::make_pair(p.vertical_min, p.vertical_max);
    horizontalCenter = p.horizontal_center;
    verticalCenter = p.vertical_center;
}
}  // namespace cart
