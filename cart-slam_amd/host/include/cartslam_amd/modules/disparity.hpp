// modules/disparity.hpp -- same names, constructor arguments, blackboard keys and error behaviour as the
// reference's include/modules/disparity.hpp:13-79; the arithmetic is the gfx950 engine behind include/cart_engine.h.
#pragma once
#include <mutex>

#include "../cartslam.hpp"
#include "cart_engine.h"

#define CARTSLAM_KEY_DISPARITY "disparity"
#define CARTSLAM_KEY_DISPARITY_DERIVATIVE "disparity_derivative"
#define CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM "disparity_derivative_histogram"
#define CARTSLAM_DISPARITY_INVALID (-32768)

namespace cart {

typedef int16_t disparity_t;
typedef int16_t derivative_t;

// Shared engine handle: the modules of one System that work on the same image size share workspaces.
class EngineHandle {
   public:
    EngineHandle(Size imageRes, const cart_engine_params &params);
    ~EngineHandle();
    cart_engine *get() const { return engine; }
    [[noreturn]] void fail(const char *what) const;  // C-ABI status -> std::runtime_error (never exit(), cuda.cuh:193-201)

   private:
    cart_engine *engine = nullptr;
};

class ImageDisparityModule : public SyncWrapperSystemModule {
   public:
    // reference ctor: disparity.hpp:26-34. `paths`, `p1`, `p2`, `uniquenessRatio` expose what the reference leaves at
    // cv::cuda::createStereoSGM's defaults (MODE_HH4 = 4 paths, P1 = 10, P2 = 120) / sets to 12 (:32).
    ImageDisparityModule(const Size imageRes, int minDisparity = 4, int numDisparities = 256, int blockSize = 3,
                         int smoothingRadius = -1, int smoothingIterations = 5, int paths = 4, int p1 = 10, int p2 = 120,
                         int uniquenessRatio = 12);
    system_data_t runInternal(System &system, SystemRunData &data) override;
    double meanFramesPerLaunch() const;  // frames per launch sequence so far (1 when coalescing is off)

   private:
    const Size imageRes;
    std::shared_ptr<EngineHandle> engine;
    std::shared_ptr<class DisparityCoalescer> coalescer;  // NULL when CARTSLAM_COALESCE=0: one launch sequence per frame
};

class ImageDisparityDerivativeModule : public SyncWrapperSystemModule {
   public:
    ImageDisparityDerivativeModule();
    system_data_t runInternal(System &system, SystemRunData &data) override;

   private:
    std::mutex engineMutex;
    std::shared_ptr<EngineHandle> engine;  // created on first use from the disparity image's size
};
}  // namespace cart
