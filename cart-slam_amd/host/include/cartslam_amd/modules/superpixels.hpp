// modules/superpixels.hpp -- mirrors include/modules/superpixels.hpp:11-41 + src/modules/superpixels.cu:19-118: the
// contour-relaxation superpixel module (same name, constructor arguments, blackboard keys, error texts).  The
// ContourRelaxation object with its three features lives behind cart_superpixels_* (include/cart_engine.h).
#pragma once
#include <set>
#include <condition_variable>
#include <mutex>

#include "disparity.hpp"

#define CARTSLAM_KEY_SUPERPIXELS "superpixels"
#define CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL "superpixels_max_label"

namespace cart {
namespace contour {
typedef uint16_t label_t;  // contourrelaxation/constants.hpp:35
}

// Stateful modules of the reference take frames in whatever order their lock is won (superpixels.cu:97-99,
// sp_planeseg.cu:356-371), which makes their output depend on thread timing.  FrameOrder admits frames in id order
// instead.  A frame that never takes its turn (its module failed, or a dependency did) is passed over explicitly: the
// System reports the end of every frame to the module (SystemModule::frameFinished -> finish), no timer is involved.
class FrameOrder {
   public:
    void finish(uint32_t id);   // idempotent: frame `id` has had (or will never take) its turn
    void startAt(uint32_t id);  // the first frame that will ever arrive (default 1); frames below it are taken as over
    class Turn {
       public:
        Turn(FrameOrder &o, uint32_t id);
        ~Turn();

       private:
        FrameOrder &order;
        uint32_t id;
    };

   private:
    std::mutex mutex;
    std::condition_variable cv;
    uint32_t next = 1;
    std::set<uint32_t> finishedAhead;   // ids > next that are already over
};

class SuperPixelModule : public SyncWrapperSystemModule {
   public:
    SuperPixelModule(const Size imageRes, const unsigned int initialIterations = 18, const unsigned int iterations = 6, const unsigned int blockSize = 12,
                     const unsigned int resetIterations = 64, const double directCliqueCost = 0.5, const double diagonalCliqueCost = 0.35355339059327373,
                     const double compactnessWeight = 0.05, const double progressiveCompactnessCost = 0.0, const double imageWeight = 1.0,
                     const double disparityWeight = 1.25);
    ~SuperPixelModule();
    system_data_t runInternal(System &system, SystemRunData &data) override;
    void frameFinished(uint32_t id) noexcept override { order.finish(id); }
    void attached(uint32_t firstFrameId) noexcept override { order.startAt(firstFrameId); }
    unsigned int getBlockSize() const { return blockSize; }

   private:
    std::shared_ptr<EngineHandle> engine;
    cart_superpixels *contourRelaxation = nullptr;
    FrameOrder order;
    const unsigned int initialIterations;
    const unsigned int iterations;
    const unsigned int resetIterations;
    const unsigned int blockSize;
    const bool requiresDisparityDerivative;
};
}  // namespace cart
