// modules/planeseg.hpp -- mirrors include/modules/planeseg.hpp:15-162 (keys, Plane, PlaneParameters, the two
// parameter providers, DisparityPlaneSegmentationModule).  Temporal smoothing needs the optical-flow module, which is
// NVIDIA fixed-function hardware in the reference and out of scope: requesting it throws at construction.
#pragma once
#include <mutex>
#include <shared_mutex>
#include <utility>

#include "disparity.hpp"

#define CARTSLAM_KEY_PLANES "planes"
#define CARTSLAM_KEY_PLANES_UNSMOOTHED "planes_unsmoothed"
#define CARTSLAM_KEY_PLANE_PARAMETERS "plane_parameters"
#define CARTSLAM_KEY_DISPARITY_DERIVATIVE_HIST "disp_derivative_histogram"
#define CARTSLAM_KEY_PLANE_COMPONENTS "plane_components"  // new: connected-component ids (no reference counterpart)
#define CARTSLAM_PLANE_COUNT 3
#define CARTSLAM_PLANE_TEMPORAL_DISTANCE_DEFAULT 3

namespace cart {

struct PlaneParameters {
    PlaneParameters(const int horizontalCenter, const int verticalCenter, const std::pair<int, int> horizontalRange, const std::pair<int, int> verticalRange)
        : horizontalRange(horizontalRange), verticalRange(verticalRange), horizontalCenter(horizontalCenter), verticalCenter(verticalCenter) {}
    const std::pair<int, int> horizontalRange;
    const std::pair<int, int> verticalRange;
    const int horizontalCenter;
    const int verticalCenter;
};

enum Plane { HORIZONTAL = 0, VERTICAL = 1, UNKNOWN = 2 };

class DisparityPlaneSegmentationModule;

class PlaneParameterProvider {
   public:
    virtual ~PlaneParameterProvider() = default;
    PlaneParameters getPlaneParameters() const { return PlaneParameters(horizontalCenter, verticalCenter, horizontalRange, verticalRange); }
    friend class DisparityPlaneSegmentationModule;

   protected:
    PlaneParameterProvider(const int horizontalCenter = 0, const int verticalCenter = 0, const std::pair<int, int> horizontalRange = std::make_pair(0, 0),
                           const std::pair<int, int> verticalRange = std::make_pair(0, 0))
        : horizontalRange(horizontalRange), verticalRange(verticalRange), horizontalCenter(horizontalCenter), verticalCenter(verticalCenter) {}
    virtual void updatePlaneParameters(System &system, SystemRunData &data, const std::vector<int32_t> &histogram) = 0;
    std::pair<int, int> horizontalRange;
    std::pair<int, int> verticalRange;
    int horizontalCenter;
    int verticalCenter;
};

class HistogramPeakPlaneParameterProvider : public PlaneParameterProvider {
   public:
    HistogramPeakPlaneParameterProvider() {}

   protected:
    void updatePlaneParameters(System &system, SystemRunData &data, const std::vector<int32_t> &histogram) override;  // planeseg.cu:405-458
};

class StaticPlaneParameterProvider : public PlaneParameterProvider {
   public:
    StaticPlaneParameterProvider(const int horizontalCenter, const int verticalCenter, const std::pair<int, int> horizontalRange, const std::pair<int, int> verticalRange)
        : PlaneParameterProvider(horizontalCenter, verticalCenter, horizontalRange, verticalRange) {}

   protected:
    void updatePlaneParameters(System &, SystemRunData &, const std::vector<int32_t> &) override {}
};

class DisparityPlaneSegmentationModule : public SyncWrapperSystemModule {
   public:
    DisparityPlaneSegmentationModule(std::shared_ptr<PlaneParameterProvider> planeParameterProvider, const int updateInterval = 30, const int resetInterval = 10,
                                     const bool useTemporalSmoothing = false, const unsigned int temporalSmoothingDistance = CARTSLAM_PLANE_TEMPORAL_DISTANCE_DEFAULT,
                                     const bool labelComponents = false);
    ~DisparityPlaneSegmentationModule();
    system_data_t runInternal(System &system, SystemRunData &data) override;

   private:
    void updatePlaneParameters(System &system, SystemRunData &data);  // planeseg.cu:379-403

    const int updateInterval;
    const int resetInterval;
    const bool labelComponents;
    std::shared_ptr<PlaneParameterProvider> planeParameterProvider;
    std::shared_mutex derivativeHistogramMutex;
    int32_t *derivativeHistogram = nullptr;  // persistent 256-bin device histogram (planeseg.hpp:160-161)
    std::mutex engineMutex;
    std::shared_ptr<EngineHandle> engine;
};
}  // namespace cart
