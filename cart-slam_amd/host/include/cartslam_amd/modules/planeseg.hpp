// modules/planeseg.hpp -- mirrors include/modules/planeseg.hpp:15-162 (keys, Plane, PlaneParameters, the two
// parameter providers, DisparityPlaneSegmentationModule incl. temporal smoothing).  Temporal smoothing consumes the
// "optflow" key (S10.5 CV_16SC2); the reference's provider of that key is NVIDIA fixed-function hardware
// (src/modules/optflow.cpp): ImageOpticalFlowModule below provides it by census block matching, and any other module
// providing "optflow" will do (OpticalFlowFileModule
// replays flow fields from files).
#pragma once
#include <mutex>
#include <shared_mutex>
#include <utility>

#include "disparity.hpp"
#include "superpixels.hpp"

#define CARTSLAM_KEY_PLANES "planes"
#define CARTSLAM_KEY_PLANES_UNSMOOTHED "planes_unsmoothed"
#define CARTSLAM_KEY_PLANE_PARAMETERS "plane_parameters"
#define CARTSLAM_KEY_DISPARITY_DERIVATIVE_HIST "disp_derivative_histogram"
#define CARTSLAM_KEY_OPTFLOW "optflow"  // include/modules/optflow.hpp
#define CARTSLAM_KEY_PLANE_COMPONENTS "plane_components"  // new: connected-component ids (no reference counterpart)
#define CARTSLAM_KEY_PLANE_COMPONENT_TABLE "plane_component_table"  // new: CV_32SC1 [4096 x 7]: {id, label, area, x0, y0, x1, y1} per row
#define CARTSLAM_KEY_PLANE_COMPONENT_COUNT "plane_component_count"  // new: int32, number of components (rows beyond 4096 are dropped)
#define CARTSLAM_PLANE_COMPONENT_TABLE_ROWS 4096
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
class SuperPixelDisparityPlaneSegmentationModule;

class PlaneParameterProvider {
   public:
    virtual ~PlaneParameterProvider() = default;
    PlaneParameters getPlaneParameters() const { return PlaneParameters(horizontalCenter, verticalCenter, horizontalRange, verticalRange); }
    friend class DisparityPlaneSegmentationModule;
    friend class SuperPixelDisparityPlaneSegmentationModule;

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

    const bool useTemporalSmoothing;
    const unsigned int temporalSmoothingDistance;
    const int updateInterval;
    const int resetInterval;
    const bool labelComponents;
    std::shared_ptr<PlaneParameterProvider> planeParameterProvider;
    std::shared_mutex derivativeHistogramMutex;
    int32_t *derivativeHistogram = nullptr;  // persistent 256-bin device histogram (planeseg.hpp:160-161)
    std::mutex engineMutex;
    std::shared_ptr<EngineHandle> engine;
    std::shared_ptr<class PlaneCoalescer> coalescer;  // frames that wait together share one launch per stage (modules.cpp); NULL when CARTSLAM_COALESCE=0
    void ensureHistogram();
};

// mirrors include/modules/planeseg.hpp:164-186 + src/modules/planeseg/sp_planeseg.cu:180-388: per-pixel classification of
// the vertical directional derivative, optional temporal vote, majority vote per superpixel.
class SuperPixelDisparityPlaneSegmentationModule : public SyncWrapperSystemModule {
   public:
    SuperPixelDisparityPlaneSegmentationModule(std::shared_ptr<PlaneParameterProvider> planeParameterProvider, const int updateInterval = 30, const int resetInterval = 10,
                                               const bool useTemporalSmoothing = false,
                                               const unsigned int temporalSmoothingDistance = CARTSLAM_PLANE_TEMPORAL_DISTANCE_DEFAULT);
    system_data_t runInternal(System &system, SystemRunData &data) override;
    void frameFinished(uint32_t id) noexcept override { order.finish(id); }
    void attached(uint32_t firstFrameId) noexcept override { order.startAt(firstFrameId); }

   private:
    void updatePlaneParameters(System &system, SystemRunData &data);  // sp_planeseg.cu:349-387

    const bool useTemporalSmoothing;
    const unsigned int temporalSmoothingDistance;
    const int updateInterval;
    const int resetInterval;
    std::shared_ptr<PlaneParameterProvider> planeParameterProvider;
    FrameOrder order;
    std::vector<int32_t> derivativeHistogram;  // running total on the host (planeseg.hpp:185)
    std::mutex engineMutex;
    std::shared_ptr<EngineHandle> engine;
};

typedef int16_t optical_flow_t;  // S10.5, two channels (include/modules/optflow.hpp)

// mirrors include/modules/optflow.hpp:23-41 + src/modules/optflow.cpp:52-140: same name, key and frame logic (no flow for
// the first frame; flow between the reference images of frame id and id-1).  The reference computes it on NVIDIA's
// fixed-function optical-flow engine (cv::cuda::NvidiaOpticalFlow_2_0, grid size 1, S10.5); here it is dense census block
// matching (cart_optical_flow, oracle S15) with the same output format.  searchRadius / blockRadius are extensions.
class ImageOpticalFlowModule : public SyncWrapperSystemModule {
   public:
    explicit ImageOpticalFlowModule(const Size imageRes, int searchRadius = 8, int blockRadius = 2);
    system_data_t runInternal(System &system, SystemRunData &data) override;

   private:
    std::shared_ptr<EngineHandle> engine;
    const int searchRadius, blockRadius;
};

// Stand-in provider of "optflow": <sequence dir>/flow/%06d.bin, raw int16 [h][w][2], frame index = run id - 1.
class OpticalFlowFileModule : public SyncWrapperSystemModule {
   public:
    OpticalFlowFileModule() : SyncWrapperSystemModule("ImageOpticalFlow") { this->providesData.push_back(CARTSLAM_KEY_OPTFLOW); }
    system_data_t runInternal(System &system, SystemRunData &data) override;
};
}  // namespace cart
