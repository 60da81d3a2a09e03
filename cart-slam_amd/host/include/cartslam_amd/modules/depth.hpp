// modules/depth.hpp -- mirrors include/modules/depth.hpp:7-19 + src/modules/depth.cpp:9-25.
#pragma once
#include <mutex>

#include "disparity.hpp"

#define CARTSLAM_KEY_DEPTH "depth"
#define CV_32FC3 21

namespace cart {
class DepthModule : public SyncWrapperSystemModule {
   public:
    DepthModule() : SyncWrapperSystemModule("Depth") {
        this->requiresData.push_back(module_dependency_t(CARTSLAM_KEY_DISPARITY));
        this->providesData.push_back(CARTSLAM_KEY_DEPTH);
    }
    system_data_t runInternal(System &system, SystemRunData &data) override;

   private:
    std::mutex engineMutex;
    std::shared_ptr<EngineHandle> engine;
};
}  // namespace cart
