// cart_slam_amd <source.json> <modules.json> [--frames N] [--dump DIR] [--sequential 1] [--inflight N] [--timing FILE.csv]
// (--sequential 1 finishes every frame before the next starts: the cumulative plane histogram then sees the frames in id
//  order, which the reference's concurrent frame loop does not guarantee)
// Frame loop of the reference's src/main.cpp:8-63 without logging/UI; --dump writes every frame's blackboard images as
// raw little-endian files (<DIR>/<id>_<key>.bin) so that tests can compare them with the oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <deque>
#include <future>
#include <fstream>
#include <iostream>

#include "cartslam_amd/cartconfig.hpp"
#include "cartslam_amd/modules/depth.hpp"
#include "cartslam_amd/timing.hpp"
#include "cartslam_amd/modules/planeseg.hpp"

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << argv[0] << " <data source config file> <module config file> [--frames N] [--dump DIR]\n";
        return 1;
    }
    int maxFrames = 1 << 30;
    std::string dump;
    bool sequential = false;
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--frames")) maxFrames = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--dump")) dump = argv[i + 1];
        else if (!std::strcmp(argv[i], "--sequential")) sequential = std::atoi(argv[i + 1]) != 0;
        else if (!std::strcmp(argv[i], "--inflight")) setenv("CARTSLAM_CONCURRENT_RUN_LIMIT", argv[i + 1], 1);  // frames in flight (reference: 12)
        else if (!std::strcmp(argv[i], "--timing")) cart::timing::Sink::instance().open(argv[i + 1]);
    }
    try {
        auto dataSource = cart::config::readDataSourceConfig(argv[1]);
        const size_t inflight = cart::concurrentRunLimit();
        auto system = std::make_shared<cart::System>(dataSource, std::max<size_t>(CARTSLAM_RUN_RETENTION, inflight + 8), inflight);
        cart::config::readModuleConfig(argv[2], system);
        std::deque<std::future<void>> pending;   // at most `inflight` + 1 entries: finished frames are reaped as the loop goes
        int frames = 0, failed = 0;
        auto reap = [&](bool all) {
            while (!pending.empty() && (all || pending.front().wait_for(std::chrono::seconds(0)) == std::future_status::ready)) {
                try { pending.front().get(); } catch (const std::exception &e) { std::cerr << "Error in processing: " << e.what() << "\n"; ++failed; }
                pending.pop_front();
            }
        };
        while (!dataSource->isFinished() && frames < maxFrames) {
            if (!dataSource->isNextReady()) continue;
            pending.push_back(system->run());
            if (sequential) pending.back().wait();
            ++frames;
            reap(false);
        }
        reap(true);
        if (!dump.empty()) {
            const char *keys[] = {CARTSLAM_KEY_DISPARITY, CARTSLAM_KEY_DISPARITY_DERIVATIVE, CARTSLAM_KEY_DISPARITY_DERIVATIVE_HISTOGRAM, CARTSLAM_KEY_PLANES,
                                  CARTSLAM_KEY_PLANE_COMPONENTS, CARTSLAM_KEY_DEPTH, CARTSLAM_KEY_PLANES_UNSMOOTHED, CARTSLAM_KEY_SUPERPIXELS, CARTSLAM_KEY_OPTFLOW,
                                  CARTSLAM_KEY_PLANE_COMPONENT_TABLE, CARTSLAM_KEY_PLANE_COMPONENT_COUNT};
            for (int id = 1; id <= frames; ++id) {
                std::shared_ptr<cart::SystemRunData> run;
                try { run = system->getRunById((uint32_t)id); } catch (const std::exception &) { continue; }  // evicted (retention ring)
                if (run->hasData(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL)) {
                    const cart::contour::label_t mx = *run->getData<cart::contour::label_t>(CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL);
                    std::ofstream o(dump + "/" + std::to_string(id) + "_" + CARTSLAM_KEY_SUPERPIXELS_MAX_LABEL + ".bin", std::ios::binary);
                    o.write(reinterpret_cast<const char *>(&mx), sizeof(mx));
                }
                for (const char *k : keys) {
                    if (!run->hasData(k)) continue;
                    auto img = run->getData<cart::image_t>(k);
                    if (!img) continue;  // e.g. "optflow" of the first frame (optflow.cpp:126-128)
                    auto bytes = img->downloadTight();
                    std::ofstream o(dump + "/" + std::to_string(id) + "_" + k + ".bin", std::ios::binary);
                    o.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size());
                }
            }
        }
        if (!dump.empty()) {
            std::ofstream q(dump + "/Q.bin", std::ios::binary);
            const cart::CameraIntrinsics K = dataSource->getCameraIntrinsics();
            q.write(reinterpret_cast<const char *>(K.Q), sizeof(K.Q));
        }
        std::cout << "frames " << frames << " failed " << failed;
        for (const auto &m : system->getModules())
            if (auto d = std::dynamic_pointer_cast<cart::ImageDisparityModule>(m)) std::cout << " frames_per_launch " << d->meanFramesPerLaunch();
        std::cout << "\n";
        return failed ? 2 : 0;
    } catch (const std::exception &e) {
        std::cerr << "fatal: " << e.what() << "\n";
        return 1;
    }
}
