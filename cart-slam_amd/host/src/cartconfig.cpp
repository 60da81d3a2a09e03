// cartconfig.cpp -- module / data-source factory. Type strings, keys and defaults are the reference's
// (src/cartconfig.cpp:56-80, :82-104, :144-152, :161-163, :198-206); GUI module types ("*_visualization") are
// accepted and skipped so that the reference's config files load unchanged; module types outside the hot path throw
// the reference's "Unknown module type" error.
#include "cartslam_amd/cartconfig.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "cartslam_amd/json.hpp"
#include "cartslam_amd/modules/depth.hpp"
#include "cartslam_amd/modules/disparity.hpp"
#include "cartslam_amd/modules/planeseg.hpp"
#include "cartslam_amd/modules/superpixels.hpp"

#define CART_CONFIG_KEY_DATA_SOURCE "data_source"
#define CART_CONFIG_KEY_MODULES "modules"

namespace cart::config {
namespace {
using json::Value;

template <typename T>
T get(const Value &data, const std::string &key, const T &defaultValue) {
    if (!data.contains(key)) return defaultValue;
    return data.at(key).get<T>();
}
template <typename T>
T get(const Value &data, const std::string &key) { return data.at(key).get<T>(); }  // throws "Key <k> not found."

std::string slurp(const std::string &path) {
    std::string p = path;
    if (!p.empty() && p[0] == '~') if (const char *home = std::getenv("HOME")) p = std::string(home) + p.substr(1);
    std::ifstream file(p);
    if (!file.is_open()) throw std::runtime_error("Could not open file " + path + ": " + std::strerror(errno));
    std::stringstream ss;
    ss << file.rdbuf();
    return ss.str();
}

std::shared_ptr<PlaneParameterProvider> readParameterProvider(const Value &data) {
    if (!data.contains("type")) throw std::runtime_error("Parameter provider type not found.");
    const std::string providerType = data.at("type").get<std::string>();
    if (providerType == "static") {
        auto horizontalRange = std::make_pair(get<int>(data, "horizontal_range_min"), get<int>(data, "horizontal_range_max"));
        auto verticalRange = std::make_pair(get<int>(data, "vertical_range_min"), get<int>(data, "vertical_range_max"));
        const int horizontalCenter = (horizontalRange.first + horizontalRange.second) / 2;
        const int verticalCenter = (verticalRange.first + verticalRange.second) / 2;
        return std::make_shared<StaticPlaneParameterProvider>(horizontalCenter, verticalCenter, horizontalRange, verticalRange);
    }
    if (providerType == "histogram_peak") return std::make_shared<HistogramPeakPlaneParameterProvider>();
    throw std::runtime_error("Unknown parameter provider type.");
}

std::shared_ptr<DataSource> createDataSource(const Value &cfg) {
    if (!cfg.is_object()) throw std::runtime_error("Data source configuration is not an object.");
    const std::string sourcePath = cfg.at("path").get<std::string>();
    const std::string type = cfg.at("type").get<std::string>();
    // image_width / image_height are an extension: the reference's factory (cartconfig.cpp:95-98) never passes its ctor's imageSize
    if (type == "kitti")
        return std::make_shared<sources::KITTIDataSource>(sourcePath, get(cfg, "sequence", 0), Size{get(cfg, "image_width", 0), get(cfg, "image_height", 0)});
    if (type == "zed") throw std::runtime_error("Data source type zed needs the proprietary ZED SDK: not supported.");
    throw std::runtime_error("Unknown data source type.");
}

bool endsWith(const std::string &s, const std::string &suffix) { return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0; }

void applyModuleConfig(const Value &modulesConfig, std::shared_ptr<System> system) {
    if (!modulesConfig.is_array()) throw std::runtime_error("Modules configuration is not an array.");
    auto dataSource = system->getDataSource();
    for (const auto &moduleConfig : modulesConfig.arr) {
        if (!moduleConfig.is_object()) throw std::runtime_error("Module configuration is not an object.");
        const std::string moduleType = moduleConfig.at("type").get<std::string>();
        if (moduleType == "disparity") {
            system->addModule<ImageDisparityModule>(dataSource->getImageSize(), get(moduleConfig, "min_disparity", 4), get(moduleConfig, "num_disparities", 256),
                                                    get(moduleConfig, "block_size", 3), get(moduleConfig, "smoothing_radius", -1),
                                                    get(moduleConfig, "smoothing_iterations", 5),
                                                    // extensions (not in the reference's JSON): OpenCV's createStereoSGM knobs
                                                    get(moduleConfig, "paths", 4), get(moduleConfig, "p1", 10), get(moduleConfig, "p2", 120),
                                                    get(moduleConfig, "uniqueness_ratio", 12));
        } else if (moduleType == "depth") {  // cartconfig.cpp:138-140
            system->addModule<DepthModule>();
        } else if (moduleType == "disparity_derivative") {
            system->addModule<ImageDisparityDerivativeModule>();
        } else if (moduleType == "disparity_planeseg") {
            const auto parameterProvider = readParameterProvider(moduleConfig.at("parameter_provider"));
            system->addModule<DisparityPlaneSegmentationModule>(parameterProvider, get(moduleConfig, "update_interval", 30), get(moduleConfig, "reset_interval", 10),
                                                                get(moduleConfig, "use_temporal_smoothing", false),
                                                                (unsigned)get(moduleConfig, "temporal_smoothing_distance", CARTSLAM_PLANE_TEMPORAL_DISTANCE_DEFAULT),
                                                                get(moduleConfig, "label_components", false));
        } else if (moduleType == "superpixels") {  // cartconfig.cpp:121-134
            const double direct = get(moduleConfig, "direct_clique_cost", 0.5);
            system->addModule<SuperPixelModule>(dataSource->getImageSize(), (unsigned)get(moduleConfig, "initial_iterations", 18), (unsigned)get(moduleConfig, "iterations", 6),
                                                (unsigned)get(moduleConfig, "block_size", 12), (unsigned)get(moduleConfig, "reset_iterations", 64), direct,
                                                get(moduleConfig, "diagonal_clique_cost", direct / std::sqrt(2.0)), get(moduleConfig, "compactness_weight", 0.1),
                                                get(moduleConfig, "progressive_compactness_cost", 0.0), get(moduleConfig, "image_weight", 1.5),
                                                get(moduleConfig, "disparity_weight", 1.0));
        } else if (moduleType == "superpixel_disparity_planeseg") {  // cartconfig.cpp:212-220
            const auto parameterProvider = readParameterProvider(moduleConfig.at("parameter_provider"));
            system->addModule<SuperPixelDisparityPlaneSegmentationModule>(parameterProvider, get(moduleConfig, "update_interval", 30), get(moduleConfig, "reset_interval", 10),
                                                                          get(moduleConfig, "use_temporal_smoothing", false),
                                                                          (unsigned)get(moduleConfig, "temporal_smoothing_distance", CARTSLAM_PLANE_TEMPORAL_DISTANCE_DEFAULT));
        } else if (moduleType == "optflow_file") {  // extension: replays flow fields from <sequence>/flow/%06d.bin
            system->addModule<OpticalFlowFileModule>();
        } else if (moduleType == "optflow") {  // cartconfig.cpp:183-185; search_radius / block_radius are extensions
            system->addModule<ImageOpticalFlowModule>(dataSource->getImageSize(), get(moduleConfig, "search_radius", 8), get(moduleConfig, "block_radius", 2));
        } else if (endsWith(moduleType, "_visualization")) {
            std::cerr << "[cartconfig] skipping GUI module type " << moduleType << " (out of scope)\n";
        } else {
            throw std::runtime_error("Unknown module type " + moduleType + ".");
        }
    }
}
}  // namespace

std::shared_ptr<DataSource> readDataSourceConfig(const std::string path) { return createDataSource(json::parse(slurp(path))); }

void readModuleConfig(const std::string path, std::shared_ptr<System> system) { applyModuleConfig(json::parse(slurp(path)), system); }

void applyModuleConfigText(const std::string &text, std::shared_ptr<System> system) { applyModuleConfig(json::parse(text), system); }

std::shared_ptr<System> readSystemConfig(const std::string path) {
    const Value data = json::parse(slurp(path));
    if (!data.contains(CART_CONFIG_KEY_DATA_SOURCE)) throw std::runtime_error("Data source not found in configuration file.");
    if (!data.contains(CART_CONFIG_KEY_MODULES)) throw std::runtime_error("Modules not found in configuration file.");
    auto system = std::make_shared<System>(createDataSource(data.at(CART_CONFIG_KEY_DATA_SOURCE)));
    applyModuleConfig(data.at(CART_CONFIG_KEY_MODULES), system);
    return system;
}
}  // namespace cart::config
