// cartconfig.hpp -- JSON -> DataSource + modules, accepting the reference's config files verbatim for the hot-path
// module types (src/cartconfig.cpp:56-80 providers, :82-104 sources, :106-228 module factory, :230-277 readers).
#pragma once
#include <memory>
#include <string>

#include "cartslam.hpp"

namespace cart::config {
std::shared_ptr<cart::DataSource> readDataSourceConfig(const std::string path);
void readModuleConfig(const std::string path, std::shared_ptr<cart::System> system);
std::shared_ptr<cart::System> readSystemConfig(const std::string path);
// same, from JSON text (used by tests)
void applyModuleConfigText(const std::string &json, std::shared_ptr<cart::System> system);
}  // namespace cart::config
