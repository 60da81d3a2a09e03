// module.hpp -- the plugin API (drop-in boundary). Mirrors include/modules/module.hpp:14-56 and
// src/modules/module.cpp:7-19 of the reference with std:: in place of boost:: futures.
#pragma once
#include <cstdint>
#include <future>
#include <string>
#include <vector>

#include "data.hpp"

namespace cart {

class System;
class SystemRunData;

struct module_dependency_t {
    std::string name;
    int8_t runOffset;  // <= 0: same / earlier frame
    bool optional;

    module_dependency_t(const std::string &name, const int runOffset, const bool optional) : name(name), runOffset((int8_t)runOffset), optional(optional) {}
    module_dependency_t(const std::string &name, const int runOffset) : module_dependency_t(name, runOffset, false) {}
    module_dependency_t(const std::string &name) : module_dependency_t(name, 0, false) {}
    module_dependency_t() : module_dependency_t("", 0, false) {}
};

class SystemModule {
   public:
    explicit SystemModule(const std::string &name) : name(name) {}
    virtual ~SystemModule() = default;

    virtual std::future<system_data_t> run(System &system, SystemRunData &data) = 0;

    // Extension over the reference's interface: the System calls this once per frame and module when the module's work
    // for that frame is over -- after run() returned, threw, or was never started because a dependency failed.  Modules
    // that admit frames in id order (FrameOrder) use it to pass the turn of a frame that will never take it.
    virtual void frameFinished(uint32_t /*id*/) noexcept {}
    // Extension as well: System::addModule tells the module the id of the first frame it will see (1 for a module list built
    // before the first run; later for a module added to a running System), so that id-ordered modules do not wait for frames
    // that ran before they existed.
    virtual void attached(uint32_t /*firstFrameId*/) noexcept {}

    const std::vector<module_dependency_t> getRequiredData() const { return requiresData; }
    const std::vector<std::string> getProvidedData() const { return providesData; }

    const std::string name;

   protected:
    std::vector<module_dependency_t> requiresData;
    std::vector<std::string> providesData;
};

// run() posts runInternal to a worker thread and returns its future (module.cpp:7-19)
class SyncWrapperSystemModule : public SystemModule {
   public:
    explicit SyncWrapperSystemModule(const std::string &name) : SystemModule(name) {}
    std::future<system_data_t> run(System &system, SystemRunData &data) override;
    virtual system_data_t runInternal(System &system, SystemRunData &data) = 0;
};
}  // namespace cart
