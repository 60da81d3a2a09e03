#include <cstdlib>
#include <algorithm>
#include "cartslam_amd/cartslam.hpp"

#include <algorithm>

#include "cartslam_amd/timing.hpp"

namespace cart {

// src/modules/module.cpp:7-19: post runInternal to the system's worker pool and hand back its future
std::future<system_data_t> SyncWrapperSystemModule::run(System &system, SystemRunData &data) {
    return system.getThreadPool().post([this, &system, &data]() { return this->runInternal(system, data); });
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> lock(mutex);
        stopping = true;
    }
    wake.notify_all();
    for (auto &t : threads) t.join();
}

size_t WorkerPool::threadCount() {
    std::lock_guard<std::mutex> lock(mutex);
    return threads.size();
}

void WorkerPool::enqueue(std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lock(mutex);
        // more queued tasks than idle workers to take them: one more thread (every queued task is guaranteed a worker
        // without waiting for a running -- possibly blocked -- task to end).  The thread comes first: if it cannot be created
        // (std::system_error under a pid limit) the caller gets the exception and nothing is left queued without a worker.
        if (queue.size() + 1 > idle) threads.emplace_back([this]() { work(); });
        queue.push_back(std::move(fn));
    }
    wake.notify_one();
}

void WorkerPool::work() {
    std::unique_lock<std::mutex> lock(mutex);
    for (;;) {
        while (queue.empty() && !stopping) {
            ++idle;
            wake.wait(lock);
            --idle;
        }
        if (queue.empty()) return;   // stopping and drained
        std::function<void()> fn = std::move(queue.front());
        queue.pop_front();
        lock.unlock();
        fn();   // packaged_task: exceptions end up in the future
        fn = nullptr;
        lock.lock();
    }
}

std::shared_ptr<SystemRunData> SystemRunData::getRelativeRun(const int8_t offset) {
    if (offset > 0) throw std::invalid_argument("Offset must be negative or zero");
    const int64_t target = (int64_t)id + offset;
    if (target <= 0) throw std::invalid_argument("Index out of range");
    return system->getRunById((uint32_t)target);
}

size_t concurrentRunLimit() {
    const char *env = std::getenv("CARTSLAM_CONCURRENT_RUN_LIMIT");
    const long n = env ? std::atol(env) : 0;
    return n > 0 ? (size_t)std::min(n, 64L) : (size_t)CARTSLAM_CONCURRENT_RUN_LIMIT;
}

System::System(std::shared_ptr<DataSource> dataSource, size_t runRetention, size_t concurrentRunLimit, size_t workerThreads)
    : runRetention(runRetention), concurrentRunLimit(concurrentRunLimit), dataSource(dataSource),
      threadPool(workerThreads ? workerThreads : 16 * concurrentRunLimit) {}

System::~System() {
    std::unique_lock<std::mutex> lock(runMutex);
    runCondition.wait(lock, [this] { return activeRuns == 0; });
}

void System::addModule(std::shared_ptr<SystemModule> module) {
    std::unique_lock<std::mutex> lock(runMutex);
    module->attached(runId + 1);   // a module added to a running System starts with the next frame, not with frame 1
    modules.push_back(module);
}

std::shared_ptr<SystemRunData> System::getRunById(const uint32_t id) {
    std::unique_lock<std::mutex> lock(runMutex);
    for (auto &r : runs)
        if (r->id == id) return r;
    throw std::invalid_argument("Index out of range");
}

void System::verifyDependencies() {
    std::map<std::string, bool> provided;
    for (const auto &m : modules)
        for (const auto &k : m->getProvidedData()) provided[k] = true;
    for (const auto &m : modules)
        for (const auto &d : m->getRequiredData()) {
            if (!d.optional && !provided.count(d.name)) throw std::invalid_argument("Module " + m->name + " requires \"" + d.name + "\" but no module provides it");
            if (d.runOffset < 0) maxBackOffset = std::max<uint32_t>(maxBackOffset, (uint32_t)(-(int)d.runOffset));
        }
    verifiedDependencies = true;
}

std::future<void> System::run() {
    if (!verifiedDependencies) verifyDependencies();
    std::shared_ptr<SystemRunData> run;
    {
        std::unique_lock<std::mutex> lock(runMutex);
        runCondition.wait(lock, [this] { return activeRuns < concurrentRunLimit; });  // cartslam.cpp:196-198
        auto sourceTiming = timing::initTiming("DataSource", runId + 1);
        timing::startTiming(sourceTiming);
        auto element = dataSource->getNext();
        timing::endTiming(sourceTiming);
        run = std::make_shared<SystemRunData>(++runId, this, element);
        runs.push_back(run);
        activeIds.insert(run->id);
        // cartslam.cpp:202-205 drops the oldest run as soon as the ring is full.  A frame that is still running may need that run
        // (its own blackboard, or the earlier frames its modules depend on): with 12 frames in flight a single slow frame is
        // overtaken by 32 newer ones in a few milliseconds.  The ring therefore keeps what the oldest active frame can still ask for.
        const uint32_t oldestActive = *activeIds.begin();
        while (runs.size() > runRetention && runs.front()->id + maxBackOffset < oldestActive) runs.erase(runs.begin());
        ++activeRuns;
    }
    std::vector<std::shared_ptr<SystemModule>> mods;
    {
        std::unique_lock<std::mutex> lock(runMutex);
        mods = modules;
    }
    // the frame could not be handed to the pool (thread creation failed): take it back, and tell every module that this id
    // will never come, so that id-ordered modules do not wait for it
    auto abandon = [this, run, &mods]() {
        for (const auto &m : mods) m->frameFinished(run->id);
        {
            std::unique_lock<std::mutex> lock(runMutex);
            --activeRuns;
            activeIds.erase(run->id);
        }
        runCondition.notify_all();
    };
    try {
    return threadPool.post([this, run, mods]() {
        auto frameTiming = timing::initTiming("Frame", run->id);  // cartslam.cpp:245-251
        timing::startTiming(frameTiming);
        std::exception_ptr first;
        std::vector<std::future<void>> done;
        for (size_t mi = 0; mi < mods.size(); ++mi) {
            const auto &m = mods[mi];
            // every module gets its own waiter: dependencies first (cartslam.cpp:96-167), then the module, then the
            // returned (key, ptr) pairs go onto the frame's blackboard (cartslam.cpp:279-301)
            try {
            done.push_back(threadPool.post([this, run, m]() {
                auto moduleTiming = timing::initTiming(m->name, run->id);  // cartslam.cpp:259-262: init before the dependency wait
                struct Finished {   // the module hears about the end of this frame on every way out, exceptions included
                    SystemModule &m; uint32_t id;
                    ~Finished() { m.frameFinished(id); }
                } finished{*m, run->id};
                std::vector<std::string> same_frame;
                for (const auto &d : m->getRequiredData()) {
                    if (d.runOffset == 0) { same_frame.push_back(d.name); continue; }
                    if ((int64_t)run->id + d.runOffset <= 0) continue;
                    try { run->getRelativeRun(d.runOffset)->waitForData({d.name}); }
                    catch (...) { if (!d.optional) throw; }
                }
                run->waitForData(same_frame);
                timing::startTiming(moduleTiming);  // :270
                system_data_t out = m->run(*this, *run).get();
                for (const auto &kv : out) run->insertData(kv);
                timing::endTiming(moduleTiming);  // :290
            }));
            } catch (...) {   // this waiter and the ones after it never start: their modules hear that the frame is over
                if (!first) first = std::current_exception();
                for (size_t k = mi; k < mods.size(); ++k) mods[k]->frameFinished(run->id);
                break;
            }
        }
        for (auto &f : done) {
            try { f.get(); } catch (...) { if (!first) first = std::current_exception(); }
        }
        timing::endTiming(frameTiming);
        {
            std::unique_lock<std::mutex> lock(runMutex);
            --activeRuns;
            activeIds.erase(run->id);
        }
        runCondition.notify_all();
        if (first) std::rethrow_exception(first);
    });
    } catch (...) {
        abandon();
        throw;
    }
}
}  // namespace cart
