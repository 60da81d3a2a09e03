// cartslam.hpp -- minimal System / SystemRunData able to drive modules exactly like the reference's
// (include/cartslam.hpp:3-5,27-117, src/cartslam.cpp:60-334): per-frame blackboard, dependency wait, one
// future per module, bounded number of frames in flight, run retention ring.  Scheduling details that are
// not on the arithmetic path (cross-frame runOffset waits, out-of-order completion bookkeeping) are reduced
// to what the three hot-path modules need.
#pragma once

#define CARTSLAM_RUN_RETENTION 32
#define CARTSLAM_CONCURRENT_RUN_LIMIT 12
#define CARTSLAM_WORKER_THREADS (16 * CARTSLAM_CONCURRENT_RUN_LIMIT)

#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "data.hpp"
#include "datasource.hpp"
#include "module.hpp"

namespace cart {
// Frames in flight: the reference's compile-time CARTSLAM_CONCURRENT_RUN_LIMIT (include/cartslam.hpp:4), overridable at run
// time by the environment variable of the same name (cart_slam_amd --inflight N sets it).  Engines size their workspace
// rings with it, so it has to be fixed before the modules are constructed.
size_t concurrentRunLimit();


class System;

// The worker pool every frame task, module waiter and SyncWrapperSystemModule::run is posted to (the reference's
// boost::asio::thread_pool, include/cartslam.hpp:5,104: 16 x run-limit threads).  Threads are started on demand and kept;
// a task that finds no idle worker gets a new one even beyond `workerThreads`, so that tasks blocked on their
// dependencies can never starve the task that provides them (a fixed pool deadlocks on a long module list with many
// frames in flight; the reference's 192 threads merely make that unlikely).
class WorkerPool {
   public:
    explicit WorkerPool(size_t workerThreads) : workerThreads(workerThreads) {}
    ~WorkerPool();   // runs what is queued, then joins
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;

    template <class F>
    auto post(F &&f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::forward<F>(f));
        auto future = task->get_future();
        enqueue([task]() { (*task)(); });
        return future;
    }
    size_t threadCount();      // threads started so far (diagnostics)
    const size_t workerThreads;   // nominal size

   private:
    void enqueue(std::function<void()> fn);
    void work();
    std::mutex mutex;
    std::condition_variable wake;
    std::deque<std::function<void()>> queue;
    std::vector<std::thread> threads;
    size_t idle = 0;
    bool stopping = false;
};

class SystemRunData : public DataContainer {
   public:
    SystemRunData(uint32_t id, System *system, std::shared_ptr<DataElement> dataElement) : dataElement(dataElement), id(id), system(system) {}
    std::shared_ptr<SystemRunData> getRelativeRun(const int8_t offset);
    std::shared_ptr<DataElement> dataElement;
    const uint32_t id;  // 1-based frame id (cartslam.cpp:194)

   private:
    System *system;
};

class System : public DataContainer {
   public:
    explicit System(std::shared_ptr<DataSource> dataSource, size_t runRetention = CARTSLAM_RUN_RETENTION,
                    size_t concurrentRunLimit = cart::concurrentRunLimit(), size_t workerThreads = 0 /* 16 x concurrentRunLimit */);
    ~System();

    // one frame: next data element, every module once (cartslam.cpp:228-334). The future resolves when all
    // modules of the frame have finished; exceptions of modules propagate through it.
    std::future<void> run();

    template <typename T, typename... Args>
    void addModule(Args... args) { addModule(std::make_shared<T>(args...)); }
    void addModule(std::shared_ptr<SystemModule> module);

    template <typename T>
    std::shared_ptr<T> getModule() {
        for (const auto &m : modules)
            if (auto c = std::dynamic_pointer_cast<T>(m)) return c;
        throw std::invalid_argument("Could not find module");
    }

    std::shared_ptr<SystemRunData> getRunById(const uint32_t id);
    void insertGlobalData(const std::string &key, std::shared_ptr<void> data) { insertData(std::make_pair(key, data)); }
    const std::shared_ptr<DataSource> getDataSource() const { return dataSource; }
    WorkerPool &getThreadPool() { return threadPool; }   // include/cartslam.hpp:86
    const std::vector<std::shared_ptr<SystemModule>> &getModules() const { return modules; }

   private:
    void verifyDependencies();  // every required same-frame key has a provider (cartslam.cpp:74-90)

    const size_t runRetention, concurrentRunLimit;
    bool verifiedDependencies = false;
    uint32_t runId = 0;
    size_t activeRuns = 0;
    std::set<uint32_t> activeIds;   // frames whose modules are still running (guarded by runMutex)
    uint32_t maxBackOffset = 0;     // largest |runOffset| any module asks for (verifyDependencies)
    std::shared_ptr<DataSource> dataSource;
    std::vector<std::shared_ptr<SystemModule>> modules;
    std::vector<std::shared_ptr<SystemRunData>> runs;
    std::mutex runMutex;
    std::condition_variable runCondition;
    WorkerPool threadPool;   // last member: destroyed (drained and joined) first, while everything its tasks touch is still alive
};
}  // namespace cart
