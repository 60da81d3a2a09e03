// runtime_test -- the host runtime (System, per-frame blackboard, cross-frame dependencies, worker pool) without a GPU:
// dummy modules, 400 frames, 12 in flight.  Prints "ok" and returns 0; any failed check prints what failed and returns 1.
// Mirrors what src/cartslam.cpp:96-334 of the reference has to guarantee: a module runs once per frame after its
// dependencies (same frame and earlier frames), results land on the frame's blackboard, an exception of one module reaches
// the frame's future and no other frame, old frames leave the retention ring.
#include <atomic>
#include <cstdio>
#include <deque>
#include <future>
#include <thread>

#include <cstring>

#include "cartslam_amd/cartslam.hpp"
#include "cartslam_amd/coalescer.hpp"
#include "cartslam_amd/modules/superpixels.hpp"
#include "cartslam_amd/png.hpp"

using namespace cart;

namespace {
int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

class CountingSource : public DataSource {
   public:
    explicit CountingSource(int n) : DataSource(Size{}), left(n) {}
    bool isNextReady() override { return left > 0; }
    bool isFinished() override { return left <= 0; }
    DataElementType getProvidedType() override { return DataElementType::STEREO; }

   protected:
    std::shared_ptr<DataElement> getNextInternal() override { --left; return std::make_shared<StereoDataElement>(); }
    int left;
};

// a = 10 * id
class ModuleA : public SyncWrapperSystemModule {
   public:
    ModuleA() : SyncWrapperSystemModule("A") { providesData.push_back("a"); }
    system_data_t runInternal(System &, SystemRunData &data) override {
        std::this_thread::sleep_for(std::chrono::microseconds(200 + 37 * (data.id % 7)));   // frames finish out of order
        return MODULE_RETURN_SHARED("a", long, 10L * data.id);
    }
};
// b = a + 1; throws on frame 7
class ModuleB : public SyncWrapperSystemModule {
   public:
    ModuleB() : SyncWrapperSystemModule("B") { requiresData.push_back(module_dependency_t("a")); providesData.push_back("b"); }
    system_data_t runInternal(System &, SystemRunData &data) override {
        if (data.id == 7) throw std::runtime_error("frame 7 breaks in B");
        return MODULE_RETURN_SHARED("b", long, *data.getData<long>("a") + 1);
    }
};
// c = a + (a of the previous frame; frame 1 has none)
class ModuleC : public SyncWrapperSystemModule {
   public:
    ModuleC() : SyncWrapperSystemModule("C") {
        requiresData.push_back(module_dependency_t("a"));
        requiresData.push_back(module_dependency_t("a", -1));
        providesData.push_back("c");
    }
    system_data_t runInternal(System &, SystemRunData &data) override {
        long prev = 0;
        if (data.id > 1) {
            auto run = data.getRelativeRun(-1);
            prev = *run->getData<long>("a");
        }
        ++calls;
        return MODULE_RETURN_SHARED("c", long, *data.getData<long>("a") + prev);
    }
    std::atomic<int> calls{0};
};
// takes its frames in id order (like the superpixel and plane modules); records the order
class ModuleOrdered : public SyncWrapperSystemModule {
   public:
    ModuleOrdered() : SyncWrapperSystemModule("Ordered") { providesData.push_back("o"); }
    system_data_t runInternal(System &, SystemRunData &data) override {
        FrameOrder::Turn turn(order, data.id);
        std::lock_guard<std::mutex> lock(m);
        seen.push_back(data.id);
        return MODULE_RETURN_SHARED("o", long, (long)data.id);
    }
    void frameFinished(uint32_t id) noexcept override { order.finish(id); }
    void attached(uint32_t first) noexcept override { order.startAt(first); }
    FrameOrder order;
    std::mutex m;
    std::vector<uint32_t> seen;
};
}  // namespace

int main(int argc, char **argv) {
    // runtime_test --png FILE: decode with the host's PNG reader and dump "w h channels\n" + the pixels (tests/test_host.py)
    if (argc == 3 && !std::strcmp(argv[1], "--png")) {
        try {
            util::HostImage img;
            if (!util::readPng(argv[2], img)) { std::printf("missing\n"); return 2; }
            std::printf("%d %d %d\n", img.w, img.h, img.channels);
            std::fwrite(img.data.data(), 1, img.data.size(), stdout);
            return 0;
        } catch (const std::exception &e) {
            std::printf("error: %s\n", e.what());
            return 3;
        }
    }
    // 1. the pool alone: tasks that block on tasks posted after them (a fixed-size pool would deadlock), reuse of threads
    {
        WorkerPool pool(4);
        std::vector<std::future<int>> outer;
        for (int i = 0; i < 64; ++i)
            outer.push_back(pool.post([&pool, i]() {
                auto inner = pool.post([i]() { return i * i; });   // queued behind 63 blocked tasks in the worst case
                return inner.get() + 1;
            }));
        long sum = 0;
        for (auto &f : outer) sum += f.get();
        CHECK(sum == 64 + 63L * 64 * 127 / 6);
        const size_t started = pool.threadCount();
        CHECK(started >= 2 && started <= 128);
        for (int round = 0; round < 50; ++round) pool.post([]() { return 0; }).get();   // sequential tasks reuse idle workers
        CHECK(pool.threadCount() == started);
        auto thrower = pool.post([]() -> int { throw std::runtime_error("boom"); });
        bool threw = false;
        try { thrower.get(); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
    // 1b. the frame coalescer: 24 threads x 40 requests of two incompatible kinds; every request runs exactly once, in a group of
    //     its own kind of at most 6, never more than 2 groups at a time; a failing group fails each of its members and nobody else
    {
        struct Req : CoalescedRequest { int kind = 0, id = 0; int ran = 0; };
        std::atomic<int> inFlight{0}, maxInFlight{0}, groups{0}, biggest{0};
        std::atomic<bool> mixed{false}, oversized{false};
        FrameCoalescer<Req> co(
            6, 2, [](const Req &a, const Req &b) { return a.kind == b.kind; },
            [&](const std::vector<Req *> &g) {
                const int now = ++inFlight;
                int seen = maxInFlight.load();
                while (now > seen && !maxInFlight.compare_exchange_weak(seen, now)) {}
                if ((int)g.size() > 6) oversized = true;
                int big = biggest.load();
                while ((int)g.size() > big && !biggest.compare_exchange_weak(big, (int)g.size())) {}
                bool fail = false;
                for (Req *q : g) { if (q->kind != g[0]->kind) mixed = true; ++q->ran; fail |= q->id == 13; }
                ++groups;
                std::this_thread::sleep_for(std::chrono::microseconds(300));   // "the GPU is busy": later requests gather meanwhile
                --inFlight;
                if (fail) throw std::runtime_error("group with request 13");
            });
        std::vector<std::thread> threads;
        std::atomic<int> thrown{0}, completed{0}, ranTwice{0};
        for (int t = 0; t < 24; ++t)
            threads.emplace_back([&, t]() {
                for (int k = 0; k < 40; ++k) {
                    Req r; r.kind = t & 1; r.id = t * 40 + k;
                    try { co.run(r); ++completed; } catch (const std::runtime_error &) { ++thrown; }
                    if (r.ran != 1) ++ranTwice;
                }
            });
        for (auto &t : threads) t.join();
        CHECK(completed + thrown == 24 * 40);
        CHECK(ranTwice == 0);
        CHECK(!mixed && !oversized);
        CHECK(maxInFlight <= 2);
        CHECK(thrown >= 1 && thrown <= 6);          // request 13 and whoever shared its group
        CHECK(biggest >= 2);                        // groups did form while two were "on the GPU"
        CHECK(co.meanGroup() > 1.0 && groups < 24 * 40);
    }
    // 1c. FrameOrder: 60 frames enter from 12 threads in scrambled order and take their turns in id order; every fifth frame
    //     never takes its turn (its module "failed") and is only reported finished -- before or after its predecessors
    {
        FrameOrder order;
        std::mutex m;
        std::vector<uint32_t> sequence;
        std::vector<std::thread> threads, turns;
        // ids 1..60 dealt to 12 threads round-robin in reverse, so that late frames arrive first
        for (int t = 0; t < 12; ++t)
            threads.emplace_back([&, t]() {
                for (int k = 4; k >= 0; --k) {
                    const uint32_t id = (uint32_t)(1 + t + 12 * k);
                    if (k != 4) std::this_thread::sleep_for(std::chrono::microseconds(50 * (12 - t)));
                    if (id % 5 == 0) { order.finish(id); order.finish(id); continue; }   // idempotent
                    // frames of one thread come in descending order: each would block the thread for its predecessors, which
                    // live on other threads (and on this one!) -- so take the turn on a helper thread like the worker pool does
                    std::thread turn([&order, &m, &sequence, id]() {
                        FrameOrder::Turn turn(order, id);
                        std::lock_guard<std::mutex> lock(m);
                        sequence.push_back(id);
                    });
                    std::lock_guard<std::mutex> lock(m);
                    turns.push_back(std::move(turn));
                }
            });
        for (auto &t : threads) t.join();
        for (auto &t : turns) t.join();
        std::lock_guard<std::mutex> lock(m);
        CHECK(sequence.size() == 48);
        bool ascending = true;
        for (size_t i = 1; i < sequence.size(); ++i) ascending &= sequence[i] > sequence[i - 1];
        CHECK(ascending);
    }
    // 1d. an id-ordered module added to a System that has ALREADY run frames: it starts with the next frame instead of waiting
    //     for frames 1..20, which it will never be given (a hang here is the failure)
    {
        auto src = std::make_shared<CountingSource>(60);
        auto sys = std::make_shared<System>(src, CARTSLAM_RUN_RETENTION, 12);
        sys->addModule<ModuleA>();
        std::deque<std::future<void>> fs;
        for (int i = 0; i < 20; ++i) fs.push_back(sys->run());
        for (auto &f : fs) f.get();
        fs.clear();
        sys->addModule<ModuleOrdered>();
        while (!src->isFinished()) fs.push_back(sys->run());
        bool hung = false;
        for (auto &f : fs)
            if (f.wait_for(std::chrono::seconds(30)) != std::future_status::ready) { hung = true; break; } else f.get();
        CHECK(!hung);
        auto ordered = sys->getModule<ModuleOrdered>();
        std::lock_guard<std::mutex> lock(ordered->m);
        CHECK(ordered->seen.size() == 40 && ordered->seen.front() == 21 && ordered->seen.back() == 60);
        bool ascending = true;
        for (size_t i = 1; i < ordered->seen.size(); ++i) ascending &= ordered->seen[i] == ordered->seen[i - 1] + 1;
        CHECK(ascending);
        if (hung) { std::printf("FAILED: frames behind a late id-ordered module never finished\n"); return 1; }
    }
    // 2. the System: consumers listed before their providers, 12 frames in flight, retention 32
    const int frames = 400;
    auto source = std::make_shared<CountingSource>(frames);
    auto system = std::make_shared<System>(source, CARTSLAM_RUN_RETENTION, 12);
    system->addModule<ModuleC>();
    system->addModule<ModuleB>();
    system->addModule<ModuleA>();
    std::deque<std::future<void>> pending;
    int failed = 0, ran = 0;
    std::string message;
    while (!source->isFinished()) {
        pending.push_back(system->run());
        ++ran;
        while (pending.size() > 40) {
            try { pending.front().get(); } catch (const std::exception &e) { ++failed; message = e.what(); }
            pending.pop_front();
        }
    }
    while (!pending.empty()) {
        try { pending.front().get(); } catch (const std::exception &e) { ++failed; message = e.what(); }
        pending.pop_front();
    }
    if (failed != 1 || system->getModule<ModuleC>()->calls != frames)
        std::printf("failed frames %d (last message: %s), C ran %d times\n", failed, message.c_str(), (int)system->getModule<ModuleC>()->calls);
    CHECK(ran == frames);
    CHECK(failed == 1);                               // frame 7 only
    CHECK(message == "frame 7 breaks in B");
    CHECK(system->getModule<ModuleC>()->calls == frames);
    for (int id = frames - CARTSLAM_RUN_RETENTION + 1; id <= frames; ++id) {   // the retained frames: values as defined
        auto run = system->getRunById((uint32_t)id);
        CHECK(*run->getData<long>("a") == 10L * id);
        CHECK(*run->getData<long>("b") == 10L * id + 1);
        CHECK(*run->getData<long>("c") == 10L * id + 10L * (id - 1));
    }
    bool evicted = false;
    try { system->getRunById(1); } catch (const std::invalid_argument &) { evicted = true; }
    CHECK(evicted);
    CHECK(system->getThreadPool().threadCount() <= 12 * (1 + 2 * 3) + 8);   // at most every task of every frame in flight at once
    system.reset();
    if (failures == 0) std::printf("ok\n");
    return failures ? 1 : 0;
}
