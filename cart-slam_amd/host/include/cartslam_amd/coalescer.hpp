// coalescer.hpp -- hands the frames that wait inside one module at the same moment to the engine as one call.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace cart {
// The reference's runtime enters runInternal from up to CARTSLAM_CONCURRENT_RUN_LIMIT worker threads at once, one frame
// each (cartslam.cpp:196).  One frame per launch sequence leaves path aggregation latency-bound on this GPU (0.67 ms per
// frame against 0.10 ms per frame in a 16-frame launch) and a one-frame plane kernel costs what a 16-frame one does, so
// the frames that are waiting inside a module at the same moment are handed to the engine as ONE *_multi call.
// Leader/follower, no extra thread and no timer: a caller whose request is still queued becomes the leader as soon as
// fewer than `maxOutstanding` groups are on the GPU, takes every compatible queued request, runs the group on its own
// stream and wakes the others.  A lone frame is dispatched at once (no added latency at low load); groups only form
// while the GPU is already busy with earlier ones.
struct CoalescedRequest {
    bool queued = true, done = false;
    std::string error;
};

template <class Request>
class FrameCoalescer {
   public:
    using Compatible = std::function<bool(const Request &, const Request &)>;
    using RunGroup = std::function<void(const std::vector<Request *> &)>;   // enqueue + wait; throws on failure
    FrameCoalescer(int maxGroup, int maxOutstanding, Compatible compatible, RunGroup runGroup, int minAhead = 1)
        : maxGroup(maxGroup), maxOutstanding(maxOutstanding), minAhead(minAhead), compatible(std::move(compatible)), runGroup(std::move(runGroup)) {}

    void run(Request &rq) {
        std::unique_lock<std::mutex> lock(mutex);
        pending.push_back(&rq);
        while (!rq.done) {
            // a group is dispatched at once while the GPU has none of this module's; a further one (queued behind it, so that
            // the GPU does not idle through the host round trip between two groups) only when `minAhead` requests have gathered
            if (!rq.queued || outstanding >= maxOutstanding || (outstanding > 0 && (int)pending.size() < minAhead)) { cv.wait(lock); continue; }
            // leader: this request + every queued one with the same image layout, oldest first
            std::vector<Request *> group{&rq}, rest;
            for (Request *q : pending) {
                if (q == &rq) continue;
                ((int)group.size() < maxGroup && compatible(*q, rq) ? group : rest).push_back(q);
            }
            pending.swap(rest);
            for (Request *q : group) q->queued = false;
            ++outstanding;
            lock.unlock();
            std::string error;
            try { runGroup(group); } catch (const std::exception &e) { error = e.what(); if (error.empty()) error = "failed"; }
            lock.lock();
            --outstanding;
            ++groups; frames += group.size();
            for (Request *q : group) { q->error = error; q->done = true; }
            cv.notify_all();
        }
        if (!rq.error.empty()) throw std::runtime_error(rq.error);
    }
    // mean frames per launch sequence so far (diagnostics)
    double meanGroup() { std::lock_guard<std::mutex> lock(mutex); return groups ? (double)frames / groups : 0.0; }

   private:
    const int maxGroup, maxOutstanding, minAhead;
    const Compatible compatible;
    const RunGroup runGroup;
    std::mutex mutex;
    std::condition_variable cv;
    std::vector<Request *> pending;
    int outstanding = 0;
    size_t groups = 0, frames = 0;
};

}  // namespace cart
