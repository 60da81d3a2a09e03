// sharder.hpp -- the batched-sequence mode (BASELINE.json configs[4]) below Python: ONE process drives the N GPUs of
// a node, frames of a sequence that starts on GPU 0 are dealt out frame k -> GPU k mod N, every GPU runs its share through
// the C ABI (disparity -> plane derivative + per-frame histogram), the histograms are all-gathered, every GPU replays the
// plane-parameter schedule of DisparityPlaneSegmentationModule::updatePlaneParameters (planeseg.cu:379-403) for the whole
// sequence and classifies its own frames, and disparity + planes come back to GPU 0 in sequence order.
// Transport: RCCL (ncclCommInitAll, ONE communicator per GPU).  Scatter and gather are grouped ncclSend / ncclRecv,
// ONE per peer and image kind (GPU 0 packs a peer's frames into a staging area first) -- GPU 0 has a direct xGMI link to every
// peer, so no ring is involved -- the histogram exchange is one ncclAllGather (1 KB per frame).
// Sequences are double-buffered: every GPU has a copy stream beside its compute stream; EVERY RCCL operation is enqueued on the
// copy stream (so a GPU never has collectives of two communicators, or of one communicator on two streams, in flight), ordered
// against the kernels by events: submit(i+1) posts the scatter of sequence i+1, then the all-gather, classification and gather of
// sequence i, then the disparity kernels of i+1 -- the transfers run beside the kernels (order table in sharder.cpp).
// The reference has no multi-GPU mode (include/cartslam.hpp:4 is frame pipelining on one GPU): this is new functionality.
// The same sharding over torch.distributed lives in cartslam/pipeline.py and must give identical results.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "cart_engine.h"

namespace cart {

class FrameSharder {
   public:
    // devices: HIP device ids, devices[0] holds the sequence and receives the results.  params.max_inflight is overwritten.
    // framesPerGpu: the largest share of one sequence a GPU may hold (capacity() = framesPerGpu * gpus()).
    // placementTries > 1 opts in to cart_engine_tune_placement on every GPU's engine (set-up time and, transiently, up to two units
    // of slab memory per GPU beyond the workspace: include/cart_engine.h); the default keeps the engines' first allocations.
    FrameSharder(const std::vector<int> &devices, cart_engine_params params, int framesPerGpu, int updateInterval = 30, int resetInterval = 10,
                 int placementTries = 1);
    ~FrameSharder();
    FrameSharder(const FrameSharder &) = delete;
    FrameSharder &operator=(const FrameSharder &) = delete;

    int gpus() const { return (int)ranks.size(); }
    int capacity() const { return framesPerGpu * gpus(); }
    // frames of an nFrames-sequence that land on GPU `rank`: the first nFrames % gpus() GPUs hold one frame more
    static int shareOf(int nFrames, int rank, int gpus) { return nFrames > rank ? (nFrames - rank + gpus - 1) / gpus : 0; }

    // Enqueues one sequence and returns at once.  left / right: gray frames [nFrames][h][w], tight, in the memory of
    // devices[0], complete on `callerStream` (a hipStream_t of devices[0]; nullptr = its null stream) when the call is made:
    // the sharder's own streams are non-blocking, so it records an event there and waits for it.  1 <= nFrames <= capacity(),
    // any remainder modulo gpus().  Frame k gets id firstId + k (ids are 1-based and continue from the previous call: the
    // schedule's histogram is cumulative).  disparity [nFrames][h][w] s16 and planes [nFrames][h][w] u8: tight, devices[0];
    // they are complete after the wait() that follows (at most two sequences are in flight: a third submit first waits for
    // the oldest).  The input and output buffers of a sequence must stay untouched until then.
    // Throws std::runtime_error on any failure.
    void submit(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes, void *callerStream = nullptr);
    // Posts the gather of the sequence submitted last and blocks until every submitted sequence's results are in place.
    // A GPU whose work does not finish within timeoutSeconds makes this throw, naming that GPU and its RCCL error state.
    void wait(double timeoutSeconds = 120.0);
    // submit + wait
    void processSequence(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes);
    // What has been enqueued so far: sequences finished (all-gather + classification + gather posted), grouped RCCL operations (three per
    // sequence: scatter, all-gather, gather), hipStreamWaitEvent calls (per sequence 1 + 4 per GPU, + 1 per GPU from the third sequence on).
    // Pipelined and one-at-a-time use issue the same operations in a different interleaving; the counts must agree (cart_shard_amd checks).
    struct Counters { long long sequences = 0, collectiveGroups = 0, streamWaits = 0; };
    Counters counters() const { return counters_; }

   private:
    struct Rank;
    struct Pending { int nFrames = 0, firstId = 1; int16_t *disparity = nullptr; uint8_t *planes = nullptr; int buf = 0; bool live = false; };
    void finish(const Pending &p);
    void streamWait(void *stream, void *event);
    Counters counters_;
    std::vector<std::unique_ptr<Rank>> ranks;
    const int framesPerGpu, width, height;
    int nextId = 1;
    long long submitted = 0;
    Pending pending;   // disparity kernels enqueued; all-gather, classification and gather not yet posted
};
}  // namespace cart
