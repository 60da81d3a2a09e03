// sharder.hpp -- the batched-sequence mode (BASELINE.json configs[4]) below Python: ONE process drives the N GPUs of
// a node, frames of a sequence that starts on GPU 0 are dealt out frame k -> GPU k mod N, every GPU runs its share through
// the C ABI (disparity -> plane derivative + per-frame histogram), the histograms are all-gathered, every GPU replays the
// plane-parameter schedule of DisparityPlaneSegmentationModule::updatePlaneParameters (planeseg.cu:379-403) for the whole
// sequence and classifies its own frames, and disparity + planes come back to GPU 0 in sequence order.
// Transport: RCCL (ncclCommInitAll, one communicator per GPU).  Scatter and gather are grouped ncclSend / ncclRecv -- GPU 0
// has a direct xGMI link to every peer, so no ring is involved -- the histogram exchange is one ncclAllGather (1 KB per frame).
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
    FrameSharder(const std::vector<int> &devices, cart_engine_params params, int framesPerGpu, int updateInterval = 30, int resetInterval = 10);
    ~FrameSharder();
    FrameSharder(const FrameSharder &) = delete;
    FrameSharder &operator=(const FrameSharder &) = delete;

    int gpus() const { return (int)ranks.size(); }
    int capacity() const { return framesPerGpu * gpus(); }

    // left / right: gray frames [nFrames][h][w], tight, in the memory of devices[0]; nFrames must be a positive multiple of
    // gpus() and <= capacity().  Frame k gets id firstId + k (ids are 1-based and must continue from the previous call: the
    // schedule's histogram is cumulative).  disparity [nFrames][h][w] s16 and planes [nFrames][h][w] u8: tight, devices[0].
    // Blocks until the results are in place.  Throws std::runtime_error on any failure.
    void processSequence(const uint8_t *left, const uint8_t *right, int nFrames, int16_t *disparity, uint8_t *planes);

   private:
    struct Rank;
    std::vector<std::unique_ptr<Rank>> ranks;
    const int framesPerGpu, width, height;
    int nextId = 1;
};
}  // namespace cart
