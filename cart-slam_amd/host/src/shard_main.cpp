// cart_shard_amd <left.bin> <right.bin> <width> <height> <frames> <num_disparities> <paths> <gpus> <frames per call> <out dir> [update_interval reset_interval]
// Batched-sequence mode of the C++ host (cartslam_amd/sharder.hpp): the gray frames of <left.bin>/<right.bin> (u8, tight) are
// uploaded to GPU 0, dealt out over <gpus> GPUs with RCCL <frames per call> at a time (any remainder: the first GPUs hold one
// frame more, the last call is shorter), and disparity.bin (s16) / planes.bin (u8) are written to <out dir>.
// Two passes over the sequence, each with a fresh sharder (frame ids start over): pass 0 one call at a time (submit + wait),
// pass 1 pipelined (every call submitted, one wait at the end: the scatter of call i+1 and the gather of call i-1 run beside
// the kernels of call i).  The two passes must agree byte for byte (exit 2 otherwise); the files hold the pipelined pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "cartslam_amd/sharder.hpp"

int main(int argc, char **argv) {
    if (argc < 11) {
        std::cerr << "usage: cart_shard_amd <left.bin> <right.bin> <width> <height> <frames> <num_disparities> <paths> <gpus> <frames per call> <out dir> [update_interval reset_interval]\n";
        return 1;
    }
    try {
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), n = std::atoi(argv[5]), D = std::atoi(argv[6]), P = std::atoi(argv[7]);
        const int gpus = std::atoi(argv[8]), perCall = std::atoi(argv[9]);
        const std::string out = argv[10];
        const int ui = argc > 11 ? std::atoi(argv[11]) : 30, ri = argc > 12 ? std::atoi(argv[12]) : 10;
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have < gpus) throw std::runtime_error("this box has " + std::to_string(have) + " GPUs, " + std::to_string(gpus) + " asked for");
        if (n < 1 || perCall < 1) throw std::runtime_error("frames and frames per call must be positive");
        const size_t npx = (size_t)w * h;
        std::vector<uint8_t> hl(npx * n), hr(npx * n);
        for (auto &pr : {std::make_pair(argv[1], &hl), std::make_pair(argv[2], &hr)}) {
            std::ifstream f(pr.first, std::ios::binary);
            if (!f.read(reinterpret_cast<char *>(pr.second->data()), (std::streamsize)pr.second->size())) throw std::runtime_error(std::string("cannot read ") + pr.first);
        }
        cart_engine_params p;
        cart_engine_default_params(&p);
        p.width = w; p.height = h; p.num_disparities = D; p.paths = P; p.smoothing_radius = 2; p.smoothing_iterations = 1;
        std::vector<int> devices;
        for (int i = 0; i < gpus; ++i) devices.push_back(i);
        std::vector<int16_t> hd(npx * n), hd0;
        std::vector<uint8_t> hp(npx * n), hp0;
        double pairsPerSecond[2] = {0, 0};
        cart::FrameSharder::Counters counted[2];
        for (int pass = 0; pass < 2; ++pass) {   // fresh sharder per pass: frame ids (and the cumulative histogram) start over
            cart::FrameSharder sharder(devices, p, (perCall + gpus - 1) / gpus, ui, ri);
            if (hipSetDevice(0) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
            uint8_t *dl, *dr, *dp; int16_t *dd;
            if (hipMalloc((void **)&dl, npx * n) || hipMalloc((void **)&dr, npx * n) || hipMalloc((void **)&dp, npx * n) || hipMalloc((void **)&dd, npx * n * 2))
                throw std::runtime_error("hipMalloc failed");
            (void)hipMemcpy(dl, hl.data(), npx * n, hipMemcpyHostToDevice);
            (void)hipMemcpy(dr, hr.data(), npx * n, hipMemcpyHostToDevice);
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int f0 = 0; f0 < n; f0 += perCall) {
                const int cnt = std::min(perCall, n - f0);
                if (pass == 0) sharder.processSequence(dl + f0 * npx, dr + f0 * npx, cnt, dd + f0 * npx, dp + f0 * npx);
                else sharder.submit(dl + f0 * npx, dr + f0 * npx, cnt, dd + f0 * npx, dp + f0 * npx);
            }
            sharder.wait();
            counted[pass] = sharder.counters();
            pairsPerSecond[pass] = n / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            (void)hipSetDevice(0);
            (void)hipMemcpy(hd.data(), dd, npx * n * 2, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hp.data(), dp, npx * n, hipMemcpyDeviceToHost);
            (void)hipFree(dl); (void)hipFree(dr); (void)hipFree(dp); (void)hipFree(dd);
            if (pass == 0) { hd0 = hd; hp0 = hp; }
        }
        if (hd0 != hd || hp0 != hp) {
            std::cerr << "fatal: the pipelined pass differs from the one-call-at-a-time pass\n";
            return 2;
        }
        // the two passes enqueue the same operations in a different interleaving: three grouped RCCL operations per sequence, all on one
        // communicator and one stream per GPU, and the same number of stream waits
        if (counted[0].sequences != counted[1].sequences || counted[0].collectiveGroups != counted[1].collectiveGroups || counted[0].streamWaits != counted[1].streamWaits ||
            counted[1].collectiveGroups != 3 * counted[1].sequences) {
            std::cerr << "fatal: the pipelined pass enqueued other operations than the one-call-at-a-time pass (" << counted[1].collectiveGroups << " / "
                      << counted[0].collectiveGroups << " RCCL groups, " << counted[1].streamWaits << " / " << counted[0].streamWaits << " stream waits)\n";
            return 2;
        }
        std::ofstream(out + "/disparity.bin", std::ios::binary).write(reinterpret_cast<const char *>(hd.data()), (std::streamsize)(npx * n * 2));
        std::ofstream(out + "/planes.bin", std::ios::binary).write(reinterpret_cast<const char *>(hp.data()), (std::streamsize)(npx * n));
        std::cout << "frames " << n << " gpus " << gpus << " frames_per_call " << perCall << " pairs_per_s " << pairsPerSecond[1]
                  << " pairs_per_s_one_call_at_a_time " << pairsPerSecond[0] << " sequences " << counted[1].sequences << " rccl_groups " << counted[1].collectiveGroups
                  << " stream_waits " << counted[1].streamWaits << " communicators_per_gpu 1\n";
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "fatal: " << e.what() << "\n";
        return 1;
    }
}
