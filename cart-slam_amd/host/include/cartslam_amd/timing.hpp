// timing.hpp -- per-module / per-frame wall-clock CSV with the reference's columns (include/timing.hpp:17-70:
// name;run_id;time_init;time_start;time_end;duration_ms). The reference compiles it in with ENABLE_TIMING and always
// writes timing/timing-<date>.csv; here it is switched on at run time by giving the sink a file name, and the duration
// is in microsecond resolution (duration_us column added) because a frame takes well under a millisecond on MI355X.
#pragma once
#include <chrono>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>

namespace cart::timing {
struct timing_handle_t {
    const std::string name;
    size_t runId;
    const std::chrono::time_point<std::chrono::high_resolution_clock> init;
    std::chrono::time_point<std::chrono::high_resolution_clock> start;
    std::chrono::time_point<std::chrono::high_resolution_clock> end;
    timing_handle_t(const std::string name, const size_t runId) : name(name), runId(runId), init(std::chrono::high_resolution_clock::now()) {}
};

class Sink {
   public:
    static Sink &instance() { static Sink s; return s; }
    void open(const std::string &path) {
        std::lock_guard<std::mutex> lock(mutex);
        file.open(path);
        if (file.is_open()) file << "name;run_id;time_init;time_start;time_end;duration_ms;duration_us\n";
    }
    bool enabled() const { return file.is_open(); }
    void write(const timing_handle_t &h) {
        using namespace std::chrono;
        auto ms = [](const time_point<high_resolution_clock> &t) { return duration_cast<milliseconds>(t.time_since_epoch()).count(); };
        std::lock_guard<std::mutex> lock(mutex);
        if (!file.is_open()) return;
        file << h.name << ';' << h.runId << ';' << ms(h.init) << ';' << ms(h.start) << ';' << ms(h.end) << ';'
             << duration_cast<milliseconds>(h.end - h.start).count() << ';' << duration_cast<microseconds>(h.end - h.start).count() << '\n';
    }

   private:
    std::ofstream file;
    std::mutex mutex;
};

inline std::shared_ptr<timing_handle_t> initTiming(const std::string &name, const size_t runId) { return std::make_shared<timing_handle_t>(name, runId); }
inline void startTiming(std::shared_ptr<timing_handle_t> h) { h->start = std::chrono::high_resolution_clock::now(); }
inline void endTiming(std::shared_ptr<timing_handle_t> h) {
    h->end = std::chrono::high_resolution_clock::now();
    Sink::instance().write(*h);
}
}  // namespace cart::timing
