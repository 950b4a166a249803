// Optional per-launch HIP-event profiler (bench.py's live roofline measurement).  When enabled, every
// instrumented launch is bracketed by two hipEvents recorded on the launch stream; durations are summed per kernel
// tag when the report is read.  Disabled (the default) it costs one branch per launch.
#pragma once
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace dtts {

struct ProfStat {
    std::string name;
    long long launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
    double union_ms = 0.0;     // length of the union of the launch intervals (== ms when launches never overlap)
    std::vector<std::pair<float, float>> spans;
};

class Profiler {
public:
    static Profiler& get() {
        static Profiler p;
        return p;
    }
    bool on = false;
    bool all = false;        // also bracket the bandwidth-only helper kernels (flops == 0); off in bench.py: fewer events
    // false inside the sampling steps dtts_profile_sampling skips.  Per THREAD: only the thread issuing dtts_diff_sample is gated, the
    // stage-A thread of infer_stream keeps bracketing its launches ("launches outside dtts_diff_sample are always bracketed").
    static bool& gate() {
        static thread_local bool g = true;
        return g;
    }
    int step_every = 1;
    // events are created here, outside any timed region
    void reserve(size_t n) {
        while (pool_.size() < n) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pool_.push_back(e);
        }
    }
    // begin / collect take the lock: a handle may be driven from two host threads (include/detail_hip.h, "Threads"), and two handles
    // from any.  begin returns the scope's stop event (nullptr: not recorded); each ProfScope lives on one thread.
    hipEvent_t begin(const char* tag, double flops, double bytes, hipStream_t s) {
        if (!on || !gate() || !(all || flops > 0.0)) return nullptr;
        std::lock_guard<std::mutex> lk(mu_);
        if (!base_) {
            (void)hipEventCreate(&base_);
            (void)hipEventRecord(base_, s);
        }
        hipEvent_t a = take(), b = take();
        recs_.push_back({tag, flops, bytes, a, b});
        (void)hipEventRecord(a, s);
        return b;
    }
    void end(hipEvent_t stop, hipStream_t s) {
        if (stop) (void)hipEventRecord(stop, s);
    }
    // synchronises, folds the pending records into the per-tag totals and recycles the events
    void collect() {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto& r : recs_) {
            (void)hipEventSynchronize(r.stop);
            float ms = 0.f, t0 = 0.f;
            (void)hipEventElapsedTime(&ms, r.start, r.stop);
            (void)hipEventElapsedTime(&t0, base_, r.start);
            ProfStat& st = stats_[r.tag];
            st.spans.push_back({t0, t0 + ms});
            st.name = r.tag;
            st.launches += 1;
            st.ms += ms;
            st.flops += r.flops;
            st.bytes += r.bytes;
            pool_.push_back(r.start);
            pool_.push_back(r.stop);
        }
        recs_.clear();
    }
    void reset() {
        collect();
        stats_.clear();
        if (base_) (void)hipEventDestroy(base_);
        base_ = nullptr;
    }
    std::vector<ProfStat> report() {
        collect();
        std::vector<ProfStat> v;
        for (auto& kv : stats_) {
            ProfStat st = kv.second;
            std::sort(st.spans.begin(), st.spans.end());
            double u = 0.0;
            float cur0 = 0.f, cur1 = -1.f;
            for (auto& sp : st.spans) {
                if (cur1 < cur0 || sp.first > cur1) {
                    if (cur1 >= cur0) u += cur1 - cur0;
                    cur0 = sp.first;
                    cur1 = sp.second;
                } else if (sp.second > cur1) cur1 = sp.second;
            }
            if (cur1 >= cur0) u += cur1 - cur0;
            st.union_ms = u;
            st.spans.clear();
            v.push_back(st);
        }
        return v;
    }

private:
    std::mutex mu_;
    struct Rec {
        const char* tag;
        double flops, bytes;
        hipEvent_t start, stop;
    };
    hipEvent_t take() {
        if (!pool_.empty()) {
            hipEvent_t e = pool_.back();
            pool_.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    hipEvent_t base_ = nullptr;
    std::vector<Rec> recs_;
    std::vector<hipEvent_t> pool_;
    std::map<std::string, ProfStat> stats_;
};

struct ProfScope {
    hipStream_t s;
    hipEvent_t stop;
    ProfScope(const char* tag, double flops, double bytes, hipStream_t st) : s(st), stop(Profiler::get().begin(tag, flops, bytes, st)) {}
    ~ProfScope() { Profiler::get().end(stop, s); }
};

}  // namespace dtts
