// Optional per-launch HIP-event profiler (bench.py's live roofline measurement).  When enabled, every
// instrumented launch is bracketed by two hipEvents recorded on the launch stream; durations are summed per kernel
// tag when the report is read.  Disabled (the default) it costs one branch per launch.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace dtts {

struct ProfStat {
    std::string name;
    long long launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};

class Profiler {
public:
    static Profiler& get() {
        static Profiler p;
        return p;
    }
    bool on = false;
    void begin(const char* tag, double flops, double bytes, hipStream_t s) {
        if (!on) return;
        hipEvent_t a = take(), b = take();
        recs_.push_back({tag, flops, bytes, a, b});
        (void)hipEventRecord(a, s);
    }
    void end(hipStream_t s) {
        if (!on) return;
        (void)hipEventRecord(recs_.back().stop, s);
    }
    // synchronises, folds the pending records into the per-tag totals and recycles the events
    void collect() {
        for (auto& r : recs_) {
            (void)hipEventSynchronize(r.stop);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.start, r.stop);
            ProfStat& st = stats_[r.tag];
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
    }
    std::vector<ProfStat> report() {
        collect();
        std::vector<ProfStat> v;
        for (auto& kv : stats_) v.push_back(kv.second);
        return v;
    }

private:
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
    std::vector<Rec> recs_;
    std::vector<hipEvent_t> pool_;
    std::map<std::string, ProfStat> stats_;
};

struct ProfScope {
    hipStream_t s;
    ProfScope(const char* tag, double flops, double bytes, hipStream_t st) : s(st) { Profiler::get().begin(tag, flops, bytes, st); }
    ~ProfScope() { Profiler::get().end(s); }
};

}  // namespace dtts
