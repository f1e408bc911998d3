// Optional per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg; off by default).
#pragma once
#include <vector>
#include "tcvn_common.h"

namespace tcvn {

struct ProfRec { char name[96]; hipEvent_t e0, e1; double flops, bytes; };
struct Profiler {
    bool enabled = false;
    char filter[96] = {0};             // when set: only scopes whose label contains it are recorded
    std::vector<ProfRec> recs;
};
Profiler& profiler();

// RAII: records an event pair around the launches issued inside its scope
struct ProfScope {
    int idx = -1; hipStream_t st;
    ProfScope(const char* name, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

}  // namespace tcvn
