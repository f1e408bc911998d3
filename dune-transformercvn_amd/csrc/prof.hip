#include <cstring>
#include "../../include/tcvn_hip.h"
#include "prof.h"

namespace tcvn {

Profiler& profiler() { static Profiler p; return p; }

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t s) : st(s) {
    Profiler& p = profiler();
    if (!p.enabled || (p.filter[0] && !strstr(name, p.filter))) return;
    ProfRec r;
    strncpy(r.name, name, sizeof(r.name) - 1); r.name[sizeof(r.name) - 1] = 0;
    r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    p.recs.push_back(r);
    idx = (int)p.recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(profiler().recs[idx].e1, st);
}

}  // namespace tcvn

using namespace tcvn;

extern "C" {
void tcvn_profile_enable(int on) { profiler().enabled = on != 0; }
void tcvn_profile_filter(const char* substr) {
    Profiler& p = profiler();
    p.filter[0] = 0;
    if (substr) { strncpy(p.filter, substr, sizeof(p.filter) - 1); p.filter[sizeof(p.filter) - 1] = 0; }
}
void tcvn_profile_reset(void) {
    for (auto& r : profiler().recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    profiler().recs.clear();
}
int tcvn_profile_count(void) { return (int)profiler().recs.size(); }
int tcvn_profile_get(int i, char* name, int cap, float* ms, double* flops, double* bytes) {
    if (i < 0 || i >= (int)profiler().recs.size()) return -1;
    ProfRec& r = profiler().recs[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) return -2;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -3;
    if (name && cap > 0) { strncpy(name, r.name, cap - 1); name[cap - 1] = 0; }
    *ms = t; *flops = r.flops; *bytes = r.bytes;
    return 0;
}
}
