// mos_api.hip — error plumbing, version and the optional kernel profiler of libmos_hip.so (include/mos_hip.h).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "mos_common.h"

namespace {
thread_local char g_err[512] = "ok";

struct ProfRec { std::string name; double flops, bytes; hipEvent_t e0, e1; };
struct ProfAgg { std::string name; double ms = 0, flops = 0, bytes = 0; long long calls = 0; };
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<ProfAgg> g_prof_aggs;
}  // namespace

int mos_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mos_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mos_set_error(MOS_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return MOS_OK;
}

MosProfScope::MosProfScope(hipStream_t s, const char* kernel, const char* key, double flops, double bytes)
    : slot(-1), st(s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    ProfRec r;
    r.name = std::string(kernel) + " " + key;
    r.flops = flops;
    r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(r);
    slot = (int)g_prof_recs.size() - 1;
}

MosProfScope::~MosProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof_recs[slot].e1, st);
}

extern "C" {
int mos_version(void) { return 100; }  // 0.1.0
const char* mos_last_error_string(void) { return g_err; }

int mos_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof_recs.clear();
    g_prof_aggs.clear();
    g_prof_on.store(1);
    return MOS_OK;
}

int mos_profile_end(void) {
    g_prof_on.store(0);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, ProfAgg> agg;
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ProfAgg& a = agg[r.name];
            a.name = r.name; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.calls += 1;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof_recs.clear();
    g_prof_aggs.clear();
    for (auto& kv : agg) g_prof_aggs.push_back(kv.second);
    return (int)g_prof_aggs.size();
}

int mos_profile_get(int idx, char* name, int name_cap, double* total_ms, long long* calls, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < 0 || idx >= (int)g_prof_aggs.size() || !name || name_cap <= 0)
        return mos_set_error(MOS_ERR_BAD_ARG, "mos_profile_get: index %d out of range", idx);
    const ProfAgg& a = g_prof_aggs[idx];
    snprintf(name, name_cap, "%s", a.name.c_str());
    if (total_ms) *total_ms = a.ms;
    if (calls) *calls = a.calls;
    if (flops) *flops = a.flops;
    if (bytes) *bytes = a.bytes;
    return MOS_OK;
}
}
