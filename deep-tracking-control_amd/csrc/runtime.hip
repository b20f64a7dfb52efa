// Error reporting + per-launch HIP-event profiler of libdtc_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace {
thread_local char g_err[512] = "";
std::mutex g_prof_mu;
bool g_prof_on = false;
struct Rec {
    const char* name;
    double work, bytes;
    hipEvent_t a, b;
};
std::vector<Rec> g_recs;
}  // namespace

namespace dtc {

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DTC_ERR_LAUNCH;
    }
    return DTC_OK;
}

const char* prof_shape_name(const char* base, int M, int N, int K) {
    // interned "base[MxNxK]" names when DTC_PROF_SHAPES is set (per-layer breakdown of the GEMM classes)
    static const bool on = getenv("DTC_PROF_SHAPES") != nullptr;
    if (!on || !g_prof_on) return base;
    static std::map<std::string, std::string> pool;
    char buf[96];
    snprintf(buf, sizeof(buf), "%s[%dx%dx%d]", base, M, N, K);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = pool.emplace(buf, buf).first;
    return it->second.c_str();
}

ProfScope::ProfScope(const char* name, double work, hipStream_t s, double bytes) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    Rec r{name, work, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].b, stream);
}

}  // namespace dtc

extern "C" {

int dtc_version(void) { return DTC_ABI_VERSION; }
static int g_gemm_split = 0;
void dtc_set_gemm_split(int on) { g_gemm_split = on ? 1 : 0; }
int dtc_get_gemm_split(void) { return g_gemm_split; }
int dtc_abi_sizes(int64_t* out, int cap) {
    const int64_t sz[] = {sizeof(DtcGridCfg), sizeof(DtcObsCfg), sizeof(DtcRowCopy), sizeof(DtcSeg), sizeof(DtcSegMat),
                          sizeof(DtcFwdLayer), sizeof(DtcWgradJob), sizeof(DtcPpoCfg), sizeof(DtcProfRec), sizeof(DtcWimgJob),
                          sizeof(DtcH2iWJob), sizeof(DtcH2iOperand), sizeof(DtcWgradH2iJob), sizeof(DtcEnvStep), sizeof(DtcH2iFwdLayer),
                          sizeof(DtcH2iDgradLayer), sizeof(DtcGruFwdItem), sizeof(DtcGruBwdItem)};
    const int n = (int)(sizeof(sz) / sizeof(sz[0]));
    for (int i = 0; i < n && i < cap && out; ++i) out[i] = sz[i];
    return n;
}
const char* dtc_last_error(void) { return g_err; }

// A HIP stream of the library's own (non-blocking; high_priority: the device's greatest priority).  The trainers' compute lanes are such streams:
// torch.cuda.Stream() hands out the entries of a 32-stream pool per priority that every other component of the process draws from as well
// (torch.distributed's gloo / NCCL work streams among them), so a "second stream" taken from there can be the very stream a collective runs on.
int dtc_stream_create(int high_priority, void** out) {
    DTC_REQUIRE(out, "null pointer");
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    hipStream_t s = nullptr;
    // (not `least`: that is the LOW priority class (numerically 1 on this stack), below torch's default streams (0))
    const int normal = (0 <= least && 0 >= greatest) ? 0 : least;
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? greatest : normal);
    if (e != hipSuccess) {
        dtc::set_error("hipStreamCreateWithPriority: %s", hipGetErrorString(e));
        return DTC_ERR_LAUNCH;
    }
    *out = (void*)s;
    return DTC_OK;
}
int dtc_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return DTC_ERR_LAUNCH;
    return DTC_OK;
}

void dtc_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

void dtc_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
}

int dtc_prof_report(DtcProfRec* out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, DtcProfRec> agg;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        DtcProfRec& d = agg[r.name];
        if (d.launches == 0) {
            memset(&d, 0, sizeof(d));
            strncpy(d.name, r.name, sizeof(d.name) - 1);
        }
        d.ms_total += ms;
        d.work += r.work;
        d.bytes += r.bytes;
        d.launches += 1;
    }
    int n = 0;
    for (auto& kv : agg) {
        if (n < cap && out) out[n] = kv.second;
        ++n;
    }
    return n;
}

}  // extern "C"
