// hgym_capi.hip -- library-level entry points (version, error string, device probe).
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "hgym_common.hpp"

namespace hgym {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int device_cus();

namespace {
struct ProfRec {
    hipEvent_t a, b;
    double work;
};
bool g_prof = false;
std::vector<ProfRec> g_rec[HGYM_PROF_CLASSES];
hipEvent_t g_open[HGYM_PROF_CLASSES];
}  // namespace

int32_t ensure_dynamic_lds(const void* fn, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> reserved;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) HG_FAIL(HGYM_E_NODEVICE, "%s: no current device", what);
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = reserved[std::make_pair(fn, dev)];
    if (bytes <= have) return HGYM_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "cannot reserve %zu bytes of LDS for %s on device %d", bytes, what, dev);
    }
    have = bytes;
    return HGYM_OK;
}

long long* g_phase_buf = nullptr;
int64_t g_phase_slots = 0;
// phase-timestamp buffer for a grid of `blocks` workgroups (8 slots each), or null when none is set / it is too small
long long* phase_buffer(int64_t blocks) { return blocks * 8 <= g_phase_slots ? g_phase_buf : nullptr; }
bool prof_on() { return g_prof; }
void prof_begin(int cls, hipStream_t s) {
    if (!g_prof) return;
    (void)hipEventCreate(&g_open[cls]);
    (void)hipEventRecord(g_open[cls], s);
}
void prof_end(int cls, hipStream_t s, double work) {
    if (!g_prof) return;
    ProfRec r;
    r.a = g_open[cls];
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.b, s);
    r.work = work;
    g_rec[cls].push_back(r);
}
}  // namespace hgym

extern "C" {
int32_t hgym_version(void) { return HGYM_VERSION; }
const char* hgym_last_error(void) { return hgym::last_error_buf(); }
int64_t hgym_sizeof(const char* name) {
    if (!name) return -1;
#define HG_SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T)
    HG_SZ(HgymEnvConfig); HG_SZ(HgymStrided); HG_SZ(HgymSimTensors); HG_SZ(HgymEnvState); HG_SZ(HgymEnvOut);
    HG_SZ(HgymEnvNoise); HG_SZ(HgymNetConfig); HG_SZ(HgymPPOConfig); HG_SZ(HgymNet); HG_SZ(HgymBatch); HG_SZ(HgymObsShadow); HG_SZ(HgymComm);
#undef HG_SZ
    return -1;
}
int32_t hgym_prof_enable(int32_t on) {
    for (int c = 0; c < HGYM_PROF_CLASSES; ++c) {
        for (auto& r : hgym::g_rec[c]) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        hgym::g_rec[c].clear();
    }
    hgym::g_prof = on != 0;
    return HGYM_OK;
}
int32_t hgym_prof_phase_buffer(void* dev, int64_t slots) {
    hgym::g_phase_buf = (long long*)dev;
    hgym::g_phase_slots = dev ? slots : 0;
    return HGYM_OK;
}
int32_t hgym_prof_summary(int32_t cls, int64_t* launches, double* total_ms, double* work) {
    using namespace hgym;
    HG_REQUIRE(cls >= 0 && cls < HGYM_PROF_CLASSES && launches && total_ms && work, HGYM_E_BADARG, "bad profile class / null output");
    *launches = 0;
    *total_ms = 0.0;
    *work = 0.0;
    for (auto& r : g_rec[cls]) {
        if (hipEventSynchronize(r.b) != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "event sync failed");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        *total_ms += ms;
        *work += r.work;
        *launches += 1;
    }
    return HGYM_OK;
}
int32_t hgym_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return hgym::device_cus();
}
}
