// csrc/kws_core.hip -- version / error plumbing of the C ABI.
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "kws_common.h"
#include "kws_build_id.h"

namespace kws {
std::string &last_error_slot()
{
    static thread_local std::string s;
    return s;
}

namespace {
struct ProfRec { const char *name; hipEvent_t e0, e1; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

const float *zero_page()
{
    static std::mutex mu;
    static void *pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!pages[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, 4096) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
        pages[dev] = p;
    }
    return static_cast<const float *>(pages[dev]);
}

bool prof_on() { return g_prof_on.load(std::memory_order_relaxed); }

const char *prof_name(const char *base, int layer)
{
    if (!prof_on()) return base;
    static std::map<std::string, std::string> interned;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const std::string key = std::string(base) + ".L" + std::to_string(layer);
    return interned.emplace(key, key).first->second.c_str();
}

ArmedEvent &armed_event()
{
    static thread_local ArmedEvent a;
    return a;
}

hipError_t &launch_error_slot()
{
    static thread_local hipError_t e = hipSuccess;
    return e;
}

int device_cus()
{
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lk(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t p;
    int n = 256;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    else (void)hipGetLastError();
    cus[dev] = n;
    return n;
}

int ensure_dynamic_lds(const void *kernel, int bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> done;
    int dev = 0;
    KWS_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int &have = done[std::make_pair(dev, kernel)];
    if (have >= bytes) return KWS_OK;
    KWS_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return KWS_OK;
}

void prof_events(const char *name, hipEvent_t *e0, hipEvent_t *e1)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, take_event(), take_event()};
    g_prof.push_back(r);
    *e0 = r.e0;
    *e1 = r.e1;
}
}  // namespace kws

extern "C" {

int kws_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(kws::g_prof_mu);
    for (auto &r : kws::g_prof) { kws::g_pool.push_back(r.e0); kws::g_pool.push_back(r.e1); }
    kws::g_prof.clear();
    kws::g_prof_on.store(on != 0);
    return KWS_OK;
}

int64_t kws_prof_report(char *buf, size_t buflen)
{
    std::lock_guard<std::mutex> lk(kws::g_prof_mu);
    std::map<std::string, std::pair<long, double>> agg;
    for (auto &r : kws::g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto &a = agg[r.name];
            a.first += 1;
            a.second += ms;
        }
    }
    (void)hipGetLastError();
    std::string out = "{";
    bool first = true;
    for (auto &kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s\"%s\": {\"count\": %ld, \"total_ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
                 kv.second.first, kv.second.second);
        out += line;
        first = false;
    }
    out += "}";
    if (buf && buflen > 0) {
        const size_t n = out.size() < buflen - 1 ? out.size() : buflen - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

const char *kws_version(void) { return "kws-amd 0.1.0 (gfx950)"; }

const char *kws_build_id(void) { return KWS_BUILD_ID; }

const char *kws_last_error(void) { return kws::last_error_slot().c_str(); }

int kws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
