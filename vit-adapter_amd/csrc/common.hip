// Error slot + launch timing for libvitadapter_hip.so.
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace vah {

static thread_local char g_err[512] = {0};

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void clear_error() { g_err[0] = 0; }

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

int allow_dynamic_lds(const void *kernel, int bytes, const char *what) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail((int)e, "%s: cannot allow %d bytes of dynamic LDS: %s", what, bytes, hipGetErrorString(e));
    return 0;
}

// ---- launch timing -----------------------------------------------------------------
struct Rec {
    const char *name;
    int64_t bytes, def_bytes, flops;
    hipEvent_t start, stop;
};

static std::mutex g_mu;
static bool g_on = false;
static char g_prefix[128] = "";                 // only scopes whose name starts with one of these are timed
static std::vector<Rec> g_recs;                 // live records of the current collection
static std::vector<hipEvent_t> g_pool;          // recycled events

// g_prefix is a comma-separated list of name prefixes ("" = everything)
static bool name_selected(const char *name) {
    const char *p = g_prefix;
    if (!*p) return true;
    while (*p) {
        const char *e = strchr(p, ',');
        const size_t n = e ? (size_t)(e - p) : strlen(p);
        if (n > 0 && strncmp(name, p, n) == 0) return true;
        if (!e) break;
        p = e + 1;
    }
    return false;
}

static hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static void recycle_all() {
    for (auto &r : g_recs) {
        g_pool.push_back(r.start);
        g_pool.push_back(r.stop);
    }
    g_recs.clear();
}

LaunchScope::LaunchScope(const char *name, int64_t bytes, hipStream_t stream, int64_t def_bytes, int64_t flops)
    : slot_(-1), stream_(stream) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || !name_selected(name)) return;
    Rec r{name, bytes, def_bytes ? def_bytes : bytes, flops, take_event(), take_event()};
    if (!r.start || !r.stop) return;
    (void)hipEventRecord(r.start, stream);
    g_recs.push_back(r);
    slot_ = (int)g_recs.size() - 1;
}

LaunchScope::~LaunchScope() {
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot_ < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot_].stop, stream_);
}

}  // namespace vah

extern "C" {

int vah_abi_version(void) { return 36; }

const char *vah_last_error(void) { return vah::g_err; }

int vah_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(vah::g_mu);
    if (on) vah::recycle_all();
    vah::g_on = on != 0;
    return VAH_OK;
}

int vah_prof_filter(const char *prefix) {
    std::lock_guard<std::mutex> lk(vah::g_mu);
    snprintf(vah::g_prefix, sizeof(vah::g_prefix), "%s", prefix ? prefix : "");
    return VAH_OK;
}

int64_t vah_prof_report(char *buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(vah::g_mu);
    struct Agg {
        int64_t calls = 0;
        double ms = 0;
        int64_t bytes = 0, def_bytes = 0, flops = 0;
    };
    std::map<std::string, Agg> agg;
    for (auto &r : vah::g_recs) {
        if (hipEventSynchronize(r.stop) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
        Agg &a = agg[r.name];
        a.calls += 1;
        a.ms += ms;
        a.bytes += r.bytes;
        a.def_bytes += r.def_bytes;
        a.flops += r.flops;
    }
    std::string out;
    char line[256];
    for (auto &kv : agg) {
        snprintf(line, sizeof(line), "%s %lld %.6f %lld %lld %lld\n", kv.first.c_str(),
                 (long long)kv.second.calls, kv.second.ms, (long long)kv.second.bytes,
                 (long long)kv.second.def_bytes, (long long)kv.second.flops);
        out += line;
    }
    if (buf && cap > 0) {
        int64_t n = (int64_t)out.size() < cap - 1 ? (int64_t)out.size() : cap - 1;
        memcpy(buf, out.data(), (size_t)n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

}  // extern "C"
