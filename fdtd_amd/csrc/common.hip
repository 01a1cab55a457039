// common.hip -- error slot, version string, kernel timer.
#include "common.hpp"

namespace fdtd {

std::string& last_error_slot() {
    static thread_local std::string s;
    return s;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

hipEvent_t KernelTimer::get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void KernelTimer::begin(int kind, hipStream_t s, int part) {
    if (!enabled) return;
    Rec r{get(), get(), kind, part};
    if (!r.a || !r.b) { enabled = false; return; }
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
}
void KernelTimer::end(hipStream_t s) {
    if (!enabled || recs.empty()) return;
    (void)hipEventRecord(recs.back().b, s);
}
int KernelTimer::collect(int kind, int* launches, double* total_ms) {
    int n = 0;
    double ms = 0;
    std::vector<Rec> keep;
    for (auto& r : recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        FDTD_HIP(hipEventSynchronize(r.b));
        float t = 0;
        FDTD_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; ++n;
        pool.push_back(r.a); pool.push_back(r.b);
    }
    recs.swap(keep);
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return FDTD_OK;
}
int KernelTimer::timeline(int cap, int* kinds, int* parts, double* start_ms, double* dur_ms, int* n_out) {
    if (n_out) *n_out = (int)recs.size();
    for (size_t q = 0; q < recs.size() && (int)q < cap; ++q) {
        FDTD_HIP(hipEventSynchronize(recs[q].b));
        float t0 = 0, t1 = 0;
        if (q > 0) FDTD_HIP(hipEventElapsedTime(&t0, recs[0].a, recs[q].a));
        FDTD_HIP(hipEventElapsedTime(&t1, recs[q].a, recs[q].b));
        if (kinds) kinds[q] = recs[q].kind;
        if (parts) parts[q] = recs[q].part;
        if (start_ms) start_ms[q] = t0;
        if (dur_ms) dur_ms[q] = t1;
    }
    return FDTD_OK;
}
void KernelTimer::release() {
    for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
    recs.clear(); pool.clear();
}

}  // namespace fdtd

extern "C" const char* fdtd_last_error(void) { return fdtd::last_error_slot().c_str(); }
extern "C" const char* fdtd_engine_version(void) { return "fdtd_engine gfx950 0.1 (wave64, 16-byte lanes, fp-contract off)"; }
extern "C" int fdtd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
