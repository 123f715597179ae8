// csrc/kws_common.h -- shared host-side helpers for the C-ABI implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "kws.h"

namespace kws {

std::string &last_error_slot();

inline int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

#define KWS_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ::kws::fail(KWS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                                \
    } while (0)

// kernel launches do not return a status; fetch the sticky launch error instead
#define KWS_LAUNCH_CHECK(what)                                                                  \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess)                                                                  \
            return ::kws::fail(KWS_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e__)); \
    } while (0)

// 4 KB of zeros on the current device (allocated once per device): padding rows of the convolutions are READ from here
// instead of being masked after the load, so a fragment load has no consumer until its MFMA (the prefetch can overlap).
const float *zero_page();

// gradient exchange on streams the train step already uses (kws_comm.hip)
int comm_allreduce_early(kws_comm *c, float *buf, int64_t n, hipStream_t s);
int comm_allreduce_late(kws_comm *c, float *grads, int64_t n, float *state, int64_t n_state, float state_weight, hipStream_t s);

// library-wide defaults of the per-model precision attributes (kws_model_set_precision)
int default_matrix_precision();
int default_infer_precision();

// Opt-in per-launch timing (kws_prof_enable / kws_prof_report): HIP events recorded on the launch stream around each
// kernel.  Disabled (the default) it costs one relaxed load per launch site.
bool prof_on();
const char *prof_name(const char *base, int layer);   // interned "<base>.L<layer>" while profiling, else base
void prof_mark(const char *name, hipStream_t s, bool begin);
struct ProfScope {
    const char *name;
    hipStream_t s;
    bool on;
    ProfScope(const char *n, hipStream_t st) : name(n), s(st), on(prof_on()) { if (on) prof_mark(name, s, true); }
    ~ProfScope() { if (on) prof_mark(name, s, false); }
};

}  // namespace kws

#define KWS_LAUNCH(name, kernel, grid, block, smem, stream, ...)          \
    do {                                                                 \
        ::kws::ProfScope prof__(name, stream);                           \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__); \
    } while (0)
