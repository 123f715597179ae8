// csrc/kws_common.h -- shared host-side helpers for the C-ABI implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <tuple>
#include <utility>

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

// <<<>>> launches do not return a status: fetch the sticky launch error; hipExtLaunchKernel (profiled launches, launches that carry a
// fork event) does return one, which the launch helpers park in launch_error_slot()
#define KWS_LAUNCH_CHECK(what)                                                                  \
    do {                                                                                        \
        hipError_t e__ = ::kws::launch_error_slot();                                            \
        ::kws::launch_error_slot() = hipSuccess;                                                \
        if (e__ == hipSuccess) e__ = hipGetLastError(); else (void)hipGetLastError();           \
        if (e__ != hipSuccess)                                                                  \
            return ::kws::fail(KWS_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e__)); \
    } while (0)

hipError_t &launch_error_slot();      // per host thread: first failed hipExtLaunchKernel since the last KWS_LAUNCH_CHECK
inline void note_launch_error(hipError_t e) { if (e != hipSuccess && launch_error_slot() == hipSuccess) launch_error_slot() = e; }
// compute units of the CURRENT device, and "this kernel may use `bytes` of dynamic LDS on the current device" -- both cached per device
// (a process may drive several devices: kws_model keeps per-device resources)
int device_cus();
int ensure_dynamic_lds(const void *kernel, int bytes);

// 4 KB of zeros on the current device (allocated once per device): padding rows of the convolutions are READ from here
// instead of being masked after the load, so a fragment load has no consumer until its MFMA (the prefetch can overlap).
const float *zero_page();

// gradient exchange on streams the train step already uses (kws_comm.hip)
int comm_allreduce_early(kws_comm *c, float *buf, int64_t n, hipStream_t s);
int comm_allreduce_late(kws_comm *c, float *grads, int64_t n, float *state, int64_t n_state, float state_weight, hipStream_t s);

// library-wide defaults of the per-model precision attributes (kws_model_set_precision)
int default_matrix_precision();
int default_infer_precision();

// Opt-in per-launch timing (kws_prof_enable / kws_prof_report).  A profiled launch goes through hipExtLaunchKernel with a start and a stop
// event bound to the dispatch itself, so the reported time is the kernel's own begin -> end (what rocprofv3 --kernel-trace reports), without
// the barrier packets event records around a launch would add between concurrent streams.  Disabled (the default) it costs one relaxed load
// per launch site.
bool prof_on();
const char *prof_name(const char *base, int layer);   // interned "<base>.L<layer>" while profiling, else base
void prof_events(const char *name, hipEvent_t *e0, hipEvent_t *e1);   // a fresh pair, remembered under `name`

// Fork events without a marker packet.  hipEventRecord on the caller's stream puts a barrier packet of its own between two kernels of the
// main chain, and every such packet cost the chain 6-8 us (rocprofv3 timeline: five forks / joins per train step).  A kernel launched
// through hipExtLaunchKernel can carry a stop event that is bound to the dispatch's own completion signal instead: arm_stop_event(ev, s)
// makes the NEXT kernel launched on s carry ev; fork code then only lets the other stream wait for ev (stop_event_bound tells whether the
// launch took it; if no kernel followed, the caller records the event the plain way).  One armed event per host thread.
struct ArmedEvent { hipEvent_t ev = nullptr; hipStream_t s = nullptr; hipEvent_t bound = nullptr; };
ArmedEvent &armed_event();
// A stream that is being CAPTURED into a hipGraph gets no armed events: an event bound to a dispatch's completion signal is not a graph node,
// so the waits on it would fall out of the capture (the side stream's kernels then ran once, eagerly, and every replay lacked them); the
// callers fall back to hipEventRecord, which the capture turns into an edge.
inline bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}
inline void arm_stop_event(hipEvent_t ev, hipStream_t s)
{
    if (stream_is_capturing(s)) return;
    ArmedEvent &a = armed_event(); a.ev = ev; a.s = s; a.bound = nullptr;
}
// Scope guard of every function that arms events: whatever way the function is left (an early KWS_TRY / KWS_HIP_CHECK return between
// arm_stop_event and the launch that should have carried the event), no armed event survives into an unrelated launch of this thread.
struct DisarmOnExit { ~DisarmOnExit() { ArmedEvent &a = armed_event(); a.ev = nullptr; a.s = nullptr; a.bound = nullptr; } };
inline bool stop_event_bound(hipEvent_t ev)      // true: the kernel launched since arm_stop_event carries ev (consumes the answer)
{
    ArmedEvent &a = armed_event();
    const bool yes = a.bound == ev && ev != nullptr;
    a.ev = nullptr; a.bound = nullptr;
    return yes;
}

template <size_t I, typename Tuple>
inline void set_launch_args(Tuple &) {}
template <size_t I, typename Tuple, typename A, typename... Rest>
inline void set_launch_args(Tuple &t, A &&a, Rest &&...rest)
{
    std::get<I>(t) = static_cast<std::tuple_element_t<I, Tuple>>(std::forward<A>(a));    // the conversion a <<<>>> call would apply
    set_launch_args<I + 1>(t, std::forward<Rest>(rest)...);
}
template <typename... KArgs, typename... Args, size_t... I>
inline void launch_timed_impl(const char *name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t s,
                              std::index_sequence<I...>, Args &&...args)
{
    // RULE: trailing parameters the call leaves out take a ZERO value (value-initialised tuple), not the kernel's C++ default argument:
    // every default argument of this library's kernels must be nullptr / 0 / false / an empty plane struct
    std::tuple<std::remove_cv_t<KArgs>...> vals{};
    set_launch_args<0>(vals, std::forward<Args>(args)...);
    void *ptrs[sizeof...(KArgs) + 1] = {static_cast<void *>(&std::get<I>(vals))..., nullptr};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    prof_events(name, &e0, &e1);
    note_launch_error(hipExtLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, smem, s, e0, e1, 0));
}
// the same launch with a stop event and no timing pair (fork events, see ArmedEvent)
template <typename... KArgs, typename... Args, size_t... I>
inline void launch_stop_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t s, hipEvent_t stop,
                             std::index_sequence<I...>, Args &&...args)
{
    std::tuple<std::remove_cv_t<KArgs>...> vals{};
    set_launch_args<0>(vals, std::forward<Args>(args)...);
    void *ptrs[sizeof...(KArgs) + 1] = {static_cast<void *>(&std::get<I>(vals))..., nullptr};
    note_launch_error(hipExtLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, smem, s, nullptr, stop, 0));
}
template <typename... KArgs, typename... Args>
inline void launch_stop(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t s, hipEvent_t stop, Args &&...args)
{
    static_assert(sizeof...(Args) <= sizeof...(KArgs), "argument count of a kernel launch");
    launch_stop_impl(kernel, grid, block, smem, s, stop, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}
template <typename... KArgs, typename... Args>
inline void launch_timed(const char *name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t s, Args &&...args)
{
    static_assert(sizeof...(Args) <= sizeof...(KArgs), "argument count of a kernel launch");
    launch_timed_impl(name, kernel, grid, block, smem, s, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}

}  // namespace kws

#define KWS_LAUNCH(name, kernel, grid, block, smem, stream, ...)                                              \
    do {                                                                                                     \
        ::kws::ArmedEvent &ae__ = ::kws::armed_event();                                                      \
        if (::kws::prof_on()) ::kws::launch_timed(name, kernel, dim3(grid), dim3(block), smem, stream, __VA_ARGS__); \
        else if (ae__.ev && ae__.s == (stream)) {                                                            \
            ::kws::launch_stop(kernel, dim3(grid), dim3(block), smem, stream, ae__.ev, __VA_ARGS__);         \
            ae__.bound = ae__.ev;                                                                            \
            ae__.ev = nullptr;                                                                               \
        } else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                           \
    } while (0)
