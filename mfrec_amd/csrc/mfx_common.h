// Shared internals of libmfx.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mfx.h"

namespace mfx {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);   // set message + remember code, returns code
int last_code();                              // code of the last fail() / MFX_HIP failure
hipStream_t stream();          // library stream (created by mfx_init)
bool ready();                  // mfx_init succeeded
int ensure_init();             // lazily bind device 0 if mfx_init was never called
void set_last_compute_ms(double ms);

#define MFX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            return mfx::fail(MFX_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,   \
                             hipGetErrorString(e_));                                    \
        }                                                                               \
    } while (0)

struct Rec {        // one rating in level-major order: 16 B, one dwordx4 load
    int32_t user;
    int32_t item;
    double r;
};

// roctx range for rocprofv3 --marker-trace (SURVEY section 5 "tracing"): the profiler's roctx
// library is looked up lazily with dlopen; without it (or with MFX_ROCTX=0) ranges are no-ops.
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
    bool on;
};

template <typename T>
static inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

}  // namespace mfx
