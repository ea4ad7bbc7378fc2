// Runtime plumbing of libmfx.so: error string, device binding, stream, event stopwatch.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include "mfx_common.h"

namespace {
char g_err[1024] = "";
int g_code = 0;
double g_last_ms = 0.0;
hipStream_t g_stream = nullptr;
bool g_ready = false;
int g_device = -1;
hipEvent_t g_t0 = nullptr, g_t1 = nullptr;
}  // namespace

namespace mfx {
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_code = code;
    return code;
}
int last_code() { return g_code; }
void set_last_compute_ms(double ms) { g_last_ms = ms; }
hipStream_t stream() { return g_stream; }
bool ready() { return g_ready; }
int ensure_init() { return g_ready ? MFX_OK : mfx_init(0); }

namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_push = nullptr;
roctx_pop_t g_pop = nullptr;
int g_roctx_state = 0;          // 0 = not looked up, 1 = available, -1 = absent / disabled
bool roctx_ready()
{
    if (g_roctx_state == 0) {
        g_roctx_state = -1;
        const char *e = getenv("MFX_ROCTX");
        if (!(e && e[0] == '0')) {
            const char *names[] = {"librocprofiler-sdk-roctx.so", "libroctx64.so"};
            for (const char *n : names) {
                void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (!h) continue;
                g_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
                g_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
                if (g_push && g_pop) {
                    g_roctx_state = 1;
                    break;
                }
            }
        }
    }
    return g_roctx_state == 1;
}
}  // namespace

TraceRange::TraceRange(const char *name) : on(roctx_ready())
{
    if (on) (void)g_push(name);
}
TraceRange::~TraceRange()
{
    if (on) (void)g_pop();
}
}  // namespace mfx

extern "C" {

int mfx_version(void) { return 100; }
double mfx_last_compute_ms(void) { return g_last_ms; }
const char *mfx_last_error(void) { return g_err; }

int mfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mfx_init(int device)
{
    if (g_ready && device == g_device) return MFX_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        return mfx::fail(MFX_E_HIP,
                         "mfx_init: no HIP device available (%s); libmfx has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) {
        return mfx::fail(MFX_E_ARG, "mfx_init: device %d outside [0,%d)", device, n);
    }
    MFX_HIP(hipSetDevice(device));
    if (g_stream) {
        (void)hipStreamDestroy(g_stream);
        g_stream = nullptr;
    }
    MFX_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (!g_t0) MFX_HIP(hipEventCreate(&g_t0));
    if (!g_t1) MFX_HIP(hipEventCreate(&g_t1));
    g_device = device;
    g_ready = true;
    return MFX_OK;
}

int mfx_sync(void)
{
    if (!g_ready) return MFX_OK;
    MFX_HIP(hipStreamSynchronize(g_stream));
    return MFX_OK;
}

int mfx_device_name(char *buf, int buflen)
{
    if (!buf || buflen <= 0) return MFX_E_ARG;
    int rc = mfx::ensure_init();
    if (rc) return rc;
    hipDeviceProp_t p;
    MFX_HIP(hipGetDeviceProperties(&p, g_device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MFX_OK;
}

int mfx_device_mem_info(int64_t *free_bytes, int64_t *total_bytes)
{
    if (mfx::ensure_init()) return mfx::last_code();
    size_t f = 0, t = 0;
    MFX_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return MFX_OK;
}

int mfx_timer_start(void)
{
    int rc = mfx::ensure_init();
    if (rc) return rc;
    MFX_HIP(hipEventRecord(g_t0, g_stream));
    return MFX_OK;
}

int mfx_timer_stop(double *elapsed_ms)
{
    if (!g_ready || !elapsed_ms) return MFX_E_ARG;
    MFX_HIP(hipEventRecord(g_t1, g_stream));
    MFX_HIP(hipEventSynchronize(g_t1));
    float ms = 0.f;
    MFX_HIP(hipEventElapsedTime(&ms, g_t0, g_t1));
    *elapsed_ms = (double)ms;
    return MFX_OK;
}

}  // extern "C"
