// What does a kernel boundary cost inside a replayed hipGraph?  Chains of N dependent kernel nodes:
//   empty (1 workgroup), empty (256 x 512 threads), and one that touches memory (a 16-byte load + store per thread).
// Prints microseconds per node.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o build/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(double2 *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double2 v = p[i]; v.x += 1.0; p[i] = v; }
}

static int run(const char *name, int blocks, int threads, bool touch, double2 *buf, int n)
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 2000;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
        if (touch) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(threads), 0, st, buf, n);
        else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, st);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.3f us per node\n", name, ms * 1e3 / (5.0 * N));
    return 0;
}

int main()
{
    const int n = 256 * 512;
    double2 *buf;
    CK(hipMalloc(&buf, (size_t)n * sizeof(double2)));
    CK(hipMemset(buf, 0, (size_t)n * sizeof(double2)));
    if (run("empty kernel, 1 x 64", 1, 64, false, buf, n)) return 1;
    if (run("empty kernel, 256 x 512", 256, 512, false, buf, n)) return 1;
    if (run("empty kernel, 2048 x 512", 2048, 512, false, buf, n)) return 1;
    if (run("16-byte load + store per thread, 256 x 512", 256, 512, true, buf, n)) return 1;
    return 0;
}
