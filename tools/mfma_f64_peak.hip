// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (4 independent accumulator chains per wave,
// 4 or 8 waves per CU x 256 CUs x many iterations) -> achieved TFLOP/s and cycles per instruction.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters, unsigned long long *ticks)
{
    double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = threadIdx.x * 1e-3, y = 1.0 + blockIdx.x * 1e-6;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
int main()
{
    double *d; unsigned long long *dt, ht;
    (void)hipMalloc(&d, 4096 * 256 * 8); (void)hipMalloc(&dt, 8);
    const int cfg[][2] = {{1, 200}, {1, 2000}, {1, 20000}, {2, 20000}, {4, 20000}, {8, 10000}, {1, 200000}};
    for (auto &c : cfg) {
        const int wg_per_cu = c[0];
        const int blocks = 256 * wg_per_cu, iters = c[1];
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<blocks, 256>>>(d, 100, dt); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<<<blocks, 256>>>(d, iters, dt); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        const double n_mfma = (double)blocks * 4 * iters * 4;          // waves x instr
        const double tflops = n_mfma * 2048 / (ms * 1e-3) / 1e12;
        printf("waves/SIMD=%d iters=%d: %.3f ms, %.1f TFLOP/s f64, s_memtime ticks per MFMA per wave slot = %.1f (tick rate %.2f GHz)\n",
               wg_per_cu, iters, ms, tflops, (double)ht / (iters * 4.0 * wg_per_cu), ht / (ms * 1e6));
    }
    return 0;
}
