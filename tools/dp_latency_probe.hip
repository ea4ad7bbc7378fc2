// Probe: latency of a dependent v_add_f64 chain (8 active lanes vs 64), and of the LDS-fed ordered sum
// used by the full-k kernel.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/dp_latency_probe.hip -o /tmp/dpl && /tmp/dpl
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(double *out, unsigned long long *ticks, int active)
{
    __shared__ __attribute__((aligned(16))) double tile[8 * 66];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8 * 66; i += 64) tile[i] = 1e-3 * i;
    __syncthreads();
    double s = out[lane];
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (lane < active) {
#pragma unroll
        for (int i = 0; i < 256; ++i) s += 1.0000001;            // register-only dependent chain
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double q = s;
    if (lane < 8) {                                               // the kernel's pattern: 64 values from LDS, 8 lanes
        const double *row = tile + lane * 66;
        for (int f = 0; f + 8 <= 64; f += 8) {
            const double2 a = *reinterpret_cast<const double2 *>(row + f), b = *reinterpret_cast<const double2 *>(row + f + 2);
            const double2 c = *reinterpret_cast<const double2 *>(row + f + 4), d = *reinterpret_cast<const double2 *>(row + f + 6);
            q += a.x; q += a.y; q += b.x; q += b.y; q += c.x; q += c.y; q += d.x; q += d.y;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    out[lane] = s + q;
    if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; }
}
int main()
{
    double *d; unsigned long long *t, h[2];
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&t, 16); (void)hipMemset(d, 0, 64 * 8);
    for (int active : {8, 64}) {
        k_chain<<<1, 64>>>(d, t, active); (void)hipDeviceSynchronize();
        k_chain<<<1, 64>>>(d, t, active); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("active lanes %2d: 256 dependent v_add_f64: %llu ticks = %.1f per add | LDS-fed ordered sum of 64: %llu ticks = %.1f per add\n",
               active, h[0], h[0] / 256.0, h[1], h[1] / 64.0);
    }
    return 0;
}
