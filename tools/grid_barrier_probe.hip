// Probe: latency of a device-wide (all-XCD) barrier built from agent-scope atomics + fences, for
// deciding whether a persistent one-launch-per-epoch kernel could beat one launch per level.
// hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ bool grid_barrier(unsigned *cnt, unsigned *gen, unsigned nblocks, unsigned &g)
{
    bool ok = true;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                       // release: write back dirty L2 lines
        if (atomicAdd(cnt, 1u) == nblocks - 1) {
            atomicExch(cnt, 0u);
            __threadfence();
            atomicAdd(gen, 1u);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { ok = false; break; }   // never hang the box
            }
        }
        __threadfence();                                       // acquire
    }
    ++g;
    __syncthreads();
    return ok;
}
__global__ void k(unsigned *ctl, int iters, double *data, int touch)
{
    unsigned g = 0;
    for (int i = 0; i < iters; ++i) {
        if (touch) data[(size_t)blockIdx.x * blockDim.x + threadIdx.x + (size_t)(i & 7) * 65536] += 1.0;   // some dirty lines
        if (!grid_barrier(ctl, ctl + 32, gridDim.x, g)) { if (threadIdx.x == 0) atomicExch(ctl + 64, 1u); return; }
    }
}
int main()
{
    unsigned *ctl; double *data;
    (void)hipMalloc(&ctl, 4096); (void)hipMalloc(&data, 8 * 65536 * 8 * 8);
    (void)hipMemset(data, 0, 8 * 65536 * 8 * 8);
    const int cfgs[][3] = {{256, 64, 0}, {256, 256, 0}, {256, 64, 1}, {512, 64, 0}, {32, 64, 0}, {8, 64, 0}};
    for (auto &c : cfgs) {
        (void)hipMemset(ctl, 0, 4096);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        k<<<c[0], c[1]>>>(ctl, 10, data, c[2]); (void)hipDeviceSynchronize();
        (void)hipMemset(ctl, 0, 4096);
        (void)hipEventRecord(e0); k<<<c[0], c[1]>>>(ctl, iters, data, c[2]); (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned flag; (void)hipMemcpy(&flag, ctl + 64, 4, hipMemcpyDeviceToHost);
        printf("blocks=%d threads=%d touch=%d: %.2f us per barrier%s\n", c[0], c[1], c[2], ms * 1e3 / iters, flag ? "  (TIMED OUT)" : "");
    }
    return 0;
}
