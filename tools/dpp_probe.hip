// Probe: semantics of the f64 DPP row_newbcast forms used by the ALS block factorisation.
// hipcc -O3 --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int N>
__device__ __forceinline__ double fnma_bcast(double acc, double l, double a)
{   // acc - l[lane N of my 16-lane row] * a   (one rounding)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(l), "v"(a), "n"(N));
    return acc;
}
template <int N>
__device__ __forceinline__ double bcast(double l)
{
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(l), "n"(N));
    return r;
}
__global__ void k(const double *in, double *out)
{
    const int t = threadIdx.x;
    double a = in[t], l = in[t + 64], acc = in[t + 128];
    out[t] = fnma_bcast<3>(acc, l, a);
    out[t + 64] = bcast<5>(l);
    out[t + 128] = fnma_bcast<15>(acc, l, a);
}
int main()
{
    double h[192], o[192], *di, *d_o;
    for (int i = 0; i < 192; ++i) h[i] = sin(i * 0.37) + 1.5;
    hipMalloc(&di, sizeof h); hipMalloc(&d_o, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, d_o);
    hipMemcpy(o, d_o, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const int row = t & ~15;
        bad += o[t] != fma(-h[64 + row + 3], h[t], h[128 + t]);
        bad += o[64 + t] != h[64 + row + 5];
        bad += o[128 + t] != fma(-h[64 + row + 15], h[t], h[128 + t]);
    }
    printf("dpp probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
