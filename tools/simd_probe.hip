// Probe (round 5): (1) which SIMD the waves of a 256-thread workgroup land on (HW_ID), under the ALS row
// kernel's footprint (39 KB LDS, 3 workgroups per CU); (2) what a DEPENDENT chain of vector instructions
// costs per instruction when other waves of the same SIMD issue v_mfma_f64_16x16x4_f64 back to back -- the
// question behind the ALS row kernel's serial phases (block factorisation, panel solve).
//   hipcc -O3 --offload-arch=gfx950 tools/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hw_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ---- (1) placement ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void k_place(unsigned *out, int spin, unsigned *xcc, unsigned long long *start)
{
    extern __shared__ double lds[];
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x * 4 + wid] = hw_id();
        if (wid == 0) xcc[blockIdx.x] = x & 15u;
        if (wid == 0) start[blockIdx.x] = now();
    }
    // stay resident for a while, so that later workgroups see a partly filled chip as in the real kernel
    const unsigned long long t0 = now();
    while (now() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 999) lds[0] = 1.0;
}

// ---- (2) interference --------------------------------------------------------------------------------
// wave w of a 512-thread workgroup: role = (roles >> (4 * w)) & 15: 0 idle, 1 f64 MFMA stream, 2..6 a chain
enum { IDLE = 0, MFMA = 1, CH_FMA64 = 2, CH_FMA32 = 3, CH_DPP64 = 4, CH_RSQ64 = 5, CH_ADD32I = 6 };
constexpr int CHAIN_N = 1024;
__global__ __launch_bounds__(512) void k_interfere(unsigned long long roles, int mfma_iters, unsigned long long *out, double *sink)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = (int)((roles >> (4 * wid)) & 15);
    __shared__ int go;
    if (threadIdx.x == 0) go = 0;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    double res = 0.0;
    if (role == MFMA) {
        double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double x = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-4;
        t0 = now();
        for (int i = 0; i < mfma_iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        t1 = now();
        res = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (role >= CH_FMA64) {
        // let the matrix waves get going first
        const unsigned long long ts = now();
        while (now() - ts < 20000ull) { }
        double v = 1.0 + lane * 1e-9;
        float f = 1.0f + lane * 1e-6f;
        int n = lane;
        const double m = 0.999999, c = 1e-7;
        t0 = now();
        if (role == CH_FMA64) {
#pragma unroll 64
            for (int i = 0; i < CHAIN_N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
        } else if (role == CH_FMA32) {
            const float mf = 0.99999f, cf = 1e-6f;
#pragma unroll 64
            for (int i = 0; i < CHAIN_N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(mf), "v"(cf));
        } else if (role == CH_DPP64) {
            double acc = v;
#pragma unroll 64
            for (int i = 0; i < CHAIN_N; ++i)
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c));
            v = acc;
        } else if (role == CH_RSQ64) {
#pragma unroll 64
            for (int i = 0; i < CHAIN_N; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(v));
        } else {
#pragma unroll 64
            for (int i = 0; i < CHAIN_N; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n) : "v"(lane));
        }
        t1 = now();
        res = v + f + n;
    }
    if (lane == 0) {
        unsigned long long *o = out + ((size_t)blockIdx.x * 8 + wid) * 2;
        o[0] = t1 - t0;
        o[1] = ((unsigned long long)role << 32) | hw_id();
    }
    if (res == 123.456) sink[threadIdx.x] = res;
}


// ---- (3) issue throughput of independent instruction streams (one wave alone on its SIMD) -------------------------
enum { TP_FMA64 = 0, TP_DPP64 = 1, TP_DPP64_NOP = 2, TP_MOVDPP64 = 3, TP_READLANE = 4, TP_RSQ64 = 5, TP_MUL64 = 6 };
template <int kind>
__global__ __launch_bounds__(64) void k_tput(unsigned long long *out, double *sink)
{
    double a[8];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + (lane + i) * 1e-9;
    const double m = 0.999999, c = 1e-7;
    int acc = 0;
    const unsigned long long t0 = now();
    for (int it = 0; it < 128; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (kind == TP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if constexpr (kind == TP_DPP64) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
            else if constexpr (kind == TP_DPP64_NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
            else if constexpr (kind == TP_MOVDPP64) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(m));
            else if constexpr (kind == TP_READLANE) { int r; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r) : "v"(lane)); acc += r; }
            else if constexpr (kind == TP_RSQ64) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(m));
            else asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    const unsigned long long t1 = now();
    double r = acc;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    if (lane == 0) out[0] = t1 - t0;
    if (r == 123.456) sink[lane] = r;
}

static int simd_of(unsigned h) { return (h >> 4) & 3; }
static int cu_of(unsigned h) { return (h >> 8) & 15; }
static int se_of(unsigned h) { return (h >> 13) & 7; }

int main()
{
    // ---- (1) ----
    {
        const int nb = 256 * 3 * 8;
        unsigned *d;
        (void)hipMalloc(&d, nb * 4 * 4);
        (void)hipFuncSetAttribute((const void *)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
        unsigned *dx; unsigned long long *dst;
        (void)hipMalloc(&dx, nb * 4); (void)hipMalloc(&dst, nb * 8);
        k_place<<<nb, 256, 39 * 1024>>>(d, 20000, dx, dst);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> hx(nb); std::vector<unsigned long long> hst(nb);
        (void)hipMemcpy(hx.data(), dx, nb * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hst.data(), dst, nb * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned> h(nb * 4);
        (void)hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
        int same[4] = {0, 0, 0, 0}, distinct = 0, rot = 0;
        std::map<int, int> pat;
        for (int b = 0; b < nb; ++b) {
            int s[4], mask = 0;
            for (int w = 0; w < 4; ++w) { s[w] = simd_of(h[b * 4 + w]); mask |= 1 << s[w]; if (s[w] == w) ++same[w]; }
            if (mask == 15) ++distinct;
            if (((s[1] - s[0]) & 3) == 1 && ((s[2] - s[0]) & 3) == 2 && ((s[3] - s[0]) & 3) == 3) ++rot;
            ++pat[s[0] | s[1] << 2 | s[2] << 4 | s[3] << 6];
        }
        printf("[placement] %d workgroups of 4 waves (39 KB LDS): wave w on SIMD w: %.3f %.3f %.3f %.3f; all four SIMDs distinct: %.3f; "
               "a rotation of 0123: %.3f\n", nb, same[0] / (double)nb, same[1] / (double)nb, same[2] / (double)nb, same[3] / (double)nb,
               distinct / (double)nb, rot / (double)nb);
        {   // TG_ID (HW_ID[19:16]) of the workgroups that share a CU AT THE SAME TIME (the first wave: 3 per CU): distinct? mod 4?
            std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned>>> by_cu;
            int tg[16] = {0};
            for (int b = 0; b < nb; ++b) {
                const unsigned hw = h[b * 4];
                by_cu[(hx[b] << 16) | (hw & 0x7f00u)].push_back({hst[b], (hw >> 16) & 15});
                ++tg[(hw >> 16) & 15];
            }
            printf("[placement] TG_ID histogram:");
            for (int i = 0; i < 16; ++i) if (tg[i]) printf("  %d:%d", i, tg[i]);
            printf("\n[placement] %zu distinct (xcc, se, sh, cu) keys; first three workgroups of each CU by start time, TG_IDs:", by_cu.size());
            int shown = 0, coll = 0, n3 = 0;
            for (auto &kv : by_cu) {
                auto v = kv.second;
                std::sort(v.begin(), v.end());
                if (v.size() >= 3) {
                    ++n3;
                    if ((v[0].second & 3) == (v[1].second & 3) || (v[0].second & 3) == (v[2].second & 3) || (v[1].second & 3) == (v[2].second & 3)) ++coll;
                    if (shown++ < 6) printf("  [%u %u %u]", v[0].second, v[1].second, v[2].second);
                }
            }
            printf("  ... CUs whose first three TG_IDs collide mod 4: %d of %d\n", coll, n3);
        }
        int slot[16] = {0};
        for (int i = 0; i < nb * 4; ++i) ++slot[h[i] & 15];
        printf("[placement] hardware wave-slot ids (HW_ID[3:0]) over all waves:");
        for (int i = 0; i < 16; ++i) if (slot[i]) printf("  %d:%d", i, slot[i]);
        printf("\n");
        printf("[placement] patterns (SIMD of wave 0,1,2,3: count):");
        for (auto &p : pat) if (p.second > nb / 100) printf("  %d%d%d%d:%d", p.first & 3, (p.first >> 2) & 3, (p.first >> 4) & 3, (p.first >> 6) & 3, p.second);
        printf("\n");
        (void)hipFree(d);
    }
    // ---- (2) ----
    {
        const int nb = 256;
        unsigned long long *d;
        double *sink;
        (void)hipMalloc(&d, nb * 8 * 2 * 8);
        (void)hipMalloc(&sink, 512 * 8);
        const char *names[] = {"idle", "mfma", "v_fma_f64", "v_fma_f32", "v_fmac_f64_dpp(+s_nop 1)", "v_rsq_f64", "v_add_u32"};
        for (int chain = CH_FMA64; chain <= CH_ADD32I; ++chain) {
            for (int nm = 0; nm <= 7; nm += (nm < 4 ? 1 : 3)) {     // number of MFMA waves in the workgroup: 0,1,2,3,4,7
                unsigned long long roles = (unsigned long long)chain;            // wave 0 = chain
                for (int w = 1; w <= nm; ++w) roles |= (unsigned long long)MFMA << (4 * w);
                k_interfere<<<nb, 512>>>(roles, 1500, d, sink);
                (void)hipDeviceSynchronize();
                std::vector<unsigned long long> h(nb * 16);
                (void)hipMemcpy(h.data(), d, nb * 16 * 8, hipMemcpyDeviceToHost);
                // group chain waves by the number of MFMA waves that share their SIMD
                double sum[9] = {0}; int cnt[9] = {0};
                double msum = 0; int mcnt = 0;
                for (int b = 0; b < nb; ++b) {
                    const int cs = simd_of((unsigned)h[(b * 8 + 0) * 2 + 1]);
                    int share = 0;
                    for (int w = 1; w <= nm; ++w) {
                        if (simd_of((unsigned)h[(b * 8 + w) * 2 + 1]) == cs) ++share;
                        msum += (double)h[(b * 8 + w) * 2] / (1500.0 * 4); ++mcnt;
                    }
                    sum[share] += (double)h[(b * 8) * 2] / CHAIN_N; ++cnt[share];
                }
                printf("[interfere] chain %-26s + %d mfma waves in the workgroup:", names[chain], nm);
                for (int s = 0; s < 9; ++s) if (cnt[s]) printf("  shares SIMD with %d: %.1f cyc/instr (%d wgs)", s, sum[s] / cnt[s], cnt[s]);
                if (mcnt) printf("  | mfma waves: %.1f cyc per MFMA per wave", msum / mcnt);
                printf("\n");
            }
        }
    }
    {
        unsigned long long *d, h;
        double *sink;
        (void)hipMalloc(&d, 8);
        (void)hipMalloc(&sink, 64 * 8);
        const char *nm[] = {"v_fma_f64", "v_fmac_f64_dpp", "s_nop 1 + v_fmac_f64_dpp", "v_mov_b64_dpp", "v_readlane_b32", "v_rsq_f64", "v_mul_f64"};
        auto run = [&](auto K) {
            constexpr int kind = decltype(K)::value;
            k_tput<kind><<<1, 64>>>(d, sink);
            (void)hipDeviceSynchronize();
            k_tput<kind><<<1, 64>>>(d, sink);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("[throughput] %-26s 8 independent streams: %.1f cyc per instruction\n", nm[kind], (double)h / (128 * 8));
        };
        run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{}); run(std::integral_constant<int, 2>{});
        run(std::integral_constant<int, 3>{}); run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 5>{});
        run(std::integral_constant<int, 6>{});
    }
    return 0;
}
