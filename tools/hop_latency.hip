// Floor of one hand-off hop on this part: two waves ping-pong an 8-byte word N times; placements (same CU / same XCD, other CU /
// other XCD) x cache-control bits of the load and of the store.  A combination whose poller never sees the update (a stale line in
// the CU's L1) is reported as "never" after a bounded number of polls.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/hop_latency tools/hop_latency.hip && tools/_bin/hop_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }
__device__ __forceinline__ unsigned hw_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }

template <int L> __device__ __forceinline__ unsigned long long ld(const unsigned long long *p)
{
    unsigned long long v;
    if constexpr (L == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (L == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int S> __device__ __forceinline__ void st(unsigned long long *p, unsigned long long v)
{
    if constexpr (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if constexpr (S == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

struct Ctl {
    unsigned registered, pad[15];
    unsigned xcc[1024], hw[1024];
    long long result[64];            // [placement][L][S] : 100 MHz ticks for N round trips, -1 = never seen
    unsigned go[64];                 // start handshake per experiment (device-scope atomics)
};

constexpr int N = 2000;
constexpr long long kMaxPolls = 100000;

template <int L, int S>
__device__ long long pingpong(unsigned long long *word, bool pinger, unsigned long long base)
{
    // lane 0 only
    long long t0 = wall_clock64();
    for (int i = 0; i < N; ++i) {
        const unsigned long long want = base + 2 * i + (pinger ? 2 : 1);
        if (pinger) st<S>(word, base + 2 * i + 1);
        long long polls = 0;
        while (ld<L>(word) != want)
            if (++polls > kMaxPolls) return -1;
        if (!pinger) st<S>(word, base + 2 * i + 2);
    }
    return wall_clock64() - t0;
}

__global__ __launch_bounds__(64) void k_probe(Ctl *c, unsigned long long *words, int nwg)
{
    const int b = blockIdx.x;
    __shared__ int partner[3];
    if (threadIdx.x == 0) {
        c->xcc[b] = xcc_id();
        c->hw[b] = hw_id();
        __threadfence();
        atomicAdd(&c->registered, 1u);
        while (__hip_atomic_load(&c->registered, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) __builtin_amdgcn_s_sleep(8);
        // partners of workgroup 0: same CU / same XCD other CU / other XCD (first match in block order)
        const unsigned x0 = __hip_atomic_load(&c->xcc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned h0 = __hip_atomic_load(&c->hw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xff00u;   // cu, sh, se
        int p[3] = {-1, -1, -1};
        for (int i = 1; i < nwg; ++i) {
            const unsigned x = __hip_atomic_load(&c->xcc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned h = __hip_atomic_load(&c->hw[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xff00u;
            const int cls = (x == x0 && h == h0) ? 0 : (x == x0 ? 1 : 2);
            if (p[cls] < 0) p[cls] = i;
        }
        partner[0] = p[0]; partner[1] = p[1]; partner[2] = p[2];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int role = -1;                                   // which placement this workgroup plays the ponger in
    for (int q = 0; q < 3; ++q) if (partner[q] == b) role = q;
    if (b != 0 && role < 0) return;
    unsigned long long base = 0;
    auto run = [&](auto Lc, auto Sc, int q, int slot) {
        constexpr int L = decltype(Lc)::value, S = decltype(Sc)::value;
        if (partner[q] < 0) { if (b == 0) c->result[slot] = -2; return; }
        unsigned long long *w = words + 64 * slot;              // a line of its own per experiment
        if (b == 0) {
            __hip_atomic_store(&c->go[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&c->go[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 2u) __builtin_amdgcn_s_sleep(2);
            c->result[slot] = pingpong<L, S>(w, true, base);
        } else if (role == q) {
            while (__hip_atomic_load(&c->go[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store(&c->go[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            (void)pingpong<L, S>(w, false, base);
        }
    };
#define RUN(L, S) { for (int q = 0; q < 3; ++q) run(std::integral_constant<int, L>{}, std::integral_constant<int, S>{}, q, (q * 4 + L) * 4 + S); }
    // loads that may hit the CU's L1 (L = 0, 1) can only be paired with a partner on the same CU: the others would spin to the bound
    RUN(2, 0) RUN(2, 1) RUN(2, 2) RUN(2, 3) RUN(3, 0) RUN(3, 2) RUN(3, 3) RUN(1, 0) RUN(1, 1) RUN(1, 2) RUN(0, 0) RUN(0, 2)
}

int main(int argc, char **argv)
{
    const int nwg = 512;
    Ctl *c; unsigned long long *words;
    CK(hipMalloc((void **)&c, sizeof(Ctl)));
    CK(hipMemset(c, 0, sizeof(Ctl)));
    CK(hipMalloc((void **)&words, 64 * 64 * 8));
    CK(hipMemset(words, 0, 64 * 64 * 8));
    hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(64), 0, 0, c, words, nwg);
    CK(hipDeviceSynchronize());
    std::vector<char> hb(sizeof(Ctl));
    CK(hipMemcpy(hb.data(), c, sizeof(Ctl), hipMemcpyDeviceToHost));
    const Ctl *h = (const Ctl *)hb.data();
    const char *pl[3] = {"same CU", "same XCD, other CU", "other XCD"};
    const char *bits[4] = {"plain", "sc0", "sc1", "sc0 sc1"};
    printf("wg0: xcc %u hw 0x%x\n", h->xcc[0], h->hw[0]);
    for (int q = 0; q < 3; ++q)
        for (int L = 0; L < 4; ++L)
            for (int S = 0; S < 4; ++S) {
                const long long t = h->result[(q * 4 + L) * 4 + S];
                if (t == 0) continue;
                if (t == -2) printf("%-20s load %-8s store %-8s : no partner\n", pl[q], bits[L], bits[S]);
                else if (t < 0) printf("%-20s load %-8s store %-8s : never seen\n", pl[q], bits[L], bits[S]);
                else printf("%-20s load %-8s store %-8s : %7.3f us per round trip (two hops)\n", pl[q], bits[L], bits[S], t * 0.01 / N);
            }
    // the floor bench.py prices a chain of hand-offs with: the fastest ONE hop between two CUs (placements 1, 2) over the
    // combinations that were seen at all; argv[1] = where to write it (profiles/rNN_hop_latency.json)
    double best[3] = {1e9, 1e9, 1e9};
    for (int q = 0; q < 3; ++q)
        for (int L = 0; L < 4; ++L)
            for (int S = 0; S < 4; ++S) {
                const long long t = h->result[(q * 4 + L) * 4 + S];
                if (t > 0 && t * 0.01 / N / 2 < best[q]) best[q] = t * 0.01 / N / 2;
            }
    const double min_hop = best[1] < best[2] ? best[1] : best[2];
    char js[512];
    snprintf(js, sizeof js, "{\"min_hop_us\": %.4f, \"same_cu_hop_us\": %.4f, \"same_xcd_other_cu_hop_us\": %.4f, \"other_xcd_hop_us\": %.4f, "
             "\"what\": \"half of the fastest 8-byte ping-pong round trip between two waves on DIFFERENT CUs, idle chip "
             "(tools/hop_latency.hip); min over the cache-control combinations whose poll saw the store\", \"round_trips\": %d}",
             min_hop, best[0], best[1], best[2], N);
    printf("%s\n", js);
    if (argc > 1) { FILE *f = fopen(argv[1], "w"); if (f) { fprintf(f, "%s\n", js); fclose(f); } }
    return 0;
}
