// Diagnostic (never shipped): where does one level of the full-k kernel spend its time?
// Builds k1_level with MFX_K1_STAMPS and prints, over all waves of one launch, the median
// s_memtime deltas between: start | record loaded | rows loaded | sums+bias done | row
// stores issued | stores drained.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#define MFX_K1_STAMPS 1
#include <algorithm>
#include <random>
#include <vector>
#include "../mfrec_amd/csrc/mfx_sgd_kernels.h"
namespace mfx { void set_error(const char *, ...) {} int fail(int c, const char *, ...) { return c; } }
using namespace mfx;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BATCH, int NT, int KP = 64, int FPL = 1, bool VEC = false>
int run(int count, int nu, int ni, int uncached = 0)
{
    const int k = 64;
    std::mt19937_64 rng(1);
    std::vector<int> pu(nu), pi(ni);
    for (int i = 0; i < nu; ++i) pu[i] = i;
    for (int i = 0; i < ni; ++i) pi[i] = i;
    std::shuffle(pu.begin(), pu.end(), rng);
    std::shuffle(pi.begin(), pi.end(), rng);
    std::vector<Rec> recs(count);
    for (int t = 0; t < count; ++t) recs[t] = {pu[t], pi[t], 3.0};
    std::vector<double> U((size_t)ni * k, 0.05), V((size_t)nu * k, 0.05);
    Rec *dr; double *dU, *dV, *dib, *dub, *dse; unsigned long long *dst;
    const int waves = (count + BATCH - 1) / BATCH, blocks = (waves + kWaves - 1) / kWaves;
    CK(hipMalloc(&dr, sizeof(Rec) * count));
    if (uncached) {
        CK(hipExtMallocWithFlags((void **)&dU, U.size() * 8, hipDeviceMallocUncached));
        CK(hipExtMallocWithFlags((void **)&dV, V.size() * 8, hipDeviceMallocUncached));
    } else { CK(hipMalloc(&dU, U.size() * 8)); CK(hipMalloc(&dV, V.size() * 8)); }
    CK(hipMalloc(&dib, ni * 8)); CK(hipMalloc(&dub, nu * 8)); CK(hipMalloc(&dse, blocks * kWaves * 8));
    CK(hipMalloc(&dst, (size_t)blocks * kWaves * 8 * 8));
    CK(hipMemcpy(dr, recs.data(), sizeof(Rec) * count, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, U.data(), U.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dib, 0, ni * 8)); CK(hipMemset(dub, 0, nu * 8));
    K1Params P{}; P.recs = dr; P.U = dU; P.V = dV; P.ib = dib; P.ub = dub; P.se_partial = dse; P.k = k;
    P.upd_u = P.upd_i = 1; P.lr = 0.01; P.Ku = P.Ki = 0.1; P.Kb = 0.007; P.stamps = dst;
    const size_t lds = (size_t)kWaves * BATCH * k1_stride(k) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 50; ++rep)
            hipLaunchKernelGGL((k1_level<KP, FPL, BATCH, K1_LINEAR, NT, double, VEC>), dim3(blocks), dim3(kWaves * 64), lds, 0, P, (int64_t)0, count, (int64_t)0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 50;
    }
    std::vector<unsigned long long> st((size_t)blocks * kWaves * 8);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> d[5], startoff, endoff;
    for (int w = 0; w < waves; ++w) { t0 = std::min(t0, st[w * 8]); t1 = std::max(t1, st[w * 8 + 5]); }
    for (int w = 0; w < waves; ++w) {
        for (int i = 0; i < 5; ++i) d[i].push_back((double)(st[w * 8 + i + 1] - st[w * 8 + i]));
        startoff.push_back((double)(st[w * 8] - t0)); endoff.push_back((double)(st[w * 8 + 5] - t0));
    }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
    printf("KP %d vec %d kwaves %d uncached %d nt %d batch %2d count %d blocks %d  event %.2f us | span first-stamp..last-stamp %.0f ticks\n", KP, (int)VEC, kWaves, uncached, NT, BATCH, count, blocks, ms * 1e3, (double)(t1 - t0));
    printf("   median ticks: rec %.0f  rows %.0f  lds+sum+bias %.0f  update+issue-stores %.0f  drain %.0f\n",
           med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]));
    hipFree(dr); hipFree(dU); hipFree(dV); hipFree(dib); hipFree(dub); hipFree(dse); hipFree(dst);
    return 0;
}

int main()
{
    for (int count : {16384, 2048}) {
        if (run<8, 0>(count, 100000, 50000, 0)) return 1;
        if (run<8, 0>(count, 100000, 50000, 1)) return 1;
        if (run<8, 3>(count, 100000, 50000, 0)) return 1;
        if (run<8, 0, 32, 2, true>(count, 100000, 50000, 0)) return 1;      // 16 B per lane, plain stores
        if (run<8, 2, 32, 2, true>(count, 100000, 50000, 0)) return 1;      // 16 B per lane, write-through stores
        if (run<8, 3, 32, 2, true>(count, 100000, 50000, 0)) return 1;      // 16 B per lane, no row stores
    }
    return 0;
}
