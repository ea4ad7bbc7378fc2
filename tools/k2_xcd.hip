// EXPERIMENT: Funk per-feature pass (K2) as ONE persistent launch that runs on a SINGLE XCD.
//
// The pass is a chain of ~618 dependency levels of ~16 K scalar updates each; with one launch per level
// it costs ~3.4 us per level whatever the level's size.  The two feature vectors
// U[f,:] + V[f,:] are 1.2 MB: they fit one XCD's 4 MB L2, and inside one XCD the L2 is the coherence
// point (plain stores stay there, sc1 loads bypass the CU's L1), so a level hand-off can cost an L2 round
// trip instead of a kernel boundary.  Variants (argv[2]):
//   1  L2 path for both vectors, level-major order as scheduled
//   2  the same with every level sorted by item (coalesced item-side accesses)
//   3  users partitioned into 32 groups, a workgroup keeps its groups' slice of V[f,:] in LDS for the whole
//      pass; ratings ordered (level, group, item); only the item side goes through L2
// A workgroup = 1 sync wave + W worker waves.  Flat XCD-local barrier: every workgroup stores its flag
// (own 128-B line), every sync wave polls all flags.  Every spin is bounded.
// Checked bit for bit against the product's level-per-launch kernel.
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k2_xcd.hip -o /tmp/k2_xcd
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "../mfrec_amd/csrc/mfx_sgd_kernels.h"
namespace mfx {
void set_error(const char *, ...) {}
int fail(int c, const char *fmt, ...) { fprintf(stderr, "fail %d: %s\n", c, fmt); return c; }
int last_code() { return 0; }
}  // namespace mfx
#include "../mfrec_amd/csrc/mfx_schedule.cpp"
using namespace mfx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// (ld_sc1_u32: mfx_sgd_kernels.h)
__device__ __forceinline__ double ld_sc1_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifndef WORKERS
#define WORKERS 8
#endif
constexpr int W = WORKERS;                 // worker waves per workgroup
constexpr int NG = 32;                     // user groups (variant 3)
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct XParams {
    const Rec *recs;
    const double *cache;
    double *uf, *vf;
    const int *seg_off;        // variant 1/2: level_off [n_levels+1]; variant 3: [(n_levels*NG)+1]
    const int *group_lo;       // variant 3: [NG+1] user bounds
    double *se_partial;        // one per (segment chunk); experiment: indexed by global position / 64
    unsigned *ctl;
    int n_levels;
    int lds_doubles;           // variant 3: capacity of the dynamic LDS
    double lr, K, mu, trail;
    unsigned long long *stamps;   // -DSTAMPS: [n_levels][8] s_memtime of workgroup slot 0
};
#ifdef STAMPS
#define STAMP(i) do { if (slot == 0 && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); P.stamps[(size_t)l * 8 + (i)] = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ctl layout (unsigned): [x*32] registrations of XCD x (own line), [512] error, [1024 + slot*32] flag of workgroup slot
template <int VARIANT>
__global__ __launch_bounds__((W + 1) * 64) void k2_xcd_pass(XParams P)
{
    extern __shared__ double vlds[];           // variant 3: this workgroup's slices of vf
    __shared__ unsigned sh[6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // ---- registration: which XCD runs the pass, my slot on it, how many workgroups share it ---------
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7u;
        const unsigned slot = atomicAdd(P.ctl + x * 32, 1u);
        unsigned spins = 0, cnt[8], tot = 0;
        bool ok = true;
        for (;;) {
            tot = 0;
            for (int y = 0; y < 8; ++y) { cnt[y] = ld_sc1_u32(P.ctl + y * 32); tot += cnt[y]; }
            if (tot >= gridDim.x) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
        unsigned best = 0;
        for (int y = 1; y < 8; ++y) if (cnt[y] > cnt[best]) best = y;
        if (!ok) atomicExch(P.ctl + 512, 1u);
        sh[0] = x; sh[1] = slot; sh[2] = cnt[best]; sh[3] = ok ? 1u : 0u; sh[4] = best;
    }
    __syncthreads();
    if (!sh[3] || sh[0] != sh[4]) return;               // not on the chosen XCD: leave
    const unsigned slot = sh[1], nx = sh[2] > 64 ? 64 : sh[2];
    if (slot >= nx) return;
    unsigned *flags = P.ctl + 1024;
    const double lr = P.lr, K = P.K, trail = P.trail;
    double *const uf = P.uf, *const vf = P.vf;

    if constexpr (VARIANT == 3) {
        // load my groups' user slices into LDS (groups slot, slot+nx, ...)
        int need = 0;
        for (int g = slot; g < NG; g += nx) need += P.group_lo[g + 1] - P.group_lo[g];
        if (need > P.lds_doubles) {                       // wave-uniform: every thread computes the same
            if (threadIdx.x == 0) atomicExch(P.ctl + 512, 2u);
            return;
        }
        int base = 0;
        for (int g = slot; g < NG; g += nx) {
            const int lo = P.group_lo[g], n = P.group_lo[g + 1] - lo;
            for (int i = threadIdx.x; i < n; i += (W + 1) * 64) vlds[base + i] = vf[lo + i];
            base += n;
        }
        __syncthreads();
    }

    const int worker = wid - 1;                          // -1 = the sync wave
    // records and cache words do not depend on the updates: the FIRST chunk of the next level is requested
    // before the barrier and is (mostly) there when the level opens
    Rec nrec; nrec.user = 0; nrec.item = 0; nrec.r = 0.0;
    double ncache = 0.0;
    auto first_chunk = [&](int l, int &tb, int &s1, int &lo, int &base) {
        if constexpr (VARIANT == 3) {
            const int g = slot;
            lo = P.group_lo[g]; base = 0;
            tb = P.seg_off[l * NG + g] + worker * 64; s1 = P.seg_off[l * NG + g + 1];
        } else {
            lo = 0; base = 0;
            tb = P.seg_off[l] + (slot * W + worker) * 64; s1 = P.seg_off[l + 1];
        }
    };
#ifdef PREFETCH
    if (worker >= 0 && P.n_levels > 0) {
        int tb, s1, lo, base;
        first_chunk(0, tb, s1, lo, base);
        if (tb + lane < s1) { nrec = P.recs[tb + lane]; ncache = P.cache[tb + lane]; }
    }
#endif
    auto one = [&](const Rec &rec, double c, int lo, int base, int l = -1) -> double {
#ifdef STAMPS
        if (l >= 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (worker == 0) STAMP(6); }
#endif
        const double mf = ld_sc1_f64(uf + rec.item);
#ifdef STAMPS
        if (l >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (worker == 0) STAMP(7); }
#endif
        double cf;
        if constexpr (VARIANT == 3) cf = vlds[base + rec.user - lo];
        else cf = ld_sc1_f64(vf + rec.user);
        const double p = funk_estimate(mf, cf, c, P.mu + 0.0 + 0.0, trail, true);
        const double err = rec.r - p;
        uf[rec.item] = mf + lr * (err * cf - K * mf);
        const double vn = cf + lr * (err * mf - K * cf);
        if constexpr (VARIANT == 3) vlds[base + rec.user - lo] = vn;
        else vf[rec.user] = vn;
        return err * err;
    };
    for (int l = 0; l < P.n_levels; ++l) {
        if (worker == 0) STAMP(0);
        if (worker >= 0) {
            if constexpr (VARIANT == 3) {
                int base = 0;
                bool first = true;
                for (int g = slot; g < NG; g += nx) {
                    const int lo = P.group_lo[g];
                    const int s0 = P.seg_off[l * NG + g], s1 = P.seg_off[l * NG + g + 1];
                    for (int tb = s0 + worker * 64; tb < s1; tb += W * 64) {      // wave-uniform loop
                        const int t = tb + lane;
                        double e2 = 0.0;
                        if (t < s1) {
#ifdef PREFETCH
                            const Rec rec = first ? nrec : P.recs[t];
                            const double c = first ? ncache : P.cache[t];
#else
                            const Rec rec = P.recs[t];
                            const double c = P.cache[t];
#endif
                            e2 = one(rec, c, lo, base, first ? l : -1);
                        }
                        first = false;
                        const double ws = wave_sum_fixed(e2);
                        if (lane == 0) P.se_partial[tb >> 6] = ws;
                    }
                    first = false;
                    base += P.group_lo[g + 1] - lo;
                }
            } else {
                const int s0 = P.seg_off[l], s1 = P.seg_off[l + 1];
                const int nw = nx * W, w = slot * W + worker;
                bool first = true;
                for (int t0 = s0 + w * 64; t0 < s1; t0 += nw * 64) {
                    const int t = t0 + lane;
                    double e2 = 0.0;
                    if (t < s1) {
#ifdef PREFETCH
                        const Rec rec = first ? nrec : P.recs[t];
                        const double c = first ? ncache : P.cache[t];
#else
                        const Rec rec = P.recs[t];
                        const double c = P.cache[t];
#endif
                        e2 = one(rec, c, 0, 0);
                    }
                    first = false;
                    const double ws = wave_sum_fixed(e2);
                    if (lane == 0) P.se_partial[t0 >> 6] = ws;
                }
            }
            if (worker == 0) STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores are in L2
            if (worker == 0) STAMP(2);
#ifdef PREFETCH
            if (l + 1 < P.n_levels) {
                int tb, s1, lo, base;
                first_chunk(l + 1, tb, s1, lo, base);
                if (tb + lane < s1) { nrec = P.recs[tb + lane]; ncache = P.cache[tb + lane]; }
            }
#endif
        }
        __syncthreads();                                          // A: every worker's stores are in L2
        if (worker < 0) {
            STAMP(3);
            const unsigned gen = (unsigned)(l + 1);
            if (lane == 0) flags[slot * 32] = gen;               // plain store: stays in this XCD's L2
            unsigned spins = 0;
            for (;;) {
                const bool ok = lane >= (int)nx || ld_sc1_u32(flags + lane * 32) >= gen;
                if (__all(ok)) break;
                if (++spins > SPIN_LIMIT || ld_sc1_u32(P.ctl + 512)) { if (lane == 0) atomicExch(P.ctl + 512, 1u); break; }
            }
            STAMP(4);
#ifdef STAMPS
            (void)spins;
#endif
        }
        __syncthreads();                                          // B: the level is complete everywhere
        if (worker == 0) STAMP(5);
    }
    if constexpr (VARIANT == 3) {
        int base = 0;
        for (int g = slot; g < NG; g += nx) {
            const int lo = P.group_lo[g], n = P.group_lo[g + 1] - lo;
            for (int i = threadIdx.x; i < n; i += (W + 1) * 64) vf[lo + i] = vlds[base + i];
            base += n;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Variants 4 / 5: no barrier at all.  Every feature-vector element travels as a 16-byte granule
// {value, tag}; tag = pass << 32 | number of updates the element has received in this pass.  A rating knows
// (from the schedule) how many updates of its item / its user precede it, polls the two granules until
// the tags say so, and publishes its own results with ONE 16-byte store each.  The first rating of an
// element in a pass reads the plain array (settled at the kernel boundary), the last one also writes it.
//   4: workgroups on ALL XCDs, granules stored write-through (sc1), read sc1
//   5: one XCD (registration as above), granules stored plain (stay in that XCD's L2), read sc1
struct Gran { double v; unsigned long long tag; };
__device__ __forceinline__ Gran ld16_sc1(const Gran *p)
{
    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
    v4u32 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return __builtin_bit_cast(Gran, r);
}
__device__ __forceinline__ void st16_sc1(Gran *p, Gran g)
{
    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
    const v4u32 v = __builtin_bit_cast(v4u32, g);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
struct DParams {
    const Rec *recs;
    const double *cache;
    const int2 *seq;            // per rating: .x = updates of its user before it (bit 31: it is the last), .y = same for its item
    double *uf, *vf;            // plain arrays
    Gran *ug, *vg;              // granule arrays
    const int *level_off;
    double *se_partial;
    unsigned *ctl;
    int n_levels;
    unsigned pass;
    double lr, K, mu, trail;
};
template <int VARIANT>
__global__ __launch_bounds__(256) void k2_flow_pass(DParams P)
{
    __shared__ unsigned sh[6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned slot = blockIdx.x, nx = gridDim.x;
    if constexpr (VARIANT == 5) {
        if (threadIdx.x == 0) {
            unsigned x;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            x &= 7u;
            const unsigned sl = atomicAdd(P.ctl + x * 32, 1u);
            unsigned spins = 0, cnt[8], tot = 0;
            bool ok = true;
            for (;;) {
                tot = 0;
                for (int y = 0; y < 8; ++y) { cnt[y] = ld_sc1_u32(P.ctl + y * 32); tot += cnt[y]; }
                if (tot >= gridDim.x) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { ok = false; break; }
            }
            unsigned best = 0;
            for (int y = 1; y < 8; ++y) if (cnt[y] > cnt[best]) best = y;
            if (!ok) atomicExch(P.ctl + 512, 1u);
            sh[0] = x; sh[1] = sl; sh[2] = cnt[best]; sh[3] = ok ? 1u : 0u; sh[4] = best;
        }
        __syncthreads();
        if (!sh[3] || sh[0] != sh[4]) return;
        slot = sh[1]; nx = sh[2];
    }
    const int nw = (int)nx * 4, w = (int)slot * 4 + wid;
    const unsigned long long hi = (unsigned long long)P.pass << 32;
    const double lr = P.lr, K = P.K, trail = P.trail;
    for (int l = 0; l < P.n_levels; ++l) {
        const int s0 = P.level_off[l], s1 = P.level_off[l + 1];
        for (int t0 = s0 + w * 64; t0 < s1; t0 += nw * 64) {
            const int t = t0 + lane;
            double e2 = 0.0;
            if (t < s1) {
                const Rec rec = P.recs[t];
                const double c = P.cache[t];
                const int2 sq = P.seq[t];
                const unsigned useq = (unsigned)sq.x & 0x7fffffffu, iseq = (unsigned)sq.y & 0x7fffffffu;
                double mf = 0.0, cf = 0.0;
                bool need_i = iseq != 0, need_u = useq != 0;
                if (!need_i) mf = P.uf[rec.item];
                if (!need_u) cf = P.vf[rec.user];
                unsigned spins = 0;
                while (need_i || need_u) {
                    if (need_i) {
                        const Gran g = ld16_sc1(P.ug + rec.item);
                        if (g.tag == (hi | iseq)) { mf = g.v; need_i = false; }
                    }
                    if (need_u) {
                        const Gran g = ld16_sc1(P.vg + rec.user);
                        if (g.tag == (hi | useq)) { cf = g.v; need_u = false; }
                    }
                    if (++spins > SPIN_LIMIT) { atomicExch(P.ctl + 512, 3u); break; }
                }
                const double p = funk_estimate(mf, cf, c, P.mu + 0.0 + 0.0, trail, true);
                const double err = rec.r - p;
                e2 = err * err;
                const double un = mf + lr * (err * cf - K * mf), vn = cf + lr * (err * mf - K * cf);
                const Gran gi{un, hi | (iseq + 1)}, gu{vn, hi | (useq + 1)};
                if constexpr (VARIANT == 4) { st16_sc1(P.ug + rec.item, gi); st16_sc1(P.vg + rec.user, gu); }
                else { P.ug[rec.item] = gi; P.vg[rec.user] = gu; }
                if (sq.y < 0) P.uf[rec.item] = un;
                if (sq.x < 0) P.vf[rec.user] = vn;
            }
            const double ws = wave_sum_fixed(e2);
            if (lane == 0) P.se_partial[t0 >> 6] = ws;
        }
        if (ld_sc1_u32(P.ctl + 512)) return;
    }
}

int main(int argc, char **argv)
{
    const int nu = 100000, ni = 50000;
    const int64_t nnz = argc > 1 ? atoll(argv[1]) : 10000000;
    const int variant = argc > 2 ? atoi(argv[2]) : 1;
    const int grid = argc > 3 ? atoi(argv[3]) : 256;
    const int passes = 4;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t t = 0; t < nnz; ++t) { idx[2 * t] = (int32_t)(rnd() % nu); idx[2 * t + 1] = (int32_t)(rnd() % ni); r[t] = 1.0 + (double)(rnd() % 5); }
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    // reference order: level-major, stable
    std::vector<Rec> recsA((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recsA[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    // experiment order
    std::vector<int> glo(NG + 1);
    for (int g = 0; g <= NG; ++g) glo[g] = (int)((int64_t)nu * g / NG);
    auto group_of = [&](int u) { int g = (int)((int64_t)u * NG / nu); while (u < glo[g]) --g; while (u >= glo[g + 1]) ++g; return g; };
    std::vector<Rec> recsX = recsA;
    std::vector<int> seg;
    if (variant == 2 || variant == 3 || (variant >= 4 && argc > 4 && atoi(argv[4]) == 1)) {      // argv[4] = 1: item-sorted levels
        for (int64_t l = 0; l < n_levels; ++l) {
            auto b = recsX.begin() + off[l], e = recsX.begin() + off[l + 1];
            if (variant != 3) std::sort(b, e, [](const Rec &a, const Rec &c) { return a.item < c.item; });
            else std::sort(b, e, [&](const Rec &a, const Rec &c) { const int ga = group_of(a.user), gc = group_of(c.user); return ga != gc ? ga < gc : a.item < c.item; });
        }
    }
    if (variant == 3) {
        seg.assign((size_t)n_levels * NG + 1, 0);
        for (int64_t l = 0; l < n_levels; ++l) {
            std::vector<int> cnt(NG, 0);
            for (int64_t j = off[l]; j < off[l + 1]; ++j) ++cnt[group_of(recsX[j].user)];
            int p = (int)off[l];
            for (int g = 0; g < NG; ++g) { seg[(size_t)l * NG + g] = p; p += cnt[g]; }
        }
        seg[(size_t)n_levels * NG] = (int)nnz;
    } else {
        seg.resize((size_t)n_levels + 1);
        for (int64_t l = 0; l <= n_levels; ++l) seg[l] = (int)off[l];
    }
    printf("nnz %lld, %lld levels, variant %d, grid %d, %d worker waves\n", (long long)nnz, (long long)n_levels, variant, grid, W);

    std::vector<double> u0((size_t)ni, 0.1), v0((size_t)nu, 0.1);
    // the cache belongs to the rating, not to the position: give A and X the same value per rating
    std::vector<double> cacheA((size_t)nnz), cacheX((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) cacheA[j] = 1.0 + (double)((uint64_t)(recsA[j].user * 2654435761u + recsA[j].item * 40503u) % 4000) * 1e-3 * (((recsA[j].user + recsA[j].item) & 3) ? 1 : 0);
    for (int64_t j = 0; j < nnz; ++j) cacheX[j] = 1.0 + (double)((uint64_t)(recsX[j].user * 2654435761u + recsX[j].item * 40503u) % 4000) * 1e-3 * (((recsX[j].user + recsX[j].item) & 3) ? 1 : 0);
    Rec *dA, *dX; double *dcA, *dcX, *duf[2], *dvf[2], *d_se; int *d_seg, *d_glo; unsigned *d_ctl; K2Dyn *d_dyn; int64_t *d_dummy;
    CK(hipMalloc(&dA, nnz * sizeof(Rec))); CK(hipMalloc(&dX, nnz * sizeof(Rec)));
    CK(hipMalloc(&dcA, nnz * 8)); CK(hipMalloc(&dcX, nnz * 8)); CK(hipMalloc(&d_se, (nnz / 64 + 65536) * 8));
    CK(hipMalloc(&d_seg, seg.size() * 4)); CK(hipMalloc(&d_glo, glo.size() * 4)); CK(hipMalloc(&d_ctl, 8192 * 4)); CK(hipMalloc(&d_dyn, sizeof(K2Dyn)));
    CK(hipMemcpy(dA, recsA.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice)); CK(hipMemcpy(dX, recsX.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(dcA, cacheA.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dcX, cacheX.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_glo, glo.data(), glo.size() * 4, hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&duf[v], ni * 8)); CK(hipMalloc(&dvf[v], nu * 8));
        CK(hipMemcpy(duf[v], u0.data(), ni * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dvf[v], v0.data(), nu * 8, hipMemcpyHostToDevice));
    }
    const double lr = 0.001, K = 0.05, mu = 1.0, trail = 0.63;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0, ms_x = 0;
    {   // reference: one launch per level (the product's kernel)
        K2Dyn dyn{duf[0], dvf[0], trail};
        CK(hipMemcpy(d_dyn, &dyn, sizeof(dyn), hipMemcpyHostToDevice));
        K2Params Pm{}; Pm.recs = dA; Pm.dyn = d_dyn; Pm.cache = dcA; Pm.se_partial = d_se; Pm.lr = lr; Pm.K = K; Pm.mu = mu; Pm.upd_u = Pm.upd_i = 1;
        for (int ps = 0; ps < passes; ++ps) {
            if (ps == passes - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int cnt = (int)(off[l + 1] - off[l]);
                hipLaunchKernelGGL(k2_level<true>, dim3((cnt + kK2Block - 1) / kK2Block), dim3(kK2Block), 0, 0, Pm, off[l], cnt, (int64_t)(off[l] / 64 + l));
            }
            if (ps == passes - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    if (variant == 4 || variant == 5) {
        std::vector<int2> seq((size_t)nnz);
        std::vector<int> uc(nu, 0), ic(ni, 0), ulast(nu, -1), ilast(ni, -1);
        for (int64_t j = 0; j < nnz; ++j) {
            const Rec &q = recsX[j];
            seq[j].x = uc[q.user]++; seq[j].y = ic[q.item]++;
            ulast[q.user] = (int)j; ilast[q.item] = (int)j;
        }
        for (int u = 0; u < nu; ++u) if (ulast[u] >= 0) seq[ulast[u]].x |= (int)0x80000000;
        for (int i = 0; i < ni; ++i) if (ilast[i] >= 0) seq[ilast[i]].y |= (int)0x80000000;
        int2 *d_seq; Gran *d_ug, *d_vg;
        CK(hipMalloc(&d_seq, nnz * sizeof(int2))); CK(hipMalloc(&d_ug, ni * sizeof(Gran))); CK(hipMalloc(&d_vg, nu * sizeof(Gran)));
        CK(hipMemcpy(d_seq, seq.data(), nnz * sizeof(int2), hipMemcpyHostToDevice));
        CK(hipMemset(d_ug, 0, ni * sizeof(Gran))); CK(hipMemset(d_vg, 0, nu * sizeof(Gran)));
        DParams D{}; D.recs = dX; D.cache = dcX; D.seq = d_seq; D.uf = duf[1]; D.vf = dvf[1]; D.ug = d_ug; D.vg = d_vg; D.level_off = d_seg;
        D.se_partial = d_se; D.ctl = d_ctl; D.n_levels = (int)n_levels; D.lr = lr; D.K = K; D.mu = mu; D.trail = trail;
        void *fn = variant == 4 ? (void *)k2_flow_pass<4> : (void *)k2_flow_pass<5>;
        unsigned err = 0;
        for (int ps = 0; ps < passes && !err; ++ps) {
            CK(hipMemsetAsync(d_ctl, 0, 8192 * 4));
            D.pass = (unsigned)(ps + 1);
            if (ps == passes - 1) CK(hipEventRecord(e0));
            void *args[] = {&D};
            CK(hipLaunchKernel(fn, dim3(grid), dim3(256), args, 0, 0));
            if (ps == passes - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_x, e0, e1)); }
            CK(hipMemcpy(&err, d_ctl + 512, 4, hipMemcpyDeviceToHost));
        }
        CK(hipDeviceSynchronize());
        std::vector<double> a(ni), b(ni), c(nu), d(nu);
        CK(hipMemcpy(a.data(), duf[0], ni * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), duf[1], ni * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), dvf[0], nu * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), dvf[1], nu * 8, hipMemcpyDeviceToHost));
        size_t du = 0, dv = 0;
        for (int i = 0; i < ni; ++i) du += memcmp(&a[i], &b[i], 8) != 0;
        for (int i = 0; i < nu; ++i) dv += memcmp(&c[i], &d[i], 8) != 0;
        printf("K2 pass: level-per-launch %.3f ms = %.2f G scalar-updates/s | tagged dataflow (variant %d, grid %d): %.3f ms = %.2f G/s = %.2f us per level | error %u | differing uf %zu vf %zu\n",
               ms_ref, nnz / ms_ref / 1e6, variant, grid, ms_x, nnz / ms_x / 1e6, ms_x * 1e3 / n_levels, err, du, dv);
        return (err || du || dv) ? 2 : 0;
    }
    XParams X{}; X.recs = dX; X.cache = dcX; X.uf = duf[1]; X.vf = dvf[1]; X.seg_off = d_seg; X.group_lo = d_glo; X.se_partial = d_se; X.ctl = d_ctl;
    X.n_levels = (int)n_levels; X.lr = lr; X.K = K; X.mu = mu; X.trail = trail;
    unsigned long long *d_stamps = nullptr;
    CK(hipMalloc(&d_stamps, (size_t)(n_levels + 1) * 8 * 8)); CK(hipMemset(d_stamps, 0, (size_t)(n_levels + 1) * 8 * 8));
    X.stamps = d_stamps;
    const size_t lds = variant == 3 ? (size_t)((nu + NG - 1) / NG + 8) * 8 * 2 : 0;      // up to two groups per workgroup (nx >= 16)
    X.lds_doubles = (int)(lds / 8);
    void *fn = variant == 3 ? (void *)k2_xcd_pass<3> : (void *)k2_xcd_pass<1>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned err = 0;
    for (int ps = 0; ps < passes && !err; ++ps) {
        CK(hipMemsetAsync(d_ctl, 0, 8192 * 4));
        if (ps == passes - 1) CK(hipEventRecord(e0));
        void *args[] = {&X};
        CK(hipLaunchKernel(fn, dim3(grid), dim3((W + 1) * 64), args, lds, 0));
        if (ps == passes - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_x, e0, e1)); }
        CK(hipMemcpy(&err, d_ctl + 512, 4, hipMemcpyDeviceToHost));
    }
    CK(hipDeviceSynchronize());
    unsigned reg[8];
    for (int y = 0; y < 8; ++y) CK(hipMemcpy(&reg[y], d_ctl + y * 32, 4, hipMemcpyDeviceToHost));
    std::vector<double> a(ni), b(ni), c(nu), d(nu);
    CK(hipMemcpy(a.data(), duf[0], ni * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), duf[1], ni * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), dvf[0], nu * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), dvf[1], nu * 8, hipMemcpyDeviceToHost));
    size_t du = 0, dv = 0;
    for (int i = 0; i < ni; ++i) du += memcmp(&a[i], &b[i], 8) != 0;
    for (int i = 0; i < nu; ++i) dv += memcmp(&c[i], &d[i], 8) != 0;
#ifdef STAMPS
    {
        std::vector<unsigned long long> st((size_t)n_levels * 8);
        CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double d[6] = {0, 0, 0, 0, 0, 0}, sp = 0, e6 = 0, e7 = 0; int n = 0;
        for (int64_t l = 20; l + 1 < n_levels - 20; ++l, ++n) {
            const unsigned long long *a = &st[(size_t)l * 8], *b = &st[(size_t)(l + 1) * 8];
            d[0] += (double)(a[1] - a[0]);      // level open -> updates issued (sc1 load + compute + store issue)
            d[1] += (double)(a[2] - a[1]);      // store drain (vmcnt(0))
            d[2] += (double)(a[3] - a[2]);      // barrier A (+ prefetch issue)
            d[3] += (double)(a[4] - a[3]);      // flag store -> all flags seen
            d[4] += (double)(a[5] - a[4]);      // barrier B
            d[5] += (double)(b[0] - a[0]);      // whole level
            if (a[6] && a[7]) { e6 += (double)(a[6] - a[0]); e7 += (double)(a[7] - a[6]); }
        }
        printf("  inside load+compute+issue: level open -> record/cache in registers %.0f | sc1 load of U[f,item] %.0f | rest (LDS, compute, wave sum, store issue) %.0f\n",
               e6 / n, e7 / n, (d[0] - e6 - e7) / n);
        printf("stamps (s_memtime ticks, workgroup slot 0, mean over %d levels): load+compute+issue %.0f | store drain %.0f | barrier A %.0f | flag->all flags %.0f (%.1f poll rounds) | barrier B %.0f | level %.0f\n",
               n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, sp / n, d[4] / n, d[5] / n);
    }
#endif
    printf("workgroups per XCD: %u %u %u %u %u %u %u %u\n", reg[0], reg[1], reg[2], reg[3], reg[4], reg[5], reg[6], reg[7]);
    printf("K2 pass: level-per-launch %.3f ms = %.2f G scalar-updates/s | single-XCD persistent (variant %d): %.3f ms = %.2f G/s = %.2f us per level | error %u | differing uf %zu vf %zu\n",
           ms_ref, nnz / ms_ref / 1e6, variant, ms_x, nnz / ms_x / 1e6, ms_x * 1e3 / n_levels, err, du, dv);
    return (err || du || dv) ? 2 : 0;
}
