// EXPERIMENT (third cut): like k1_xcdflow.hip, but the rows CARRY their version: every feature is a 16-byte
// granule {f64 value, u64 tag} written by one store and read by one load (observed untorn), tag = number of
// updates the row has received this epoch.  A consumer loads the rows it needs and checks the tags -- no
// separate counters, no drain of the stores before a flag: one store -> load hop per chain step.
// EXPERIMENT (second cut of k1_xcd.hip): block order + XCD-local DATAFLOW -- no barrier between the levels of
// a block: a rating waits on its user's and its item's sequence counters, which live (like the rows) in the
// XCD's L2: plain stores + sc1 loads.  See k1_xcd.hip for the block order and the placement logic.
//
// Users and items are split into P = 8 groups; round r consists of the blocks (a, (a + r) % 8), which touch
// pairwise disjoint user and item groups -- any interleaving of a round's blocks equals their sequential
// concatenation, so the block order [round 0: blocks 0..7; round 1: ...] is an ordinary (legitimate) rating
// order and executing a round's eight blocks concurrently is exact.  One launch per round; the workgroups
// that landed on XCD x (s_getreg_b32 HW_REG_XCC_ID) process block x of the round level by level (the block's
// own dependency levels) with an XCD-LOCAL barrier in between: rows are written with plain stores (the line
// stays in that XCD's L2) and read with sc1 loads (bypass the CU's L1, served by the same L2), so inside a
// round no row crosses the fabric and a level costs an L2 round trip instead of a kernel boundary.
// Checked bit for bit against the product's level-per-launch kernel on the SAME block order.
// Every spin is bounded (error flag -> everybody leaves).
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k1_xcd.hip -o /tmp/k1_xcd
#include <algorithm>
#include <cstring>
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

__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifndef XCD_WAVES
#define XCD_WAVES 8
#endif
constexpr int P = 8, B = 8, K = 64, WAVES = XCD_WAVES;
constexpr unsigned SPIN_LIMIT = 1u << 20;
struct __attribute__((aligned(16))) Gran { double v; unsigned long long tag; };
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// 16-byte sc1 load (bypasses the CU's L1, served by the XCD's L2) the compiler can see and schedule: a raw
// buffer load with the sc1 cache-policy bit; `base` is the array, `idx` the granule index (< 2^28)
__device__ __forceinline__ Gran ld_gran(__amdgpu_buffer_rsrc_t rsrc, unsigned idx)
{
    const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx * 16u, 0, 16);
    return __builtin_bit_cast(Gran, r);
}
__device__ __forceinline__ void st_gran(Gran *p, double v, unsigned long long tag)
{
    Gran g; g.v = v; g.tag = tag;
    *reinterpret_cast<v4u *>(p) = __builtin_bit_cast(v4u, g);          // plain 16-byte store: stays in this XCD's L2
}
#ifdef XCD_STAMPS
#define STAMP(i) do { if (lane == 0 && c < 4096 * 64 && (c % 64) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); stamps[(size_t)(round * 4096 + c / 64) * 8 + (i)] = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

struct BlockDesc { int64_t rec_off; int chunk_ptr; int n_chunks; };
struct Chunk { int start; int cnt; };

// ctl layout (unsigned): [0] started, [8 + x] arrivals of XCD x, [200] error,
// [256 + x*16] release word of XCD x, [1024 + (x*128 + slot)*16] arrival flag of workgroup `slot` of XCD x (own 64-B line)
__global__ __launch_bounds__(WAVES * 64) void k1_xcd_round(const Rec *__restrict__ recs, const int2 *__restrict__ seq,
                                                           const BlockDesc *__restrict__ blocks, const Chunk *__restrict__ chunks,
                                                           int round, Gran *U, Gran *V, Gran *ib, Gran *ub,
                                                           unsigned *cnt_u, unsigned *cnt_i, unsigned *ctl, double lr,
                                                           double Ku, double Ki, double Kb, unsigned long long *stamps)
{
    __shared__ __attribute__((aligned(16))) double prod_all[WAVES * B * 66];
    __shared__ unsigned sh[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double *prod = prod_all + wid * B * 66;
    // ---- who am I: XCD id, my slot on it, how many workgroups share it --------------------------------
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7u;
        const unsigned slot = atomicAdd(ctl + 8 + x, 1u);
        atomicAdd(ctl, 1u);
        unsigned spins = 0;
        bool ok = true;
        while (ld_sc1_u32(ctl) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
        unsigned nx = ld_sc1_u32(ctl + 8 + x);
        for (int y = 0; y < P; ++y) if (ld_sc1_u32(ctl + 8 + y) == 0) ok = false;      // an XCD without workers
        if (!ok) atomicExch(ctl + 200, 1u);
        sh[0] = x; sh[1] = slot; sh[2] = nx; sh[3] = ok ? 1u : 0u;
    }
    __syncthreads();
    if (!sh[3]) return;
    const unsigned x = sh[0], slot = sh[1], nx = sh[2];
    const BlockDesc bd = blocks[round * P + x];
    const int64_t rbase = bd.rec_off;
    const unsigned wave_id = slot * WAVES + wid, n_waves = nx * WAVES;
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(U, 0, 0x7fffffff, 0x27000), rV = __builtin_amdgcn_make_buffer_rsrc(V, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rib = __builtin_amdgcn_make_buffer_rsrc(ib, 0, 0x7fffffff, 0x27000), rub = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 0x7fffffff, 0x27000);
    for (int c = wave_id; c < bd.n_chunks; c += n_waves) {
        const Chunk ch = chunks[bd.chunk_ptr + c];
        const int cnt = ch.cnt;
        Rec rec; rec.user = 0; rec.item = 0; rec.r = 0.0;
        int2 sq = make_int2(0, 0);
        if (lane < cnt) { rec = recs[rbase + ch.start + lane]; sq = seq[rbase + ch.start + lane]; }
        STAMP(0);
        // ---- load the rows and biases until every granule carries the expected version (bounded) ------------
        double uval[B], vval[B];
        Gran *ur[B], *vr[B];
        unsigned long long tu[B], ti[B];
#pragma unroll
        for (int s = 0; s < B; ++s) {
            ur[s] = U + (size_t)__builtin_amdgcn_readlane(rec.item, s) * K;
            vr[s] = V + (size_t)__builtin_amdgcn_readlane(rec.user, s) * K;
            tu[s] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(sq.x, s);
            ti[s] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(sq.y, s);
            uval[s] = 0.0; vval[s] = 0.0;
        }
        double ibv = 0.0, ubv = 0.0;
        unsigned pending = (1u << cnt) - 1u;                       // ratings of the chunk whose rows are not in yet
        bool bias_ok = lane >= cnt;
        unsigned spins = 0;
        {   // cheap wait first: lane j polls granule 0 of item row j, lane 8 + j of user row j (16 granules, not 16 rows)
            const int j = lane & 7;
            const bool mine = lane < 16 && j < cnt;
            const unsigned hidx = (lane < 8) ? (unsigned)rec.item : 0u;
            const unsigned item_j = (unsigned)__shfl(rec.item, j, 64), user_j = (unsigned)__shfl(rec.user, j, 64);
            const unsigned wy = (unsigned)__shfl(sq.y, j, 64), wx = (unsigned)__shfl(sq.x, j, 64);   // both by every lane
            const unsigned long long want = (lane < 8) ? (unsigned long long)wy : (unsigned long long)wx;
            (void)hidx;
            bool ok = !mine;
            while (!__all(ok)) {
                if (!ok) {
                    const Gran h = (lane < 8) ? ld_gran(rU, item_j * K) : ld_gran(rV, user_j * K);   // per-lane choice: no cross-lane op inside
                    ok = h.tag == want;
                }
                if (++spins > SPIN_LIMIT || ld_sc1_u32(ctl + 200)) { if (lane == 0) atomicExch(ctl + 200, 1u); return; }
            }
        }
        while (pending || !__all(bias_ok)) {
            Gran ga[B], gb[B];
#pragma unroll
            for (int s = 0; s < B; ++s) {                        // all loads of the attempt in flight together
                ga[s].v = 0.0; ga[s].tag = ~0ull; gb[s] = ga[s];
                if ((pending >> s) & 1u) {
                    ga[s] = ld_gran(rU, (unsigned)(ur[s] - U) + lane);
                    gb[s] = ld_gran(rV, (unsigned)(vr[s] - V) + lane);
                }
            }
#pragma unroll
            for (int s = 0; s < B; ++s) {
                if ((pending >> s) & 1u) {
                    if (__all(ga[s].tag == ti[s] && gb[s].tag == tu[s])) { uval[s] = ga[s].v; vval[s] = gb[s].v; pending &= ~(1u << s); }
                }
            }
            if (!bias_ok) {
                const Gran a = ld_gran(rib, (unsigned)rec.item), b2 = ld_gran(rub, (unsigned)rec.user);
                if (a.tag == (unsigned long long)(unsigned)sq.y && b2.tag == (unsigned long long)(unsigned)sq.x) { ibv = a.v; ubv = b2.v; bias_ok = true; }
            }
            if (++spins > SPIN_LIMIT || ld_sc1_u32(ctl + 200)) { if (lane == 0) atomicExch(ctl + 200, 1u); return; }
        }
        STAMP(1);
#ifdef XCD_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(2);
#pragma unroll
        for (int s = 0; s < B; ++s) prod[s * 66 + lane] = uval[s] * vval[s];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double grad = 0.0;
        if (lane < cnt) {
            double sum = 0.0 + ibv + ubv;
            const double *row = prod + lane * 66;
#pragma unroll 8
            for (int f = 0; f < K; ++f) sum += row[f];
            grad = rec.r - sum;
            st_gran(ub + rec.user, ubv + lr * (grad - Kb * ubv), (unsigned long long)(unsigned)sq.x + 1ull);
            st_gran(ib + rec.item, ibv + lr * (grad - Kb * ibv), (unsigned long long)(unsigned)sq.y + 1ull);
        }
        STAMP(3);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < B; ++s) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(grad), s), hi = __builtin_amdgcn_readlane(__double2hiint(grad), s);
            const double g = __hiloint2double(hi, lo);
            if (s < cnt) {
                const double mf = uval[s], cf = vval[s];
                st_gran(ur[s] + lane, mf + lr * (g * cf - Ki * mf), ti[s] + 1ull);
                st_gran(vr[s] + lane, cf + lr * (g * mf - Ku * cf), tu[s] + 1ull);
            }
        }
        STAMP(4);
        STAMP(5);
    }
}

__global__ void k_pack(const double *src, Gran *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { dst[i].v = src[i]; dst[i].tag = 0; } }
__global__ void k_retag(Gran *g, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i].tag = 0; }
__global__ void k_unpack(const Gran *src, double *dst, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i].v; }

int main(int argc, char **argv)
{
    const int nu = 100000, ni = 50000;
    const int64_t nnz = argc > 1 ? atoll(argv[1]) : 10000000;
    const int epochs = 3;
    std::vector<int32_t> idx0((size_t)nnz * 2);
    std::vector<double> r0((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t t = 0; t < nnz; ++t) { idx0[2 * t] = (int32_t)(rnd() % nu); idx0[2 * t + 1] = (int32_t)(rnd() % ni); r0[t] = 1.0 + (double)(rnd() % 5); }
    // ---- the block order -------------------------------------------------------------------------
    std::vector<std::vector<int64_t>> blk((size_t)P * P);
    for (int64_t t = 0; t < nnz; ++t) blk[(size_t)(idx0[2 * t] % P) * P + idx0[2 * t + 1] % P].push_back(t);
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    std::vector<BlockDesc> bds((size_t)P * P);
    std::vector<Chunk> chunks;
    std::vector<int2> seqs((size_t)nnz);
    std::vector<unsigned> cu((size_t)nu, 0), ci((size_t)ni, 0);
    std::vector<Rec> recs_blk((size_t)nnz);
    int64_t pos = 0, max_lv = 0;
    for (int rd = 0; rd < P; ++rd)
        for (int a = 0; a < P; ++a) {
            const std::vector<int64_t> &v = blk[(size_t)a * P + (a + rd) % P];
            const int64_t n = (int64_t)v.size(), off = pos;
            for (int64_t j = 0; j < n; ++j, ++pos) { idx[2 * pos] = idx0[2 * v[j]]; idx[2 * pos + 1] = idx0[2 * v[j] + 1]; r[pos] = r0[v[j]]; }
            // the block's own level schedule
            std::vector<int32_t> lev((size_t)std::max<int64_t>(n, 1));
            int64_t nl = 0;
            if (mfx_schedule_levels(idx.data() + 2 * off, n, nu, ni, lev.data(), &nl)) return 1;
            std::vector<int64_t> perm((size_t)std::max<int64_t>(n, 1)), lo((size_t)nl + 1);
            if (mfx_schedule_permutation(lev.data(), n, nl, perm.data(), lo.data())) return 1;
            // chain positions over the WHOLE block order (counters persist across the rounds of an epoch)
            std::vector<int2> sq_in((size_t)std::max<int64_t>(n, 1));
            for (int64_t j = 0; j < n; ++j) { const int64_t t = off + j; sq_in[j] = make_int2((int)cu[idx[2 * t]]++, (int)ci[idx[2 * t + 1]]++); }
            for (int64_t j = 0; j < n; ++j) { const int64_t t = off + perm[j]; recs_blk[off + j] = {idx[2 * t], idx[2 * t + 1], r[t]}; seqs[off + j] = sq_in[perm[j]]; }
            bds[(size_t)rd * P + a] = {off, (int)chunks.size(), 0};
            for (int64_t l = 0; l < nl; ++l)
                for (int64_t q = lo[l]; q < lo[l + 1]; q += B) chunks.push_back({(int)q, (int)std::min<int64_t>(B, lo[l + 1] - q)});
            bds[(size_t)rd * P + a].n_chunks = (int)chunks.size() - bds[(size_t)rd * P + a].chunk_ptr;
            max_lv = std::max(max_lv, nl);
        }
    // ---- reference: the product kernel on the block order (global level schedule) --------------------
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    std::vector<Rec> recs_lvl((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs_lvl[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    printf("nnz %lld: block order has %lld global levels; deepest block %lld levels\n", (long long)nnz, (long long)n_levels, (long long)max_lv);

    std::vector<double> U0((size_t)ni * K), V0((size_t)nu * K);
    for (auto &x : U0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    for (auto &x : V0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    Rec *d_lvl, *d_blk; BlockDesc *d_bd; Chunk *d_ch; int2 *d_seq; unsigned *d_cu, *d_ci; double *dU[2], *dV[2], *dib[2], *dub[2], *d_se; unsigned *d_ctl;
    CK(hipMalloc(&d_lvl, nnz * sizeof(Rec))); CK(hipMalloc(&d_blk, nnz * sizeof(Rec))); CK(hipMalloc(&d_bd, bds.size() * sizeof(BlockDesc)));
    CK(hipMalloc(&d_ch, chunks.size() * sizeof(Chunk))); CK(hipMalloc(&d_seq, nnz * sizeof(int2))); CK(hipMalloc(&d_cu, nu * 4)); CK(hipMalloc(&d_ci, ni * 4)); CK(hipMalloc(&d_se, ((size_t)nnz / 8 + 65536) * 8)); CK(hipMalloc(&d_ctl, 32768 * 4));
    CK(hipMemcpy(d_lvl, recs_lvl.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_blk, recs_blk.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bd, bds.data(), bds.size() * sizeof(BlockDesc), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ch, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice)); CK(hipMemcpy(d_seq, seqs.data(), nnz * sizeof(int2), hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dU[v], U0.size() * 8)); CK(hipMalloc(&dV[v], V0.size() * 8)); CK(hipMalloc(&dib[v], ni * 8)); CK(hipMalloc(&dub[v], nu * 8));
        CK(hipMemcpy(dU[v], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[v], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dib[v], 0, ni * 8)); CK(hipMemset(dub[v], 0, nu * 8));
    }
    unsigned long long *d_stamps; CK(hipMalloc(&d_stamps, (size_t)P * 4096 * 8 * 8)); CK(hipMemset(d_stamps, 0, (size_t)P * 4096 * 8 * 8));
    Gran *gU, *gV, *gib, *gub;
    CK(hipMalloc(&gU, U0.size() * 16)); CK(hipMalloc(&gV, V0.size() * 16)); CK(hipMalloc(&gib, (size_t)ni * 16)); CK(hipMalloc(&gub, (size_t)nu * 16));
    k_pack<<<2048, 256>>>(dU[1], gU, U0.size()); k_pack<<<2048, 256>>>(dV[1], gV, V0.size());
    k_pack<<<256, 256>>>(dib[1], gib, (size_t)ni); k_pack<<<256, 256>>>(dub[1], gub, (size_t)nu);
    CK(hipDeviceSynchronize());
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0, ms_x = 0;
    {
        K1Params Pm{}; Pm.recs = d_lvl; Pm.U = dU[0]; Pm.V = dV[0]; Pm.ib = dib[0]; Pm.ub = dub[0]; Pm.se_partial = d_se; Pm.k = K;
        Pm.upd_u = Pm.upd_i = 1; Pm.lr = lr; Pm.Ku = Ku; Pm.Ki = Ki; Pm.Kb = Kb;
        const size_t lds = (size_t)kWaves * 8 * k1_stride(K) * 8;
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == epochs - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int cnt = (int)(off[l + 1] - off[l]);
                const int blocks = (cnt + 8 * kWaves - 1) / (8 * kWaves);
                hipLaunchKernelGGL((k1_level<32, 2, 8, K1_LINEAR, 2, double, true>), dim3(blocks), dim3(kWaves * 64), lds, 0, Pm, off[l], cnt, (int64_t)0);
            }
            if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    unsigned err = 0;
    for (int ep = 0; ep < epochs && !err; ++ep) {
        if (ep == epochs - 1) CK(hipEventRecord(e0));
        k_retag<<<2048, 256>>>(gU, U0.size()); k_retag<<<2048, 256>>>(gV, V0.size()); k_retag<<<256, 256>>>(gib, (size_t)ni); k_retag<<<256, 256>>>(gub, (size_t)nu);
        for (int rd = 0; rd < P; ++rd) {
            CK(hipMemsetAsync(d_ctl, 0, 32768 * 4));
            hipLaunchKernelGGL(k1_xcd_round, dim3(grid), dim3(WAVES * 64), 0, 0, d_blk, d_seq, d_bd, d_ch, rd, gU, gV, gib, gub, d_cu, d_ci, d_ctl, lr, Ku, Ki, Kb, d_stamps);
            if (ep == 0) { CK(hipMemcpy(&err, d_ctl + 200, 4, hipMemcpyDeviceToHost)); if (err) break; }
        }
        if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_x, e0, e1)); }
    }
    CK(hipDeviceSynchronize());
    unsigned arr[P]; CK(hipMemcpy(arr, d_ctl + 8, sizeof arr, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&err, d_ctl + 200, 4, hipMemcpyDeviceToHost));
    k_unpack<<<2048, 256>>>(gU, dU[1], U0.size()); k_unpack<<<2048, 256>>>(gV, dV[1], V0.size());
    CK(hipDeviceSynchronize());
    std::vector<double> a(U0.size()), b(U0.size()), c(V0.size()), d(V0.size());
    CK(hipMemcpy(a.data(), dU[0], a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dU[1], b.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), dV[0], c.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), dV[1], d.size() * 8, hipMemcpyDeviceToHost));
    size_t diffU = 0, diffV = 0;
    for (size_t i = 0; i < a.size(); ++i) diffU += (memcmp(&a[i], &b[i], 8) != 0);
    for (size_t i = 0; i < c.size(); ++i) diffV += (memcmp(&c[i], &d[i], 8) != 0);
#ifdef XCD_STAMPS
    {
        std::vector<unsigned long long> hs((size_t)P * 4096 * 8);
        CK(hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> d[5];
        for (size_t w = 0; w < (size_t)P * 4096; ++w) if (hs[w * 8 + 5]) for (int i = 0; i < 5; ++i) d[i].push_back((double)(hs[w * 8 + i + 1] - hs[w * 8 + i]));
        auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("median ticks per chunk (%zu samples): rows+tags arrive %.0f | (stamp 1->2) %.0f | lds+sum+bias %.0f | update+issue stores %.0f | store ack %.0f\n", d[0].size(), med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]));
    }
#endif
    printf("workgroups per XCD (last round): %u %u %u %u %u %u %u %u\n", arr[0], arr[1], arr[2], arr[3], arr[4], arr[5], arr[6], arr[7]);
    printf("block order: level-per-launch %.3f ms/epoch = %.3f G/s | XCD-local tagged-granule rounds (grid %d): %.3f ms/epoch = %.3f G/s | error flag %u | differing U %zu V %zu\n",
           ms_ref, nnz / ms_ref / 1e6, grid, ms_x, nnz / ms_x / 1e6, err, diffU, diffV);
    return 0;
}
