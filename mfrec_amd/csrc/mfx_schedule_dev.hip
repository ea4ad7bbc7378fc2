// Device-side dependency-level schedule: the same levels and the
// same level-major record order as the host builder (mfx_schedule.cpp), computed on the GPU
// straight from the (user, item, rating) arrays.
//
//   level(t) = 1 + max(level(previous rating of the same user), level(previous rating of the same item))
//
// is a longest-path layering of a DAG in which every node has at most two predecessors and two
// successors (its neighbours in the user's chain and in the item's chain).  Steps:
//   1. stable radix sort of t by user, and of t by item  -> the chains (next_u, next_i, in-degree)
//   2. Kahn's algorithm, one launch per level: a rating joins the frontier when its last
//      predecessor has been levelled; cost O(nnz) in total, ~3 us per level
//   3. stable radix sorts of t by item, then by level -> level-major order, ascending ITEM inside a
//      level (an item occurs once per level; the plan's host path sorts the same way), gather into
//      the 16-byte records
// The host pass is sequential (3 ns per rating, cache-missing); this one wins while the level count
// stays below ~nnz/1000 and gives up (caller falls back to the host) beyond a level budget, which
// is what heavily skewed data (a user with 10^5 ratings) produces.
// Radix sort: rocprim::radix_sort_pairs (ROCm's native primitive library, stable) -- off the hot path.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>
#include "mfx_common.h"
#include "mfx_sgd_internal.h"

namespace mfx {
namespace {

constexpr int kBfsBlocks = 256;      // 64 K threads cover a typical frontier; grid-stride beyond
constexpr int kBfsThreads = 256;

__global__ void k_split_check(const int2 *__restrict__ idx, int64_t nnz, int nu, int ni,
                              int32_t *users, int32_t *items, int32_t *tvals,
                              unsigned long long *first_bad)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nnz;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int2 p = idx[t];
        users[t] = p.x;
        items[t] = p.y;
        tvals[t] = (int32_t)t;
        if (p.x < 0 || p.x >= nu || p.y < 0 || p.y >= ni) atomicMin(first_bad, (unsigned long long)t);
    }
}

// sorted (key, t) pairs -> chain links.  Thread p owns t = vals[p], so no two threads write the
// same slot; the second call (items) adds to the in-degree the first one stored.
__global__ void k_link(const int32_t *__restrict__ keys, const int32_t *__restrict__ vals,
                       int64_t nnz, int32_t *next, int32_t *indeg, int first)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t key = keys[p], t = vals[p];
        const int pred = (p > 0 && keys[p - 1] == key) ? 1 : 0;
        next[t] = (p + 1 < nnz && keys[p + 1] == key) ? vals[p + 1] : -1;
        indeg[t] = first ? pred : indeg[t] + pred;
    }
}

// append with one atomic per wave
__device__ __forceinline__ void push(int32_t *out, unsigned long long *cnt, bool have, int32_t v)
{
    const unsigned long long m = __ballot(have);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(cnt, (unsigned long long)__popcll(m));
    base = __shfl(base, leader, 64);
    if (have) out[base + __popcll(m & ((1ull << lane) - 1))] = v;
}

__global__ void k_seed(const int32_t *__restrict__ indeg, int64_t nnz, int32_t *out,
                       unsigned long long *cnt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (nnz + stride - 1) / stride;          // whole waves stay together
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = 0; i < rounds; ++i, t += stride) {
        const bool have = t < nnz && indeg[t] == 0;
        push(out, cnt, have, (int32_t)t);
    }
}

// one level of Kahn's algorithm.  cnt[3] rotate: read cnt[L%3], append to cnt[(L+1)%3], clear
// cnt[(L+2)%3] (read by the previous launch, appended to by the next).
__global__ void k_bfs_level(int32_t L, const int32_t *__restrict__ fr_in, int32_t *fr_out,
                            unsigned long long *cnt, const int32_t *__restrict__ next_u,
                            const int32_t *__restrict__ next_i, int32_t *indeg, int32_t *level,
                            unsigned long long *sizes, unsigned long long *total)
{
    const unsigned long long n = cnt[L % 3];
    unsigned long long *cnt_out = cnt + (L + 1) % 3;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sizes[L - 1] = n;
        *total += n;
        cnt[(L + 2) % 3] = 0;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = ((int64_t)n + stride - 1) / stride;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t q = 0; q < rounds; ++q, i += stride) {
        int32_t a = -1, b = -1;
        if (i < (int64_t)n) {
            const int32_t t = fr_in[i];
            level[t] = L;
            a = next_u[t];
            b = next_i[t];
            if (a >= 0 && atomicSub(&indeg[a], 1) != 1) a = -1;
            if (b >= 0 && atomicSub(&indeg[b], 1) != 1) b = -1;
        }
        push(fr_out, cnt_out, a >= 0, a);
        push(fr_out, cnt_out, b >= 0, b);
    }
}

__global__ void k_gather_recs(const int32_t *__restrict__ order, const int2 *__restrict__ idx,
                              const double *__restrict__ r, int64_t nnz, Rec *recs)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = order[p];
        const int2 q = idx[t];
        Rec o;
        o.user = q.x;
        o.item = q.y;
        o.r = r[t];
        recs[p] = o;
    }
}

// First-fit recolouring of ONE level (= a set of ratings with pairwise distinct users and items): every
// rating moves to the smallest colour no earlier-processed rating of its user or item holds.  Levels are
// processed in order, so colour (level - 1) is always free: the scan ends inside the first `level` bits.
// No two threads touch the same bitset row, so the result is deterministic.
__global__ void k_compact_level(const int32_t *__restrict__ order, int64_t off, int count,
                                const int32_t *__restrict__ users, const int32_t *__restrict__ items,
                                unsigned long long *ubits, unsigned long long *ibits, int W, int32_t *colour)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int32_t t = order[off + e];
    unsigned long long *bu = ubits + (size_t)users[t] * W, *bi = ibits + (size_t)items[t] * W;
    for (int w = 0; w < W; ++w) {
        const unsigned long long a = bu[w], b = bi[w], m = ~(a | b);
        if (m) {
            const int bit = __ffsll((long long)m) - 1;
            bu[w] = a | (1ull << bit);
            bi[w] = b | (1ull << bit);
            colour[t] = w * 64 + bit + 1;                 // colours are 1-based like levels
            return;
        }
    }
    colour[t] = W * 64;                                   // unreachable (see above)
}

// sorted keys (1-based colours, no gaps) -> start offset of every colour class
__global__ void k_gather_i32(const int32_t *__restrict__ src, const int32_t *__restrict__ at, int64_t nnz, int32_t *out)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz;
         p += (int64_t)gridDim.x * blockDim.x)
        out[p] = src[at[p]];
}

__global__ void k_class_bounds(const int32_t *__restrict__ keys, int64_t nnz, unsigned long long *start)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz;
         p += (int64_t)gridDim.x * blockDim.x)
        if (p == 0 || keys[p - 1] != keys[p]) start[keys[p] - 1] = (unsigned long long)p;
}

// stable LSD radix sort of (key, value) pairs on bits [0, end_bit)
template <typename K, typename V>
hipError_t sort_pairs(void *tmp, size_t &tmp_bytes, const K *kin, K *kout, const V *vin, V *vout, int64_t n,
                      int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (unsigned int)n, 0u, (unsigned int)end_bit, st);
}

int bits_for(int64_t n)      // radix bits that cover keys 0 .. n-1
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n) ++b;
    return b;
}

struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int get(T **out, size_t n)
    {
        void *q = nullptr;
        if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(MFX_E_NOMEM, "device schedule: out of device memory (%zu bytes)", n * sizeof(T));
        }
        ptrs.push_back(q);
        *out = (T *)q;
        return MFX_OK;
    }
};

}  // namespace

// Returns MFX_OK and fills level_off (n_levels + 1 entries, level l+1 = [level_off[l], level_off[l+1]))
// and, when given, d_recs (device, nnz records, level-major) and h_level (host, nnz entries, 1-based).
// *over_budget = true (with MFX_OK) when the level count exceeded level_budget: nothing was produced.
// compact = true: the levels are first-fit recoloured (k_compact_level) -- an execution ORDER (colour-major,
// ascending item inside a colour) with close to max-degree classes instead of the ~2.3x max-degree
// levels the input order forces; level_off / d_recs / h_level then describe that order and h_order
// (nullable, nnz entries) receives it: h_order[j] = input index of the j-th rating.
int schedule_device(const int32_t *h_idx, const double *h_r, int64_t nnz, int nu, int ni,
                    Rec *d_recs, int32_t *h_level, std::vector<int64_t> &level_off,
                    int64_t *n_levels, int64_t level_budget, bool *over_budget, bool compact,
                    int64_t *h_order)
{
    mfx::TraceRange trace_("mfx:schedule_device (sorts + frontier levels)");
    *over_budget = false;
    *n_levels = 0;
    level_off.assign(1, 0);
    if (nnz == 0) return MFX_OK;
    if (nnz >= ((int64_t)1 << 31) - 1)
        return fail(MFX_E_ARG, "device schedule: nnz must be < 2^31 (use the host builder)");
    if (level_budget < 1) level_budget = 1;
    hipStream_t st = stream();
    Scratch S;
    int2 *d_idx = nullptr;
    double *d_r = nullptr;
    int32_t *users, *items, *tvals, *ka, *va, *next_u, *next_i, *indeg, *level, *fr2;
    unsigned long long *ctl, *sizes;
    const size_t n = (size_t)nnz;
    if (S.get(&d_idx, n) || S.get(&users, n) || S.get(&items, n) || S.get(&tvals, n) || S.get(&ka, n) ||
        S.get(&va, n) || S.get(&next_u, n) || S.get(&next_i, n) || S.get(&indeg, n) ||
        S.get(&level, n) || S.get(&fr2, n) || S.get(&ctl, 8) || S.get(&sizes, (size_t)level_budget + 1))
        return last_code();
    if (d_recs && S.get(&d_r, n)) return last_code();
    MFX_HIP(hipMemcpyAsync(d_idx, h_idx, n * sizeof(int2), hipMemcpyHostToDevice, st));
    if (d_recs) MFX_HIP(hipMemcpyAsync(d_r, h_r, n * sizeof(double), hipMemcpyHostToDevice, st));
    // ctl: [0..2] frontier counters, [3] total levelled, [4] first out-of-range rating
    MFX_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(unsigned long long), st));
    MFX_HIP(hipMemsetAsync(ctl + 4, 0xff, sizeof(unsigned long long), st));
    const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(nnz, kBfsThreads), 8192);
    k_split_check<<<blocks, kBfsThreads, 0, st>>>(d_idx, nnz, nu, ni, users, items, tvals, ctl + 4);
    unsigned long long bad = 0;
    MFX_HIP(hipMemcpyAsync(&bad, ctl + 4, sizeof bad, hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    if (bad != ~0ull) {
        const int32_t a = h_idx[2 * bad], b = h_idx[2 * bad + 1];
        return fail(MFX_E_RANGE, "rating %llu: (user %d, item %d) outside [0,%d) x [0,%d)", bad, a, b, nu, ni);
    }

    // ---- chains -------------------------------------------------------------------------------
    size_t tmp_bytes = 0, tb2 = 0;
    MFX_HIP(sort_pairs(nullptr, tmp_bytes, users, ka, tvals, va, nnz, 32, st));
    MFX_HIP(sort_pairs(nullptr, tb2, level, ka, tvals, va, nnz, 32, st));
    if (tb2 > tmp_bytes) tmp_bytes = tb2;
    char *tmp = nullptr;
    if (S.get(&tmp, tmp_bytes)) return last_code();
    MFX_HIP(sort_pairs(tmp, tmp_bytes, users, ka, tvals, va, nnz, bits_for(nu), st));
    k_link<<<blocks, kBfsThreads, 0, st>>>(ka, va, nnz, next_u, indeg, 1);
    MFX_HIP(sort_pairs(tmp, tmp_bytes, items, ka, tvals, va, nnz, bits_for(ni), st));
    k_link<<<blocks, kBfsThreads, 0, st>>>(ka, va, nnz, next_i, indeg, 0);

    // ---- levels (frontier buffers rotate over ka / va / fr2, like the counters) ----------------
    int32_t *fr[3] = {fr2, ka, va};          // level L reads fr[L%3], writes fr[(L+1)%3]
    k_seed<<<blocks, kBfsThreads, 0, st>>>(indeg, nnz, fr[1], ctl + 1);
    int64_t L = 1;
    unsigned long long total = 0;
    while (true) {
        int64_t batch = 128;
        if (L + batch > level_budget + 1) batch = level_budget + 1 - L;
        for (int64_t q = 0; q < batch; ++q, ++L)
            k_bfs_level<<<kBfsBlocks, kBfsThreads, 0, st>>>((int32_t)L, fr[L % 3], fr[(L + 1) % 3], ctl,
                                                             next_u, next_i, indeg, level, sizes, ctl + 3);
        MFX_HIP(hipMemcpyAsync(&total, ctl + 3, sizeof total, hipMemcpyDeviceToHost, st));
        MFX_HIP(hipStreamSynchronize(st));
        if ((int64_t)total >= nnz) break;
        if (L > level_budget) {
            *over_budget = true;
            return MFX_OK;
        }
    }
    std::vector<unsigned long long> hs((size_t)(L - 1));
    MFX_HIP(hipMemcpy(hs.data(), sizes, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int64_t nl = 0;
    while (nl < (int64_t)hs.size() && hs[(size_t)nl] > 0) ++nl;
    level_off.assign((size_t)nl + 1, 0);
    for (int64_t l = 0; l < nl; ++l) level_off[(size_t)l + 1] = level_off[(size_t)l] + (int64_t)hs[(size_t)l];
    if (level_off[(size_t)nl] != nnz)
        return fail(MFX_E_STATE, "device schedule: levelled %lld of %lld ratings", (long long)level_off[(size_t)nl],
                    (long long)nnz);
    if (compact && nl > 1) {
        const int W = (int)ceil_div<int64_t>(nl, 64);
        const size_t words = ((size_t)nu + (size_t)ni) * (size_t)W;
        if (words > ((size_t)1 << 29))
            return fail(MFX_E_NOMEM, "device schedule: colour bitsets of %zu MB", words >> 17);
        unsigned long long *bits = nullptr, *starts = nullptr;
        if (S.get(&bits, words) || S.get(&starts, (size_t)nl + 1)) return last_code();
        MFX_HIP(hipMemsetAsync(bits, 0, words * sizeof(unsigned long long), st));
        MFX_HIP(sort_pairs(tmp, tmp_bytes, level, ka, tvals, va, nnz, bits_for(nl + 1), st));
        int32_t *colour = next_u;           // the chain arrays are free now
        for (int64_t l = 0; l < nl; ++l) {
            const int cnt = (int)(level_off[(size_t)l + 1] - level_off[(size_t)l]);
            k_compact_level<<<ceil_div(cnt, 256), 256, 0, st>>>(va, level_off[(size_t)l], cnt, users, items, bits,
                                                                bits + (size_t)nu * W, W, colour);
        }
        // colour-major keys -> the class offsets
        MFX_HIP(sort_pairs(tmp, tmp_bytes, colour, ka, tvals, va, nnz, bits_for(nl + 1), st));
        MFX_HIP(hipMemsetAsync(starts, 0xff, ((size_t)nl + 1) * sizeof(unsigned long long), st));
        k_class_bounds<<<blocks, kBfsThreads, 0, st>>>(ka, nnz, starts);
        std::vector<unsigned long long> hb((size_t)nl);
        MFX_HIP(hipMemcpyAsync(hb.data(), starts, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        MFX_HIP(hipStreamSynchronize(st));
        int64_t nc = 0;
        while (nc < nl && hb[(size_t)nc] != ~0ull) ++nc;                     // first-fit leaves no gaps
        level_off.assign((size_t)nc + 1, 0);
        for (int64_t c = 0; c < nc; ++c) level_off[(size_t)c] = (int64_t)hb[(size_t)c];
        level_off[(size_t)nc] = nnz;
        for (int64_t c = nc; c < nl; ++c)
            if (hb[(size_t)c] != ~0ull) return fail(MFX_E_STATE, "device schedule: gap in the colour classes");
        nl = nc;
        level = colour;
    }
    *n_levels = nl;
    if (h_level) MFX_HIP(hipMemcpyAsync(h_level, level, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));

    // ---- level-major order and records -----------------------------------------------------------
    // Inside a level the ratings go by ASCENDING ITEM (an item occurs once per level, so this fixes the
    // order completely; the host path of mfx_sgd_plan_create does the same).  The ratings of a level are
    // independent, any order inside it is the same algorithm -- but neighbouring lanes that touch
    // neighbouring item rows make the row gathers DRAM-page friendly: +3.5 % (cfg 2) to +10 % (cfg 5 shard
    // shape) on k1_level against ascending input index, same bits (tools/level_sort_probe.py).
    // LSD: stable sort by item, then stable sort by level.
    if (d_recs || h_order) {
        int32_t *lv_in = next_i, *lv_out = indeg, *ord = fr2;      // chain / frontier arrays are free now
        MFX_HIP(sort_pairs(tmp, tmp_bytes, items, ka, tvals, va, nnz, bits_for(ni), st));
        k_gather_i32<<<blocks, kBfsThreads, 0, st>>>(level, va, nnz, lv_in);
        MFX_HIP(sort_pairs(tmp, tmp_bytes, lv_in, lv_out, va, ord, nnz, bits_for(nl + 1), st));
        if (d_recs) k_gather_recs<<<blocks, kBfsThreads, 0, st>>>(ord, d_idx, d_r, nnz, d_recs);
        if (h_order) {
            std::vector<int32_t> o32(n);
            MFX_HIP(hipMemcpyAsync(o32.data(), ord, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            MFX_HIP(hipStreamSynchronize(st));
            for (size_t j = 0; j < n; ++j) h_order[j] = o32[j];
        }
    }
    MFX_HIP(hipGetLastError());
    MFX_HIP(hipStreamSynchronize(st));
    return MFX_OK;
}

}  // namespace mfx

// ---------------------------------------------------------------------------------------------
// Records of the resident-side epoch (mfx_sgd_res.hip), built in HBM from the plan's level-major records.  The
// level-major order lists every user's ratings and every item's ratings in their input order (a chain's ratings
// have strictly increasing levels), so a STABLE sort by entity gives each entity's list in chain order, and a
// rating's rank in its travelling entity's chain is its position in that entity's sorted segment.
namespace mfx {
namespace {
__global__ void k_res_keys(const Rec *__restrict__ recs, int64_t nnz, int res_item, int32_t *kres, int32_t *ktrav, int32_t *pos)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const Rec q = recs[p];
        kres[p] = res_item ? q.item : q.user;
        ktrav[p] = res_item ? q.user : q.item;
        pos[p] = (int32_t)p;
    }
}
__global__ void k_res_first(const int32_t *__restrict__ ks, int64_t nnz, int32_t *first)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x)
        if (s == 0 || ks[s] != ks[s - 1]) first[ks[s]] = (int32_t)s;
}
__global__ void k_res_seq(const int32_t *__restrict__ ks, const int32_t *__restrict__ ps, int64_t nnz,
                          const int32_t *__restrict__ first, int32_t *seq)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x)
        seq[ps[s]] = (int32_t)s - first[ks[s]];
}
__global__ void k_res_pack(const int32_t *__restrict__ ks, const int32_t *__restrict__ ps, int64_t nnz,
                           const Rec *__restrict__ recs, const int32_t *__restrict__ ktrav, const int32_t *__restrict__ seq,
                           ResRec *out, int32_t *lo, int32_t *hi)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = ps[s], e = ks[s];
        ResRec o;
        o.trav = ktrav[p]; o.seq = seq[p]; o.r = recs[p].r;
        out[s] = o;
        if (s == 0 || ks[s - 1] != e) lo[e] = (int32_t)s;
        if (s == nnz - 1 || ks[s + 1] != e) hi[e] = (int32_t)s + 1;
    }
}
// links of the travelling chains in GROUPED positions (the Funk pass hands V[f,user] from a rating straight to the user's next one)
__global__ void k_res_succ(const int32_t *__restrict__ ks, const int32_t *__restrict__ ps, int64_t nnz, int32_t *succ)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x)
        succ[ps[s]] = (s + 1 < nnz && ks[s + 1] == ks[s]) ? ps[s + 1] : -1;
}
__global__ void k_res_firstp(const int32_t *__restrict__ ks, const int32_t *__restrict__ ps, int64_t nnz, int32_t *firstp)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x)
        if (s == 0 || ks[s] != ks[s - 1]) firstp[ks[s]] = ps[s];
}
__global__ void k_res_gpos(const int32_t *__restrict__ ps, int64_t nnz, int32_t *gpos)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) gpos[ps[s]] = (int32_t)s;
}
__global__ void k_res_next(const int32_t *__restrict__ ps, const int32_t *__restrict__ succ, const int32_t *__restrict__ gpos, int64_t nnz,
                           int32_t *next)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t q = succ[ps[s]];
        next[s] = q >= 0 ? gpos[q] : -1;
    }
}
__global__ void k_res_first_g(int32_t *firstp, const int32_t *__restrict__ gpos, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int32_t q = firstp[i]; firstp[i] = q >= 0 ? gpos[q] : -1; }
}
}  // namespace

namespace {
__global__ void k_deg_count(const Rec *__restrict__ recs, int64_t nnz, int *cu, int *ci)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const Rec q = recs[p];
        atomicAdd(cu + q.user, 1);
        atomicAdd(ci + q.item, 1);
    }
}
__global__ void k_deg_max(const int *__restrict__ c, int n, int *out)
{
    int m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = c[i] > m ? c[i] : m;
    atomicMax(out, m);
}
}  // namespace

// largest user / item degree (how skewed the data is: the resident-side epoch's cost model)
int res_max_degrees(const Rec *d_recs, int64_t nnz, int nu, int ni, int *max_user_deg, int *max_item_deg)
{
    hipStream_t st = stream();
    Scratch S;
    int *cu, *ci, *mx;
    if (S.get(&cu, (size_t)nu) || S.get(&ci, (size_t)ni) || S.get(&mx, 2)) return last_code();
    MFX_HIP(hipMemsetAsync(cu, 0, (size_t)nu * sizeof(int), st));
    MFX_HIP(hipMemsetAsync(ci, 0, (size_t)ni * sizeof(int), st));
    MFX_HIP(hipMemsetAsync(mx, 0, 2 * sizeof(int), st));
    const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(nnz, 256), 8192);
    k_deg_count<<<blocks, 256, 0, st>>>(d_recs, nnz, cu, ci);
    k_deg_max<<<256, 256, 0, st>>>(cu, nu, mx);
    k_deg_max<<<256, 256, 0, st>>>(ci, ni, mx + 1);
    int h[2] = {0, 0};
    MFX_HIP(hipMemcpyAsync(h, mx, sizeof h, hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    *max_user_deg = h[0];
    *max_item_deg = h[1];
    return MFX_OK;
}

// d_next (optional, [nnz]): grouped position of the NEXT rating of the same travelling entity, -1 at the end of its chain;
// d_first_pos (optional, [n_trav]): grouped position of its first rating, -1 for an entity without ratings
int res_build_device(const Rec *d_recs, int64_t nnz, int n_res, int n_trav, bool res_item, ResRec *d_out, int32_t *d_lo, int32_t *d_hi,
                     int32_t *d_pos_of, int32_t *d_ent_of, int32_t *d_next, int32_t *d_first_pos)
{
    if (nnz <= 0 || nnz >= ((int64_t)1 << 31) - 1) return fail(MFX_E_ARG, "res_build_device: unsupported size");
    hipStream_t st = stream();
    Scratch S;
    const size_t n = (size_t)nnz;
    int32_t *kres, *ktrav, *pos, *ks, *ps, *seq, *first;
    if (S.get(&kres, n) || S.get(&ktrav, n) || S.get(&pos, n) || S.get(&ks, n) || S.get(&ps, n) || S.get(&seq, n) ||
        S.get(&first, (size_t)n_trav))
        return last_code();
    size_t tmp_bytes = 0;
    MFX_HIP(sort_pairs(nullptr, tmp_bytes, kres, ks, pos, ps, nnz, 32, st));
    char *tmp = nullptr;
    if (S.get(&tmp, tmp_bytes)) return last_code();
    const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(nnz, 256), 8192);
    k_res_keys<<<blocks, 256, 0, st>>>(d_recs, nnz, res_item ? 1 : 0, kres, ktrav, pos);
    MFX_HIP(sort_pairs(tmp, tmp_bytes, ktrav, ks, pos, ps, nnz, bits_for(n_trav), st));
    k_res_first<<<blocks, 256, 0, st>>>(ks, nnz, first);
    k_res_seq<<<blocks, 256, 0, st>>>(ks, ps, nnz, first, seq);
    int32_t *succ = nullptr, *gpos = nullptr;
    if (d_next) {
        if (S.get(&succ, n) || S.get(&gpos, n)) return last_code();
        k_res_succ<<<blocks, 256, 0, st>>>(ks, ps, nnz, succ);
        if (d_first_pos) {
            MFX_HIP(hipMemsetAsync(d_first_pos, 0xff, (size_t)n_trav * sizeof(int32_t), st));
            k_res_firstp<<<blocks, 256, 0, st>>>(ks, ps, nnz, d_first_pos);
        }
    }
    MFX_HIP(sort_pairs(tmp, tmp_bytes, kres, ks, pos, ps, nnz, bits_for(n_res), st));
    if (d_next) {
        k_res_gpos<<<blocks, 256, 0, st>>>(ps, nnz, gpos);
        k_res_next<<<blocks, 256, 0, st>>>(ps, succ, gpos, nnz, d_next);
        if (d_first_pos) k_res_first_g<<<ceil_div(n_trav, 256), 256, 0, st>>>(d_first_pos, gpos, n_trav);
    }
    MFX_HIP(hipMemsetAsync(d_lo, 0, (size_t)n_res * sizeof(int32_t), st));
    MFX_HIP(hipMemsetAsync(d_hi, 0, (size_t)n_res * sizeof(int32_t), st));
    k_res_pack<<<blocks, 256, 0, st>>>(ks, ps, nnz, d_recs, ktrav, seq, d_out, d_lo, d_hi);
    // (position of every grouped record in the level-major array: the Funk pass keeps its residual cache in grouped order)
    if (d_pos_of) MFX_HIP(hipMemcpyAsync(d_pos_of, ps, n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (d_ent_of) MFX_HIP(hipMemcpyAsync(d_ent_of, ks, n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));     // resident entity of every grouped record
    MFX_HIP(hipGetLastError());
    MFX_HIP(hipStreamSynchronize(st));
    return MFX_OK;
}
}  // namespace mfx

// ---------------------------------------------------------------------------------------------
// Re-order the ratings INSIDE every dependency level by (user group, late / early, item) and return the
// segment tables: seg_off[l * n_groups + g] = first position of (level l, group g), seg_mid[...] = first
// EARLY position of that segment, seg_off[n_levels * n_groups] = nnz.  A rating is LATE when its item was
// updated in the level just before its own (it can only run once that level is complete everywhere) and
// EARLY otherwise (its item value has been final for a whole level already).  The order inside a level is
// free (its ratings share no user and no item), so every kernel keeps its exact result; the XCD-resident
// Funk pass (mfx_sgd.hip: k2_xcd_pass) wants it this way: a workgroup owns a user group (its slice of
// V[f,:] lives in LDS), runs the late ratings of a level right after the barrier and the early ratings of
// the NEXT level while the barrier is in flight.  User groups are equal contiguous id ranges:
// group(u) = u * n_groups / nu.
namespace mfx {
namespace {
// one level of the late flags (host loop over the levels: they must be walked in order; inside a level
// every item occurs once)
__global__ __launch_bounds__(256) void k_late_level(const Rec *__restrict__ recs, int64_t start, int count, int32_t level,
                                                    int32_t *last_level, uint8_t *late)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int item = recs[start + t].item;
    late[start + t] = (level > 0 && last_level[item] == level - 1) ? 1 : 0;
    last_level[item] = level;
}
__global__ void k_group_keys(const Rec *__restrict__ recs, const uint8_t *__restrict__ late, int64_t nnz,
                             const int64_t *__restrict__ level_off, int n_levels, int nu, int n_groups,
                             unsigned long long *keys)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_levels;                       // level_off[lo] <= p < level_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (level_off[mid] <= p) lo = mid; else hi = mid;
        }
        const Rec q = recs[p];
        const unsigned long long g = (unsigned long long)((int64_t)q.user * n_groups / nu);
        const unsigned long long early = late[p] ? 0ull : 1ull;
        keys[p] = ((unsigned long long)lo << 37) | (g << 32) | (early << 31) | (unsigned long long)(unsigned)q.item;
    }
}
__global__ void k_seg_bounds(const unsigned long long *__restrict__ keys, int64_t nnz, int n_levels, int n_groups,
                             int32_t *seg_off, int32_t *seg_mid)
{
    const int64_t nseg = (int64_t)n_levels * n_groups;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q <= nseg; q += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long base = ((unsigned long long)(q / n_groups) << 37) | ((unsigned long long)(q % n_groups) << 32);
        for (int half = 0; half < 2; ++half) {
            const unsigned long long want = base | ((unsigned long long)half << 31);
            int64_t lo = 0, hi = nnz;                    // first position with key >= want
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] < want) lo = mid + 1; else hi = mid;
            }
            if (half == 0) seg_off[q] = (int32_t)lo;
            else if (q < nseg) seg_mid[q] = (int32_t)lo;
        }
    }
}
}  // namespace

int levels_group_sort(Rec *d_recs, int64_t nnz, const std::vector<int64_t> &level_off, const int64_t *d_level_off,
                      int64_t n_levels, int nu, int ni, int n_groups, int32_t *d_seg_off, int32_t *d_seg_mid)
{
    mfx::TraceRange trace_("mfx:levels_group_sort");
    if (nnz <= 0 || nnz >= ((int64_t)1 << 31) - 1 || n_levels >= ((int64_t)1 << 24) || n_groups < 1 || n_groups > 32)
        return fail(MFX_E_ARG, "levels_group_sort: unsupported size");
    hipStream_t st = stream();
    Scratch S;
    unsigned long long *k0, *k1;
    Rec *r1;
    uint8_t *late;
    int32_t *last;
    const size_t n = (size_t)nnz;
    if (S.get(&k0, n) || S.get(&k1, n) || S.get(&r1, n) || S.get(&late, n) || S.get(&last, (size_t)ni)) return last_code();
    MFX_HIP(hipMemsetAsync(last, 0xff, (size_t)ni * sizeof(int32_t), st));       // -1
    for (int64_t l = 0; l < n_levels; ++l) {
        const int cnt = (int)(level_off[(size_t)l + 1] - level_off[(size_t)l]);
        if (cnt > 0) k_late_level<<<ceil_div(cnt, 256), 256, 0, st>>>(d_recs, level_off[(size_t)l], cnt, (int32_t)l, last, late);
    }
    const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(nnz, 256), 8192);
    k_group_keys<<<blocks, 256, 0, st>>>(d_recs, late, nnz, d_level_off, (int)n_levels, nu, n_groups, k0);
    size_t tmp_bytes = 0;
    const int key_bits = 37 + bits_for(n_levels + 1);
    MFX_HIP(sort_pairs(nullptr, tmp_bytes, k0, k1, d_recs, r1, nnz, key_bits, st));
    char *tmp = nullptr;
    if (S.get(&tmp, tmp_bytes)) return last_code();
    MFX_HIP(sort_pairs(tmp, tmp_bytes, k0, k1, d_recs, r1, nnz, key_bits, st));
    MFX_HIP(hipMemcpyAsync(d_recs, r1, n * sizeof(Rec), hipMemcpyDeviceToDevice, st));
    k_seg_bounds<<<(int)std::min<int64_t>(ceil_div<int64_t>(n_levels * n_groups + 1, 256), 4096), 256, 0, st>>>(
        k1, nnz, (int)n_levels, n_groups, d_seg_off, d_seg_mid);
    MFX_HIP(hipGetLastError());
    MFX_HIP(hipStreamSynchronize(st));
    return MFX_OK;
}
}  // namespace mfx

// include/mfx.h: the GPU counterpart of mfx_schedule_levels (same arguments, same result)
extern "C" int mfx_schedule_levels_gpu(const int32_t *idx, int64_t nnz, int nu, int ni,
                                       int32_t *level_out, int64_t *n_levels_out)
{
    if (!idx || !level_out || nnz < 0 || nu <= 0 || ni <= 0)
        return mfx::fail(MFX_E_ARG, "mfx_schedule_levels_gpu: bad argument");
    if (mfx::ensure_init()) return mfx::last_code();
    std::vector<int64_t> off;
    int64_t nl = 0;
    bool over = false;
    try {
        const int64_t cap = (int64_t)1 << 20;
        const int rc = mfx::schedule_device(idx, nullptr, nnz, nu, ni, nullptr, level_out, off, &nl,
                                            nnz < cap ? (nnz > 0 ? nnz : 1) : cap, &over, false, nullptr);
        if (rc) return rc;
        if (over) return mfx::fail(MFX_E_STATE, "mfx_schedule_levels_gpu: more than %lld levels; use mfx_schedule_levels", (long long)cap);
    } catch (const std::bad_alloc &) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_levels_gpu: out of host memory");
    }
    if (n_levels_out) *n_levels_out = nl;
    return MFX_OK;
}

// include/mfx.h: an input order with close to the minimum number of dependency levels, built on the GPU
extern "C" int mfx_schedule_balanced_order_gpu(const int32_t *idx, int64_t nnz, int nu, int ni,
                                               int64_t *perm_out, int64_t *n_rounds_out)
{
    if (!idx || !perm_out || nnz < 0 || nu <= 0 || ni <= 0)
        return mfx::fail(MFX_E_ARG, "mfx_schedule_balanced_order_gpu: bad argument");
    if (mfx::ensure_init()) return mfx::last_code();
    std::vector<int64_t> off;
    int64_t nl = 0;
    bool over = false;
    try {
        const int64_t cap = (int64_t)1 << 20;
        const int rc = mfx::schedule_device(idx, nullptr, nnz, nu, ni, nullptr, nullptr, off, &nl,
                                            nnz < cap ? (nnz > 0 ? nnz : 1) : cap, &over, true, perm_out);
        if (rc) return rc;
        if (over) return mfx::fail(MFX_E_STATE, "mfx_schedule_balanced_order_gpu: more than %lld levels; use mfx_schedule_balanced_order", (long long)cap);
    } catch (const std::bad_alloc &) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_balanced_order_gpu: out of host memory");
    }
    if (n_rounds_out) *n_rounds_out = nl;
    return MFX_OK;
}
