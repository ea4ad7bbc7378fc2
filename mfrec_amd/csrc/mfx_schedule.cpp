// Host-side dependency-level schedule (include/mfx.h "host-side scheduling").
//
// The reference applies ratings strictly in array order (kmf_train.pyx:243,
// gd_estimator.pyx:745).  Two updates commute bit-for-bit iff they share
// neither user nor item, so the rating sequence is a DAG whose longest-path
// layering ("levels") is a conflict-free parallel schedule that reproduces the
// sequential result exactly ("Hard parts").
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "mfx_common.h"

extern "C" int mfx_schedule_levels(const int32_t *idx, int64_t nnz, int nu, int ni,
                                   int32_t *level_out, int64_t *n_levels_out)
{
    if (!idx || !level_out || nnz < 0 || nu < 0 || ni < 0) {
        return mfx::fail(MFX_E_ARG, "mfx_schedule_levels: bad argument");
    }
    std::vector<int32_t> lu, li;
    try {
        lu.assign((size_t)nu, 0);
        li.assign((size_t)ni, 0);
    } catch (...) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_levels: out of host memory");
    }
    int32_t top = 0;
    for (int64_t t = 0; t < nnz; ++t) {
        const int32_t a = idx[2 * t], b = idx[2 * t + 1];
        if (a < 0 || a >= nu || b < 0 || b >= ni) {
            return mfx::fail(MFX_E_RANGE, "rating %lld: (user %d, item %d) outside [0,%d) x [0,%d)",
                           (long long)t, a, b, nu, ni);
        }
        const int32_t L = 1 + (lu[a] > li[b] ? lu[a] : li[b]);
        level_out[t] = L;
        lu[a] = L;
        li[b] = L;
        if (L > top) top = L;
    }
    if (n_levels_out) *n_levels_out = top;
    return MFX_OK;
}

// Stable counting sort by level: perm[j] = original position of the j-th rating in
// level-major order; level_off[l] .. level_off[l+1] is level l+1's slice.
extern "C" int mfx_schedule_permutation(const int32_t *level, int64_t nnz, int64_t n_levels,
                                        int64_t *perm_out, int64_t *level_off_out)
{
    if (!level || !perm_out || !level_off_out || nnz < 0 || n_levels < 0) {
        return mfx::fail(MFX_E_ARG, "mfx_schedule_permutation: bad argument");
    }
    for (int64_t l = 0; l <= n_levels; ++l) level_off_out[l] = 0;
    for (int64_t t = 0; t < nnz; ++t) {
        const int32_t L = level[t];
        if (L < 1 || L > n_levels) {
            return mfx::fail(MFX_E_RANGE, "mfx_schedule_permutation: level %d outside [1,%lld]", L,
                           (long long)n_levels);
        }
        ++level_off_out[L];
    }
    for (int64_t l = 0; l < n_levels; ++l) level_off_out[l + 1] += level_off_out[l];
    // level_off_out[l] now = end of level l (1-based) = start of level l+1
    std::vector<int64_t> cur;
    try {
        cur.assign(level_off_out, level_off_out + n_levels + 1);
    } catch (...) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_permutation: out of host memory");
    }
    for (int64_t t = 0; t < nnz; ++t) perm_out[cur[level[t] - 1]++] = t;
    return MFX_OK;
}

// Stable counting sort by level straight into 16-byte records {int32 user, int32 item, f64 r}
// (the device layout), on up to 16 host threads: per-thread histograms over contiguous slices
// of the rating sequence, then every thread scatters its slice.  No permutation array, no gather.
// Output order == mfx_schedule_permutation's order.
namespace {
struct PackRec {
    int32_t user, item;
    double r;
};
}  // namespace

extern "C" int mfx_schedule_pack(const int32_t *idx, const double *r, const int32_t *level,
                                 int64_t nnz, int64_t n_levels, void *recs_out,
                                 int64_t *level_off_out)
{
    if (!level_off_out || nnz < 0 || n_levels < 0 || (nnz > 0 && (!idx || !r || !level || !recs_out)))
        return mfx::fail(MFX_E_ARG, "mfx_schedule_pack: bad argument");
    PackRec *recs = (PackRec *)recs_out;
    for (int64_t t = 0; t < nnz; ++t)
        if (level[t] < 1 || level[t] > n_levels)
            return mfx::fail(MFX_E_RANGE, "mfx_schedule_pack: level %d outside [1,%lld]", level[t],
                             (long long)n_levels);
    try {
        unsigned hw = std::thread::hardware_concurrency();
        int T = (int)(hw ? hw : 1);
        if (T > 16) T = 16;
        if (nnz < (1 << 20)) T = 1;
        const size_t NL = (size_t)n_levels + 1;
        while (T > 1 && (size_t)T * NL > ((size_t)1 << 26)) T /= 2;
        std::vector<int64_t> hist((size_t)T * NL, 0);
        auto slice = [&](int i, int64_t &a, int64_t &b) { a = nnz * i / T; b = nnz * (i + 1) / T; };
        auto run = [&](auto &&fn) {
            if (T == 1) { fn(0); return; }
            std::vector<std::thread> th;
            for (int i = 0; i < T; ++i) th.emplace_back(fn, i);
            for (auto &x : th) x.join();
        };
        run([&](int i) {
            int64_t a, b; slice(i, a, b);
            int64_t *h = hist.data() + (size_t)i * NL;
            for (int64_t t = a; t < b; ++t) ++h[level[t]];
        });
        // level_off[l] = start of level l+1 (levels are 1-based); one cursor per (thread, level)
        int64_t acc = 0;
        for (size_t L = 1; L < NL; ++L) {
            level_off_out[L - 1] = acc;
            for (int i = 0; i < T; ++i) {
                const int64_t c = hist[(size_t)i * NL + L];
                hist[(size_t)i * NL + L] = acc;
                acc += c;
            }
        }
        level_off_out[NL - 1] = acc;
        run([&](int i) {
            int64_t a, b; slice(i, a, b);
            int64_t *cur = hist.data() + (size_t)i * NL;
            for (int64_t t = a; t < b; ++t) {
                PackRec &q = recs[cur[level[t]]++];
                q.user = idx[2 * t];
                q.item = idx[2 * t + 1];
                q.r = r[t];
            }
        });
    } catch (...) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_pack: out of host memory / threads");
    }
    return MFX_OK;
}

// A rating ORDER with (close to) the minimum number of dependency levels.  The reference shuffles
// the ratings with an unseeded RNG before every train() (base.py:1127-1129), so any permutation
// is a legitimate input order; this one is a first-fit edge colouring of the bipartite
// user-item multigraph taken over a seeded random visiting order: every rating gets the smallest
// round in which neither its user nor its item is already busy, and the order is round-major.
// Rounds ~ max degree + a few (the level schedule of an i.i.d. shuffle needs ~2.3x max degree).
extern "C" int mfx_schedule_balanced_order(const int32_t *idx, int64_t nnz, int nu, int ni,
                                           uint64_t seed, int64_t *perm_out, int64_t *n_rounds_out)
{
    if (nnz < 0 || nu < 0 || ni < 0 || (nnz > 0 && (!idx || !perm_out)))
        return mfx::fail(MFX_E_ARG, "mfx_schedule_balanced_order: bad argument");
    try {
        std::vector<int32_t> du((size_t)nu, 0), di((size_t)ni, 0);
        for (int64_t t = 0; t < nnz; ++t) {
            const int32_t a = idx[2 * t], b = idx[2 * t + 1];
            if (a < 0 || a >= nu || b < 0 || b >= ni)
                return mfx::fail(MFX_E_RANGE, "rating %lld: (user %d, item %d) outside [0,%d) x [0,%d)",
                                 (long long)t, a, b, nu, ni);
            ++du[a];
            ++di[b];
        }
        // seeded Fisher-Yates visiting order (xorshift64*)
        std::vector<int64_t> visit((size_t)nnz);
        for (int64_t t = 0; t < nnz; ++t) visit[t] = t;
        uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        auto rnd = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; };
        for (int64_t t = nnz - 1; t > 0; --t) {
            const int64_t j = (int64_t)(rnd() % (uint64_t)(t + 1));
            std::swap(visit[t], visit[j]);
        }
        // busy-round bitsets per user / item.  First fit never needs a round >= deg(user) + deg(item),
        // so a node's bitset covers its own degree + the largest degree of the other side.  On very
        // skewed data that is too much memory: refuse (the caller then keeps a plain shuffle).
        int64_t mdu = 0, mdi = 0;
        for (int a = 0; a < nu; ++a) mdu = du[a] > mdu ? du[a] : mdu;
        for (int b = 0; b < ni; ++b) mdi = di[b] > mdi ? di[b] : mdi;
        std::vector<int64_t> ou((size_t)nu + 1, 0), oi((size_t)ni + 1, 0);
        for (int a = 0; a < nu; ++a) ou[a + 1] = ou[a] + (du[a] ? ((int64_t)du[a] + mdi + 64) / 64 : 0);
        for (int b = 0; b < ni; ++b) oi[b + 1] = oi[b] + (di[b] ? ((int64_t)di[b] + mdu + 64) / 64 : 0);
        if ((ou[nu] + oi[ni]) * 8 > ((int64_t)1 << 30))
            return mfx::fail(MFX_E_STATE, "mfx_schedule_balanced_order: degrees too skewed (bitsets would need "
                             "%lld MB); use a plain shuffle", (long long)((ou[nu] + oi[ni]) * 8 >> 20));
        std::vector<uint64_t> bu((size_t)ou[nu], 0), bi((size_t)oi[ni], 0);
        std::vector<int32_t> round((size_t)nnz);
        int64_t top = 0;
        for (int64_t v = 0; v < nnz; ++v) {
            const int64_t t = visit[v];
            const int32_t a = idx[2 * t], b = idx[2 * t + 1];
            uint64_t *wa = bu.data() + ou[a], *wb = bi.data() + oi[b];
            const int64_t na = ou[a + 1] - ou[a], nb = oi[b + 1] - oi[b];
            int64_t c = -1;
            for (int64_t w = 0;; ++w) {
                const uint64_t ma = w < na ? wa[w] : 0, mb = w < nb ? wb[w] : 0;
                const uint64_t freebits = ~(ma | mb);
                if (freebits) { c = w * 64 + __builtin_ctzll(freebits); break; }
            }
            if (c / 64 >= na || c / 64 >= nb)
                return mfx::fail(MFX_E_STATE, "mfx_schedule_balanced_order: internal bitset overflow");
            wa[c / 64] |= 1ull << (c % 64);
            wb[c / 64] |= 1ull << (c % 64);
            round[t] = (int32_t)c + 1;
            if (c + 1 > top) top = c + 1;
        }
        // round-major, visiting order inside a round (already random)
        std::vector<int64_t> off((size_t)top + 2, 0);
        for (int64_t t = 0; t < nnz; ++t) ++off[round[t] + 1];
        for (int64_t l = 1; l <= top + 1; ++l) off[l] += off[l - 1];
        for (int64_t v = 0; v < nnz; ++v) {
            const int64_t t = visit[v];
            perm_out[off[round[t]]++] = t;
        }
        if (n_rounds_out) *n_rounds_out = top;
    } catch (...) {
        return mfx::fail(MFX_E_NOMEM, "mfx_schedule_balanced_order: out of host memory");
    }
    return MFX_OK;
}
