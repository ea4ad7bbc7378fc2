// Exact multi-GPU SGD: stratified (DSGD) block rotation (include/mfx.h "exact multi-GPU mode").
//
// Sequential SGD is order dependent, and the user-sharded averaging scheme of mfx_comm.hip (the
// reference's own two-"machine" prototype, gradient_descent.py:124-211) changes the trajectory.  This
// mode does not: users are cut into G blocks (one per GPU) and items into B blocks, B = G or B = 2 G.
// In step h (0 <= h < B) GPU g trains the ratings of block (g, (stride * g + h) mod B), stride = B / G
// -- the G blocks of a step share neither a user nor an item, so running them at the same time IS
// running them one after the other -- and then hands that item block (rows + biases) to GPU g - 1, which
// needs it `stride` steps later.  With B = 2 G every block rests one whole step in transit: the
// ncclSend/ncclRecv pair of a step rides a second HIP stream under the next step's levels (events order
// the two streams per block); with B = G the hand-off sits between two steps.  After B steps every
// rating has been applied once and every item block is home again.  The result is bit-identical to the
// reference's sequential loop (kmf_train.pyx:243-270) run on the rating order
//     for h in 0..B-1: for g in 0..G-1: ratings of block (g, (stride * g + h) mod B), in block order
// where "block order" is the given order, or (MFX_DSGD_BALANCED) the block's first-fit recolouring with
// the fewest dependency levels -- the reference draws its order from an unseeded RNG (base.py:1127-1129), i.e. leaves it open;
// how this structured order trains against uniform draws is measured in profiles/r04_block_order_quality.json.
// mfx_dsgd_get_order returns that order for the tests to feed the oracle with.
//
// One DSGD rank = B sub-plans (one per item block) that share the rank's user rows.  rank = -1 keeps
// all G ranks in this process on this GPU ("fake world": the hand-off is a device copy on the second
// stream, same events); it runs the same code path except for the transport and is what the single-GPU
// tests drive.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "mfx_sgd_internal.h"

using namespace mfx;

namespace {

struct DsgdRank {
    int g = 0;                              // this rank's user block
    int user_lo = 0, nu_local = 0;
    std::vector<mfx_sgd_plan *> plans;      // [B]: block (g, j); plans[0] owns the user-side buffers
    std::vector<hipEvent_t> trained;        // [B]: recorded on the compute stream after block j's levels
    std::vector<hipEvent_t> arrived;        // [B]: recorded on the transfer stream after block j has landed
    std::vector<char> in_flight;            // [B]: block j has a pending arrival to wait for
    std::vector<std::vector<int64_t>> orig; // [B]: original rating index of every position of block j
};

int block_of(const std::vector<int64_t> &bounds, int64_t x)
{
    // bounds is ascending, bounds[0] = 0, bounds.back() = n; blocks may be empty
    int lo = 0, hi = (int)bounds.size() - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (x >= bounds[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace

struct mfx_dsgd {
    int world = 1, rank = 0;                // rank < 0: fake world
    int B = 1, stride = 1;                  // item blocks, B / world
    int nu = 0, ni = 0, k = 0;
    std::vector<int64_t> ub, ib;            // user [world + 1] / item [B + 1] block bounds
    std::vector<DsgdRank> ranks;            // 1 (a real rank) or world (fake world)
    bool have_factors = false;
    int64_t nnz_local = 0;
    std::vector<double *> avg_startU, avg_startib;   // averaging mode on the block order: the epoch's start of every item block
    bool avg_fresh = false;                          // start buffers hold the current rows
    hipStream_t xfer = nullptr;             // transfer stream
    // recovery of a run whose static block epochs lost their co-residency check (the launch modifies nothing then, but the OTHER blocks
    // of the run went on): the factors as they stood when the run call began, on the device
    std::vector<double *> snapV, snapub;                 // [rank of this process]
    std::vector<std::vector<double *>> snapU, snapib;    // [rank of this process][item block]
    int guard = -1;                         // -1: not decided; 1: some block of the JOB runs a resident-side epoch -> snapshot per run call
    int guard_collective = -1;              // the kind of call (inside a communicator or not) the decision was taken for
};


// ---- recovery: a static block epoch that finds its grid not co-resident modifies nothing and marks its plan (sticky: the plan's later
// launches of the run do nothing either) -- but the other blocks of the run go on, so the run as a whole is void.  Every run call
// therefore starts from a device snapshot of the factors (a few hundred MB of device copies against tens of milliseconds of epochs)
// while any block of the JOB is on a resident-side epoch; when the ranks agree that one of them lost a check, every rank restores its
// snapshot, puts ALL its blocks on the level schedule for good, and runs the call again.  Same process, fresh launches, no re-exec.
static bool any_resident(const mfx_dsgd *d)
{
    for (const DsgdRank &R : d->ranks)
        for (mfx_sgd_plan *p : R.plans)
            if (plan_epoch_kernel(p) != MFX_EPOCH_LEVELS) return true;
    return false;
}
// `collective`: the call is one that every rank of the communicator makes together -- the block rotation over real ranks, the averaging
// mode inside a communicator -- so the ranks can (and must) agree; a one-process object decides for itself
// decided once (among real ranks: every rank asks in its first run call), cleared for good by a recovery
// (decided again whenever the KIND of call changes -- a first run outside a communicator, later ones inside one: ranks that decided
//  alone could disagree on whether the verdict's agreement step runs at all)
static int dsgd_guard(mfx_dsgd *d, bool collective)
{
    if (d->guard >= 0 && d->guard_collective == (collective ? 1 : 0)) return MFX_OK;
    if (d->guard == 0 && !collective) { d->guard_collective = 0; return MFX_OK; }     // recovered for good: nothing resident is left
    int g = any_resident(d) ? 1 : 0;
    if (collective) {
        int rc = comm_agree_max(&g);
        if (rc) return rc;
    }
    d->guard = g;
    d->guard_collective = collective ? 1 : 0;
    return MFX_OK;
}
static int dsgd_copy_state(mfx_dsgd *d, bool save)
{
    hipStream_t s = stream();
    if (save && d->snapV.empty()) {
        d->snapV.assign(d->ranks.size(), nullptr);
        d->snapub.assign(d->ranks.size(), nullptr);
        d->snapU.assign(d->ranks.size(), std::vector<double *>((size_t)d->B, nullptr));
        d->snapib.assign(d->ranks.size(), std::vector<double *>((size_t)d->B, nullptr));
    }
    if (!save && d->snapV.empty()) return fail(MFX_E_STATE, "mfx_dsgd: no snapshot to restore");
    for (size_t gi = 0; gi < d->ranks.size(); ++gi) {
        DsgdRank &R = d->ranks[gi];
        mfx_sgd_plan *p0 = R.plans[0];
        const size_t nV = (size_t)plan_nu(p0) * d->k, nb = (size_t)plan_nu(p0);
        if (save && !d->snapV[gi]) {
            MFX_HIP(hipMalloc((void **)&d->snapV[gi], nV * sizeof(double)));
            MFX_HIP(hipMalloc((void **)&d->snapub[gi], nb * sizeof(double)));
        }
        MFX_HIP(hipMemcpyAsync(save ? d->snapV[gi] : plan_V(p0), save ? plan_V(p0) : d->snapV[gi], nV * sizeof(double), hipMemcpyDeviceToDevice, s));
        MFX_HIP(hipMemcpyAsync(save ? d->snapub[gi] : plan_ub(p0), save ? plan_ub(p0) : d->snapub[gi], nb * sizeof(double), hipMemcpyDeviceToDevice, s));
        for (int j = 0; j < d->B; ++j) {
            mfx_sgd_plan *p = R.plans[j];
            const size_t nU = (size_t)plan_ni(p) * d->k, ni = (size_t)plan_ni(p);
            if (save && !d->snapU[gi][j]) {
                MFX_HIP(hipMalloc((void **)&d->snapU[gi][j], nU * sizeof(double)));
                MFX_HIP(hipMalloc((void **)&d->snapib[gi][j], ni * sizeof(double)));
            }
            MFX_HIP(hipMemcpyAsync(save ? d->snapU[gi][j] : plan_U(p), save ? plan_U(p) : d->snapU[gi][j], nU * sizeof(double), hipMemcpyDeviceToDevice, s));
            MFX_HIP(hipMemcpyAsync(save ? d->snapib[gi][j] : plan_ib(p), save ? plan_ib(p) : d->snapib[gi][j], ni * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
    }
    return MFX_OK;
}
// the snapshot step allocates on first use: a rank that cannot must not leave its peers alone in the collectives of the body -- the
// ranks agree on the outcome first, and every rank takes the error path together
static int dsgd_snapshot(mfx_dsgd *d, bool collective, const char *what)
{
    if (!d->guard) return MFX_OK;
    const int rc_local = dsgd_copy_state(d, true);
    if (!collective) return rc_local;
    int bad = rc_local ? 1 : 0;
    const int rc = comm_agree_max(&bad);
    if (rc) return rc;
    if (rc_local) return rc_local;
    if (bad) return fail(MFX_E_STATE, "%s: another rank could not take the snapshot of its factors; nothing was run", what);
    return MFX_OK;
}
// after the run's stream syncs: *redo = 1 when some rank of the job lost a co-residency check (every rank then restores and goes to
// the level schedule); a chain-counter time-out anywhere is an error on every rank
static int dsgd_verdict(mfx_dsgd *d, bool collective, const char *what, int *redo)
{
    *redo = 0;
    int worst = 0, rc;
    for (DsgdRank &R : d->ranks)
        for (mfx_sgd_plan *p : R.plans) {
            unsigned st = 0;
            if ((rc = sgd_res_poll(p, &st))) return rc;
            const int v = st == 0 ? 0 : (st == 1 ? 1 : 2);
            if (v > worst) worst = v;
        }
    if (collective && d->guard != 0 && (rc = comm_agree_max(&worst))) return rc;
    if (worst >= 2) return fail(MFX_E_HIP, "%s: a resident-side block epoch timed out waiting for a chain counter (on this rank or another)", what);
    if (worst == 1) {
        // a PERMANENT change of schedule for this object: say so once, where a bench log or an operator sees it (the kernel labels
        // say it too, but only to who asks)
        {
            int lost = 0, first = -1;
            for (DsgdRank &R : d->ranks)
                for (size_t j = 0; j < R.plans.size(); ++j) {
                    unsigned st = 0;
                    if (!sgd_res_poll(R.plans[j], &st) && st == 1) { ++lost; if (first < 0) first = (int)j; }
                }
            fprintf(stderr, "[mfx] %s: rank %d: %d block epoch(s) lost the co-residency check on this rank (first: block %d; status 1 = launch "
                            "not applied)%s -- factors restored, ALL blocks of the job go to the level schedule for good, the call is repeated\n",
                    what, d->rank < 0 ? 0 : (comm_active() ? comm_rank() : d->rank), lost, first, lost ? "" : " [another rank reported it]");
        }
        if ((rc = dsgd_copy_state(d, false))) return rc;
        for (DsgdRank &R : d->ranks)
            for (mfx_sgd_plan *p : R.plans) plan_force_levels(p);
        d->guard = 0;
        d->avg_fresh = false;
        MFX_HIP(hipStreamSynchronize(stream()));
        *redo = 1;
    }
    return MFX_OK;
}

extern "C" {

void mfx_dsgd_destroy(mfx_dsgd *d)
{
    if (!d) return;
    if (ready()) {
        (void)hipStreamSynchronize(stream());
        if (d->xfer) (void)hipStreamSynchronize(d->xfer);
    }
    for (DsgdRank &r : d->ranks) {
        for (hipEvent_t e : r.trained) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : r.arrived) if (e) (void)hipEventDestroy(e);
        // sub-plans that alias the user side first, their owner (plans[0]) last
        for (size_t j = r.plans.size(); j-- > 0;)
            if (r.plans[j]) mfx_sgd_plan_destroy(r.plans[j]);
    }
    if (d->xfer) (void)hipStreamDestroy(d->xfer);
    for (double *q : d->snapV) if (q) (void)hipFree(q);
    for (double *q : d->snapub) if (q) (void)hipFree(q);
    for (auto &v : d->snapU) for (double *q : v) if (q) (void)hipFree(q);
    for (auto &v : d->snapib) for (double *q : v) if (q) (void)hipFree(q);
    for (double *q : d->avg_startU) if (q) (void)hipFree(q);
    for (double *q : d->avg_startib) if (q) (void)hipFree(q);
    delete d;
}

mfx_dsgd *mfx_dsgd_create(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni, int dim,
                          const int64_t *user_bounds, const int64_t *item_bounds, int n_item_blocks, int rank,
                          int world, int flags)
{
    TraceRange trace_("mfx:dsgd_create (block split + schedules)");
    if ((nnz > 0 && (!idx || !r)) || nnz < 0 || nu <= 0 || ni <= 0 || dim < 1 || !user_bounds || !item_bounds ||
        world < 1 || rank < -1 || rank >= world ||
        n_item_blocks < 1 || n_item_blocks % world != 0 ||
        (flags & ~MFX_DSGD_BALANCED)) {
        // (any multiple of world item blocks: stride = n_item_blocks / world, a block rests stride - 1 steps in transit.  In ONE
        //  process -- world = 1, or every rank in this process, rank = -1 -- the epoch is the sequential loop on that block order:
        //  the way to put a catalogue that is larger than the register file -- and, with several user blocks, a user side whose
        //  rows exceed the 2 GB a travelling side may have -- on the static resident-side epoch, block by block; over real ranks
        //  the same choice keeps every GPU's blocks small enough for it: tests/test_gpu_multiproc.py rehearses stride 3)
        fail(MFX_E_ARG, "mfx_dsgd_create: bad argument (item blocks must be a positive multiple of world)");
        return nullptr;
    }
    const int B = n_item_blocks;
    if (user_bounds[0] != 0 || user_bounds[world] != nu || item_bounds[0] != 0 || item_bounds[B] != ni) {
        fail(MFX_E_ARG, "mfx_dsgd_create: bounds must run from 0 to nu / ni");
        return nullptr;
    }
    for (int g = 0; g < world; ++g)
        if (user_bounds[g + 1] < user_bounds[g]) {
            fail(MFX_E_ARG, "mfx_dsgd_create: bounds must be ascending");
            return nullptr;
        }
    for (int j = 0; j < B; ++j)
        if (item_bounds[j + 1] < item_bounds[j]) {
            fail(MFX_E_ARG, "mfx_dsgd_create: bounds must be ascending");
            return nullptr;
        }
    if (rank >= 0 && world > 1 && (!comm_active() || comm_world() != world || comm_rank() != rank)) {
        fail(MFX_E_STATE, "mfx_dsgd_create: mfx_comm_init(rank=%d, world=%d) has not been called", rank, world);
        return nullptr;
    }
    if (ensure_init()) return nullptr;
    mfx_dsgd *d = new (std::nothrow) mfx_dsgd();
    if (!d) {
        fail(MFX_E_NOMEM, "mfx_dsgd_create: out of host memory");
        return nullptr;
    }
    try {
        d->world = world;
        d->rank = rank;
        d->B = B;
        d->stride = B / world;
        d->nu = nu;
        d->ni = ni;
        d->k = dim;
        d->ub.assign(user_bounds, user_bounds + world + 1);
        d->ib.assign(item_bounds, item_bounds + B + 1);
        if (hipStreamCreateWithFlags(&d->xfer, hipStreamNonBlocking) != hipSuccess) {
            fail(MFX_E_HIP, "mfx_dsgd_create: cannot create the transfer stream");
            mfx_dsgd_destroy(d);
            return nullptr;
        }
        const int g0 = rank < 0 ? 0 : rank, g1 = rank < 0 ? world : rank + 1;
        const int nr = g1 - g0;
        // split the ratings (order preserved) into this process's blocks, ids made block-local.  Two passes over contiguous
        // chunks on up to 16 host threads: block slot + per-thread counts, then -- from per-(thread, block) offsets, so that a
        // block keeps the input order -- the fill (a billion ratings: 52 s on one thread)
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)std::thread::hardware_concurrency(), nnz / 1000000 + 1}));
        const size_t NB = (size_t)nr * B;
        std::vector<int32_t> blk((size_t)(nnz ? nnz : 1));       // block slot per rating, -1 = not ours
        std::vector<int64_t> tcnt((size_t)T * NB, 0);            // [thread][block]
        std::atomic<int64_t> bad(-1);
        auto chunk_lo = [&](int t) { return nnz * t / T; };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    int64_t *c = &tcnt[(size_t)t * NB];
                    for (int64_t x = chunk_lo(t); x < chunk_lo(t + 1); ++x) {
                        const int32_t u = idx[2 * x], i = idx[2 * x + 1];
                        if (u < 0 || u >= nu || i < 0 || i >= ni) {
                            int64_t exp = -1;
                            bad.compare_exchange_strong(exp, x);
                            blk[x] = -1;
                            continue;
                        }
                        const int g = block_of(d->ub, u);
                        if (g < g0 || g >= g1) { blk[x] = -1; continue; }
                        const int b = (g - g0) * B + block_of(d->ib, i);
                        blk[x] = b;
                        ++c[b];
                    }
                });
            for (std::thread &t : th) t.join();
        }
        if (bad.load() >= 0) {
            int64_t t = nnz;                                     // (report the FIRST offending rating, as the one-thread loop did)
            for (int64_t x = 0; x < nnz; ++x) {
                const int32_t u = idx[2 * x], i = idx[2 * x + 1];
                if (u < 0 || u >= nu || i < 0 || i >= ni) { t = x; break; }
            }
            fail(MFX_E_RANGE, "mfx_dsgd_create: rating %lld: (user %d, item %d) outside [0,%d) x [0,%d)",
                 (long long)t, idx[2 * t], idx[2 * t + 1], nu, ni);
            mfx_dsgd_destroy(d);
            return nullptr;
        }
        std::vector<int64_t> cnt(NB, 0);
        for (size_t b = 0; b < NB; ++b) {                        // tcnt becomes the start offset of (thread, block)
            int64_t run = 0;
            for (int t = 0; t < T; ++t) { const int64_t c = tcnt[(size_t)t * NB + b]; tcnt[(size_t)t * NB + b] = run; run += c; }
            cnt[b] = run;
            d->nnz_local += run;
        }
        std::vector<std::vector<int32_t>> bidx(NB);
        std::vector<std::vector<double>> br(NB);
        std::vector<std::vector<int64_t>> borig(NB);
        for (size_t b = 0; b < NB; ++b) {
            bidx[b].resize((size_t)cnt[b] * 2 + 2);
            br[b].resize((size_t)cnt[b] + 1);
            borig[b].resize((size_t)cnt[b]);
        }
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    int64_t *off = &tcnt[(size_t)t * NB];
                    for (int64_t x = chunk_lo(t); x < chunk_lo(t + 1); ++x) {
                        const int b = blk[x];
                        if (b < 0) continue;
                        const int g = g0 + b / B, j = b % B;
                        const int64_t c = off[b]++;
                        bidx[b][2 * c] = idx[2 * x] - (int32_t)d->ub[g];
                        bidx[b][2 * c + 1] = idx[2 * x + 1] - (int32_t)d->ib[j];
                        br[b][c] = r[x];
                        borig[b][c] = x;
                    }
                });
            for (std::thread &t : th) t.join();
        }
        std::vector<int32_t>().swap(blk);
        d->ranks.resize((size_t)nr);
        for (int gi = 0; gi < nr; ++gi) {
            DsgdRank &R = d->ranks[gi];
            R.g = g0 + gi;
            R.user_lo = (int)d->ub[R.g];
            R.nu_local = (int)(d->ub[R.g + 1] - d->ub[R.g]);
            R.plans.assign((size_t)B, nullptr);
            R.trained.assign((size_t)B, nullptr);
            R.arrived.assign((size_t)B, nullptr);
            R.in_flight.assign((size_t)B, 0);
            R.orig.resize((size_t)B);
            for (int j = 0; j < B; ++j) {
                const size_t b = (size_t)gi * B + j;
                const int nij = (int)(d->ib[j + 1] - d->ib[j]);
                const int nul = R.nu_local > 0 ? R.nu_local : 1, nil = nij > 0 ? nij : 1;
                if ((flags & MFX_DSGD_BALANCED) && cnt[b] >= 2) {
                    // the block's own minimum-level order (first-fit recolouring of its dependency levels);
                    // a refusal (too many levels, scratch) keeps the given order -- never a wrong one
                    std::vector<int64_t> perm((size_t)cnt[b]);
                    int64_t rounds = 0;
                    if (mfx_schedule_balanced_order_gpu(bidx[b].data(), cnt[b], nul, nil, perm.data(), &rounds) == MFX_OK) {
                        std::vector<int32_t> ni2((size_t)cnt[b] * 2 + 2);
                        std::vector<double> nr2((size_t)cnt[b] + 1);
                        std::vector<int64_t> no2((size_t)cnt[b]);
                        for (int64_t c = 0; c < cnt[b]; ++c) {
                            const int64_t s = perm[(size_t)c];
                            ni2[2 * c] = bidx[b][2 * s];
                            ni2[2 * c + 1] = bidx[b][2 * s + 1];
                            nr2[c] = br[b][s];
                            no2[c] = borig[b][s];
                        }
                        bidx[b].swap(ni2);
                        br[b].swap(nr2);
                        borig[b].swap(no2);
                    }
                }
                // (headroom: an eighth of the wave slots stays free for the transfer kernels that run beside a block -- none with world = 1)
                // (... and none for a one-rank object outside a communicator; inside one -- a rank of the averaging mode -- the
                //  all-reduce of block j runs on the transfer stream beside block j + 1)
                const bool beside = world > 1 || (comm_active() && comm_world() > 1);
                R.plans[j] = plan_create_ex(bidx[b].data(), br[b].data(), cnt[b], nul, nil, dim, MFX_SCHED_AUTO,
                                            MFX_EPOCH_AUTO | (beside ? kEpochHeadroom : 0));
                if (!R.plans[j] || (j > 0 && plan_alias_users(R.plans[j], R.plans[0])) ||
                    hipEventCreateWithFlags(&R.trained[j], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&R.arrived[j], hipEventDisableTiming) != hipSuccess) {
                    if (R.plans[j]) fail(MFX_E_HIP, "mfx_dsgd_create: block (%d, %d) set-up failed", R.g, j);
                    mfx_dsgd_destroy(d);
                    return nullptr;
                }
                R.orig[j].swap(borig[b]);
                std::vector<int32_t>().swap(bidx[b]);
                std::vector<double>().swap(br[b]);
            }
        }
    } catch (const std::bad_alloc &) {
        fail(MFX_E_NOMEM, "mfx_dsgd_create: out of host memory");
        mfx_dsgd_destroy(d);
        return nullptr;
    }
    return d;
}

/* info[0] = ratings held by this process, [1] = dependency levels summed over its blocks,
 * [2] = largest block (ratings), [3] = smallest block, [4] = launches per epoch (all blocks), [5] = item blocks */
int mfx_dsgd_info(const mfx_dsgd *d, int64_t info[8])
{
    if (!d || !info) return fail(MFX_E_ARG, "mfx_dsgd_info: NULL");
    memset(info, 0, 8 * sizeof(int64_t));
    info[0] = d->nnz_local;
    info[3] = INT64_MAX;
    info[5] = d->B;
    for (const DsgdRank &R : d->ranks)
        for (mfx_sgd_plan *p : R.plans) {
            int64_t pi[8];
            mfx_sgd_plan_info(p, pi);
            info[1] += pi[0];
            info[4] += pi[2];
            const int64_t n = plan_nnz(p);
            if (n > info[2]) info[2] = n;
            if (n < info[3]) info[3] = n;
            if (plan_epoch_kernel(p) != MFX_EPOCH_LEVELS) ++info[6];      // blocks on the resident-side epoch
        }
    return MFX_OK;
}

// The rating order this object's epoch is bit-identical to, restricted to the ratings this process holds:
// for h: for its ranks g (ascending): block (g, (stride g + h) mod B) in block order.  perm_out[p] = index
// (into the arrays given to mfx_dsgd_create) of the p-th rating; nnz_local entries.
int mfx_dsgd_get_order(const mfx_dsgd *d, int64_t *perm_out)
{
    if (!d || !perm_out) return fail(MFX_E_ARG, "mfx_dsgd_get_order: NULL");
    int64_t p = 0;
    for (int h = 0; h < d->B; ++h)
        for (const DsgdRank &R : d->ranks) {
            const std::vector<int64_t> &o = R.orig[(size_t)((d->stride * R.g + h) % d->B)];
            for (int64_t t : o) perm_out[p++] = t;
        }
    return MFX_OK;
}

// u [k, ni] / items_bias [ni]: the FULL item side (every rank passes the same arrays; a rank keeps the
// blocks it starts with).  v [k, nu_local] / users_bias [nu_local]: this rank's users -- in the fake
// world (rank -1) the full [k, nu] / [nu].
int mfx_dsgd_set_factors(mfx_dsgd *d, const double *u, const double *v, const double *items_bias,
                         const double *users_bias)
{
    TraceRange trace_("mfx:dsgd_set_factors");
    if (!d || !u || !v) return fail(MFX_E_ARG, "mfx_dsgd_set_factors: NULL argument");
    const int k = d->k;
    const int64_t v_ld = d->rank < 0 ? d->nu : d->ranks[0].nu_local;      // row length of the v passed in
    MFX_HIP(hipStreamSynchronize(d->xfer));
    std::vector<double> tmp;
    for (DsgdRank &R : d->ranks) {
        // users: the rank's column range of v
        const int64_t off = d->rank < 0 ? R.user_lo : 0;
        const int n = R.nu_local > 0 ? R.nu_local : 1;
        tmp.assign((size_t)k * n, 0.0);
        for (int f = 0; f < k; ++f)
            if (R.nu_local > 0) memcpy(&tmp[(size_t)f * n], v + (size_t)f * v_ld + off, (size_t)R.nu_local * sizeof(double));
        std::vector<double> ubt((size_t)n, 0.0);
        if (users_bias && R.nu_local > 0) memcpy(ubt.data(), users_bias + off, (size_t)R.nu_local * sizeof(double));
        int rc = plan_upload_users(R.plans[0], tmp.data(), ubt.data());
        if (rc) return rc;
        // items: every sub-plan gets its block here (a rank starts with its home blocks; the others are
        // overwritten by the rotation before they are trained), so no plan ever runs on uninitialised rows
        for (int j = 0; j < d->B; ++j) {
            const int nij = (int)(d->ib[j + 1] - d->ib[j]);
            const int m = nij > 0 ? nij : 1;
            tmp.assign((size_t)k * m, 0.0);
            for (int f = 0; f < k; ++f)
                if (nij > 0) memcpy(&tmp[(size_t)f * m], u + (size_t)f * d->ni + d->ib[j], (size_t)nij * sizeof(double));
            std::vector<double> ibt((size_t)m, 0.0);
            if (items_bias && nij > 0) memcpy(ibt.data(), items_bias + d->ib[j], (size_t)nij * sizeof(double));
            rc = plan_upload_items(R.plans[j], tmp.data(), ibt.data());
            if (rc) return rc;
            plan_mark_ready(R.plans[j]);
            R.in_flight[j] = 0;
        }
    }
    d->have_factors = true;
    d->avg_fresh = false;
    return MFX_OK;
}

// one run of the block rotation (mfx_dsgd_run_kmf below wraps it in the snapshot / verdict / recovery steps)
static int dsgd_rotation_body(mfx_dsgd *d, int mode, int nbr_epochs, double lr, double Ku, double Ki, double Kb, double *se_out)
{
    const int G = d->world, B = d->B, stride = d->stride;
    hipStream_t s = stream(), x = d->xfer;
    int rc;
    for (DsgdRank &R : d->ranks)
        for (mfx_sgd_plan *p : R.plans)
            if ((rc = sgd_reset_se(p, nbr_epochs))) return rc;
    for (int ep = 0; ep < nbr_epochs; ++ep) {
        for (int h = 0; h < B; ++h) {
            // ---- compute: every rank of this process trains its block of the step -------------------
            for (DsgdRank &R : d->ranks) {
                const int j = (stride * R.g + h) % B;
                if (R.in_flight[j]) {                             // the block travelled here: wait for it
                    MFX_HIP(hipStreamWaitEvent(s, R.arrived[j], 0));
                    R.in_flight[j] = 0;
                }
                if ((rc = sgd_enqueue_k1(R.plans[j], mode, lr, Ku, Ki, Kb))) return rc;
                if (G > 1) MFX_HIP(hipEventRecord(R.trained[j], s));
            }
            if (G == 1) continue;
            // ---- hand-off on the transfer stream: block j leaves rank g for rank g - 1, which trains it
            //      `stride` steps later; with stride 2 it travels under the next step's levels ---------
            if (d->rank < 0) {
                for (DsgdRank &R : d->ranks) {
                    const int j = (stride * R.g + h) % B;
                    DsgdRank &dst = d->ranks[(size_t)((R.g + G - 1) % G)];
                    mfx_sgd_plan *from = R.plans[j], *to = dst.plans[j];
                    MFX_HIP(hipStreamWaitEvent(x, R.trained[j], 0));
                    MFX_HIP(hipMemcpyAsync(plan_U(to), plan_U(from), (size_t)plan_ni(from) * d->k * sizeof(double),
                                           hipMemcpyDeviceToDevice, x));
                    MFX_HIP(hipMemcpyAsync(plan_ib(to), plan_ib(from), (size_t)plan_ni(from) * sizeof(double),
                                           hipMemcpyDeviceToDevice, x));
                    MFX_HIP(hipEventRecord(dst.arrived[j], x));
                    dst.in_flight[j] = 1;
                }
            } else {
                DsgdRank &R = d->ranks[0];
                const int j_out = (stride * R.g + h) % B, j_in = (stride * (R.g + 1) + h) % B;
                mfx_sgd_plan *po = R.plans[j_out], *pi = R.plans[j_in];
                // empty item blocks are stored as one padding row: nothing travels for them
                const size_t n_out = (size_t)(d->ib[j_out + 1] - d->ib[j_out]), n_in = (size_t)(d->ib[j_in + 1] - d->ib[j_in]);
                const CommSeg out[2] = {{plan_U(po), n_out * d->k}, {plan_ib(po), n_out}};
                const CommSeg in[2] = {{plan_U(pi), n_in * d->k}, {plan_ib(pi), n_in}};
                MFX_HIP(hipStreamWaitEvent(x, R.trained[j_out], 0));
                if ((rc = comm_sendrecv_f64(out, 2, (R.g + G - 1) % G, in, 2, (R.g + 1) % G, x))) return rc;
                MFX_HIP(hipEventRecord(R.arrived[j_in], x));
                R.in_flight[j_in] = 1;
            }
        }
    }
    // the epoch ends with every block home again; join the transfer stream
    MFX_HIP(hipStreamSynchronize(x));
    MFX_HIP(hipStreamSynchronize(s));
    for (DsgdRank &R : d->ranks)
        for (int j = 0; j < B; ++j) R.in_flight[j] = 0;
    if (se_out) {
        std::vector<double> se((size_t)nbr_epochs, 0.0);
        // per epoch: blocks in step order, ranks in rank order (fake world: exactly the order of the
        // equivalent sequential run; real ranks: each rank's partial, then a rank-ordered sum)
        std::vector<std::vector<double>> per((size_t)d->ranks.size() * B);
        for (size_t gi = 0; gi < d->ranks.size(); ++gi)
            for (int j = 0; j < B; ++j) {
                per[gi * B + j].resize((size_t)nbr_epochs);
                if ((rc = sgd_fetch_se(d->ranks[gi].plans[j], nbr_epochs, per[gi * B + j].data()))) return rc;
            }
        for (int ep = 0; ep < nbr_epochs; ++ep)
            for (int h = 0; h < B; ++h)
                for (size_t gi = 0; gi < d->ranks.size(); ++gi)
                    se[ep] += per[gi * B + (size_t)((stride * d->ranks[gi].g + h) % B)][ep];
        if (d->rank >= 0 && G > 1) {
            // all-gather the per-rank partials and add them in rank order (deterministic)
            double *dbuf = nullptr;
            MFX_HIP(hipMalloc((void **)&dbuf, (size_t)G * nbr_epochs * sizeof(double)));
            hipError_t e = hipMemcpyAsync(dbuf + (size_t)d->rank * nbr_epochs, se.data(), (size_t)nbr_epochs * sizeof(double),
                                          hipMemcpyHostToDevice, s);
            rc = e == hipSuccess ? comm_allgather_f64(dbuf, (size_t)nbr_epochs) : fail(MFX_E_HIP, "mfx_dsgd_run_kmf: copy failed");
            std::vector<double> all((size_t)G * nbr_epochs);
            if (!rc && hipMemcpyAsync(all.data(), dbuf, all.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
                rc = fail(MFX_E_HIP, "mfx_dsgd_run_kmf: copy failed");
            if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(MFX_E_HIP, "mfx_dsgd_run_kmf: sync failed");
            (void)hipFree(dbuf);
            if (rc) return rc;
            for (int ep = 0; ep < nbr_epochs; ++ep) {
                double t = 0.0;
                for (int g = 0; g < G; ++g) t += all[(size_t)g * nbr_epochs + ep];
                se[ep] = t;
            }
        }
        memcpy(se_out, se.data(), (size_t)nbr_epochs * sizeof(double));
    }
    return MFX_OK;
}

// se_out (nullable): f64[nbr_epochs], the squared-error sum of each epoch over ALL ratings of the job
// (online, as the reference accumulates it, kmf_train.pyx:257); blocks are added in step order,
// ranks in rank order.
int mfx_dsgd_run_kmf(mfx_dsgd *d, int mode, int nbr_epochs, double lr, double Ku, double Ki, double Kb,
                     double *se_out)
{
    TraceRange trace_("mfx:dsgd_run_kmf (block rotation)");
    if (!d) return fail(MFX_E_ARG, "mfx_dsgd_run_kmf: NULL");
    if (!d->have_factors) return fail(MFX_E_STATE, "mfx_dsgd_run_kmf: mfx_dsgd_set_factors has not been called");
    if (mode != MFX_MODE_LINEAR && mode != MFX_MODE_LOGISTIC) return fail(MFX_E_ARG, "mfx_dsgd_run_kmf: mode %d", mode);
    if (nbr_epochs <= 0) return MFX_OK;
    int rc, redo = 0;
    const bool collective = d->rank >= 0 && d->world > 1;            // real ranks: every rank is in this call
    if ((rc = dsgd_guard(d, collective))) return rc;
    if ((rc = dsgd_snapshot(d, collective, "mfx_dsgd_run_kmf"))) return rc;
    if ((rc = dsgd_rotation_body(d, mode, nbr_epochs, lr, Ku, Ki, Kb, se_out))) return rc;
    if ((rc = dsgd_verdict(d, collective, "mfx_dsgd_run_kmf", &redo))) return rc;
    if (redo) rc = dsgd_rotation_body(d, mode, nbr_epochs, lr, Ku, Ki, Kb, se_out);     // every block on the level schedule now
    return rc;
}

// one run of the averaging mode on the block order.  Block j's all-reduce rides the TRANSFER stream behind trained[j] while block
// j + 1 trains on the compute stream (round 3 enqueued both on one stream: no overlap); block j's next training -- the next epoch --
// waits for arrived[j].  The start buffers and the combine kernels of block j touch block j's rows only.
static int dsgd_averaged_body(mfx_dsgd *d, int mode, int nbr_epochs, double lr, double Ku, double Ki, double Kb, double scale, double *se_out)
{
    const int B = d->B;
    DsgdRank &R = d->ranks[0];
    hipStream_t s = stream(), x = d->xfer;
    int rc;
    const bool comm = comm_active() && comm_world() > 1;
    if (comm) {
        if (d->avg_startU.empty()) { d->avg_startU.assign((size_t)B, nullptr); d->avg_startib.assign((size_t)B, nullptr); }
        for (int j = 0; j < B; ++j) {
            const size_t nb = (size_t)plan_ni(R.plans[j]), nU = nb * d->k;
            if (!d->avg_startU[j]) {
                MFX_HIP(hipMalloc((void **)&d->avg_startU[j], nU * sizeof(double)));
                MFX_HIP(hipMalloc((void **)&d->avg_startib[j], nb * sizeof(double)));
            }
            if (!d->avg_fresh) {
                MFX_HIP(hipMemcpyAsync(d->avg_startU[j], plan_U(R.plans[j]), nU * sizeof(double), hipMemcpyDeviceToDevice, s));
                MFX_HIP(hipMemcpyAsync(d->avg_startib[j], plan_ib(R.plans[j]), nb * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
        }
        d->avg_fresh = true;
    }
    for (mfx_sgd_plan *p : R.plans)
        if ((rc = sgd_reset_se(p, nbr_epochs))) return rc;
    for (int ep = 0; ep < nbr_epochs; ++ep)
        for (int j = 0; j < B; ++j) {
            if (R.in_flight[j]) {                                 // the block's all-reduce of the epoch before
                MFX_HIP(hipStreamWaitEvent(s, R.arrived[j], 0));
                R.in_flight[j] = 0;
            }
            if ((rc = sgd_enqueue_k1(R.plans[j], mode, lr, Ku, Ki, Kb))) return rc;
            if (comm) {
                const size_t nb = (size_t)plan_ni(R.plans[j]);
                MFX_HIP(hipEventRecord(R.trained[j], s));
                MFX_HIP(hipStreamWaitEvent(x, R.trained[j], 0));
                if ((rc = comm_average_f64(plan_U(R.plans[j]), d->avg_startU[j], nb * d->k, scale, x))) return rc;
                if ((rc = comm_average_f64(plan_ib(R.plans[j]), d->avg_startib[j], nb, scale, x))) return rc;
                MFX_HIP(hipEventRecord(R.arrived[j], x));
                R.in_flight[j] = 1;
            }
        }
    MFX_HIP(hipStreamSynchronize(x));
    MFX_HIP(hipStreamSynchronize(s));
    for (int j = 0; j < B; ++j) R.in_flight[j] = 0;
    if (se_out) {
        std::vector<double> per((size_t)nbr_epochs);
        for (int ep = 0; ep < nbr_epochs; ++ep) se_out[ep] = 0.0;
        for (int j = 0; j < B; ++j) {
            if ((rc = sgd_fetch_se(R.plans[j], nbr_epochs, per.data()))) return rc;
            for (int ep = 0; ep < nbr_epochs; ++ep) se_out[ep] += per[(size_t)ep];
        }
    }
    return MFX_OK;
}

// north_star's averaging scheme (user shards + an RCCL all-reduce of the item deltas per epoch, mfx_sgd_run_kmf_sharded) with the
// shard's local epoch on the BLOCK ORDER: this object was created with world = 1 over ONE rank's ratings (local user ids) and any
// number of item blocks, so that every block runs the static resident-side epoch instead of the level schedule (a shard holds ALL
// items: 200 K at cfg 3).  Per epoch and item block j: train the block, then U_j <- U_start_j + scale * allreduce_sum(U_j -
// U_start_j) (items_bias alike) -- element for element the arithmetic of the whole-array all-reduce; with no communicator (one
// rank) nothing is combined and the epoch is the sequential loop on the block order.  se_out (nullable): this rank's squared-error
// sums per epoch.
int mfx_dsgd_run_kmf_averaged(mfx_dsgd *d, int mode, int nbr_epochs, double lr, double Ku, double Ki, double Kb, double scale,
                              double *se_out)
{
    TraceRange trace_("mfx:dsgd_run_kmf_averaged (block order + RCCL all-reduce per item block)");
    if (!d) return fail(MFX_E_ARG, "mfx_dsgd_run_kmf_averaged: NULL");
    if (d->world != 1) return fail(MFX_E_ARG, "mfx_dsgd_run_kmf_averaged: the object must hold ONE rank's ratings (created with world = 1)");
    if (!d->have_factors) return fail(MFX_E_STATE, "mfx_dsgd_run_kmf_averaged: mfx_dsgd_set_factors has not been called");
    if (mode != MFX_MODE_LINEAR && mode != MFX_MODE_LOGISTIC) return fail(MFX_E_ARG, "mfx_dsgd_run_kmf_averaged: mode %d", mode);
    if (nbr_epochs <= 0) return MFX_OK;
    int rc, redo = 0;
    const bool collective = comm_active() && comm_world() > 1;       // the all-reduces make this call a collective one
    if ((rc = dsgd_guard(d, collective))) return rc;
    if ((rc = dsgd_snapshot(d, collective, "mfx_dsgd_run_kmf_averaged"))) return rc;
    if ((rc = dsgd_averaged_body(d, mode, nbr_epochs, lr, Ku, Ki, Kb, scale, se_out))) return rc;
    if ((rc = dsgd_verdict(d, collective, "mfx_dsgd_run_kmf_averaged", &redo))) return rc;
    if (redo) rc = dsgd_averaged_body(d, mode, nbr_epochs, lr, Ku, Ki, Kb, scale, se_out);
    return rc;
}

// The Funk family (estimator_loop_without_bias / _with_bias / _with_bias_dev, gd_estimator.pyx:691-779, :489-685) on the ITEM-BLOCK
// order of a ONE-process object (world = 1, or rank = -1 with world = 1 user block): a pass of a feature walks the item blocks one
// after the other on the one shared user side, so it IS the reference's loop on the order mfx_dsgd_get_order reports, with one
// stopping rule over the whole pass -- and every block is small enough for the resident-side pass (a lane per item) where the whole
// catalogue is not (more than ~450 K items).  use_bias: the biases given to mfx_dsgd_set_factors are the inputs of _with_bias.
int mfx_dsgd_run_funk(mfx_dsgd *d, int min_epochs, double min_improvement, double f_init, double learning_rate, double K,
                      double overall_avg, int use_bias, int update_users, int update_items, int epoch_cap, double *rmse_hist,
                      int64_t hist_cap, int64_t *n_hist)
{
    TraceRange trace_("mfx:dsgd_run_funk (item-block order)");
    if (!d) return fail(MFX_E_ARG, "mfx_dsgd_run_funk: NULL");
    if (d->world != 1 || d->ranks.size() != 1)
        return fail(MFX_E_ARG, "mfx_dsgd_run_funk: the object must live in one process with ONE user block (created with world = 1)");
    if (!d->have_factors) return fail(MFX_E_STATE, "mfx_dsgd_run_funk: mfx_dsgd_set_factors has not been called");
    if (n_hist) *n_hist = 0;
    DsgdRank &R = d->ranks[0];
    MFX_HIP(hipStreamSynchronize(d->xfer));
    return funk_blocks_core(R.plans.data(), (int)R.plans.size(), min_epochs, min_improvement, f_init, learning_rate, K, overall_avg,
                            use_bias, update_users, update_items, epoch_cap, rmse_hist, hist_cap, n_hist);
}

// u [k, ni] / items_bias [ni] (nullable): the full item side, gathered from the ranks that hold the
// blocks (collective: every rank must call).  v / users_bias: as in mfx_dsgd_set_factors.
int mfx_dsgd_get_factors(mfx_dsgd *d, double *u, double *v, double *items_bias, double *users_bias)
{
    TraceRange trace_("mfx:dsgd_get_factors");
    if (!d) return fail(MFX_E_ARG, "mfx_dsgd_get_factors: NULL");
    if (!d->have_factors) return fail(MFX_E_STATE, "mfx_dsgd_get_factors: no factors");
    const int k = d->k, G = d->world, B = d->B;
    const int64_t v_ld = d->rank < 0 ? d->nu : d->ranks[0].nu_local;
    MFX_HIP(hipStreamSynchronize(d->xfer));
    std::vector<double> tmp, bt;
    int rc;
    for (DsgdRank &R : d->ranks) {
        if (v || users_bias) {
            const int n = R.nu_local > 0 ? R.nu_local : 1;
            tmp.resize((size_t)k * n);
            bt.resize((size_t)n);
            if ((rc = plan_download_users(R.plans[0], tmp.data(), bt.data()))) return rc;
            const int64_t off = d->rank < 0 ? R.user_lo : 0;
            if (v && R.nu_local > 0)
                for (int f = 0; f < k; ++f) memcpy(v + (size_t)f * v_ld + off, &tmp[(size_t)f * n], (size_t)R.nu_local * sizeof(double));
            if (users_bias && R.nu_local > 0) memcpy(users_bias + off, bt.data(), (size_t)R.nu_local * sizeof(double));
        }
    }
    if (!u && !items_bias) return MFX_OK;
    // between epochs block j is at its home rank j / stride; a real rank broadcasts its blocks to the others
    for (int j = 0; j < B; ++j) {
        const int nij = (int)(d->ib[j + 1] - d->ib[j]);
        if (nij <= 0) continue;
        const int home = j / d->stride;
        mfx_sgd_plan *p = nullptr;
        if (d->rank < 0) p = d->ranks[(size_t)home].plans[j];
        else {
            p = d->ranks[0].plans[j];
            if (G > 1) {
                if ((rc = comm_broadcast_f64(plan_U(p), (size_t)nij * k, home))) return rc;
                if ((rc = comm_broadcast_f64(plan_ib(p), (size_t)nij, home))) return rc;
            }
        }
        tmp.resize((size_t)k * nij);
        bt.resize((size_t)nij);
        if ((rc = plan_download_items(p, tmp.data(), bt.data()))) return rc;
        if (u)
            for (int f = 0; f < k; ++f) memcpy(u + (size_t)f * d->ni + d->ib[j], &tmp[(size_t)f * nij], (size_t)nij * sizeof(double));
        if (items_bias) memcpy(items_bias + d->ib[j], bt.data(), (size_t)nij * sizeof(double));
    }
    return MFX_OK;
}

}  // extern "C"
