// Multi-GPU path: one process per GPU, RCCL over xGMI (include/mfx.h "multi-GPU").
//
// The reference has no communication layer at all; its only precedent is the two-"machine"
// parameter-averaging prototype (gradient_descent.py:124-211).  Here users are sharded
// across ranks (each rank's plan holds only its users' ratings), item factors and item
// biases are replicated, and after every local epoch
//     U <- U_start + scale * allreduce_sum(U_local - U_start)
// RCCL is loaded lazily (dlopen) so single-GPU users never pay for it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include "mfx_sgd_internal.h"
#include "mfx_rccl_abi.h"

namespace {

// the slice of the RCCL (NCCL-compatible) ABI this file binds by dlsym lives in mfx_rccl_abi.h, where
// tests/rccl_abi_check.cpp static_asserts it against <rccl/rccl.h> (tests/test_abi.py, CPU suite)
using namespace mfx_rccl;

struct Rccl {
    void *h = nullptr;
    GetUniqueId_t GetUniqueId = nullptr;
    CommInitRank_t CommInitRank = nullptr;
    CommDestroy_t CommDestroy = nullptr;
    AllReduce_t AllReduce = nullptr;
    AllGather_t AllGather = nullptr;
    Broadcast_t Broadcast = nullptr;
    Send_t Send = nullptr;
    Recv_t Recv = nullptr;
    GroupStart_t GroupStart = nullptr;
    GroupEnd_t GroupEnd = nullptr;
    GetErrorString_t GetErrorString = nullptr;
    GetLastError_t GetLastError = nullptr;
} R;

ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
double *g_token = nullptr;

int load_rccl()
{
    if (R.h) return MFX_OK;
    // MFX_RCCL_LIB: a library with the same twelve entry points -- the tests' shared-memory stand-in that lets
    // several processes on ONE GPU rehearse the N > 1 paths (tests/mock_rccl); the product never sets it
    const char *over = getenv("MFX_RCCL_LIB");
    if (over && over[0]) {
        // a test hook, never a deployment knob: honoured only when the process says it is a test
        const char *t = getenv("MFX_TESTING");
        if (!(t && t[0] == '1'))
            return mfx::fail(MFX_E_RCCL, "MFX_RCCL_LIB is a test hook (tests/mock_rccl) and needs MFX_TESTING=1; unset it to use librccl");
    }
    const char *names[] = {over && over[0] ? over : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (R.h || (over && over[0])) break;              // an override that does not load is an error, not a fallback
    }
    if (!R.h) return mfx::fail(MFX_E_RCCL, "cannot dlopen librccl: %s", dlerror());
    R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.h, "ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.h, "ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
    R.AllReduce = (decltype(R.AllReduce))dlsym(R.h, "ncclAllReduce");
    R.AllGather = (decltype(R.AllGather))dlsym(R.h, "ncclAllGather");
    R.Broadcast = (decltype(R.Broadcast))dlsym(R.h, "ncclBroadcast");
    R.Send = (decltype(R.Send))dlsym(R.h, "ncclSend");
    R.Recv = (decltype(R.Recv))dlsym(R.h, "ncclRecv");
    R.GroupStart = (decltype(R.GroupStart))dlsym(R.h, "ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))dlsym(R.h, "ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
    R.GetLastError = (decltype(R.GetLastError))dlsym(R.h, "ncclGetLastError");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce || !R.AllGather ||
        !R.Broadcast || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
        dlclose(R.h);
        R.h = nullptr;
        return mfx::fail(MFX_E_RCCL, "librccl lacks a required symbol");
    }
    return MFX_OK;
}

// the failure line carries RCCL's own last-error text (what NCCL_DEBUG=WARN would print) and the rank
#define MFX_NCCL(call)                                                                    \
    do {                                                                                  \
        int r_ = (call);                                                                  \
        if (r_ != ncclSuccess) {                                                          \
            const char *le_ = R.GetLastError ? R.GetLastError(g_comm) : nullptr;          \
            return mfx::fail(MFX_E_RCCL, "rank %d/%d: %s:%d: %s -> %s [%s]", g_rank,      \
                             g_world, __FILE__, __LINE__, #call,                          \
                             R.GetErrorString ? R.GetErrorString(r_) : "rccl error",      \
                             (le_ && *le_) ? le_ : "no further detail");                  \
        }                                                                                 \
    } while (0)

// a <- a - b
__global__ __launch_bounds__(256) void k_sub(double *a, const double *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = a[i] - b[i];
}
// a <- b + scale * a ; b <- that value too (next epoch's start)
__global__ __launch_bounds__(256) void k_combine(double *a, double *b, double scale, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double x = b[i] + scale * a[i];
        a[i] = x;
        b[i] = x;
    }
}

// a <- a + b   (fake world: the rank-ordered sum that stands in for the all-reduce)
__global__ __launch_bounds__(256) void k_add(double *a, const double *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = a[i] + b[i];
}

int grid_for(size_t n) { return (int)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096); }

struct ShardCtx {
    double scale;
    double *U_start, *ib_start;
};

int shard_hook(mfx_sgd_plan *p, int, void *vctx)
{
    if (!g_comm) return MFX_OK;           // a single shard is the sequential algorithm itself
    ShardCtx *c = (ShardCtx *)vctx;
    const size_t nU = (size_t)mfx::plan_ni(p) * mfx::plan_k(p), nb = (size_t)mfx::plan_ni(p);
    double *U = mfx::plan_U(p), *ib = mfx::plan_ib(p);
    hipStream_t s = mfx::stream();
    hipLaunchKernelGGL(k_sub, dim3(grid_for(nU)), dim3(256), 0, s, U, c->U_start, nU);
    hipLaunchKernelGGL(k_sub, dim3(grid_for(nb)), dim3(256), 0, s, ib, c->ib_start, nb);
    MFX_NCCL(R.AllReduce(U, U, nU, ncclFloat64, ncclSum, g_comm, s));
    MFX_NCCL(R.AllReduce(ib, ib, nb, ncclFloat64, ncclSum, g_comm, s));
    hipLaunchKernelGGL(k_combine, dim3(grid_for(nU)), dim3(256), 0, s, U, c->U_start, c->scale, nU);
    hipLaunchKernelGGL(k_combine, dim3(grid_for(nb)), dim3(256), 0, s, ib, c->ib_start, c->scale, nb);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

}  // namespace

namespace mfx {
bool comm_active() { return g_comm != nullptr; }
int comm_rank() { return g_rank; }
int comm_world() { return g_world; }
// in-place all-gather of f64: rank g's contribution sits at buf + g*count
int comm_allgather_f64(double *buf, size_t count)
{
    if (!g_comm) return MFX_OK;
    MFX_NCCL(R.AllGather(buf + (size_t)g_rank * count, buf, count, ncclFloat64, g_comm, mfx::stream()));
    return MFX_OK;
}
// cur <- start + scale * allreduce_sum(cur - start), start <- the same value (the next epoch's start): the combine step of the
// averaging scheme (shard_hook above) for one array on the library stream.  No communicator: nothing happens (a single shard is the
// sequential algorithm itself).
int comm_average_f64(double *cur, double *start, size_t n, double scale, hipStream_t st)
{
    if (!g_comm || !n) return MFX_OK;
    hipStream_t s = st ? st : mfx::stream();
    hipLaunchKernelGGL(k_sub, dim3(grid_for(n)), dim3(256), 0, s, cur, (const double *)start, n);
    MFX_NCCL(R.AllReduce(cur, cur, n, ncclFloat64, ncclSum, g_comm, s));
    hipLaunchKernelGGL(k_combine, dim3(grid_for(n)), dim3(256), 0, s, cur, start, scale, n);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}
// every rank passes its own flag and gets the maximum over the ranks (a collective: every rank must call; library stream, synchronous)
int comm_agree_max(int *flag)
{
    if (!g_comm) return MFX_OK;
    double v = (double)*flag;
    int rc = mfx_comm_allreduce_max(&v);
    if (rc) return rc;
    *flag = (int)v;
    return MFX_OK;
}
int comm_allreduce_sum_f64(double *buf, size_t count)
{
    if (!g_comm || !count) return MFX_OK;
    MFX_NCCL(R.AllReduce(buf, buf, count, ncclFloat64, ncclSum, g_comm, mfx::stream()));
    return MFX_OK;
}
int comm_broadcast_f64(double *buf, size_t count, int root)
{
    if (!g_comm || !count) return MFX_OK;
    MFX_NCCL(R.Broadcast(buf, buf, count, ncclFloat64, root, g_comm, mfx::stream()));
    return MFX_OK;
}
int comm_sendrecv_f64(const CommSeg *out, int n_out, int to, const CommSeg *in, int n_in, int from, hipStream_t st)
{
    if (!g_comm) return fail(MFX_E_STATE, "comm_sendrecv_f64: no communicator");
    if (!st) st = mfx::stream();
    MFX_NCCL(R.GroupStart());
    for (int i = 0; i < n_out; ++i)
        if (out[i].count) MFX_NCCL(R.Send(out[i].ptr, out[i].count, ncclFloat64, to, g_comm, st));
    for (int i = 0; i < n_in; ++i)
        if (in[i].count) MFX_NCCL(R.Recv(in[i].ptr, in[i].count, ncclFloat64, from, g_comm, st));
    MFX_NCCL(R.GroupEnd());
    return MFX_OK;
}
}  // namespace mfx

extern "C" {

int mfx_comm_unique_id(void *id_out)
{
    if (!id_out) return mfx::fail(MFX_E_ARG, "mfx_comm_unique_id: NULL");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    MFX_NCCL(R.GetUniqueId(&id));
    static_assert(sizeof(id) == MFX_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return MFX_OK;
}

int mfx_comm_init(int rank, int world, const void *id)
{
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id))
        return mfx::fail(MFX_E_ARG, "mfx_comm_init: rank %d / world %d", rank, world);
    int rc = mfx::ensure_init();
    if (rc) return rc;
    if (g_comm) return mfx::fail(MFX_E_STATE, "mfx_comm_init: a communicator exists already (mfx_comm_destroy first)");
    g_rank = rank;
    g_world = world;
    // world == 1 needs no communicator; MFX_FORCE_COMM=1 builds a 1-rank one anyway so that
    // the RCCL plumbing can be exercised on a single-GPU box (tests/test_gpu_comm.py)
    const char *force = getenv("MFX_FORCE_COMM");
    if (world == 1 && !(force && force[0] == '1' && id)) return MFX_OK;
    rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    MFX_NCCL(R.CommInitRank(&g_comm, world, uid, rank));
    if (!g_token) MFX_HIP(hipMalloc((void **)&g_token, sizeof(double)));
    return MFX_OK;
}

int mfx_comm_destroy(void)
{
    if (g_comm) {
        (void)hipStreamSynchronize(mfx::stream());
        R.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    if (g_token) {
        (void)hipFree(g_token);
        g_token = nullptr;
    }
    g_world = 1;
    g_rank = 0;
    return MFX_OK;
}

int mfx_comm_barrier(void)
{
    if (!mfx::ready()) return MFX_OK;
    if (g_comm) {
        MFX_HIP(hipMemsetAsync(g_token, 0, sizeof(double), mfx::stream()));
        MFX_NCCL(R.AllReduce(g_token, g_token, 1, ncclFloat64, ncclSum, g_comm, mfx::stream()));
    }
    MFX_HIP(hipStreamSynchronize(mfx::stream()));
    return MFX_OK;
}

int mfx_comm_allreduce_max(double *value)
{
    if (!value) return mfx::fail(MFX_E_ARG, "mfx_comm_allreduce_max: NULL");
    if (!g_comm) return MFX_OK;                      // single rank: the value is its own maximum
    MFX_HIP(hipMemcpyAsync(g_token, value, sizeof(double), hipMemcpyHostToDevice, mfx::stream()));
    MFX_NCCL(R.AllReduce(g_token, g_token, 1, ncclFloat64, ncclMax, g_comm, mfx::stream()));
    MFX_HIP(hipMemcpyAsync(value, g_token, sizeof(double), hipMemcpyDeviceToHost, mfx::stream()));
    MFX_HIP(hipStreamSynchronize(mfx::stream()));
    return MFX_OK;
}

// ---- user-sharded Funk pass: per feature-epoch all-reduce of the item-side vector U[f,:] ----
namespace {
struct FunkShard {
    double scale;
    double *start;     // [ni] snapshot of U[f,:] before the epoch
    double *pair;      // device [2]: {se, nnz} for the global RMSE
};
int funk_pre(void *vc, double *uf, int ni)
{
    if (!g_comm) return MFX_OK;
    FunkShard *c = (FunkShard *)vc;
    MFX_HIP(hipMemcpyAsync(c->start, uf, (size_t)ni * sizeof(double), hipMemcpyDeviceToDevice, mfx::stream()));
    return MFX_OK;
}
int funk_post(void *vc, double *uf, int ni)
{
    if (!g_comm) return MFX_OK;
    FunkShard *c = (FunkShard *)vc;
    hipStream_t s = mfx::stream();
    hipLaunchKernelGGL(k_sub, dim3(grid_for(ni)), dim3(256), 0, s, uf, c->start, (size_t)ni);
    MFX_NCCL(R.AllReduce(uf, uf, (size_t)ni, ncclFloat64, ncclSum, g_comm, s));
    hipLaunchKernelGGL(k_combine, dim3(grid_for(ni)), dim3(256), 0, s, uf, c->start, c->scale, (size_t)ni);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}
int funk_global_se(void *vc, double *se, double *nnz)
{
    if (!g_comm) return MFX_OK;
    FunkShard *c = (FunkShard *)vc;
    hipStream_t s = mfx::stream();
    double h[2] = {*se, *nnz};
    MFX_HIP(hipMemcpyAsync(c->pair, h, sizeof(h), hipMemcpyHostToDevice, s));
    MFX_NCCL(R.AllReduce(c->pair, c->pair, 2, ncclFloat64, ncclSum, g_comm, s));
    MFX_HIP(hipMemcpyAsync(h, c->pair, sizeof(h), hipMemcpyDeviceToHost, s));
    MFX_HIP(hipStreamSynchronize(s));
    *se = h[0];
    *nnz = h[1];
    return MFX_OK;
}
}  // namespace

int mfx_sgd_run_funk_sharded(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init,
                             double lr, double K, double mu, int use_bias, int epoch_cap,
                             double scale, double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
    mfx::TraceRange trace_("mfx:run_funk_sharded");
    if (!p) return mfx::fail(MFX_E_ARG, "mfx_sgd_run_funk_sharded: NULL plan");
    if (g_world > 1 && !g_comm) return mfx::fail(MFX_E_STATE, "mfx_comm_init has not been called");
    FunkShard c{scale, nullptr, nullptr};
    if (g_comm) {
        MFX_HIP(hipMalloc((void **)&c.start, (size_t)mfx::plan_ni(p) * sizeof(double)));
        MFX_HIP(hipMalloc((void **)&c.pair, 2 * sizeof(double)));
    }
    mfx::FunkHooks hk{funk_pre, funk_post, funk_global_se, &c};
    int rc = mfx::funk_core(p, min_epochs, min_improvement, f_init, lr, K, mu, use_bias, 1, 1, epoch_cap,
                            rmse_hist, hist_cap, n_hist, &hk);
    if (c.start) (void)hipFree(c.start);
    if (c.pair) (void)hipFree(c.pair);
    return rc;
}

int mfx_sgd_run_kmf_sharded(mfx_sgd_plan *p, int mode, int nbr_epochs, double lr, double Ku,
                            double Ki, double Kb, double scale, double *rmse_out, double *se_out)
{
    mfx::TraceRange trace_("mfx:run_kmf_sharded (epochs + RCCL all-reduce)");
    if (!p) return mfx::fail(MFX_E_ARG, "mfx_sgd_run_kmf_sharded: NULL plan");
    if (g_world > 1 && !g_comm) return mfx::fail(MFX_E_STATE, "mfx_comm_init has not been called");
    ShardCtx c{scale, nullptr, nullptr};
    if (g_comm) {
        int rc = mfx::plan_prepare_nk(p);
        if (!rc) rc = mfx::plan_start_buffers(p, &c.U_start, &c.ib_start);
        if (rc) return rc;
        const size_t nU = (size_t)mfx::plan_ni(p) * mfx::plan_k(p), nb = (size_t)mfx::plan_ni(p);
        MFX_HIP(hipMemcpyAsync(c.U_start, mfx::plan_U(p), nU * sizeof(double),
                               hipMemcpyDeviceToDevice, mfx::stream()));
        MFX_HIP(hipMemcpyAsync(c.ib_start, mfx::plan_ib(p), nb * sizeof(double),
                               hipMemcpyDeviceToDevice, mfx::stream()));
    }
    return mfx::sgd_run_k1_hooked(p, mode, nbr_epochs, lr, Ku, Ki, Kb, shard_hook, &c, rmse_out,
                                  se_out);
}

// The arithmetic of mfx_sgd_run_kmf_sharded for `world` shards that all live on THIS GPU (test and
// rehearsal hook; no communicator): per epoch every shard runs its local epoch from the same item
// state, the deltas are summed in rank order on the device -- the stand-in for the all-reduce -- and
// every shard receives U_start + scale * sum.  Same k_sub / k_combine kernels as the RCCL path.
int mfx_sgd_run_kmf_fake_world(mfx_sgd_plan **plans, int world, int mode, int nbr_epochs, double lr,
                               double Ku, double Ki, double Kb, double scale, double *se_out)
{
    mfx::TraceRange trace_("mfx:run_kmf_fake_world");
    if (!plans || world < 1 || nbr_epochs < 0) return mfx::fail(MFX_E_ARG, "mfx_sgd_run_kmf_fake_world: bad argument");
    for (int g = 0; g < world; ++g)
        if (!plans[g] || mfx::plan_ni(plans[g]) != mfx::plan_ni(plans[0]) || mfx::plan_k(plans[g]) != mfx::plan_k(plans[0]))
            return mfx::fail(MFX_E_ARG, "mfx_sgd_run_kmf_fake_world: shard %d disagrees on the item side", g);
    const size_t nb = (size_t)mfx::plan_ni(plans[0]), nU = nb * mfx::plan_k(plans[0]);
    hipStream_t s = mfx::stream();
    double *U_start = nullptr, *ib_start = nullptr, *accU = nullptr, *accib = nullptr;
    int rc = MFX_OK;
    for (int g = 0; g < world && !rc; ++g) rc = mfx::plan_prepare_nk(plans[g]);
    if (!rc) rc = mfx::plan_start_buffers(plans[0], &U_start, &ib_start);
    if (rc) return rc;
    struct Free {
        double *&a, *&b;
        ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } guard{accU, accib};
    MFX_HIP(hipMalloc((void **)&accU, nU * sizeof(double)));
    MFX_HIP(hipMalloc((void **)&accib, nb * sizeof(double)));
    MFX_HIP(hipMemcpyAsync(U_start, mfx::plan_U(plans[0]), nU * sizeof(double), hipMemcpyDeviceToDevice, s));
    MFX_HIP(hipMemcpyAsync(ib_start, mfx::plan_ib(plans[0]), nb * sizeof(double), hipMemcpyDeviceToDevice, s));
    for (int ep = 0; ep < nbr_epochs; ++ep) {
        MFX_HIP(hipMemsetAsync(accU, 0, nU * sizeof(double), s));          // +0.0
        MFX_HIP(hipMemsetAsync(accib, 0, nb * sizeof(double), s));
        for (int g = 0; g < world; ++g) {
            double *U = mfx::plan_U(plans[g]), *ib = mfx::plan_ib(plans[g]);
            rc = mfx::sgd_run_k1_hooked(plans[g], mode, 1, lr, Ku, Ki, Kb, nullptr, nullptr, nullptr,
                                        se_out ? se_out + (size_t)g * nbr_epochs + ep : nullptr);
            if (rc) return rc;
            hipLaunchKernelGGL(k_sub, dim3(grid_for(nU)), dim3(256), 0, s, U, U_start, nU);
            hipLaunchKernelGGL(k_sub, dim3(grid_for(nb)), dim3(256), 0, s, ib, ib_start, nb);
            hipLaunchKernelGGL(k_add, dim3(grid_for(nU)), dim3(256), 0, s, accU, (const double *)U, nU);
            hipLaunchKernelGGL(k_add, dim3(grid_for(nb)), dim3(256), 0, s, accib, (const double *)ib, nb);
        }
        hipLaunchKernelGGL(k_combine, dim3(grid_for(nU)), dim3(256), 0, s, accU, U_start, scale, nU);
        hipLaunchKernelGGL(k_combine, dim3(grid_for(nb)), dim3(256), 0, s, accib, ib_start, scale, nb);
        MFX_HIP(hipGetLastError());
        for (int g = 0; g < world; ++g) {
            MFX_HIP(hipMemcpyAsync(mfx::plan_U(plans[g]), U_start, nU * sizeof(double), hipMemcpyDeviceToDevice, s));
            MFX_HIP(hipMemcpyAsync(mfx::plan_ib(plans[g]), ib_start, nb * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
    }
    MFX_HIP(hipStreamSynchronize(s));
    return MFX_OK;
}

}  // extern "C"
