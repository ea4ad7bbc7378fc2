// TEST INFRASTRUCTURE -- a stand-in for librccl.so that lets several PROCESSES SHARING ONE GPU run libmfx's
// multi-GPU code paths (RCCL itself refuses two ranks on one device).  It implements the twelve nccl* entry points
// libmfx binds (mfrec_amd/csrc/mfx_comm.hip) over a shared-memory file: every call synchronises the caller's stream,
// stages the data through the host, exchanges it with the other ranks and delivers the result LATE and asynchronously
// in stream order (deliver()).  Reductions add the ranks' contributions in
// RANK ORDER (deterministic, the order of oracle/sharded_sim.py).  Point-to-point calls inside a group make progress
// together, chunk by chunk, so two ranks that both send first cannot deadlock.  Never linked into the product;
// libmfx only finds it when MFX_RCCL_LIB names it (tests/test_gpu_multiproc.py).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <vector>

namespace {

constexpr int kMaxWorld = 8;
constexpr size_t kChunk = 1 << 20;                   // bytes per rank slot / mailbox

struct Mailbox {
    std::atomic<uint64_t> written, consumed;          // chunks posted / taken
    std::atomic<uint64_t> bytes;                      // size of the chunk in flight
};

struct Header {
    std::atomic<int> world, attached;
    std::atomic<uint64_t> bar_arrived, bar_gen;
    std::atomic<int> failed;
    Mailbox box[kMaxWorld][kMaxWorld];                // [src][dst]
};

struct Comm {
    int rank = 0, world = 1;
    Header *h = nullptr;
    char *slots = nullptr;                            // [world] x kChunk (collectives)
    char *boxes = nullptr;                            // [world][world] x kChunk (point to point)
    size_t map_bytes = 0;
    char name[96] = {0};
};

struct P2P { bool send; char *dev; size_t bytes; int peer; hipStream_t st; };

// Results are DELIVERED LATE, in stream order: a kernel that idles for ~150 us goes first, then an asynchronous copy
// from a pinned buffer, and the call returns at once -- like the real library, whose data lands when the stream gets
// there.  A caller that forgets to order its next kernel behind the communication stream (a missing or wrong event)
// reads stale rows and fails the bit-for-bit comparison instead of being saved by a blocking copy.
__global__ void k_idle(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
struct Pending { void *pinned; hipEvent_t done; };
std::vector<Pending> g_pending;

void reap(bool all)
{
    for (size_t i = 0; i < g_pending.size();) {
        if (all) (void)hipEventSynchronize(g_pending[i].done);
        if (all || hipEventQuery(g_pending[i].done) == hipSuccess) {
            (void)hipEventDestroy(g_pending[i].done);
            (void)hipHostFree(g_pending[i].pinned);
            g_pending[i] = g_pending.back();
            g_pending.pop_back();
        } else ++i;
    }
}

int deliver(void *dev, const void *host, size_t bytes, hipStream_t st)
{
    reap(false);
    if (!bytes) return 0;
    void *pin = nullptr;
    hipEvent_t ev;
    if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess) return 1;
    memcpy(pin, host, bytes);
    k_idle<<<1, 1, 0, st>>>(15000);                                       // wall_clock64 ticks at 100 MHz: ~150 us
    if (hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess) return 1;
    g_pending.push_back({pin, ev});
    return 0;
}
thread_local int g_group = 0;
thread_local std::vector<P2P> g_ops;
thread_local Comm *g_group_comm = nullptr;
const char *g_last = "ok";

size_t map_size(int world)
{
    return sizeof(Header) + (size_t)world * kChunk + (size_t)world * world * kChunk;
}

bool spin(const std::atomic<int> &failed, double t0)
{
    if (failed.load()) return false;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    if (ts.tv_sec + ts.tv_nsec * 1e-9 - t0 > 120.0) return false;    // a peer died: do not hang the test
    return true;
}
double now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int barrier(Comm *c)
{
    Header *h = c->h;
    const uint64_t gen = h->bar_gen.load();
    if (h->bar_arrived.fetch_add(1) + 1 == (uint64_t)c->world) {
        h->bar_arrived.store(0);
        h->bar_gen.fetch_add(1);
        return 0;
    }
    const double t0 = now();
    while (h->bar_gen.load() == gen)
        if (!spin(h->failed, t0)) { h->failed.store(1); g_last = "mock rccl: barrier timed out"; return 1; }
    return 0;
}

int run_p2p(Comm *c, std::vector<P2P> &ops)
{
    struct St { std::vector<char> host; size_t done = 0; };
    std::vector<St> st(ops.size());
    for (size_t i = 0; i < ops.size(); ++i) {
        if (hipStreamSynchronize(ops[i].st) != hipSuccess) return 1;
        st[i].host.resize(ops[i].bytes);
        if (ops[i].send && ops[i].bytes && hipMemcpy(st[i].host.data(), ops[i].dev, ops[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    }
    const double t0 = now();
    bool busy = true;
    while (busy) {
        busy = false;
        for (size_t i = 0; i < ops.size(); ++i) {
            P2P &o = ops[i];
            if (st[i].done >= o.bytes) continue;
            busy = true;
            // messages between the same pair, in the same direction, travel one after the other (as in NCCL)
            bool blocked = false;
            for (size_t e = 0; e < i; ++e)
                if (st[e].done < ops[e].bytes && ops[e].send == o.send && ops[e].peer == o.peer) blocked = true;
            if (blocked) continue;
            const int src = o.send ? c->rank : o.peer, dst = o.send ? o.peer : c->rank;
            Mailbox &m = c->h->box[src][dst];
            char *area = c->boxes + ((size_t)src * c->world + dst) * kChunk;
            if (o.send) {
                if (m.written.load() == m.consumed.load()) {              // free
                    const size_t n = o.bytes - st[i].done < kChunk ? o.bytes - st[i].done : kChunk;
                    memcpy(area, st[i].host.data() + st[i].done, n);
                    m.bytes.store(n);
                    m.written.fetch_add(1);
                    st[i].done += n;
                }
            } else if (m.written.load() > m.consumed.load()) {             // a chunk waits
                const size_t n = m.bytes.load();
                if (st[i].done + n > o.bytes) { g_last = "mock rccl: message longer than the receive"; c->h->failed.store(1); return 1; }
                memcpy(st[i].host.data() + st[i].done, area, n);
                st[i].done += n;
                m.consumed.fetch_add(1);
            }
        }
        if (busy && !spin(c->h->failed, t0)) { c->h->failed.store(1); g_last = "mock rccl: send/recv timed out"; return 1; }
    }
    for (size_t i = 0; i < ops.size(); ++i)
        if (!ops[i].send && deliver(ops[i].dev, st[i].host.data(), ops[i].bytes, ops[i].st)) return 1;
    return 0;
}

// every rank contributes `bytes`; fn(r, chunk_of_rank_r, n, offset) is called for r = 0 .. world-1 in rank order
template <typename F>
int exchange(Comm *c, const char *mine, size_t bytes, F &&fn)
{
    for (size_t off = 0; off < bytes || off == 0; off += kChunk) {
        const size_t n = bytes - off < kChunk ? bytes - off : kChunk;
        if (n) memcpy(c->slots + (size_t)c->rank * kChunk, mine + off, n);
        if (barrier(c)) return 1;
        for (int r = 0; r < c->world; ++r) fn(r, c->slots + (size_t)r * kChunk, n, off);
        if (barrier(c)) return 1;
        if (bytes == 0) break;
    }
    return 0;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/mfx_mock_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}

int ncclCommInitRank(void **comm, int world, ncclUniqueId id, int rank)
{
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) { g_last = "mock rccl: bad rank / world"; return 4; }
    Comm *c = new Comm();
    c->rank = rank; c->world = world; c->map_bytes = map_size(world);
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = -1;
    const double t0 = now();
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { g_last = "mock rccl: shm_open failed"; return 2; }
    } else {
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0 || [&] { struct stat s; return fstat(fd, &s) != 0 || (size_t)s.st_size < c->map_bytes; }()) {
            if (fd >= 0) { close(fd); fd = -1; }
            if (now() - t0 > 120.0) { g_last = "mock rccl: rank 0 never created the segment"; return 2; }
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_last = "mock rccl: mmap failed"; return 2; }
    c->h = (Header *)p;                                   // a fresh segment is zero-filled: valid initial state
    c->slots = (char *)p + sizeof(Header);
    c->boxes = c->slots + (size_t)world * kChunk;
    if (rank == 0) c->h->world.store(world);
    c->h->attached.fetch_add(1);
    while (c->h->attached.load() < world)
        if (now() - t0 > 120.0) { g_last = "mock rccl: not every rank attached"; return 2; }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    reap(true);
    if (c->h->attached.fetch_sub(1) == 1) shm_unlink(c->name);
    munmap((void *)c->h, c->map_bytes);
    delete c;
    return 0;
}

static size_t tsize(int dtype) { return dtype == 8 ? 8 : 0; }      // ncclFloat64 only: all libmfx uses

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (tsize(dtype) != 8 || (op != 0 && op != 2)) { g_last = "mock rccl: f64 sum / max only"; return 4; }
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    std::vector<double> mine(count), acc(count);
    if (count && hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int rc = exchange(c, (const char *)mine.data(), count * 8, [&](int r, const char *chunk, size_t n, size_t off) {
        const double *x = (const double *)chunk;
        double *a = acc.data() + off / 8;
        for (size_t i = 0; i < n / 8; ++i) a[i] = r == 0 ? x[i] : (op == 0 ? a[i] + x[i] : (x[i] > a[i] ? x[i] : a[i]));
    });
    if (rc) return 1;
    return deliver(recv, acc.data(), count * 8, st);
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (tsize(dtype) != 8) return 4;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    std::vector<double> mine(count), all(count * c->world);
    if (count && hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int rc = exchange(c, (const char *)mine.data(), count * 8, [&](int r, const char *chunk, size_t n, size_t off) {
        memcpy((char *)all.data() + (size_t)r * count * 8 + off, chunk, n);
    });
    if (rc) return 1;
    return deliver(recv, all.data(), all.size() * 8, st);
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (tsize(dtype) != 8 || root < 0 || root >= c->world) return 4;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    std::vector<double> mine(count), out(count);
    if (count && c->rank == root && hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int rc = exchange(c, (const char *)mine.data(), count * 8, [&](int r, const char *chunk, size_t n, size_t off) {
        if (r == root) memcpy((char *)out.data() + off, chunk, n);
    });
    if (rc) return 1;
    return deliver(recv, out.data(), count * 8, st);
}

int ncclGroupStart() { ++g_group; return 0; }

int ncclGroupEnd()
{
    if (--g_group > 0) return 0;
    int rc = 0;
    if (!g_ops.empty()) rc = run_p2p(g_group_comm, g_ops);
    g_ops.clear();
    return rc;
}

static int p2p(bool send, void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (tsize(dtype) != 8 || peer < 0 || peer >= c->world || peer == c->rank) { g_last = "mock rccl: bad peer"; return 4; }
    g_group_comm = c;
    g_ops.push_back({send, (char *)buf, count * 8, peer, st});
    if (g_group == 0) {
        int rc = run_p2p(c, g_ops);
        g_ops.clear();
        return rc;
    }
    return 0;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return p2p(true, (void *)buf, count, dtype, peer, comm, st); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return p2p(false, buf, count, dtype, peer, comm, st); }

const char *ncclGetErrorString(int code) { return code == 0 ? "success" : g_last; }
const char *ncclGetLastError(void *) { return g_last; }

}  // extern "C"
