// EXPERIMENT (round 3, second session): tools/k1_ur.hip with the step of a wave PIPELINED -- ONE memory round trip per
// step instead of three.  The old step: wait for the polls -> row loads -> wait -> compute -> row stores -> wait for the
// acknowledge -> publish -> next polls.  Here every load of a step (the rows of the ready slots, the poll of every slot's
// NEXT rating -- or the re-poll of a waiting slot --, the record three ahead) is issued in one block, ONE s_waitcnt
// vmcnt(0) returns them AND acknowledges the row stores of the step before (loads return in issue order behind the
// stores), the granules of the step before are published behind that wait, then the arithmetic and the row stores.
// A poll that matched stays valid (only this rating's own update advances the counter), so polling a rating ahead is
// sound; one that missed is simply repeated.  Same protocol otherwise (sc1 rows, drained, then the sc1 granule).
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k1_pipe.hip -o /tmp/k1_pipe
// /tmp/k1_pipe <nu> <ni> <nnz> [epochs] [shuffled]
#include <algorithm>
#include <chrono>
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

struct __attribute__((aligned(16))) URec { int item; int seq; double r; };
struct __attribute__((aligned(16))) IHdr { double ib; int cnt; int pad; };
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

#ifndef UR_AUX
#define UR_AUX 16          // sc1
#endif
__device__ __forceinline__ v4u ld16c(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, UR_AUX); }
__device__ __forceinline__ void st16c(v4u v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, UR_AUX); }

constexpr int K = 64, KP = 32, SPW = 2, STRIDE = 66;
#ifndef HL
#define HL 16          // bytes per granule slot: 16, or 64 = a whole line written by four lanes (no partial-line write)
#endif
constexpr int SPIN_LIMIT = 1 << 18;

// "resident" side R (rows + biases in registers for a whole run), "travelling" side T (rows through the coherent path,
// bias + chain counter in one 16-byte granule).  RES_ITEM: the resident side is the item side.
// one step of the B slots of a wave.  FULL: every slot advances (no branches around memory operations).
template <int B, bool FULL, bool RES_ITEM>
__device__ __forceinline__ void res_step(unsigned long long mask, bool ready, const URec &rec, double tbv, double &rbv, double &se,
                                         d2 (&vv)[B / SPW], double *prod, __amdgpu_buffer_rsrc_t rT, int lane, int g, int q,
                                         double lr, double Kres, double Ktrav, double Kb, double &tbn)
{
    constexpr int NR = B / SPW;
    unsigned uoff[NR];
    bool rdy[NR];
    d2 uu[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int src = j * SPW + g;
        const int itm = __shfl(rec.item, src, 64);
        uoff[j] = (unsigned)itm * (unsigned)(K * 8) + (unsigned)q * 16u;
        rdy[j] = FULL || ((mask >> src) & 1ull);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        uu[j] = d2{0.0, 0.0};
        if (FULL || rdy[j]) uu[j] = __builtin_bit_cast(d2, ld16c(rT, uoff[j]));
    }
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (FULL || rdy[j])
            *reinterpret_cast<d2 *>(prod + (j * SPW + g) * STRIDE + 2 * q) = d2{uu[j].x * vv[j].x, uu[j].y * vv[j].y};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double grad = 0.0;
    if (FULL ? (lane < B) : ready) {
        double sum = RES_ITEM ? (0.0 + rbv + tbv) : (0.0 + tbv + rbv);   // kmf_train.pyx:92: item bias first
        const double *row = prod + lane * STRIDE;
#pragma unroll 2
        for (int f = 0; f < K; f += 8) {
            const d2 a = *reinterpret_cast<const d2 *>(row + f), b = *reinterpret_cast<const d2 *>(row + f + 2);
            const d2 c = *reinterpret_cast<const d2 *>(row + f + 4), d = *reinterpret_cast<const d2 *>(row + f + 6);
            sum += a.x; sum += a.y; sum += b.x; sum += b.y; sum += c.x; sum += c.y; sum += d.x; sum += d.y;
        }
        const double e = rec.r - sum;                        // kmf_train.pyx:251-256
        grad = e;
        se += e * e;
        rbv = rbv + lr * (grad - Kb * rbv);                  // :259-260
        tbn = tbv + lr * (grad - Kb * tbv);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double gr = shfl_f64(grad, j * SPW + g);
        if (FULL || rdy[j]) {
            const d2 mf = uu[j], cf = vv[j];
            d2 un, vn;
            un.x = mf.x + lr * (gr * cf.x - Ktrav * mf.x);   // :267-270
            un.y = mf.y + lr * (gr * cf.y - Ktrav * mf.y);
            vn.x = cf.x + lr * (gr * mf.x - Kres * cf.x);
            vn.y = cf.y + lr * (gr * mf.y - Kres * cf.y);
            st16c(__builtin_bit_cast(v4u, un), rT, uoff[j]);
            vv[j] = vn;
        }
    }
}

// STATIC: every wave owns fixed entities for the whole epoch (run index = slot * waves + wave): any input order, needs the
// whole grid resident.  !STATIC: runs taken by ticket in input order (input grouped by the resident side).
// OVL: the polls of the next heads are issued before the wait for the row stores' acknowledgement.
template <int B, bool STATIC, bool RES_ITEM, bool OVL, bool NOPOLL, int MINW, int SLEEP = 1>
__global__ __launch_bounds__(64, MINW) void k1_res(const URec *__restrict__ recs, const int64_t *__restrict__ run_off,
                                            const int *__restrict__ run_ent, int n_runs, double *Tm, double *Rm, IHdr *hdr,
                                            double *rb, int *ticket, double *se_run, int *err, double lr, double Kres, double Ktrav,
                                            double Kb)
{
    constexpr int NR = B / SPW;
    __shared__ __attribute__((aligned(16))) double prod[B * STRIDE];
    const int lane = threadIdx.x, g = lane / KP, q = lane % KP;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(Tm, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 0x7fffffff, 0x27000);
    constexpr unsigned long long ALL = (B == 64) ? ~0ull : ((1ull << B) - 1ull);
    for (int pass = 0;; ++pass) {
        int64_t run = -1;
        if (STATIC) {
            if (pass) break;
            run = (int64_t)lane * gridDim.x + blockIdx.x;
        } else {
            int b = 0;
            if (lane == 0) b = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0x7fffffff : atomicAdd(ticket, 1);
            b = __builtin_amdgcn_readfirstlane(b);
            if ((int64_t)b * B >= n_runs) break;
            run = (int64_t)b * B + lane;
        }
        int pos = 0, end = 0;                              // relative to the run's first record
        int64_t base = 0;
        int my_ent = 0;
        bool have = false;
        if (lane < B && run < n_runs) { base = run_off[run]; end = (int)(run_off[run + 1] - base); my_ent = run_ent[run]; have = true; }
        const URec *rp = recs + base;
        URec rec{0, 0, 0.0}, nxt{0, 0, 0.0};
        if (pos < end) rec = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(rp + pos)));
        if (pos + 1 < end) nxt = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(rp + pos + 1)));
        double rbv = have ? rb[my_ent] : 0.0;
        d2 vv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            vv[j] = d2{0.0, 0.0};
            if (hv) vv[j] = *reinterpret_cast<const d2 *>(Rm + (size_t)ent * K + 2 * q);
        }
        double se = 0.0;
        int spins = 0;
        bool bail = false;
        IHdr h; h.ib = 0.0; h.cnt = -1; h.pad = 0;
        if (pos < end) h = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)rec.item * (unsigned)HL));
        while (__any(pos < end)) {
            const bool act = pos < end;
            const bool ready = act && (NOPOLL || h.cnt == rec.seq);
            const unsigned long long mask = __ballot(ready);
            if (!mask) {
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1); bail = true; break; }
                __builtin_amdgcn_s_sleep(SLEEP);
                if (act) h = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)rec.item * (unsigned)HL));
                continue;
            }
            double tbn = 0.0;
            if (!STATIC && mask == ALL) res_step<B, true, RES_ITEM>(mask, ready, rec, h.ib, rbv, se, vv, prod, rT, lane, g, q, lr, Kres, Ktrav, Kb, tbn);
            else res_step<B, false, RES_ITEM>(mask, ready, rec, h.ib, rbv, se, vv, prod, rT, lane, g, q, lr, Kres, Ktrav, Kb, tbn);
            IHdr hn; hn.ib = 0.0; hn.cnt = -1; hn.pad = 0;
            const bool more = ready ? (pos + 1 < end) : act;
            const int nitem = ready ? nxt.item : rec.item;
            if (OVL) { if (more) hn = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)nitem * (unsigned)HL)); }
            // publish: every row store of this wave has completed, then bias + chain counter in one granule
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (HL == 64) {
                // whole-line publish: four lanes write the 64-byte slot of one granule
#pragma unroll
                for (int hh = 0; hh < (B * 4 + 63) / 64; ++hh) {
                    const int sl = (hh * 64 + lane) >> 2;
                    const int it = __shfl(rec.item, sl & 63, 64), cn = __shfl(rec.seq, sl & 63, 64);
                    const double bb = shfl_f64(tbn, sl & 63);
                    if (sl < B && ((mask >> sl) & 1ull)) {
                        IHdr nh; nh.ib = bb; nh.cnt = cn + 1; nh.pad = 0;
                        st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)it * 64u + (unsigned)(lane & 3) * 16u);
                    }
                }
            }
            if (ready) {
                if (HL != 64) {
                    IHdr nh; nh.ib = tbn; nh.cnt = rec.seq + 1; nh.pad = 0;
                    st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)rec.item * (unsigned)HL);
                }
                ++pos;
                rec = nxt;
                if (pos + 1 < end) nxt = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(rp + pos + 1)));
            }
            if (!OVL) { if (more) hn = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)nitem * (unsigned)HL)); }
            h = hn;
            spins = 0;
        }
        if (bail) break;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            if (hv) *reinterpret_cast<d2 *>(Rm + (size_t)ent * K + 2 * q) = vv[j];
        }
        if (have) { rb[my_ent] = rbv; se_run[run] = se; }
    }
}


// ---- pipelined step: loads of a step in one block (res_load), arithmetic + stores behind the wait (res_apply) --------------
template <int B, bool FULL>
__device__ __forceinline__ void res_load(unsigned long long mask, const URec &rec, d2 (&uu)[B / SPW], __amdgpu_buffer_rsrc_t rT, int g, int q)
{
    constexpr int NR = B / SPW;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int src = j * SPW + g;
        const int itm = __shfl(rec.item, src, 64);
        const unsigned o = (unsigned)itm * (unsigned)(K * 8) + (unsigned)q * 16u;
        uu[j] = d2{0.0, 0.0};
        if (FULL || ((mask >> src) & 1ull)) uu[j] = __builtin_bit_cast(d2, ld16c(rT, o));
    }
}
template <int B, bool FULL, bool RES_ITEM>
__device__ __forceinline__ void res_apply(unsigned long long mask, bool ready, const URec &rec, double tbv, double &rbv, double &se,
                                          d2 (&vv)[B / SPW], d2 (&uu)[B / SPW], double *prod, __amdgpu_buffer_rsrc_t rT, int lane, int g, int q,
                                          double lr, double Kres, double Ktrav, double Kb, double &tbn)
{
    constexpr int NR = B / SPW;
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (FULL || ((mask >> (j * SPW + g)) & 1ull))
            *reinterpret_cast<d2 *>(prod + (j * SPW + g) * STRIDE + 2 * q) = d2{uu[j].x * vv[j].x, uu[j].y * vv[j].y};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double grad = 0.0;
    if (FULL ? (lane < B) : ready) {
        double sum = RES_ITEM ? (0.0 + rbv + tbv) : (0.0 + tbv + rbv);
        const double *row = prod + lane * STRIDE;
#pragma unroll 2
        for (int f = 0; f < K; f += 8) {
            const d2 a = *reinterpret_cast<const d2 *>(row + f), b = *reinterpret_cast<const d2 *>(row + f + 2);
            const d2 c = *reinterpret_cast<const d2 *>(row + f + 4), d = *reinterpret_cast<const d2 *>(row + f + 6);
            sum += a.x; sum += a.y; sum += b.x; sum += b.y; sum += c.x; sum += c.y; sum += d.x; sum += d.y;
        }
        const double e = rec.r - sum;
        grad = e;
        se += e * e;
        rbv = rbv + lr * (grad - Kb * rbv);
        tbn = tbv + lr * (grad - Kb * tbv);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double gr = shfl_f64(grad, j * SPW + g);
        const unsigned o = (unsigned)__shfl(rec.item, j * SPW + g, 64) * (unsigned)(K * 8) + (unsigned)q * 16u;
        if (FULL || ((mask >> (j * SPW + g)) & 1ull)) {
            const d2 mf = uu[j], cf = vv[j];
            d2 un, vn;
            un.x = mf.x + lr * (gr * cf.x - Ktrav * mf.x);
            un.y = mf.y + lr * (gr * cf.y - Ktrav * mf.y);
            vn.x = cf.x + lr * (gr * mf.x - Kres * cf.x);
            vn.y = cf.y + lr * (gr * mf.y - Kres * cf.y);
            st16c(__builtin_bit_cast(v4u, un), rT, o);
            vv[j] = vn;
        }
    }
}

__device__ __forceinline__ URec ldrec_nt(const URec *p) { return __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(p))); }

template <int B, bool STATIC, bool RES_ITEM, bool NOPOLL, int MINW, int SLEEP = 1>
__global__ __launch_bounds__(64, MINW) void k1_pipe(const URec *__restrict__ recs, const int64_t *__restrict__ run_off,
                                             const int *__restrict__ run_ent, int n_runs, double *Tm, double *Rm, IHdr *hdr,
                                             double *rb, int *ticket, double *se_run, int *err, double lr, double Kres, double Ktrav,
                                             double Kb)
{
    constexpr int NR = B / SPW;
    __shared__ __attribute__((aligned(16))) double prod[B * STRIDE];
    const int lane = threadIdx.x, g = lane / KP, q = lane % KP;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(Tm, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 0x7fffffff, 0x27000);
    constexpr unsigned long long ALL = (B == 64) ? ~0ull : ((1ull << B) - 1ull);
    for (int pass = 0;; ++pass) {
        int64_t run = -1;
        if (STATIC) {
            if (pass) break;
            run = (int64_t)lane * gridDim.x + blockIdx.x;
        } else {
            int b = 0;
            if (lane == 0) b = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0x7fffffff : atomicAdd(ticket, 1);
            b = __builtin_amdgcn_readfirstlane(b);
            if ((int64_t)b * B >= n_runs) break;
            run = (int64_t)b * B + lane;
        }
        int pos = 0, end = 0;
        int64_t base = 0;
        int my_ent = 0;
        bool have = false;
        if (lane < B && run < n_runs) { base = run_off[run]; end = (int)(run_off[run + 1] - base); my_ent = run_ent[run]; have = true; }
        const URec *rp = recs + base;
        URec rec{0, 0, 0.0}, nxt{0, 0, 0.0};
        if (pos < end) rec = ldrec_nt(rp + pos);
        if (pos + 1 < end) nxt = ldrec_nt(rp + pos + 1);
        double rbv = have ? rb[my_ent] : 0.0;
        d2 vv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            vv[j] = d2{0.0, 0.0};
            if (hv) vv[j] = *reinterpret_cast<const d2 *>(Rm + (size_t)ent * K + 2 * q);
        }
        double se = 0.0;
        int spins = 0;
        bool bail = false;
        IHdr h; h.ib = 0.0; h.cnt = -1; h.pad = 0;
        if (pos < end) h = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)rec.item * 16u));
        // the publish that waits for its row stores' acknowledgement
        bool pend = false;
        int ptrav = 0, pcnt = 0;
        double pbias = 0.0;
        while (__any(pos < end || pend)) {
            const bool act = pos < end;
            const bool ready = act && (NOPOLL || h.cnt == rec.seq);
            const unsigned long long mask = __ballot(ready);
            if (!mask) {
                // nothing can move: acknowledge + publish what is pending, poll again
                IHdr hq; hq.ib = 0.0; hq.cnt = -1; hq.pad = 0;
                if (act) hq = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)rec.item * 16u));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (pend) {
                    IHdr nh; nh.ib = pbias; nh.cnt = pcnt; nh.pad = 0;
                    st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)ptrav * 16u);
                    pend = false;
                }
                if (__any(act)) {
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1); bail = true; break; }
                    __builtin_amdgcn_s_sleep(SLEEP);
                }
                h = hq;
                continue;
            }
            // ---- every load of the step ------------------------------------------------------------------------------------
            d2 uu[NR];
            const bool full = !STATIC && mask == ALL;
            if (full) res_load<B, true>(mask, rec, uu, rT, g, q);
            else res_load<B, false>(mask, rec, uu, rT, g, q);
            IHdr hp; hp.ib = 0.0; hp.cnt = -1; hp.pad = 0;
            const bool pollit = ready ? (pos + 1 < end) : act;
            if (pollit) hp = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)(ready ? nxt.item : rec.item) * 16u));
            URec n3{0, 0, 0.0};
            if (ready && pos + 2 < end) n3 = ldrec_nt(rp + pos + 2);      // becomes nxt behind this step
            // ---- ONE wait: the loads above are back, the row stores of the step before are acknowledged ----------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (pend) {
                IHdr nh; nh.ib = pbias; nh.cnt = pcnt; nh.pad = 0;
                st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)ptrav * 16u);
                pend = false;
            }
            double tbn = 0.0;
            if (full) res_apply<B, true, RES_ITEM>(mask, ready, rec, h.ib, rbv, se, vv, uu, prod, rT, lane, g, q, lr, Kres, Ktrav, Kb, tbn);
            else res_apply<B, false, RES_ITEM>(mask, ready, rec, h.ib, rbv, se, vv, uu, prod, rT, lane, g, q, lr, Kres, Ktrav, Kb, tbn);
            if (ready) {
                pend = true; ptrav = rec.item; pcnt = rec.seq + 1; pbias = tbn;
                ++pos;
                rec = nxt; nxt = n3;
            }
            h = hp;
            spins = 0;
        }
        if (bail) break;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            if (hv) *reinterpret_cast<d2 *>(Rm + (size_t)ent * K + 2 * q) = vv[j];
        }
        if (have) { rb[my_ent] = rbv; se_run[run] = se; }
    }
}

__global__ void k_hdr_reset(IHdr *hdr, const double *ib, int ni)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ni) { IHdr h; h.ib = ib[i]; h.cnt = 0; h.pad = 0; hdr[(size_t)i * (HL / 16)] = h; }
}
__global__ void k_hdr_finish(const IHdr *hdr, double *ib, int ni)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ni) ib[i] = hdr[(size_t)i * (HL / 16)].ib;
}

struct Dev {
    URec *d_in; int64_t *d_run; int *d_rent; int n_runs;       // grouped records of one (order, side)
    double *dU, *dV, *dib, *dub, *d_se; IHdr *d_hdr; int *d_ticket, *d_err;
    int nu, ni; int64_t nnz;
};

template <int B, bool STATIC, bool RES_ITEM, bool OVL, bool NOPOLL, int MINW, int SLEEP = 1, bool PIPE = false>
static int run_variant(const char *name, int waves, int epochs, const Dev &D, const std::vector<double> &U0, const std::vector<double> &V0,
                       const std::vector<double> &a, const std::vector<double> &c, float ms_ref)
{
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    const int n_trav = RES_ITEM ? D.nu : D.ni;
    double *Tm = RES_ITEM ? D.dV : D.dU, *Rm = RES_ITEM ? D.dU : D.dV, *tb = RES_ITEM ? D.dub : D.dib, *rb = RES_ITEM ? D.dib : D.dub;
    if (STATIC && (int64_t)waves * B < D.n_runs) { printf("%-34s skipped: %d waves x %d slots < %d entities\n", name, waves, B, D.n_runs); return 0; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemcpy(D.dU, U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(D.dV, V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(D.dib, 0, (size_t)D.ni * 8)); CK(hipMemset(D.dub, 0, (size_t)D.nu * 8)); CK(hipMemset(D.d_err, 0, 4));
    float ms = 0;
    for (int ep = 0; ep < epochs; ++ep) {
        CK(hipMemsetAsync(D.d_ticket, 0, 4));
        if (ep == epochs - 1) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_hdr_reset, dim3((n_trav + 255) / 256), dim3(256), 0, 0, D.d_hdr, tb, n_trav);
        if (PIPE)
            hipLaunchKernelGGL((k1_pipe<B, STATIC, RES_ITEM, NOPOLL, MINW, SLEEP>), dim3(waves), dim3(64), 0, 0, D.d_in, D.d_run, D.d_rent, D.n_runs, Tm, Rm,
                               D.d_hdr, rb, D.d_ticket, D.d_se, D.d_err, lr, RES_ITEM ? Ki : Ku, RES_ITEM ? Ku : Ki, Kb);
        else
            hipLaunchKernelGGL((k1_res<B, STATIC, RES_ITEM, OVL, NOPOLL, MINW, SLEEP>), dim3(waves), dim3(64), 0, 0, D.d_in, D.d_run, D.d_rent, D.n_runs, Tm, Rm,
                               D.d_hdr, rb, D.d_ticket, D.d_se, D.d_err, lr, RES_ITEM ? Ki : Ku, RES_ITEM ? Ku : Ki, Kb);
        hipLaunchKernelGGL(k_hdr_finish, dim3((n_trav + 255) / 256), dim3(256), 0, 0, D.d_hdr, tb, n_trav);
        if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    }
    CK(hipDeviceSynchronize());
    int err = 0; CK(hipMemcpy(&err, D.d_err, 4, hipMemcpyDeviceToHost));
    size_t diffU = 0, diffV = 0;
    if (!NOPOLL) {
        std::vector<double> U1(U0.size()), V1(V0.size());
        CK(hipMemcpy(U1.data(), D.dU, U1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(V1.data(), D.dV, V1.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); ++i) diffU += (memcmp(&a[i], &U1[i], 8) != 0);
        for (size_t i = 0; i < c.size(); ++i) diffV += (memcmp(&c[i], &V1[i], 8) != 0);
    }
    printf("%-34s %5d waves: %8.3f ms/epoch = %.3f G/s (level-per-launch %.3f ms = %.3f G/s) | timeout %d | differing U %zu V %zu%s\n",
           name, waves, ms, D.nnz / ms / 1e6, ms_ref, D.nnz / ms_ref / 1e6, err, diffU, diffV, NOPOLL ? "  [no waits: WRONG results, throughput bound only]" : "");
    fflush(stdout);
    return err ? 2 : 0;
}

// group the ratings by the resident side (stable), rank them in the travelling side's chains (input order)
static void prep(const std::vector<int32_t> &idx, const std::vector<double> &r, int64_t nnz, int nu, int ni, bool res_item, bool all_entities,
                 std::vector<URec> &recs, std::vector<int64_t> &run_off, std::vector<int32_t> &run_ent)
{
    const int n_res = res_item ? ni : nu, n_trav = res_item ? nu : ni;
    const int rc = res_item ? 1 : 0, tc = res_item ? 0 : 1;
    std::vector<int64_t> cnt((size_t)n_res + 1, 0);
    for (int64_t t = 0; t < nnz; ++t) ++cnt[idx[2 * t + rc] + 1];
    for (int e = 0; e < n_res; ++e) cnt[e + 1] += cnt[e];
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    std::vector<int32_t> ctr((size_t)n_trav, 0);
    recs.resize((size_t)nnz);
    for (int64_t t = 0; t < nnz; ++t) {
        const int tr = idx[2 * t + tc];
        recs[(size_t)fill[idx[2 * t + rc]]++] = URec{tr, ctr[tr]++, r[t]};
    }
    run_off.clear(); run_ent.clear();
    for (int e = 0; e < n_res; ++e)
        if (all_entities || cnt[e + 1] > cnt[e]) { run_off.push_back(cnt[e]); run_ent.push_back(e); }
    run_off.push_back(nnz);
}

int main(int argc, char **argv)
{
    const int nu = argc > 1 ? atoi(argv[1]) : 100000, ni = argc > 2 ? atoi(argv[2]) : 50000;
    const int64_t nnz = argc > 3 ? atoll(argv[3]) : 10000000;
    const int epochs = argc > 4 ? atoi(argv[4]) : 2;
    const int shuffled = argc > 5 ? atoi(argv[5]) : 0;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    if (shuffled) {
        for (int64_t t = 0; t < nnz; ++t) { idx[2 * t] = (int32_t)(rnd() % nu); idx[2 * t + 1] = (int32_t)(rnd() % ni); }
    } else {   // get_ratings(randomize_order=False): user-major, ascending items inside a user
        std::vector<int32_t> uu((size_t)nnz), ii((size_t)nnz);
        std::vector<int64_t> cnt((size_t)nu + 1, 0);
        for (int64_t t = 0; t < nnz; ++t) { uu[t] = (int32_t)(rnd() % nu); ii[t] = (int32_t)(rnd() % ni); ++cnt[uu[t] + 1]; }
        for (int u = 0; u < nu; ++u) cnt[u + 1] += cnt[u];
        std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
        for (int64_t t = 0; t < nnz; ++t) { const int64_t p = fill[uu[t]]++; idx[2 * p] = uu[t]; idx[2 * p + 1] = ii[t]; }
        for (int u = 0; u < nu; ++u) {
            std::vector<int32_t> tmp;
            for (int64_t p = cnt[u]; p < cnt[u + 1]; ++p) tmp.push_back(idx[2 * p + 1]);
            std::sort(tmp.begin(), tmp.end());
            for (int64_t p = cnt[u]; p < cnt[u + 1]; ++p) idx[2 * p + 1] = tmp[p - cnt[u]];
        }
    }
    for (int64_t t = 0; t < nnz; ++t) r[t] = 1.0 + (double)(rnd() % 5);
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    std::vector<Rec> recs_lvl((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs_lvl[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    for (int64_t l = 0; l < n_levels; ++l)
        std::sort(recs_lvl.begin() + off[l], recs_lvl.begin() + off[l + 1], [](const Rec &a, const Rec &b) { return a.item < b.item; });
    printf("nu %d ni %d nnz %lld %s: levels (= DAG depth) %lld\n", nu, ni, (long long)nnz, shuffled ? "shuffled" : "user-major", (long long)n_levels);

    std::vector<double> U0((size_t)ni * K), V0((size_t)nu * K);
    for (auto &x : U0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    for (auto &x : V0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    Rec *d_lvl; double *dU[2], *dV[2], *dib[2], *dub[2], *d_se; IHdr *d_hdr; int *d_err, *d_ticket;
    CK(hipMalloc(&d_lvl, nnz * sizeof(Rec)));
    CK(hipMemcpy(d_lvl, recs_lvl.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dU[v], U0.size() * 8)); CK(hipMalloc(&dV[v], V0.size() * 8)); CK(hipMalloc(&dib[v], (size_t)ni * 8)); CK(hipMalloc(&dub[v], (size_t)nu * 8));
    }
    CK(hipMemcpy(dU[0], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[0], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dib[0], 0, (size_t)ni * 8)); CK(hipMemset(dub[0], 0, (size_t)nu * 8));
    CK(hipMalloc(&d_se, ((size_t)nnz / 8 + (size_t)nu + ni + 8192) * 8)); CK(hipMalloc(&d_hdr, (size_t)(nu > ni ? nu : ni) * HL)); CK(hipMalloc(&d_err, 4)); CK(hipMalloc(&d_ticket, 4));
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0;
    {   // reference: the product kernel, one launch per level, on the same input
        K1Params P{}; P.recs = d_lvl; P.U = dU[0]; P.V = dV[0]; P.ib = dib[0]; P.ub = dub[0]; P.se_partial = d_se; P.k = K;
        P.upd_u = P.upd_i = 1; P.lr = lr; P.Ku = Ku; P.Ki = Ki; P.Kb = Kb;
        const size_t lds = (size_t)kWaves * 8 * k1_stride(K) * 8;
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == epochs - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int cnt = (int)(off[l + 1] - off[l]);
                const int blocks = (cnt + 8 * kWaves - 1) / (8 * kWaves);
                hipLaunchKernelGGL((k1_level<32, 2, 8, K1_LINEAR, 2, double, true>), dim3(blocks), dim3(kWaves * 64), lds, 0, P, off[l], cnt, (int64_t)0);
            }
            if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    std::vector<double> a(U0.size()), c(V0.size());
    CK(hipMemcpy(a.data(), dU[0], a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), dV[0], c.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_lvl);

    std::vector<URec> recs; std::vector<int64_t> run_off; std::vector<int32_t> run_ent;
    Dev D{}; D.dU = dU[1]; D.dV = dV[1]; D.dib = dib[1]; D.dub = dub[1]; D.d_se = d_se; D.d_hdr = d_hdr; D.d_ticket = d_ticket; D.d_err = d_err;
    D.nu = nu; D.ni = ni; D.nnz = nnz;
    CK(hipMalloc(&D.d_in, nnz * sizeof(URec))); CK(hipMalloc(&D.d_run, ((size_t)(nu > ni ? nu : ni) + 1) * 8)); CK(hipMalloc(&D.d_rent, (size_t)(nu > ni ? nu : ni) * 4));
    auto upload = [&](bool res_item, bool all) {
        prep(idx, r, nnz, nu, ni, res_item, all, recs, run_off, run_ent);
        D.n_runs = (int)run_ent.size();
        if (hipMemcpy(D.d_in, recs.data(), nnz * sizeof(URec), hipMemcpyHostToDevice) != hipSuccess) return 1;
        if (hipMemcpy(D.d_run, run_off.data(), run_off.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
        if (hipMemcpy(D.d_rent, run_ent.data(), run_ent.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
        return 0;
    };
#define RUN(B, ST, RI, OVL, NP, MINW, W, SL) if (run_variant<B, ST, RI, OVL, NP, MINW, SL>(#B " slots " #ST "/" #RI "/" #OVL "/" #NP "/" #MINW "/sleep" #SL, W, epochs, D, U0, V0, a, c, ms_ref)) return 2
#define RUNP(B, ST, RI, NP, MINW, W, SL) if (run_variant<B, ST, RI, false, NP, MINW, SL, true>("PIPE " #B " slots " #ST "/" #RI "/" #NP "/" #MINW "/sleep" #SL, W, epochs, D, U0, V0, a, c, ms_ref)) return 2
    if (!shuffled) {
        if (upload(false, false)) return 1;               // users resident, tickets
        RUN(8, false, false, false, false, 4, 4096, 1);
        RUN(8, false, false, false, true, 4, 4096, 1);
#if HL == 16
        RUNP(8, false, false, false, 4, 4096, 1);
        RUNP(8, false, false, true, 4, 4096, 1);
#endif
    }
    if ((int64_t)ni * 8 * K <= 64ll << 20) {               // items resident, static (any order)
        if (upload(true, true)) return 1;
        RUN(18, true, true, false, false, 3, 3072, 1);
        RUN(18, true, true, false, true, 3, 3072, 1);
#if HL == 16
        RUNP(18, true, true, false, 3, 3072, 1);
        RUNP(18, true, true, true, 3, 3072, 1);
#endif
    }
    return 0;
}
