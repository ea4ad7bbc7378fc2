// What the memory side sustains for the resident epoch's access pattern: random 512-byte rows of a table read (16 bytes per
// lane, two rows per wave instruction) and written back, nothing else -- no dependencies, no arithmetic beyond one add, no polls.
// Variants: sc1 loads + sc1 (write-through) stores (the travelling rows of k1_res), plain loads + sc1 stores (the level kernel),
// plain + plain, read only, write only.  Table sizes: cfg 2's user rows (51 MB, Infinity-Cache resident), cfg 3's item rows
// (102 MB), cfg 5's item rows (512 MB, beyond the Infinity Cache).
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/rmw_ceiling.hip -o /tmp/rmw_ceiling && /tmp/rmw_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// NR rows in flight per half-wave (2 NR per wave), like the NR register sets of k1_res
// GR: the per-row granule traffic of k1_res on top: 1 = a 16-byte sc1 poll per row (lane s polls slot s's granule, before the rows),
// 2 = + a 16-byte sc1 publish per row behind the acknowledge wait, 3 = the publish as a whole 64-byte line (granules padded to 64 B)
template <int NR, int LD, int ST, int GR = 0>      // LD / ST: 0 = none, 1 = plain, 2 = sc1
__global__ __launch_bounds__(64) void k_rmw(double *tab, const int *__restrict__ ids, int n_ids_per_wave, int iters, double *gran = nullptr)
{
    const int lane = threadIdx.x, g = lane >> 5, q = lane & 31;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tab, 0, 0x7fffffff, 0x27000);
    const int *my = ids + (size_t)blockIdx.x * n_ids_per_wave;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(gran ? gran : tab, 0, 0x7fffffff, 0x27000);
    constexpr unsigned GS = GR == 3 ? 64u : 16u;
    d2 acc = d2{0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
        d2 v[NR];
        unsigned off[NR];
        d2 gv = d2{0.0, 0.0};
        int gid = 0;
        if (GR) {
            if (lane < 2 * NR) { gid = my[it * NR * 2 + lane]; gv = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rg, (unsigned)gid * GS, 0, 16)); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the poll's round trip
            acc.x += gv.x;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int id = my[(it * NR + j) * 2 + g];
            off[j] = (unsigned)id * 512u + (unsigned)q * 16u;
            v[j] = d2{1.0, 2.0};
            if (LD == 1) v[j] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off[j], 0, 0));
            if (LD == 2) v[j] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off[j], 0, 16));
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            v[j].x += 1.0; v[j].y += 1.0;
            if (ST == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v[j]), r, off[j], 0, 0);
            if (ST == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v[j]), r, off[j], 0, 16);
            if (ST == 0) { acc.x += v[j].x; acc.y += v[j].y; }
        }
        if (ST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the step's acknowledge wait
        if (GR == 2 && lane < 2 * NR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, gv), rg, (unsigned)gid * 16u, 0, 16);
        if (GR == 3) {                                              // four lanes write one granule's whole 64-byte line
#pragma unroll
            for (int h = 0; h < (2 * NR * 4 + 63) / 64; ++h) {
                const int sl = (h * 64 + lane) >> 2;
                const int id = __shfl(gid, sl & 63, 64);
                if (sl < 2 * NR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, gv), rg, (unsigned)id * 64u + (unsigned)(lane & 3) * 16u, 0, 16);
            }
        }
    }
    if (ST == 0 && acc.x == 12345.678) tab[0] = acc.y;
}

template <int NR, int LD, int ST, int GR = 0>
static int run(const char *name, double *d_tab, const int *d_ids, int waves, int per_wave, int iters, int reps, double *d_gran = nullptr)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rmw<NR, LD, ST, GR>), dim3(waves), dim3(64), 0, 0, d_tab, d_ids, per_wave, iters, d_gran);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rmw<NR, LD, ST, GR>), dim3(waves), dim3(64), 0, 0, d_tab, d_ids, per_wave, iters, d_gran);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = (double)waves * iters * NR * 2 * reps;
    const double bytes = rows * 512.0 * ((LD ? 1 : 0) + (ST ? 1 : 0));
    printf("  %-44s %2d rows/wave %5d waves: %7.2f G rows/s, %6.2f TB/s of row bytes\n", name, NR * 2, waves, rows / ms / 1e6, bytes / ms / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    const int waves_list[] = {3072, 4096};
    const size_t rows_list[] = {100000, 200000, 1000000};
    for (size_t n_rows : rows_list) {
        double *d_tab; CK(hipMalloc(&d_tab, n_rows * 512)); CK(hipMemset(d_tab, 0, n_rows * 512));
        const int iters = 200, NRMAX = 9, per_wave = iters * NRMAX * 2;
        const int max_waves = 8192;
        std::vector<int> ids((size_t)max_waves * per_wave);
        uint64_t st = 88172645463325252ull;
        for (auto &x : ids) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (int)(st % n_rows); }
        int *d_ids; CK(hipMalloc(&d_ids, ids.size() * 4)); CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        double *d_gran; CK(hipMalloc(&d_gran, n_rows * 64)); CK(hipMemset(d_gran, 0, n_rows * 64));
        printf("table: %zu rows x 512 B = %.0f MB, uniformly random rows\n", n_rows, n_rows * 512 / 1e6);
        for (int waves : waves_list) {
            if (run<9, 2, 2>("sc1 load + sc1 store (k1_res, 18 slots)", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<9, 2, 2, 1>("  + 16-B sc1 poll per row", d_tab, d_ids, waves, per_wave, iters, 3, d_gran)) return 1;
            if (run<9, 2, 2, 2>("  + poll + 16-B sc1 publish per row", d_tab, d_ids, waves, per_wave, iters, 3, d_gran)) return 1;
            if (run<9, 2, 2, 3>("  + poll + 64-B (whole line) publish", d_tab, d_ids, waves, per_wave, iters, 3, d_gran)) return 1;
            if (run<4, 2, 2, 2>("  8 slots + poll + 16-B sc1 publish", d_tab, d_ids, waves, per_wave, iters, 3, d_gran)) return 1;
            if (run<4, 2, 2>("sc1 load + sc1 store (k1_res, 8 slots)", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<4, 1, 2>("plain load + sc1 store (k1_level)", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<4, 1, 1>("plain load + plain store", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<4, 2, 0>("sc1 load only", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<4, 1, 0>("plain load only", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
            if (run<4, 0, 2>("sc1 store only", d_tab, d_ids, waves, per_wave, iters, 3)) return 1;
        }
        (void)hipFree(d_tab); (void)hipFree(d_ids); (void)hipFree(d_gran);
    }
    return 0;
}
