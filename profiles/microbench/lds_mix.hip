// lds_mix.hip -- does a ds_add_u32 occupy the CU's LDS pipeline for its whole ~13 clocks?  Half of the waves of a block issue
// ds_add_u32 (conflict-free), the other half ds_read_b64 (conflict-free); rates alone and together.  Also: ds_add_u32 with half of
// the lanes masked off, and ds_add_u64.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_mix lds_mix.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 4000;

// mode bit 0: atomic waves active, bit 1: read waves active; amask: 0 = all lanes, 1 = lanes < 32 only, 2 = even lanes only ; wide: ds_add_u64
__global__ void k_mix(uint32_t *out, unsigned long long *clk, int mode, int amask, int wide, int ratio)
{
    extern __shared__ uint32_t sm[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = 0;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool atomic_wave = (wv % ratio) == 0;
    uint32_t a0 = 0, a1 = 0;
    uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    const uint32_t d0 = (lane * (wide ? 8u : 4u) + wv * 512u) & 0xffffu, d1 = d0 + 16384u, d2 = d0 + 32768u, d3 = d0 + 49152u - 2048u;
    const uint32_t r0 = lane * 8u + wv * 512u;
    uint32_t one = 1;
    uint64_t one64 = 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (atomic_wave) {
        if (mode & 1) {
            if (amask == 1 && lane >= 32) goto done;
            if (amask == 2 && (lane & 1)) goto done;
            for (int i = 0; i < ITERS; ++i) {
                if (!wide)
                    asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %4\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %4\n\t"
                                 "ds_add_u32 %0, %4 offset:256\n\tds_add_u32 %1, %4 offset:256\n\tds_add_u32 %2, %4 offset:256\n\tds_add_u32 %3, %4 offset:256\n\t"
                                 "s_waitcnt lgkmcnt(0)" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(one) : "memory");
                else
                    asm volatile("ds_add_u64 %0, %4\n\tds_add_u64 %1, %4\n\tds_add_u64 %2, %4\n\tds_add_u64 %3, %4\n\t"
                                 "ds_add_u64 %0, %4 offset:512\n\tds_add_u64 %1, %4 offset:512\n\tds_add_u64 %2, %4 offset:512\n\tds_add_u64 %3, %4 offset:512\n\t"
                                 "s_waitcnt lgkmcnt(0)" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(one64) : "memory");
            }
        }
    } else if (mode & 2) {
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:2048\n\tds_read_b64 %3, %4 offset:3072\n\t"
                         "ds_read_b64 %0, %4 offset:4096\n\tds_read_b64 %1, %4 offset:5120\n\tds_read_b64 %2, %4 offset:6144\n\tds_read_b64 %3, %4 offset:7168\n\t"
                         "s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(r0) : "memory");
        }
    }
done:
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3);
    if (lane == 0) clk[blockIdx.x * 16 + wv] = t1 - t0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint32_t *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)n_cu * 1024 * 4));
    CHECK(hipMalloc(&clk, (size_t)n_cu * 16 * 8));
    struct Cfg { const char *name; int mode, amask, wide, ratio, waves; };
    const Cfg cfgs[] = {
        {"12 waves: 6 ds_add_u32 waves alone", 1, 0, 0, 2, 12}, {"12 waves: 6 ds_read_b64 waves alone", 2, 0, 0, 2, 12},
        {"12 waves: 6 add + 6 read together", 3, 0, 0, 2, 12},
        {"12 waves: 3 add + 9 read together", 3, 0, 0, 4, 12}, {"12 waves: 3 add alone", 1, 0, 0, 4, 12}, {"12 waves: 9 read alone", 2, 0, 0, 4, 12},
        {"12 waves: all ds_add_u32", 1, 0, 0, 1, 12}, {"12 waves: all ds_add_u32, lanes < 32 only", 1, 1, 0, 1, 12},
        {"12 waves: all ds_add_u32, even lanes only", 1, 2, 0, 1, 12}, {"12 waves: all ds_add_u64", 1, 0, 1, 1, 12},
        {"4 waves: all ds_add_u32", 1, 0, 0, 1, 4}, {"16 waves: all ds_add_u32", 1, 0, 0, 1, 16},
    };
    for (const Cfg &c : cfgs) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_mix, dim3(n_cu), dim3(c.waves * 64), 65536, 0, out, clk, c.mode, c.amask, c.wide, c.ratio);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix, dim3(n_cu), dim3(c.waves * 64), 65536, 0, out, clk, c.mode, c.amask, c.wide, c.ratio);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        int n_add = 0, n_rd = 0;
        for (int w = 0; w < c.waves; ++w) { if ((w % c.ratio) == 0) n_add += (c.mode & 1) ? 1 : 0; else n_rd += (c.mode & 2) ? 1 : 0; }
        const double adds = (double)n_add * ITERS * 8 * n_cu, rds = (double)n_rd * ITERS * 8 * n_cu;
        {   // per role: median wave clocks (s_memtime) -> LDS clocks per wave-instruction per CU while that role was running
            static unsigned long long h[256 * 16];
            CHECK(hipMemcpy(h, clk, (size_t)n_cu * 16 * 8, hipMemcpyDeviceToHost));
            double ca = 0, cr = 0; int na = 0, nr = 0;
            for (int b = 0; b < n_cu; ++b) for (int w = 0; w < c.waves; ++w) { if ((w % c.ratio) == 0) { ca += h[b * 16 + w]; ++na; } else { cr += h[b * 16 + w]; ++nr; } }
            ca /= na ? na : 1; cr /= nr ? nr : 1;
            printf("   mean wave clocks: add waves %.0f (%.2f clk per instr per CU over %d waves)  read waves %.0f (%.2f clk per instr per CU over %d waves)\n", ca,
                   n_add ? ca / (ITERS * 8.0 * n_add) : 0.0, n_add, cr, n_rd ? cr / (ITERS * 8.0 * n_rd) : 0.0, n_rd);
        }
        printf("%-46s %.3f ms : ds_add %.1f G/s (%.2f clk/instr/CU at 2.4 GHz)  ds_read_b64 %.1f G/s (%.2f clk/instr/CU)\n", c.name, ms, adds / ms * 1e-6,
               adds > 0 ? ms * 1e-3 * 2.4e9 * n_cu / adds : 0.0, rds / ms * 1e-6, rds > 0 ? ms * 1e-3 * 2.4e9 * n_cu / rds : 0.0);
        fflush(stdout);
    }
    return 0;
}
