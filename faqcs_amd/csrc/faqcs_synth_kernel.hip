// faqcs_synth_kernel.hip -- measurement helper (bench.py): fills device arenas with the synthetic 2x150 /
// 2x250 reads of SURVEY.md section 8(d) directly in HBM (FASTQ text for 100 M pairs would be ~64 GB).
// Counter-based generator keyed by (seed, absolute read index, position, stream) so any shard of any size
// regenerates exactly its slice.  Recipe: bases iid ACGT, each N w.p. 0.002; Phred+33 qualities: first 3
// bases U[2,37], plateau U[30,40] up to a breakpoint b ~ U[L/2, L+40], tail Q2 w.p. 0.7 (per read) else
// U[3,15]; with probability adapter_frac one of the 9 built-in adapters or poly-A (options.cpp:583-625) is
// read through from a position U[40, L-10] with 5 % substitutions.  With genome_len > 0 (the k-mer configuration) a read
// is a window of a fixed synthetic genome (base at position g = hash(seed, g)), on either strand, with 0.5 % substitutions,
// so that distinct k-mers grow like they do on real data.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__device__ const char k_ad0[] = "TCGTATAACTTCGTATAATGTATGCTATACGAAGTTATTACG";
__device__ const char k_ad1[] = "AGCATATTGAAGCATATTACATACGATATGCTTCAATAATGC";
__device__ const char k_ad2[] = "GGGGTAGTGTGGATCCTCCTCTAGGCAGTTGGGTTATTCTAGAAGCAGATGTGTTGGCTGTTTCTGAAACTCTGGAAAA";
__device__ const char k_ad3[] = "CAACAGCCGGTCAAAACATCTGGAGGGTAAGCCATAAACACCTCAACAGAAAA";
__device__ const char k_ad4[] = "CGATAACTTCGTATAATGTATGCTATACGAAGTTATTACG";
__device__ const char k_ad5[] = "GCATAACTTCGTATAGCATACATTATACGAAGTTATACGA";
__device__ const char k_ad6[] = "GATCGGAAGAGCACACGTCTGAACTCCAGTCAC";
__device__ const char k_ad7[] = "GATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT";
__device__ const char k_ad8[] = "CTGTCTCTTATACACATCTAGATGTGTATAAGAGACAG";
__device__ const char k_ad9[] = "AAAAAAAAAAAAAAAAAAAA";

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t rnd(uint64_t seed, uint64_t read, uint32_t a, uint32_t b)
{
    return (uint32_t)(mix(mix(seed ^ (read * 0xd1342543de82ef95ull)) + ((uint64_t)a << 32 | b)) >> 32);
}
__device__ __forceinline__ uint32_t below(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }
} // namespace

// at_frac < 0: bases uniform over ACGT (the SURVEY recipe); otherwise A+T make up at_frac of the bases (AT-rich genomes: the
// dinucleotide-candidate path of the low-complexity filter becomes visible)
__global__ void synth_fill(uint8_t *seq, uint8_t *qual, uint32_t *offset, uint32_t n_reads, uint32_t L, uint64_t seed,
                           uint64_t first_read, float adapter_frac, uint64_t genome_len, float at_frac)
{
    const uint64_t total = (uint64_t)n_reads * L;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total + n_reads + 1; g += (uint64_t)gridDim.x * blockDim.x) {
        if (g >= total) { const uint64_t i = g - total; offset[i] = (uint32_t)(i * L); continue; }
        const uint32_t r = (uint32_t)(g / L), p = (uint32_t)(g % L);
        const uint64_t R = first_read + r;
        // per-read draws (stream 0)
        const uint32_t brk = L / 2 + below(rnd(seed, R, 0, 1), L + 41 - L / 2);
        const bool tail_q2 = rnd(seed, R, 0, 2) < (uint32_t)(0.7 * 4294967296.0);
        const bool has_ad = adapter_frac > 0.f && rnd(seed, R, 0, 3) < (uint32_t)((double)adapter_frac * 4294967296.0);
        // base
        uint32_t x = rnd(seed, R, 1, p);
        uint8_t b = "ACGT"[x & 3u];
        if (at_frac >= 0.f) b = ((x >> 8) < (uint32_t)((double)at_frac * 16777216.0)) ? "AT"[x & 1u] : "CG"[x & 1u];
        if (genome_len > L) {
            const uint64_t start = mix(mix(seed ^ 0x67656e6f6d65ull) + R) % (genome_len - L);
            const bool rc = (rnd(seed, R, 0, 6) & 1u) != 0;
            const uint64_t gp = rc ? start + (L - 1 - p) : start + p;
            const uint32_t gb = (uint32_t)(mix(seed * 0x2545f4914f6cdd1dull + gp) >> 62); // the genome's base at gp
            if (rnd(seed, R, 5, p) >= (uint32_t)(0.005 * 4294967296.0)) b = rc ? "TGCA"[gb] : "ACGT"[gb];
        }
        if (has_ad) {
            const uint32_t ai = below(rnd(seed, R, 0, 4), 10);
            const uint32_t ap = 40 + below(rnd(seed, R, 0, 5), L - 10 - 40 + 1);
            const char *ad = ai == 0 ? k_ad0 : ai == 1 ? k_ad1 : ai == 2 ? k_ad2 : ai == 3 ? k_ad3 : ai == 4 ? k_ad4
                           : ai == 5 ? k_ad5 : ai == 6 ? k_ad6 : ai == 7 ? k_ad7 : ai == 8 ? k_ad8 : k_ad9;
            const uint32_t alen = ai == 0 ? 42 : ai == 1 ? 42 : ai == 2 ? 79 : ai == 3 ? 53 : ai == 4 ? 40 : ai == 5 ? 40
                                : ai == 6 ? 33 : ai == 7 ? 32 : ai == 8 ? 38 : 20;
            if (p >= ap && p - ap < alen && rnd(seed, R, 2, p) >= (uint32_t)(0.05 * 4294967296.0)) b = (uint8_t)ad[p - ap];
        }
        if (rnd(seed, R, 3, p) < (uint32_t)(0.002 * 4294967296.0)) b = 'N';
        // quality
        const uint32_t y = rnd(seed, R, 4, p);
        uint32_t q;
        if (p >= brk) q = tail_q2 ? 2u : 3u + below(y, 13);
        else if (p < 3) q = 2u + below(y, 36);
        else q = 30u + below(y, 11);
        seq[g] = b;
        qual[g] = (uint8_t)(q + 33u);
    }
}

hipError_t faqcs_launch_synth(uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads, uint32_t L,
                              uint64_t seed, uint64_t first_read, float adapter_frac, uint64_t genome_len, float at_frac, hipStream_t st)
{
    hipLaunchKernelGGL(synth_fill, dim3(256 * 16), dim3(256), 0, st, d_seq, d_qual, d_offset, n_reads, L, seed, first_read, adapter_frac, genome_len, at_frac);
    return hipGetLastError();
}
