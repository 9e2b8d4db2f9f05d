// faqcs_kmer_kernel.hip -- kmer_count: canonical k-mer counting into a device hash table (gfx950, wave64).
//
// Replaces update_kmer() (trim.cpp:887-931) and the std::unordered_map<size_t,size_t> tables
// (trim.cpp:82,133-135) whose serial merge dominates the reference's k-mer mode.  One wavefront per read,
// lane = position inside a 64-base chunk.  Per chunk three ballots give the 2-bit code planes (bit0, bit1)
// and the "valid ACGT inside the kept window" plane as 64-bit scalars; lane l extracts the k-bit windows
// ending at its position with 64-bit funnel shifts -- no LDS, no per-base loop.
//
// Key encoding: the reference keys its map by min(w, comp) of 2-bit-packed words.  Only the PARTITION of
// k-mer occurrences into {k-mer, reverse complement} classes is observable (distinct / total / histogram of
// counts), so any injective encoding with a consistent class representative yields identical integers.
// Here enc = plane1 << 32 | plane0 (window bit t = t-th base), rc = reversed planes with plane0 inverted
// (codes A=0,T=1,C=2,G=3: complement flips bit0, trim.cpp:904-917), key = min(enc, enc_rc).
#include "faqcs_dev.h"

// One 16-byte slot per key so that a probe touches ONE 64-byte sector (separate key / count arrays cost two random sectors
// per insert).  An empty slot is {key = ~0, count_m1 = 0, first_epoch = ~0} (kmer_table_init): the count is stored MINUS ONE,
// so the compare-and-swap that claims a slot already leaves the right count for a key seen once.
// The path is bound by the chip's L2 atomic rate (~14 G atomics/s measured, independent of table size and key reuse), so
// an insert costs ONE atomic wherever possible: a plain 16-byte load classifies the slot first (keys never change once
// written, so a stale view can only say "empty" and fall through to the CAS); a new key costs the CAS only, a known key
// the count add only, and the epoch min is issued only when it would lower the stored epoch.
struct __attribute__((aligned(16))) KmerSlot {
    unsigned long long key;
    uint32_t count_m1;     // occurrences - 1
    uint32_t first_epoch;  // owner-partitioned (multi-GPU) mode: smallest epoch that inserted the key
};
struct KmerTable {
    KmerSlot *slots;           // [mask + 1]
    uint64_t mask;             // slots - 1
    unsigned long long *stats; // [0] distinct keys, [1] total occurrences, [2] overflow flag
    uint32_t partitioned;      // maintain first_epoch
};

// multi-GPU exchange buffers of one submission (owner-partitioned mode)
struct KmerOutbox {
    ulonglong2 *items;              // (key, epoch) pairs, grouped by destination rank
    unsigned long long *dest_count; // [world]  occurrences per destination (pass 1)
    unsigned long long *dest_offset;// [world]  exclusive prefix of dest_count
    unsigned long long *dest_cursor;// [world]  (unused by the kernels; kept zero)
    uint32_t world;
    // The fill pass takes NO atomics: the count pass leaves every wave's per-destination count in wave_count, a scan turns
    // them into wave_offset (start of the wave's slice inside the destination's bucket), and a wave then advances private
    // cursors.  Both passes use the same grid per launch, so a wave sees the same reads in both.
    uint32_t *wave_count;            // [total waves of the submission][world]
    unsigned long long *wave_offset; // same shape
    uint32_t total_waves;
};

__device__ __forceinline__ uint64_t kmer_mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
// owner rank of a key: the HIGH half of the mix (the slot index uses the low bits), multiply-shift into [0, world)
__device__ __forceinline__ uint32_t kmer_owner(uint64_t key, uint32_t world)
{
    return (uint32_t)(((kmer_mix(key) >> 32) * (uint64_t)world) >> 32);
}

// bits [p-k+1, p] (p = 64*c + lane) of the bit string whose 64-bit words are ... prev, cur
__device__ __forceinline__ uint32_t window_bits(uint64_t cur, uint64_t prev, int lane, int k)
{
    const int lo = lane - (k - 1); // first bit relative to cur's bit 0 (may be negative: comes from prev)
    uint64_t w;
    if (lo >= 0) w = cur >> lo;
    else w = (cur << (-lo)) | (prev >> (64 + lo));
    return (uint32_t)(w & ((1ull << k) - 1ull));
}

// Calls emit(ok, key) once per 64-base chunk of read r in EVERY lane (ok = a canonical k-mer ends at this lane's
// position), so emit may use wave-wide ballots.
template <class F>
__device__ __forceinline__ void kmer_enumerate(const DevParams &P, const uint32_t k, const uint8_t *__restrict__ seq,
                                               const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off,
                                               const uint32_t r, const uint2 *__restrict__ results, const int lane, F &&emit)
{
    const uint32_t o = off[r];
    const int len = (int)(off[r + 1] - o);
    int a = 0, n = len;
    if (!P.qc_only) { // trimmed read of a valid record (trim.cpp:545-547); raw read under --qc_only (:260-262)
        const uint2 res = results[r];
        if (!(res.y & FAQCS_F_VALID)) return;
        a = (int)(res.x & 0xffffu);
        n = (int)(res.x >> 16);
    }
    uint64_t pv = 0, p0 = 0, p1 = 0;
    const int c_begin = a >> 6, c_end = (a + n + 63) >> 6;
#pragma unroll 1
    for (int c = c_begin; c < c_end; ++c) {
        const int p = c * 64 + lane;
        const bool in = p >= a && p < a + n;
        uint32_t b = in ? seq[(size_t)o + p] : 0u;
        if (in && !P.qc_only && P.replace_q > 0 && b == 'G') { // G -> N precedes k-mer counting (trim.cpp:390-403)
            int qv = (int)(int8_t)qual[(size_t)o + p] - P.in_off;
            qv = qv < 0 ? 0 : qv;
            if (qv < (int)P.replace_q) b = 'N';
        }
        const uint32_t l = b | 0x20u;
        const bool isA = l == 'a', isT = l == 't', isC = l == 'c', isG = l == 'g';
        const uint64_t cv = __ballot(isA | isT | isC | isG);
        const uint64_t c0 = __ballot(isT | isG); // codes A=0 T=1 C=2 G=3 (FaQCs.h:35-42)
        const uint64_t c1 = __ballot(isC | isG);
        const uint32_t wv = window_bits(cv, pv, lane, (int)k);
        const uint32_t w0 = window_bits(c0, p0, lane, (int)k);
        const uint32_t w1 = window_bits(c1, p1, lane, (int)k);
        pv = cv; p0 = c0; p1 = c1;
        const uint32_t kmask = (uint32_t)((1ull << k) - 1ull);
        const bool ok = wv == kmask; // k valid bases ending here (word_len >= k, trim.cpp:924)
        const uint32_t r0 = __brev(~w0 & kmask) >> (32 - k), r1 = __brev(w1) >> (32 - k);
        const uint64_t fwd = ((uint64_t)w1 << 32) | w0, rc = ((uint64_t)r1 << 32) | r0;
        emit(ok, fwd < rc ? fwd : rc);
    }
}

// The table's atomics, device scope.  (Checked in round 2: workgroup scope compiles to the SAME instructions on gfx950 -- the
// atomics carry no scope bit below "device" -- and the counters show every one of them leaving the XCD's L2 for the memory
// side (TCC_EA0_ATOMIC == TCC_ATOMIC, profiles/r2c/pmc_kmer_atomics.txt): with eight L2s that is where device-wide atomicity
// lives.  There is no cheaper L2-local atomic to route XCD-partitioned slots to.)
#define FAQCS_KMER_SCOPE __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned long long slot_cas(unsigned long long *p, unsigned long long expect, unsigned long long v)
{
    __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FAQCS_KMER_SCOPE);
    return expect;
}
__device__ __forceinline__ void slot_add(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, FAQCS_KMER_SCOPE); }
__device__ __forceinline__ void slot_min(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, FAQCS_KMER_SCOPE); }

// open-addressing insert; returns false when the probe budget is exhausted (table full)
__device__ __forceinline__ bool kmer_insert(const KmerTable &T, const uint64_t key, const uint32_t epoch, bool &is_new)
{
    uint64_t h = kmer_mix(key) & T.mask;
    is_new = false;
#pragma unroll 1
    for (uint32_t probe = 0; probe < 4096; ++probe) {
        KmerSlot *sl = &T.slots[h];
        // device-scope relaxed loads (bypass the per-CU L1): one 16-byte read of the slot
        typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));
        const ull2_t cur = __builtin_nontemporal_load(reinterpret_cast<const ull2_t *>(sl));
        unsigned long long seen = cur.x;
        if (seen == ~0ull) {
            seen = slot_cas(&sl->key, ~0ull, (unsigned long long)key);
            if (seen == ~0ull) { // claimed: count_m1 = 0 already says "seen once"
                is_new = true;
                if (T.partitioned) slot_min(&sl->first_epoch, epoch);
                return true;
            }
            if (seen == key) { // lost the race against the same key
                slot_add(&sl->count_m1, 1u);
                if (T.partitioned) slot_min(&sl->first_epoch, epoch);
                return true;
            }
        } else if (seen == key) {
            slot_add(&sl->count_m1, 1u);
            if (T.partitioned && epoch < (uint32_t)(cur.y >> 32)) slot_min(&sl->first_epoch, epoch);
            return true;
        }
        h = (h + 1) & T.mask;
    }
    return false;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_count(
    const DevParams P, const uint32_t k, const KmerTable T, const uint8_t *__restrict__ seq,
    const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin, const uint32_t r_end,
    const uint2 *__restrict__ results)
{
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * NW;
    unsigned long long my_total = 0, my_new = 0;
    bool full = false;
#pragma unroll 1
    for (uint32_t r = r_begin + blockIdx.x * NW + wave; r < r_end; r += n_waves) {
        kmer_enumerate(P, k, seq, qual, off, r, results, lane, [&](bool ok, uint64_t key) {
            if (ok) {
                bool is_new;
                if (kmer_insert(T, key, 0u, is_new)) { ++my_total; my_new += is_new ? 1u : 0u; }
                else full = true;
            }
        });
    }
    // one atomic per wave for the two rarefaction sums
    const unsigned long long tot = (unsigned long long)wave_sum_i32((int)my_total);
    const unsigned long long nw = (unsigned long long)wave_sum_i32((int)my_new);
    if (lane == 0) {
        if (nw) atomicAdd(&T.stats[0], nw);
        if (tot) atomicAdd(&T.stats[1], tot);
    }
    if (__any(full) && lane == 0) atomicOr(&T.stats[2], 1ull);
}

// ---- owner-partitioned (multi-GPU) mode -----------------------------------------------------------------------
// distinct counts are not additive, so every canonical k-mer has ONE owner rank
// (kmer_owner).  A rank enumerates the k-mers of its shard, buckets (key, epoch) by owner (two passes: count, then
// fill at device-computed offsets), the buckets travel by all-to-all, and the owner inserts them keeping the smallest
// epoch per key.  epoch = index of the first rarefaction point that includes the read's trim() call.
template <int NW, bool FILL>
__global__ __launch_bounds__(NW * 64) void kmer_extract(
    const DevParams P, const uint32_t k, const KmerOutbox O, const uint8_t *__restrict__ seq,
    const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin, const uint32_t r_end,
    const uint2 *__restrict__ results, const uint32_t epoch, const uint32_t wave_base)
{
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * NW;
    const uint32_t gw = wave_base + blockIdx.x * NW + (uint32_t)wave; // this wave's row in wave_count / wave_offset
    const uint64_t lt = (1ull << lane) - 1ull;
    unsigned long long my_count = 0; // pass 1: lane d accumulates the count for destination d
    unsigned long long cursor = 0;   // pass 2: lane d holds the next free item index of destination d for this wave
    if (FILL && (uint32_t)lane < O.world) cursor = O.wave_offset[(size_t)gw * O.world + lane];
#pragma unroll 1
    for (uint32_t r = r_begin + blockIdx.x * NW + wave; r < r_end; r += n_waves) {
        kmer_enumerate(P, k, seq, qual, off, r, results, lane, [&](bool ok, uint64_t key) {
            const uint32_t dest = ok ? kmer_owner(key, O.world) : 0xffffffffu;
#pragma unroll 1
            for (uint32_t d = 0; d < O.world; ++d) {
                const uint64_t m = __ballot(dest == d);
                if (m == 0) continue;
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (!FILL) {
                    if ((uint32_t)lane == d) my_count += cnt;
                } else {
                    const unsigned long long base = (unsigned long long)__shfl((long long)cursor, (int)d);
                    if (dest == d) O.items[base + (unsigned long long)__popcll(m & lt)] = make_ulonglong2(key, (unsigned long long)epoch);
                    if ((uint32_t)lane == d) cursor += cnt;
                }
            }
        });
    }
    if (!FILL && (uint32_t)lane < O.world) {
        O.wave_count[(size_t)gw * O.world + lane] = (uint32_t)my_count;
        if (my_count) atomicAdd(&O.dest_count[lane], my_count);
    }
}

// wave_offset[gw][d] = dest_offset[d] + sum of wave_count[gw'][d] over gw' < gw.  One block per destination.
__global__ __launch_bounds__(1024) void kmer_outbox_wave_offsets(const KmerOutbox O)
{
    __shared__ unsigned long long part[1024];
    const uint32_t d = blockIdx.x, tid = threadIdx.x;
    const uint32_t per = (O.total_waves + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = lo + per < O.total_waves ? lo + per : O.total_waves;
    unsigned long long sum = 0;
    for (uint32_t g = lo; g < hi; ++g) sum += O.wave_count[(size_t)g * O.world + d];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { // 1024 partial sums: a serial exclusive scan is plenty
        unsigned long long run = O.dest_offset[d];
        for (uint32_t i = 0; i < 1024; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    unsigned long long run = part[tid];
    for (uint32_t g = lo; g < hi; ++g) { O.wave_offset[(size_t)g * O.world + d] = run; run += O.wave_count[(size_t)g * O.world + d]; }
}

__global__ void kmer_outbox_offsets(const KmerOutbox O)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long o = 0;
        for (uint32_t d = 0; d < O.world; ++d) { O.dest_offset[d] = o; o += O.dest_count[d]; O.dest_cursor[d] = 0; }
    }
}

// owner side: insert received (key, epoch) pairs; tot_by_epoch[e] += occurrences (wave-aggregated per distinct epoch)
__global__ __launch_bounds__(256) void kmer_insert_items(const KmerTable T, const ulonglong2 *__restrict__ items,
                                                         const unsigned long long n, unsigned long long *tot_by_epoch,
                                                         const uint32_t n_epochs)
{
    const int lane = threadIdx.x & 63;
    unsigned long long my_new = 0;
    bool full = false;
    // occurrences per epoch: a wave keeps the running count of the epoch it is seeing in (uniform) registers and only
    // touches the global counter when the epoch changes -- one same-address atomic per item batch would serialise in L2
    uint32_t acc_e = 0xffffffffu;
    unsigned long long acc_n = 0;
    for (unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < n;
         i0 += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long i = i0 + lane;
        const bool on = i < n;
        ulonglong2 it = make_ulonglong2(0, 0);
        if (on) it = items[i];
        uint32_t e = on ? (uint32_t)it.y : 0xffffffffu;
        if (on) {
            bool is_new;
            if (kmer_insert(T, it.x, e, is_new)) my_new += is_new ? 1u : 0u;
            else full = true;
        }
        uint64_t todo = __ballot(on);
        while (todo) { // one atomic per distinct epoch in the wave
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t e0 = (uint32_t)__shfl((int)e, leader);
            const uint64_t m = __ballot(on && e == e0);
            if (e0 != acc_e) {
                if (lane == 0 && acc_n && acc_e < n_epochs) atomicAdd(&tot_by_epoch[acc_e], acc_n);
                acc_e = e0; acc_n = 0;
            }
            acc_n += (unsigned long long)__popcll(m);
            todo &= ~m;
        }
    }
    if (lane == 0 && acc_n && acc_e < n_epochs) atomicAdd(&tot_by_epoch[acc_e], acc_n);
    const unsigned long long nw = (unsigned long long)wave_sum_i32((int)my_new);
    if (lane == 0 && nw) atomicAdd(&T.stats[0], nw);
    if (__any(full) && lane == 0) atomicOr(&T.stats[2], 1ull);
}

// owner side: distinct_by_first_epoch[e] = number of keys whose smallest epoch is e
__global__ __launch_bounds__(256) void kmer_first_epoch_histogram(const KmerTable T, unsigned long long *hist, const uint32_t n_epochs)
{
    constexpr uint32_t LOCAL = 4096; // (block-local first: a handful of epochs from 10^8+ slots would serialise in L2)
    __shared__ uint32_t h[LOCAL];
    for (uint32_t i = threadIdx.x; i < LOCAL; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint64_t slots = T.mask + 1;
    const uint64_t per_block = (slots + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < slots ? lo + per_block : slots;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const KmerSlot sl = T.slots[i];
        if (sl.key != ~0ull) {
            const uint32_t e = sl.first_epoch;
            if (e < LOCAL && e < n_epochs) atomicAdd(&h[e], 1u);
            else if (e < n_epochs) atomicAdd(&hist[e], 1ull);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LOCAL && i < n_epochs; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// histogram of counts over the table (FaQCs.cpp:518-521): dense[c] for c < dense_n, (count) list otherwise.
// Nearly every key of a real run has one of a few hundred small counts: the block counts those in LDS first (global
// atomics on a few hundred addresses from 10^8..10^9 slots serialise in L2 -- seconds on a 34 GB table).
__global__ __launch_bounds__(256) void kmer_count_histogram(const KmerTable T, unsigned long long *dense, uint32_t dense_n,
                                                            unsigned long long *big, unsigned long long *n_big, uint32_t big_cap)
{
    constexpr uint32_t LOCAL = 4096;
    __shared__ uint32_t h[LOCAL];
    for (uint32_t i = threadIdx.x; i < LOCAL; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint64_t slots = T.mask + 1;
    const uint64_t per_block = (slots + gridDim.x - 1) / gridDim.x; // contiguous slice per block: < 2^32 slots each
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < slots ? lo + per_block : slots;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const KmerSlot sl = T.slots[i];
        if (sl.key != ~0ull) {
            const uint32_t c = sl.count_m1 + 1u;
            if (c < LOCAL && c < dense_n) atomicAdd(&h[c], 1u);
            else if (c < dense_n) atomicAdd(&dense[c], 1ull);
            else {
                const unsigned long long s = atomicAdd(n_big, 1ull);
                if (s < big_cap) big[s] = c;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LOCAL; i += blockDim.x)
        if (h[i]) atomicAdd(&dense[i], (unsigned long long)h[i]);
}

hipError_t faqcs_launch_kmer(const DevParams &P, uint32_t k, const KmerTable &T, const uint8_t *seq, const uint8_t *qual,
                             const uint32_t *off, uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results,
                             int n_cu, hipStream_t st)
{
    if (r_end <= r_begin) return hipSuccess;
    constexpr int NW = 4;
    uint32_t grid = (r_end - r_begin + NW - 1) / NW;
    const uint32_t cap = (uint32_t)n_cu * 8u;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((kmer_count<NW>), dim3(grid), dim3(NW * 64), 0, st, P, k, T, seq, qual, off, r_begin, r_end,
                       reinterpret_cast<const uint2 *>(results));
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_histogram(const KmerTable &T, unsigned long long *dense, uint32_t dense_n,
                                       unsigned long long *big, unsigned long long *n_big, uint32_t big_cap, int n_cu,
                                       hipStream_t st)
{
    hipLaunchKernelGGL(kmer_count_histogram, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T, dense, dense_n, big, n_big, big_cap);
    return hipGetLastError();
}

// waves a launch over n_reads reads uses (identical for the count and the fill pass)
uint32_t faqcs_kmer_extract_waves(uint32_t n_reads, int n_cu)
{
    constexpr uint32_t NW = 4;
    uint32_t grid = (n_reads + NW - 1) / NW;
    const uint32_t cap = (uint32_t)n_cu * 8u;
    if (grid > cap) grid = cap;
    return grid * NW;
}

hipError_t faqcs_launch_kmer_extract(const DevParams &P, uint32_t k, const KmerOutbox &O, bool fill, const uint8_t *seq,
                                     const uint8_t *qual, const uint32_t *off, uint32_t r_begin, uint32_t r_end,
                                     const faqcs_read_result *results, uint32_t epoch, uint32_t wave_base, int n_cu, hipStream_t st)
{
    if (r_end <= r_begin) return hipSuccess;
    constexpr int NW = 4;
    const uint32_t grid = faqcs_kmer_extract_waves(r_end - r_begin, n_cu) / NW;
    if (fill)
        hipLaunchKernelGGL((kmer_extract<NW, true>), dim3(grid), dim3(NW * 64), 0, st, P, k, O, seq, qual, off, r_begin, r_end,
                           reinterpret_cast<const uint2 *>(results), epoch, wave_base);
    else
        hipLaunchKernelGGL((kmer_extract<NW, false>), dim3(grid), dim3(NW * 64), 0, st, P, k, O, seq, qual, off, r_begin, r_end,
                           reinterpret_cast<const uint2 *>(results), epoch, wave_base);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_outbox_offsets(const KmerOutbox &O, hipStream_t st)
{
    hipLaunchKernelGGL(kmer_outbox_offsets, dim3(1), dim3(64), 0, st, O);
    if (O.total_waves) hipLaunchKernelGGL(kmer_outbox_wave_offsets, dim3(O.world), dim3(1024), 0, st, O);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_insert_items(const KmerTable &T, const void *items, unsigned long long n,
                                          unsigned long long *tot_by_epoch, uint32_t n_epochs, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    unsigned long long blocks = (n + 255) / 256;
    const unsigned long long cap = (unsigned long long)n_cu * 16ull;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(kmer_insert_items, dim3((uint32_t)blocks), dim3(256), 0, st, T, reinterpret_cast<const ulonglong2 *>(items), n,
                       tot_by_epoch, n_epochs);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_first_epoch_histogram(const KmerTable &T, unsigned long long *hist, uint32_t n_epochs, int n_cu,
                                                   hipStream_t st)
{
    hipLaunchKernelGGL(kmer_first_epoch_histogram, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T, hist, n_epochs);
    return hipGetLastError();
}

__global__ void kmer_table_init(KmerSlot *slots, const uint64_t n)
{
    const ulonglong2 empty = make_ulonglong2(~0ull, 0xffffffff00000000ull); // key, {count_m1 = 0, first_epoch = ~0}
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2 *>(slots)[i] = empty;
}

hipError_t faqcs_launch_kmer_table_init(const KmerTable &T, int n_cu, hipStream_t st)
{
    hipLaunchKernelGGL(kmer_table_init, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T.slots, T.mask + 1);
    return hipGetLastError();
}
