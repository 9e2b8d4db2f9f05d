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
#include "faqcs_kmer.h"

// The path of kmer_count / kmer_insert_items is bound by the chip's memory-side atomic rate (~14 G atomics/s measured,
// independent of table size and key reuse), so an insert costs ONE atomic wherever possible: a plain 16-byte load classifies
// the slot first (keys never change once written, so a stale view can only say "empty" and fall through to the CAS); a new
// key costs the CAS only, a known key the count add only, and the epoch min is issued only when it would lower the stored
// epoch.  (The single-GPU path no longer inserts per occurrence at all: faqcs_kmer_skm_kernel.hip.)

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
    const uint64_t slots = kmer_table_total(T);
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
// (Read-only: round 4 cleared the live sectors in the same pass; the scattered stores between the reads cost more than kmer_table_init's
// stream of stores afterwards -- 17.5 against 7 + 6.6 ms on a 2^31-slot table.)
__global__ __launch_bounds__(256) void kmer_count_histogram(const KmerTable T, unsigned long long *dense, uint32_t dense_n,
                                                            unsigned long long *big, unsigned long long *n_big, uint32_t big_cap, const uint64_t first_slot)
{
    constexpr uint32_t LOCAL = 4096;
    __shared__ uint32_t h[LOCAL];
    for (uint32_t i = threadIdx.x; i < LOCAL; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint64_t slots = kmer_table_total(T);
    // (the grid sweeps the table front to back, every block 4 x 256 consecutive slots per turn: the DRAM sees one stream, not 2 048)
    const uint64_t hi = slots;
    // Four slots per thread in flight (the pass is a stream over the whole table: 36 GB for 2^31 slots).  Most keys of a real run have
    // been seen ONCE (sequencing errors): 64 lanes adding to h[1] serialise in the LDS, so the lanes of a wave whose key has count 1 add
    // their number once (round 5: 18.5 -> see profiles/r5*/ ms per pass on the bench's table).
    constexpr int U = 4;
    for (uint64_t i0 = first_slot + (uint64_t)blockIdx.x * U * blockDim.x + threadIdx.x; i0 < hi; i0 += (uint64_t)gridDim.x * U * blockDim.x) {
        KmerSlot sl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * blockDim.x;
            sl[u].key = ~0ull; sl[u].count_m1 = 0; sl[u].first_epoch = 0;
            if (i < hi) sl[u] = T.slots[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * blockDim.x;
            const bool live = sl[u].key != ~0ull;
            const uint32_t c = sl[u].count_m1 + 1u;
            const unsigned long long once = __ballot(live && c == 1u);
            if (once != 0ull && (int)(threadIdx.x & 63u) == __builtin_ctzll(once) && 1u < dense_n) atomicAdd(&h[1], (uint32_t)__popcll(once));
            if (live && (c != 1u || 1u >= dense_n)) {
                if (c < LOCAL && c < dense_n) atomicAdd(&h[c], 1u);
                else if (c < dense_n) atomicAdd(&dense[c], 1ull);
                else {
                    const unsigned long long s = atomicAdd(n_big, 1ull);
                    if (s < big_cap) big[s] = c;
                }
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

// overflow_only: the area behind the table alone (a pass counted in one piece leaves the table's slices empty: skm_combine<KS_REDO>)
hipError_t faqcs_launch_kmer_histogram(const KmerTable &T, unsigned long long *dense, uint32_t dense_n,
                                       unsigned long long *big, unsigned long long *n_big, uint32_t big_cap, int n_cu,
                                       hipStream_t st, bool overflow_only)
{
    hipLaunchKernelGGL(kmer_count_histogram, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T, dense, dense_n, big, n_big, big_cap,
                       overflow_only ? T.mask + 1 : 0ull);
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

hipError_t faqcs_launch_kmer_table_init(const KmerTable &T, int n_cu, hipStream_t st, bool overflow_only)
{
    if (overflow_only) { if (T.ovf_mask) hipLaunchKernelGGL(kmer_table_init, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T.slots + T.mask + 1, T.ovf_mask + 1); }
    else hipLaunchKernelGGL(kmer_table_init, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, T.slots, kmer_table_total(T));
    return hipGetLastError();
}
