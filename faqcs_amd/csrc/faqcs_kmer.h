// faqcs_kmer.h -- structures and device helpers shared by the k-mer kernels (faqcs_kmer_kernel.hip: the table, the
// per-occurrence diagnostic path; faqcs_kmer_skm_kernel.hip: super-k-mers, combine-before-insert) and the host side (faqcs_capi.hip).
//
// Replaces update_kmer() (trim.cpp:887-931) and the std::unordered_map<size_t,size_t> tables (trim.cpp:82,133-135).
#pragma once
#include "faqcs_dev.h"

// One 16-byte slot per key so that a probe touches ONE 64-byte sector.  An empty slot is {key = ~0, count_m1 = 0,
// first_epoch = ~0} (kmer_table_init): the count is stored MINUS ONE, so the compare-and-swap that claims a slot already
// leaves the right count for a key seen once.
struct __attribute__((aligned(16))) KmerSlot {
    unsigned long long key;
    uint32_t count_m1;     // occurrences - 1
    uint32_t first_epoch;  // smallest epoch (index of the first rarefaction point that includes the key) that inserted the key
};
struct KmerTable {
    KmerSlot *slots;           // [mask + 1]
    uint64_t mask;             // slots - 1
    unsigned long long *stats; // [0] distinct keys, [1] total occurrences, [2] overflow flag
    uint32_t partitioned;      // (kmer_count / kmer_insert_items) maintain first_epoch
    uint32_t shift;            // 62 - log2(slots): combine-before-insert keys are 62-bit mixes, slot = key >> shift
    uint64_t ovf_mask;         // overflow area behind the table: slots[mask + 1 .. mask + 1 + ovf_mask] (0: none).  A key whose probe window
                               // inside its partition's slice is full lives there (faqcs_kmer_skm_kernel.hip): shared by all partitions, atomics only
    uint32_t fine;             // F = 0 .. 3: the table is cut into 2^(16 + F) slices, one per FINE partition (the top 16 + F bits of an item's
                               // 19-bit partition); a pass that is counted in one piece at its end has one workgroup turn per fine partition
    uint32_t *dirty;           // [2^(16 + F) / 32] bit p: the slice of fine partition p holds keys that reached the table occurrence by occurrence
                               // (what did not fit a sub-region of the group buffers) -- the counting pass must then go through the table for p
};
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t kmer_table_total(const KmerTable &T) { return T.mask + 1 + (T.ovf_mask ? T.ovf_mask + 1 : 0); }

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

// ---- combine-before-insert (k-mer counting; the items and their arithmetic: faqcs_skm.h) ---------------------
// A k-mer occurrence is not inserted when it is seen.  Runs of consecutive k-mers that share their minimizer travel as ONE 16-byte
// item (a super-k-mer); the partition of the key space is a function of the minimizer: 16 bits, the top 8 = one of 256 level-1 buckets
// the extraction kernels append to.  When a group of runs is full, a second scatter splits every bucket 256 ways (the low 8 bits):
// 65 536 partitions, each the ONLY holder of its keys and of the table slice they live in.  One workgroup per partition then expands
// its items, counts the keys (h = mix62 of the canonical k-mer: a bijection, so h IS the key from there on) in an LDS hash table and
// applies ONE update per DISTINCT key to the table: plain loads and stores -- a new slot is claimed through the workgroup's LDS bitmap.
//
// Round 6: a pass whose items fit the group buffers (they are sized from the free HBM) is not flushed before it ENDS.  Then the second
// scatter splits a bucket (256 << F) ways -- a sort of 8 192-item tiles in LDS -- and one workgroup turn per FINE partition counts the
// keys in LDS and adds them straight to the histogram of counts and to the keys-by-first-epoch histogram: the table is not touched at
// all (skm_split_sort / skm_combine<FINAL>).  A fine partition with more distinct keys than the LDS table takes, or one that the
// per-occurrence path has written to (KmerTable::dirty), goes through its slice of the table as before and sweeps it afterwards.
enum {
    KG_FAN = 256,          // fan-out of either scatter level
    KG_MAX_RUNS = 1000,    // extraction launches (runs of segments with one epoch) per group (an item has 10 bits for the run / epoch)
    KG_EPOCH_SPAN = 1000,  // epochs a group may span (the combine kernel's LDS histogram)
    KG_MIN_CAP = 128,      // smallest sub-region
    KG_SLICE_MAX = 65536,  // largest table slice of a partition (table <= 2^32 slots): the combine kernel's claim bitmap is 8 KB of LDS
    KG_SLICE_MIN = 64      // smallest (table >= 2^22 slots)
};
#define KG_M62 ((1ull << 62) - 1ull)

struct KmerGroupDev {
    unsigned long long *l1;   // [256 buckets][256 sub-regions][cap1]  16-byte items (faqcs_skm.h), run field = extraction launch; sub-region = the block that wrote it
    unsigned long long *l2;   // [65536 partitions][split sub-regions][cap2]  items, run field = epoch - epoch_base
                              // (a pass counted at its end: [2^(16 + F) fine partitions][cap2f], same memory)
    uint32_t *cur1;           // [256 sub-regions][256 buckets]  items a level-1 sub-region holds
    uint32_t *cur2;           // [65536][split]
    uint32_t *run_epoch;      // [KG_MAX_RUNS]  epoch of run j minus epoch_base
    uint32_t cap1, cap2;      // items per sub-region
    uint32_t cap2f;           // items per FINE partition when the level-2 buffer is cut 2^(16 + F) ways (the pass counted at its end)
    // histogram of counts of a pass counted at its end (FaQCs.cpp:518-521): dense[c] for c < dense_n, a list of the larger counts
    unsigned long long *dense, *big, *n_big;
    uint32_t dense_n, big_cap;
    uint32_t *redo, *n_redo;  // [2^19], [1]: fine partitions the count-only kernel leaves to the one that goes through the table
    uint32_t split;           // blocks per bucket of the level-2 scatter = sub-regions per partition (1, 2, 4 or 8)
    uint32_t n_runs, epoch_base;
    unsigned long long *first_hist;   // [n_epochs] keys by first epoch
    unsigned long long *tot_by_epoch; // [n_epochs] occurrences by epoch
    uint32_t n_epochs;
    // sender staging of the multi-GPU exchange (super-k-mer items only): what does not fit a level-1 sub-region is appended here instead
    // of being counted into the table -- the keys belong to other ranks; first_hist / tot_by_epoch are null in this mode
    unsigned long long *spill;        // [spill_cap] 16-byte items
    uint32_t *spill_n;                // items appended (may run past spill_cap: the surplus is dropped and bit 1 of the table's overflow flag set)
    uint32_t spill_cap;
    // owner side of the multi-GPU exchange (skm_items): the 19-bit partitions this rank owns are [part_lo, part_lo + n) and an item's
    // partition becomes the LOCAL one, ((part - part_lo) * part_mul) >> 32 in [0, 2^19), part_mul = floor(2^51 / n): the owner's buckets,
    // level-2 regions and table slices then spread over ALL of its buffers and its whole table.  part_mul == 0: as they are.
    uint32_t part_lo;
    unsigned long long part_mul;
};

// launchers
uint32_t faqcs_kmer_extract_waves(uint32_t n_reads, int n_cu);
hipError_t faqcs_launch_kmer_extract(const DevParams &P, uint32_t k, const KmerOutbox &O, bool fill, const uint8_t *seq,
                                     const uint8_t *qual, const uint32_t *off, uint32_t r_begin, uint32_t r_end,
                                     const faqcs_read_result *results, uint32_t epoch, uint32_t wave_base, int n_cu, hipStream_t st);
hipError_t faqcs_launch_kmer_outbox_offsets(const KmerOutbox &O, hipStream_t st);
hipError_t faqcs_launch_kmer_insert_items(const KmerTable &T, const void *items, unsigned long long n,
                                          unsigned long long *tot_by_epoch, uint32_t n_epochs, int n_cu, hipStream_t st);
hipError_t faqcs_launch_kmer_table_init(const KmerTable &T, int n_cu, hipStream_t st, bool overflow_only = false);
hipError_t faqcs_launch_kmer_first_epoch_histogram(const KmerTable &T, unsigned long long *hist, uint32_t n_epochs, int n_cu,
                                                   hipStream_t st);
hipError_t faqcs_launch_kmer(const DevParams &P, uint32_t k, const KmerTable &T, const uint8_t *seq, const uint8_t *qual,
                             const uint32_t *off, uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results,
                             int n_cu, hipStream_t st);
hipError_t faqcs_launch_kmer_histogram(const KmerTable &T, unsigned long long *dense, uint32_t dense_n,
                                       unsigned long long *big, unsigned long long *n_big, uint32_t big_cap, int n_cu,
                                       hipStream_t st, bool overflow_only = false);
// super-k-mers (round 5; faqcs_kmer_skm_kernel.hip, faqcs_skm.h): the same group buffers with 16-byte items, a run of up to 17
// consecutive k-mers each; l1 / l2 / cap1 / cap2 of KmerGroupDev count 16-byte items in this mode
uint32_t faqcs_skm_grid(uint32_t n_reads, int n_cu);
hipError_t faqcs_launch_skm_extract(const DevParams &P, uint32_t k, const KmerGroupDev &G, const KmerTable &T, uint32_t run, uint32_t rot,
                                    uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                    uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, int n_cu, hipStream_t st,
                                    const uint32_t *list = nullptr, const uint32_t *list_n = nullptr, uint32_t grid_blocks = 0);
uint32_t faqcs_skm_grid16(uint32_t n_reads, int n_cu);
hipError_t faqcs_launch_skm_extract16(const DevParams &P, const KmerGroupDev &G, const KmerTable &T, uint32_t run, uint32_t rot,
                                      uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                      uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, uint32_t *defer, uint32_t *defer_n,
                                      int n_cu, hipStream_t st);
hipError_t faqcs_launch_skm_flush(const KmerGroupDev &G, const KmerTable &T, uint32_t k, hipStream_t st, uint32_t stages = 7u);
// the pass ends with this group and nothing of it has reached the table but through the per-occurrence path: fine split, count, cursors
// back to zero (stages as above).  The table is clean afterwards unless its overflow area was used (stats[3] counts those inserts).
hipError_t faqcs_launch_skm_finish(const KmerGroupDev &G, const KmerTable &T, uint32_t k, int n_cu, hipStream_t st, uint32_t stages = 7u);
hipError_t faqcs_launch_skm_reset(const KmerGroupDev &G, hipStream_t st);
uint32_t faqcs_skm_items_grid(unsigned long long n_items, int n_cu);
hipError_t faqcs_launch_skm_items(const KmerGroupDev &G, const KmerTable &T, uint32_t k, uint32_t rot, const void *items, unsigned long long n_items,
                                  int n_cu, hipStream_t st);
// sender side: dest_count[world] = items per destination rank (bucket b belongs to rank (b * world) >> 8), out = the items grouped by
// destination, run fields replaced by absolute epochs; scratch: region_offset[65536 + 2 * 64] u64
hipError_t faqcs_launch_skm_outbox(const KmerGroupDev &G, uint32_t world, unsigned long long *dest_count, unsigned long long *region_offset,
                                   void *out, hipStream_t st);
#ifdef __HIPCC__
__device__ __forceinline__ uint64_t kmer_mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
// A bijection of [0, 2^62): xor-shifts and multiplications by odd constants modulo 2^62 are each invertible.  Canonical keys
// of k <= 31 bases occupy 62 bits, so the mixed value can stand for the key (distinct / total / count histogram only see
// the partition of occurrences into keys), and its TOP bits select the bucket, the partition and the table slot.
__device__ __forceinline__ uint64_t kmer_mix62(uint64_t x)
{
    // ONE multiplication: only the TOP bits of h are used as an index (bucket 8, partition 16, table slot <= 32, LDS slot 12 bits of
    // the remainder), and the top bits of a product depend on every bit below them; the first xor-shift folds the high plane into the
    // low one so that keys that differ in high bits only still move the low bits.  Bucket / partition / cell occupancies of random,
    // AT-rich and repeat-rich k-mer sets are Poisson to within 5 % (profiles/microbench/kmer_mix_check.py), like the two-round form's;
    // a 62-bit multiplication is 3-4 quarter-rate instructions on gfx950, a sixth of the extraction's issue cycles.
    x ^= x >> 31; x = (x * 0x9E3779B97F4A7C15ull) & KG_M62;
    x ^= x >> 29;
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

// Key encoding: the reference keys its map by min(w, comp) of 2-bit-packed words.  Only the PARTITION of k-mer occurrences
// into {k-mer, reverse complement} classes is observable (distinct / total / histogram of counts), so any injective encoding
// with a consistent class representative yields identical integers.  Here enc = plane1 << 32 | plane0 (window bit t = t-th
// base), rc = reversed planes with plane0 inverted (codes A=0,T=1,C=2,G=3: complement flips bit0, trim.cpp:904-917),
// key = min(enc, enc_rc).
// One 64-base chunk: b = this lane's base byte (0 outside the kept window); the three ballots give the 2-bit code planes and
// the "valid ACGT" plane as 64-bit scalars, lane l extracts the k-bit windows ending at its position with funnel shifts.
struct KmerPlanes { uint64_t pv, p0, p1; };
__device__ __forceinline__ bool kmer_chunk_key(const uint32_t b, const int lane, const uint32_t k, KmerPlanes &S, uint64_t &key)
{
    const uint32_t l = b | 0x20u;
    const bool isA = l == 'a', isT = l == 't', isC = l == 'c', isG = l == 'g';
    const uint64_t cv = __ballot(isA | isT | isC | isG);
    const uint64_t c0 = __ballot(isT | isG); // codes A=0 T=1 C=2 G=3 (FaQCs.h:35-42)
    const uint64_t c1 = __ballot(isC | isG);
    const uint32_t wv = window_bits(cv, S.pv, lane, (int)k);
    const uint32_t w0 = window_bits(c0, S.p0, lane, (int)k);
    const uint32_t w1 = window_bits(c1, S.p1, lane, (int)k);
    S.pv = cv; S.p0 = c0; S.p1 = c1;
    const uint32_t kmask = (uint32_t)((1ull << k) - 1ull);
    const uint32_t r0 = __brev(~w0 & kmask) >> (32 - k), r1 = __brev(w1) >> (32 - k);
    const uint64_t fwd = ((uint64_t)w1 << 32) | w0, rc = ((uint64_t)r1 << 32) | r0;
    key = fwd < rc ? fwd : rc;
    return wv == kmask; // k valid bases ending here (word_len >= k, trim.cpp:924)
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
    KmerPlanes S{0, 0, 0};
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
        uint64_t key;
        const bool ok = kmer_chunk_key(b, lane, k, S, key);
        emit(ok, key);
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
__device__ __forceinline__ uint32_t slot_min_rtn(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, FAQCS_KMER_SCOPE); }
#endif
