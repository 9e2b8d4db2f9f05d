// faqcs_kmer_skm_kernel.hip -- k-mer counting over super-k-mers (gfx950, wave64; round 5).
//
// update_kmer() (trim.cpp:887-931) increments a hash-map entry per k-mer occurrence.  Only (distinct, total) at the sampling
// points (trim.cpp:157-185) and the final histogram of counts (FaQCs.cpp:518-521) are observable, and both are functions of
// {key -> (count, first epoch)}.  Round 4 moved every occurrence as an 8-byte item through two scatter passes before one
// workgroup per partition combined them: 61 bytes of HBM traffic per occurrence.  Here the unit that travels is a RUN of up
// to 17 consecutive k-mers of a read that share their minimizer (faqcs_skm.h): a 16-byte item per ~9 occurrences, partitioned
// by a hash of the minimizer so that every occurrence of a canonical k-mer still reaches the same workgroup.
//
//   skm_extract   one wave per read piece (256 positions, a lane owns four); codes, minimizers, run boundaries through the
//                 wave's LDS exchange rows; items staged per level-1 bucket in LDS and written as whole 256-byte granules into
//                 sub-regions that belong to the writing block (no atomics)
//   skm_split     every bucket 256 ways by the next 8 bits of the partition; the run number becomes the epoch
//   skm_combine   one workgroup per partition: expands the items in registers (rolling forward / reverse-complement words),
//                 counts the mixed canonical keys in an LDS hash table (key, count, smallest epoch) and applies ONE plain
//                 read-modify-write per distinct key to the slice of the table the partition owns.  An LDS table that fills up
//                 is written out and the workgroup goes on with an empty one: a partition of any size is exact without a
//                 per-occurrence path.
//   skm_items     (owner side of the multi-GPU exchange) items received from other ranks join the level-1 buckets
//   skm_outbox    (sender side) the level-1 sub-regions of the buckets a rank owns, packed back to back
//
// What does not fit a sub-region (heavy hitters of a skewed input) is expanded and inserted occurrence by occurrence
// (kmer_insert_atomic), so exactness never depends on a capacity.
#include "faqcs_kmer.h"
#include "faqcs_skm.h"
#include "faqcs_trim_common.h" // ensure_dynamic_lds

#include <stdlib.h>

namespace {

typedef unsigned long long u64;
typedef ulonglong2 Item;
#define KS_NONE (~0ull) /* w1 of "no item" in prefetch registers (a real item's run field never reaches 2^13 - 1 together with all other bits set) */

enum { KS_STAGE = 32, KS_GRAN = 16 }; // staging slots per bucket / items per global write (256 bytes)

__device__ __forceinline__ void hist_add(u64 *h, uint32_t e, uint32_t n_epochs, long long v)
{
    if (e < n_epochs) atomicAdd(&h[e], (u64)v);
}

// The table is cut into 2^(16 + F) slices, one per FINE partition (the top 16 + F bits of an item's 19-bit partition; F = KmerTable::fine);
// a key lives in the slice of ITS fine partition (slot inside the slice = top bits of the mixed key) and a probe sequence wraps inside
// the slice.  That is what lets skm_combine update the table without device-scope atomics: the workgroup that counts a partition --
// a fine one at the end of a pass, the eight (2^F) fine ones of a 16-bit partition in a group that is flushed while the pass goes on --
// is the only one in its slices for the length of the launch.
struct Slice { u64 base, mask; int shift; };
__device__ __forceinline__ Slice slice_of_fine(const KmerTable &T, const uint32_t fine_part)
{
    const u64 size = (T.mask + 1) >> (16u + T.fine);
    return Slice{(u64)fine_part * size, size - 1, (int)(T.shift + 16u + T.fine)};
}
__device__ __forceinline__ uint32_t fine_of(const KmerTable &T, const uint32_t part19) { return part19 >> (3u - T.fine); }
__device__ __forceinline__ Slice slice_of(const KmerTable &T, const uint32_t part19) { return slice_of_fine(T, fine_of(T, part19)); }

// One slot, atomically: key h gets `count` more occurrences and epoch as a candidate first epoch.  true: done (the key was there or
// the slot was free); false: the slot holds another key.  This is also where the general first-epoch rule lives: old = atomic min;
// a successful lowering moves the key from hist[old] to hist[epoch] -- the lowerings of one key form a chain, so the moves telescope
// to exactly one count at the key's final first epoch.
__device__ __forceinline__ bool slot_insert_atomic(KmerSlot *sl, const u64 h, const uint32_t epoch, const uint32_t count, u64 *first_hist,
                                                   const uint32_t n_epochs)
{
    u64 seen = __hip_atomic_load(&sl->key, __ATOMIC_RELAXED, FAQCS_KMER_SCOPE);
    uint32_t add = count;
    if (seen == ~0ull) {
        seen = slot_cas(&sl->key, ~0ull, h);
        if (seen == ~0ull) { seen = h; add = count - 1u; } // claimed: count_m1 = 0 already says "seen once"
    }
    if (seen != h) return false;
    if (add) slot_add(&sl->count_m1, add);
    const uint32_t old = slot_min_rtn(&sl->first_epoch, epoch);
    if (epoch < old) {
        if (old != 0xffffffffu) hist_add(first_hist, old, n_epochs, -1);
        hist_add(first_hist, epoch, n_epochs, 1);
    }
    return true;
}
// A key is looked for in a WINDOW of at most KS_PROBE_MAX slots behind its home slot, inside its partition's slice.  Minimizer
// partitions are not hash partitions: reads that share a sequence (an adapter, a repeat) with different flanks put thousands of
// distinct k-mers into one partition, and its slice can fill up while the table is half empty.  A key that finds its window full
// lives in the overflow area behind the table instead -- slots never become free during a pass, so a window that was full when a
// key arrived is full whenever the key is looked up again: the rule "window, else overflow area" always finds the same place.
enum { KS_PROBE_MAX = 128 };
__device__ void ovf_insert_atomic(const KmerTable &T, const u64 h, const uint32_t epoch, const uint32_t count, u64 *first_hist, const uint32_t n_epochs)
{
    // (a probe sequence of the overflow area is cut at 4 096 slots and nothing is tried once the table has been declared full: an area
    // that is full would otherwise be scanned end to end for every key -- the run fails with FAQCS_E_KMER_FULL either way, but at once)
    if (T.ovf_mask && !(__hip_atomic_load(&T.stats[2], __ATOMIC_RELAXED, FAQCS_KMER_SCOPE) & 1ull)) {
        atomicAdd(&T.stats[3], 1ull); // (the area is in use: the end of the pass has to sweep it)
        KmerSlot *ovf = T.slots + T.mask + 1;
        u64 g = (h ^ (h >> 23)) & T.ovf_mask;
        const u64 n_probe = T.ovf_mask < 4095ull ? T.ovf_mask + 1 : 4096ull;
#pragma unroll 1
        for (u64 probe = 0; probe < n_probe; ++probe) {
            if (slot_insert_atomic(&ovf[g], h, epoch, count, first_hist, n_epochs)) return;
            g = (g + 1) & T.ovf_mask;
        }
    }
    atomicOr(&T.stats[2], 1ull); // table full
}
// per-occurrence insert of a mixed key (what does not fit a sub-region of the group buffers); part = the item's 19-bit partition.
// The fine partition's dirty bit tells the counting pass at the end that this slice holds keys.
__device__ void kmer_insert_atomic(const KmerTable &T, const uint32_t part, const u64 h, const uint32_t epoch, const uint32_t count,
                                   u64 *first_hist, const uint32_t n_epochs)
{
    const uint32_t fp = fine_of(T, part);
    const Slice sc = slice_of_fine(T, fp);
    if (T.dirty) { const uint32_t bit = 1u << (fp & 31u); if (!(__hip_atomic_load(&T.dirty[fp >> 5], __ATOMIC_RELAXED, FAQCS_KMER_SCOPE) & bit)) atomicOr(&T.dirty[fp >> 5], bit); }
    u64 g = (h >> sc.shift) & sc.mask;
    const u64 win = sc.mask + 1 < (u64)KS_PROBE_MAX ? sc.mask + 1 : (u64)KS_PROBE_MAX;
#pragma unroll 1
    for (u64 probe = 0; probe < win; ++probe) {
        if (slot_insert_atomic(&T.slots[sc.base + g], h, epoch, count, first_hist, n_epochs)) return;
        g = (g + 1) & sc.mask;
    }
    ovf_insert_atomic(T, h, epoch, count, first_hist, n_epochs);
}
// every k-mer of an item, one by one
__device__ void skm_insert_item_atomic(const KmerTable &T, const Item it, const uint32_t epoch, const SkmGeom &g, u64 *first_hist, const uint32_t n_epochs)
{
    const uint32_t nk = skm_item_kmers(it.y), part = skm_item_part(it.y);
    SkmRoll r = skm_roll_begin(it.x, it.y, g);
#pragma unroll 1
    for (uint32_t j = 0; j < nk; ++j) {
        kmer_insert_atomic(T, part, skm_mix62(skm_roll_key(r)), epoch, 1u, first_hist, n_epochs);
        skm_roll_next(r, g);
    }
}

// ---- LDS staging shared by the scatter kernels --------------------------------------------------------------------------------
// 256 buckets x 32 slots of 16 bytes.  put(): a ticket from the bucket's LDS counter; a lane whose ticket is past the last slot
// keeps its item for the next round.  drain(): wave w owns buckets [16 w, 16 w + 16) and writes every full granule (16 items,
// 256 bytes) of them, FOUR buckets per step -- lane i of a quarter wave holds item i of its bucket's granule -- then moves what
// is left to the front.  WHERE a granule goes needs no atomic: a block appends to sub-regions that are its own (one per bucket),
// so the cursors live in its LDS (round 4).
template <int NW> struct Staging16 {
    static constexpr int BPW = KG_FAN / NW;
    Item *items;    // [KG_FAN][KS_STAGE]
    uint32_t *cnt;  // [KG_FAN]
    uint32_t *cur;  // [KG_FAN] items this block's sub-region of every bucket holds
    uint32_t *flag; // [3] block_or
    __device__ __forceinline__ bool put(const uint32_t b, const Item item) const
    {
        const uint32_t pos = atomicAdd(&cnt[b], 1u);
        if (pos < (uint32_t)KS_STAGE) { items[b * KS_STAGE + pos] = item; return true; }
        return false;
    }
    template <class Write, class Slow>
    __device__ __forceinline__ void drain(const int wave, const int lane, const bool final, const uint32_t cap, Write &&write, Slow &&slow) const
    {
        const uint32_t n_l = lane < BPW ? cnt[wave * BPW + lane] : 0u;
        uint64_t m = __ballot(final ? n_l > 0u : n_l >= (uint32_t)KS_GRAN);
        const int q = lane >> 4, l16 = lane & 15;
#pragma unroll 1
        while (m) {
            int sel[4];
            uint64_t t = m;
#pragma unroll
            for (int i = 0; i < 4; ++i) { sel[i] = t ? uni(__ffsll((long long)t) - 1) : -1; t = t ? t & (t - 1) : 0ull; }
            const int bsel = q == 0 ? sel[0] : (q == 1 ? sel[1] : (q == 2 ? sel[2] : sel[3]));
            const bool act = bsel >= 0;
            const uint32_t b = (uint32_t)(wave * BPW + (act ? bsel : sel[0]));
            uint32_t n = cnt[b];
            n = n < (uint32_t)KS_STAGE ? n : (uint32_t)KS_STAGE;
            const uint32_t pos = cur[b];
            const uint32_t take = n < (uint32_t)KS_GRAN ? n : (uint32_t)KS_GRAN;
            const bool mine = act && (uint32_t)l16 < take;
            const Item it = items[b * KS_STAGE + (uint32_t)l16];
            const bool fits = pos + take <= cap;
            if (mine) { if (fits) write(b, pos + (uint32_t)l16, it); else slow(b, it); }
            const uint32_t left = n - take;
            const bool mv = act && (uint32_t)l16 < left;
            const Item nx = items[b * KS_STAGE + (uint32_t)KS_GRAN + (uint32_t)l16]; // (one wave: the reads of an instruction complete before the writes of the next)
            if (mv) items[b * KS_STAGE + (uint32_t)l16] = nx;
            if (act && l16 == 0) { cnt[b] = left; cur[b] = fits ? pos + take : pos; }
            const bool again = act && l16 == 0 && (left >= (uint32_t)KS_GRAN || (final && left > 0u));
            const uint64_t keep = __ballot(again); // bits 0, 16, 32, 48: the step's buckets stay
            m = t;
#pragma unroll
            for (int i = 0; i < 4; ++i) if ((keep >> (16 * i)) & 1ull) m |= 1ull << sel[i];
        }
    }
    // OR of `bits` over the block; one barrier.  Three flag words in rotation: call k writes word k % 3, reads it behind the
    // barrier, and clears the word of call k - 1, which every wave has read before it arrived here.
    __device__ __forceinline__ uint32_t block_or(const uint32_t bits, uint32_t &phase, const int tid) const
    {
        const uint32_t wb = (__ballot(bits & 1u) ? 1u : 0u) | (__ballot(bits & 2u) ? 2u : 0u);
        if ((tid & 63) == 0 && wb) atomicOr(&flag[phase], wb);
        __syncthreads();
        const uint32_t v = flag[phase];
        const uint32_t prev = phase == 0 ? 2u : phase - 1u;
        if (tid == 0) flag[prev] = 0u;
        phase = phase == 2 ? 0u : phase + 1u;
        return v;
    }
};
constexpr size_t KS_STAGE_BYTES = (size_t)KG_FAN * KS_STAGE * 16 + (size_t)KG_FAN * 8 + 16;

// a wave's LDS operations execute in order; this only stops the compiler from moving them across the point
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// b where the lane's bit of m is set, else a (one v_cndmask_b32)
__device__ __forceinline__ uint32_t sel_(const uint32_t a, const uint32_t b, const uint64_t m)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}

__device__ __forceinline__ Item *l1_region(const KmerGroupDev &G, const uint32_t b, const uint32_t sub)
{
    return reinterpret_cast<Item *>(G.l1) + ((size_t)b * KG_FAN + sub) * G.cap1;
}

// ---- level 1: reads -> runs -> 256 buckets ------------------------------------------------------------------------------------
// A wave works on one PIECE at a time: 256 consecutive positions of a read's kept window, lane l owning positions 4 l .. 4 l + 3
// (one dword load).  A lane looks FORWARD: the k-mer at position p is bases p .. p + k - 1, its minimizer the smallest ord among
// the m-mers at p .. p + w - 1, and a run that starts at p needs bases up to p + w + k - 2 -- at most 14 lanes ahead, read from
// the wave's exchange row.  A read of up to 256 bases is one piece; a longer one advances by 200 positions per piece (the k-mer
// starts of the last 56 positions belong to the next piece, which sees all their bases).  An N (or any non-ACGT byte, or a G
// turned into N by --replace_to_N_q) inside the window cuts the piece into STRETCHES of valid bases, each handled like a window
// of its own: inside a stretch every base is valid, so "k valid bases" is arithmetic on positions.
// Per-wave LDS rows (dwords): A [80] codes of the lanes (8 bits each), later the run-break nibbles; B [288] the minima exchange.
enum { KS_ROW_A = 80, KS_ROW_B = 288, KS_ROWS = KS_ROW_A + KS_ROW_B };

template <int NW, bool K31>
__global__ __launch_bounds__(NW * 64) void skm_extract(
    const DevParams P, const uint32_t k_arg, const KmerGroupDev G, const KmerTable T, const uint32_t run, const uint32_t rot, const uint32_t epoch,
    const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin_arg,
    const uint32_t r_end_arg, const uint2 *__restrict__ results, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n)
{
    // list != null: the reads list[0 .. *list_n) (what skm_extract16 left to this kernel) instead of the range [r_begin, r_end)
    const uint32_t r_begin = list ? 0u : r_begin_arg, r_end = list ? uniu(*list_n) : r_end_arg;
    if (r_end <= r_begin) return;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    Item *s_items = reinterpret_cast<Item *>(lds);
    uint32_t *w32 = reinterpret_cast<uint32_t *>(s_items + KG_FAN * KS_STAGE);
    const Staging16<NW> S{s_items, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_total = w32 + 2 * KG_FAN + 3; // occurrences of this block
    uint32_t *s_rows = w32 + 2 * KG_FAN + 4;
    const SkmGeom g = skm_geom(K31 ? 31u : k_arg);
    const int k = (int)g.k, w = (int)g.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    uint32_t *rowA = s_rows + wave * KS_ROWS, *rowB = rowA + KS_ROW_A;
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN; // this block's sub-region of every bucket
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    if (tid < 4) S.flag[tid] = 0u; // (flags and s_total)
    for (int i = tid; i < NW * KS_ROWS; i += NW * 64) s_rows[i] = 0xfu; // (the pads behind the 64 lanes read as "break" / unused codes)
    const uint32_t n_waves = gridDim.x * NW;
    const bool g2n = !P.qc_only && P.replace_q > 0;
    auto write = [&](const uint32_t b, const uint32_t pos, const Item it) { l1_region(G, b, sub)[pos] = it; };
    auto slow = [&](const uint32_t, const Item it) { // the item does not fit its sub-region
        if (G.spill) { // sender staging: it still has to travel
            const uint32_t at = atomicAdd(G.spill_n, 1u);
            if (at < G.spill_cap) reinterpret_cast<Item *>(G.spill)[at] = it; else atomicOr(&T.stats[2], 2ull);
        } else skm_insert_item_atomic(T, it, epoch, g, G.first_hist, G.n_epochs);
    };
    __syncthreads();

    struct Hdr { uint32_t o; int a, n; }; // kept window [a, a + n) of the read at byte o; n == 0: nothing to count
    auto load_hdr = [&](const uint32_t ri) -> Hdr {
        Hdr h{0u, 0, 0};
        if (ri < r_end) {
            const uint32_t r = list ? uniu(list[ri]) : ri;
            h.o = uniu(off[r]);
            h.n = (int)(uniu(off[r + 1]) - h.o);
            if (!P.qc_only) { // trimmed read of a valid record (trim.cpp:545-547); raw read under --qc_only (:260-262)
                const uint2 rs = results[r];
                h.a = uni((int)(rs.x & 0xffffu));
                h.n = uni((rs.y & FAQCS_F_VALID) ? (int)(rs.x >> 16) : 0);
            }
            if (h.n < k) h.n = 0;
        }
        return h;
    };
    struct __attribute__((packed, aligned(1))) U32u { uint32_t w; };
    const uint32_t safe_o = off[r_begin_arg]; // (a byte of the arena that is there whatever the cursor says)
    // the piece being fetched: read hn_ (header), first position pn; the piece in work: hc, pc
    uint32_t r_nxt = r_begin + blockIdx.x * NW + wave;
    Hdr hf = load_hdr(r_nxt), hc{0u, 0, 0};
    int pf = hf.a, pc = 0;
    bool f_live = r_nxt < r_end;   // the fetched piece exists
    Hdr ha = load_hdr(r_nxt + n_waves); // the header after the fetched piece's read
    uint32_t nw_ = 0, nqw = 0;
    auto fetch = [&]() { // the fetched piece's four bytes of this lane (what lies past the window is not read)
        const int p0 = pf + 4 * lane;
        const bool need = f_live && hf.n > 0 && p0 < hf.a + hf.n;
        const size_t at = need ? (size_t)hf.o + (uint32_t)p0 : (size_t)safe_o;
        const uint32_t v = reinterpret_cast<const U32u *>(seq + at)->w;
        nw_ = need ? v : 0u;
        if (g2n) { const uint32_t qv = reinterpret_cast<const U32u *>(qual + at)->w; nqw = need ? qv : 0u; }
    };
    auto advance_fetch = [&]() { // the piece after the fetched one
        if (f_live && hf.n > 0 && pf + SKM_PIECE < hf.a + hf.n) { pf += SKM_ADVANCE; return; }
        r_nxt += n_waves;
        f_live = r_nxt < r_end;
        hf = ha; pf = hf.a;
        ha = load_hdr(r_nxt + n_waves);
    };
    fetch();

    // wave state of the piece in work
    bool c_live = false;            // a piece is in work
    int st_next = 0, st_end = 0;    // stretch cursor inside the piece's window
    bool has_bad = false;
    uint64_t bad[4] = {0, 0, 0, 0}; // bit l of bad[j]: position 4 l + j of the piece is inside the window and not a base
    int limit = 0;                  // k-mer starts at or past it belong to the next piece
    uint32_t codes = 0;             // this lane's four 2-bit codes
    // the stretch in work, per lane
    u64 c_lo = 0, c_hi = 0;
    uint32_t omin[4] = {0, 0, 0, 0}, mask24 = 0, sbits = 0;
    Item pend[2];
    uint32_t n_pend = 0;
    uint32_t my_total = 0, phase = 0, round_no = 0;
    const u64 tag = (u64)run;

    auto stretch = [&](const int sa, const int sn) { // runs of the valid bases [sa, sa + sn) of piece pc
        rowA[lane] = codes;
        lds_wave_sync();
        uint32_t x[SKM_LOOK];
#pragma unroll
        for (int i = 0; i < SKM_LOOK; ++i) x[i] = rowA[lane + i];
        lds_wave_sync();
        c_lo = 0; c_hi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) c_lo |= (u64)x[i] << (8 * i);
#pragma unroll
        for (int i = 8; i < SKM_LOOK; ++i) c_hi |= (u64)x[i] << (8 * (i - 8));
        uint32_t o[4];
        if (K31) skm_mmer_ords15(c_lo, g, o);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = skm_mmer_ord_at(c_lo, j, g);
        }
        if (K31) { // w = 17: positions p .. p + 16 = the lane's own from j on, three whole lanes, the first j + 1 of the fourth
            const uint32_t p2 = umin_(o[0], o[1]), p3 = umin_(p2, o[2]), m4 = umin_(p3, o[3]);
            const uint32_t s2 = umin_(o[2], o[3]), s1 = umin_(o[1], s2);
            rowB[lane] = m4; rowB[72 + lane] = o[0]; rowB[144 + lane] = p2; rowB[216 + lane] = p3;
            lds_wave_sync();
            const uint32_t mid = umin_(umin_(rowB[lane + 1], rowB[lane + 2]), rowB[lane + 3]);
            omin[0] = umin_(umin_(m4, mid), rowB[72 + lane + 4]);
            omin[1] = umin_(umin_(s1, mid), rowB[144 + lane + 4]);
            omin[2] = umin_(umin_(s2, mid), rowB[216 + lane + 4]);
            omin[3] = umin_(umin_(o[3], mid), rowB[lane + 4]);
            lds_wave_sync();
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) rowB[4 * lane + j] = o[j];
            lds_wave_sync();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t mn = o[j];
#pragma unroll 1
                for (int t = 1; t < w; ++t) mn = umin_(mn, rowB[4 * lane + j + t]);
                omin[j] = mn;
            }
            lds_wave_sync();
        }
        // valid k-mer starts of the stretch in this piece: [vlo, vhi)
        const int vlo = sa > pc ? sa : pc;
        int vhi = sa + sn - k + 1;
        vhi = vhi < limit ? vhi : limit;
        const uint32_t prev = (uint32_t)__shfl_up((int)omin[3], 1);
        uint32_t vb = 0, st = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pc + 4 * lane + j;
            const bool v = p >= vlo && p < vhi;
            const bool s = v && (p == vlo || omin[j] != (j ? omin[j - 1] : prev));
            vb |= v ? 1u << j : 0u;
            st |= s ? 1u << j : 0u;
        }
        my_total += (uint32_t)__popc(vb);
        auto exchange_breaks = [&](const uint32_t brk) {
            rowA[lane] = brk;
            lds_wave_sync();
            uint32_t m = brk;
#pragma unroll
            for (int i = 1; i <= 5; ++i) m |= rowA[lane + i] << (4 * i);
            lds_wave_sync();
            return m;
        };
        auto too_long = [&](const uint32_t m) { // some run of this lane's starts has more than w k-mers
            bool bad_ = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) bad_ |= ((st >> j) & 1u) && ((m >> (j + 1)) & ((1u << w) - 1u)) == 0u;
            return bad_;
        };
        mask24 = exchange_breaks(st | (~vb & 0xfu));
        if (__any(too_long(mask24))) { // equal minimizers over more than w k-mers (a repeat): cut the stretch's runs on a grid of w
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = pc + 4 * lane + j;
                if (((vb >> j) & 1u) && (uint32_t)(p - vlo) % (uint32_t)w == 0u) st |= 1u << j;
            }
            mask24 = exchange_breaks(st | (~vb & 0xfu));
        }
        sbits = st;
    };
    // first valid stretch at or after st_next; false: none left in this piece
    auto next_stretch = [&]() -> bool {
#pragma unroll 1
        while (st_next < st_end) {
            const int lo = st_next;
            int nb = st_end;
            if (has_bad) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rel = lo - pc - j;
                    const int ln = rel <= 0 ? 0 : (rel + 3) >> 2;
                    if (ln < 64) {
                        const uint64_t m = bad[j] >> ln;
                        if (m) { const int cand = pc + 4 * (ln + (__ffsll((long long)m) - 1)) + j; nb = cand < nb ? cand : nb; }
                    }
                }
            }
            st_next = nb + 1;
            if (nb - lo >= k) { stretch(lo, nb - lo); return true; }
        }
        return false;
    };

#pragma unroll 1
    for (;;) {
        // ---- produce: the lanes' next runs ----
        if (!__any(n_pend != 0u || sbits != 0u)) {
            bool got = c_live && next_stretch();
#pragma unroll 1
            while (!got && f_live) { // the fetched piece becomes the piece in work
                asm volatile("" ::"v"(nw_), "v"(nqw));
                const uint32_t bw = nw_, bqw = nqw;
                hc = hf; pc = pf; c_live = true;
                const bool last = !(hc.n > 0 && pc + SKM_PIECE < hc.a + hc.n);
                limit = last ? 0x7fffffff : pc + SKM_ADVANCE;
                advance_fetch();
                fetch();
                uint32_t valid;
                skm_classify4(bw, codes, valid);
                if (g2n) { // G -> N precedes k-mer counting (trim.cpp:390-403)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int qv = (int)(int8_t)((bqw >> (8 * j)) & 0xffu) - P.in_off;
                        qv = qv < 0 ? 0 : qv;
                        if (((bw >> (8 * j)) & 0xffu) == (uint32_t)'G' && qv < (int)P.replace_q) valid &= ~(1u << j);
                    }
                }
                st_next = pc;
                st_end = hc.a + hc.n < pc + SKM_PIECE ? hc.a + hc.n : pc + SKM_PIECE;
                int inw = st_end - (pc + 4 * lane); // the lane's positions inside the window: the first inw of its four
                inw = inw < 0 ? 0 : (inw > 4 ? 4 : inw);
                const uint32_t nb = ((1u << inw) - 1u) & ~valid;
                has_bad = __any(nb != 0u);
                if (has_bad) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bad[j] = __ballot((nb >> j) & 1u);
                }
                got = hc.n > 0 && next_stretch();
            }
            if (!got) c_live = false;
        }
        // ---- emit: up to two runs per lane and round ----
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (n_pend == (uint32_t)e && sbits) {
                const uint32_t j = (uint32_t)__ffs((int)sbits) - 1u;
                sbits &= sbits - 1u;
                const uint32_t len = (uint32_t)__ffs((int)((mask24 >> (j + 1u)) | (1u << w)));
                const uint32_t om = j == 0 ? omin[0] : (j == 1 ? omin[1] : (j == 2 ? omin[2] : omin[3]));
                u64 w0, w1;
                skm_pack(c_lo, c_hi, j, len, skm_part(om), (uint32_t)tag, w0, w1);
                pend[e] = make_ulonglong2(w0, w1);
                n_pend = (uint32_t)e + 1u;
            }
        }
        // ---- stage ----
        if (n_pend) {
            bool ok0 = S.put(skm_item_bucket(pend[0].y), pend[0]);
            if (n_pend == 2u) {
                const bool ok1 = S.put(skm_item_bucket(pend[1].y), pend[1]);
                if (ok0 && !ok1) { pend[0] = pend[1]; }
                if (ok0 != ok1) n_pend = 1u; else n_pend = ok0 ? 0u : 2u;
            } else n_pend = ok0 ? 0u : 1u;
        }
        __syncthreads();
        // (a bucket gets about 1.5 of the block's items per round and holds 32: the owners write the full granules every fourth round,
        // four buckets per step; a put that finds its bucket full waits a round)
        if ((++round_no & 3u) == 0u) S.drain(wave, lane, false, G.cap1, write, slow);
        const uint32_t f = S.block_or((n_pend != 0u || sbits != 0u || c_live || f_live) ? 1u : 0u, phase, tid);
        if (!(f & 1u)) break;
    }
    S.drain(wave, lane, true, G.cap1, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur1[sub * KG_FAN + i] = S.cur[i];
    // occurrences of this launch's epoch (total_kmer of the sampling points, trim.cpp:170-176)
    const uint32_t wt = (uint32_t)wave_sum_i32((int)my_total); // (a wave sees < 2^31 occurrences per launch)
    if (lane == 0 && wt) atomicAdd(s_total, wt);                // (a block sees < 2^32)
    __syncthreads();
    if (tid == 0 && s_total[0]) {
        if (G.tot_by_epoch) { // (sender staging of the exchange: the owner counts what it receives)
            hist_add(G.tot_by_epoch, epoch, G.n_epochs, (long long)s_total[0]);
            atomicAdd(&T.stats[1], (u64)s_total[0]);
        }
    }
}

// ---- level 1 for reads of up to 256 bases, k = 31: a 16-lane DPP row per read, a lane owns SIXTEEN positions ------------------
// skm_extract spends about 130 of its 530 vector instructions per round on the arithmetic of the runs; the rest -- classification,
// the exchange rows, emission, staging, the round's barriers -- is paid per ROUND and shared by four positions per lane
// (profiles/r5a/pmc_skm_*.txt: 2.4 vector + 1.8 scalar instructions per occurrence, issue-bound on both).  Here a wave works on FOUR
// reads per round, one per row of 16 lanes, and a lane owns 16 consecutive positions (one 16-byte load): the same per-round costs are
// shared by four times the positions, and everything a lane needs from its neighbours comes through DPP row shifts, which fill with
// zeros past the end of the row -- exactly "no bases behind the read": the three following lanes' codes (row_shl 1 .. 3), the next
// lane's prefix minima (the window of 17 m-mers at position j = the lane's own suffix j .. 15 and the next lane's prefix 0 .. j: one
// v_min with a DPP source per position), the previous lane's last minimum, the following lanes' run bits.  No LDS but the staging.
// A byte that is not a base cuts a read by bit arithmetic (the "unusable position" bits smeared over 31 places).  A read with a run of
// more than 17 k-mers (a repeat) is not handled here: its index goes to `defer` and skm_extract takes it afterwards (grid cuts).
template <int NW>
__global__ __launch_bounds__(NW * 64) void skm_extract16(
    const DevParams P, const KmerGroupDev G, const KmerTable T, const uint32_t run, const uint32_t rot, const uint32_t epoch,
    const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin,
    const uint32_t r_end, const uint2 *__restrict__ results, uint32_t *__restrict__ defer, uint32_t *__restrict__ defer_n)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    Item *s_items = reinterpret_cast<Item *>(lds);
    uint32_t *w32 = reinterpret_cast<uint32_t *>(s_items + KG_FAN * KS_STAGE);
    const Staging16<NW> S{s_items, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_total = w32 + 2 * KG_FAN + 3; // occurrences of this block
    const SkmGeom g = skm_geom(31u);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), row = lane >> 4, l16 = lane & 15;
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN; // this block's sub-region of every bucket
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    if (tid < 4) S.flag[tid] = 0u; // (flags and s_total)
    const uint32_t stride = gridDim.x * NW * 4u;
    const bool g2n = !P.qc_only && P.replace_q > 0;
    auto write = [&](const uint32_t b, const uint32_t pos, const Item it) { l1_region(G, b, sub)[pos] = it; };
    auto slow = [&](const uint32_t, const Item it) { // the item does not fit its sub-region
        if (G.spill) { // sender staging: it still has to travel
            const uint32_t at = atomicAdd(G.spill_n, 1u);
            if (at < G.spill_cap) reinterpret_cast<Item *>(G.spill)[at] = it; else atomicOr(&T.stats[2], 2ull);
        } else skm_insert_item_atomic(T, it, epoch, g, G.first_hist, G.n_epochs);
    };
    __syncthreads();

    struct Hdr { uint32_t o; int a, n; }; // kept window [a, a + n) of the row's read at byte o; n == 0: nothing to count
    auto load_hdr = [&](const uint32_t r) -> Hdr {
        Hdr h{0u, 0, 0};
        if (r < r_end) {
            h.o = off[r];
            h.n = (int)(off[r + 1] - h.o);
            if (!P.qc_only) { // trimmed read of a valid record (trim.cpp:545-547); raw read under --qc_only (:260-262)
                const uint2 rs = results[r];
                h.a = (int)(rs.x & 0xffffu);
                h.n = (rs.y & FAQCS_F_VALID) ? (int)(rs.x >> 16) : 0;
            }
            if (h.n < 31) h.n = 0;
        }
        return h;
    };
    struct __attribute__((packed, aligned(1))) U128u { uint32_t w[4]; };
    const uint32_t safe_o = off[r_begin];
    uint32_t r_f = r_begin + (blockIdx.x * NW + (uint32_t)wave) * 4u + (uint32_t)row; // the row's read being fetched
    Hdr hf = load_hdr(r_f), hn = load_hdr(r_f + stride);
    uint32_t nb[4] = {0, 0, 0, 0}, nq[4] = {0, 0, 0, 0};
    auto fetch = [&]() { // the lane's sixteen bytes of the fetched read (what lies past the window is not read)
        const bool need = hf.n > 0 && 16 * l16 < hf.n;
        const size_t at = need ? (size_t)hf.o + (uint32_t)(hf.a + 16 * l16) : (size_t)safe_o;
        const U128u v = *reinterpret_cast<const U128u *>(seq + at);
#pragma unroll
        for (int d = 0; d < 4; ++d) nb[d] = need ? v.w[d] : 0u;
        if (g2n) {
            const U128u q = *reinterpret_cast<const U128u *>(qual + at);
#pragma unroll
            for (int d = 0; d < 4; ++d) nq[d] = need ? q.w[d] : 0u;
        }
    };
    fetch();
    auto shl = [](const uint32_t v, const int n) -> uint32_t { // the value of the lane n places on in the row, 0 past its end
        return n == 1 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true)
             : n == 2 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true)
                      : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x103, 0xf, 0xf, true);
    };
    // the read in work, per lane
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0; // codes of the positions 16 l .. 16 l + 63
    uint32_t omin[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) omin[j] = 0u;
    u64 brk64 = 0;     // bit i: position 16 l + i starts a run or holds no k-mer
    uint32_t sb = 0;   // run starts of this lane still to be written
    Item pend = make_ulonglong2(0ull, 0ull);
    bool has_pend = false;
    uint32_t my_total = 0, phase = 0;
    bool more_reads = r_f < r_end; // (row-uniform) the fetched read exists

#pragma unroll 1
    for (;;) {
        if (!__any(sb != 0u || has_pend) && __any(more_reads)) {
            // ---- the fetched reads become the reads in work ----
            asm volatile("" ::"v"(nb[0]), "v"(nb[1]), "v"(nb[2]), "v"(nb[3]));
            const Hdr hc = hf;
            const uint32_t r_c = r_f;
            uint32_t bw[4], bq[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) { bw[d] = nb[d]; bq[d] = nq[d]; }
            r_f += stride; hf = hn; hn = load_hdr(r_f + stride);
            more_reads = r_f < r_end;
            fetch();
            uint32_t valid = 0;
            c0 = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t cd, vd;
                skm_classify4(bw[d], cd, vd);
                if (g2n) { // G -> N precedes k-mer counting (trim.cpp:390-403)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int qv = (int)(int8_t)((bq[d] >> (8 * j)) & 0xffu) - P.in_off;
                        qv = qv < 0 ? 0 : qv;
                        if (((bw[d] >> (8 * j)) & 0xffu) == (uint32_t)'G' && qv < (int)P.replace_q) vd &= ~(1u << j);
                    }
                }
                c0 |= cd << (8 * d); valid |= vd << (4 * d);
            }
            int inw = hc.n - 16 * l16; // the lane's positions inside the window: the first inw of its sixteen
            inw = inw < 0 ? 0 : (inw > 16 ? 16 : inw);
            // positions that cannot be part of a k-mer: outside the kept window or not a base.  A position holds a k-mer iff none of
            // the 31 positions from it on is one of them: the bits smeared over 31 places (an N cuts the read without any loop)
            const uint32_t usable = ((1u << inw) - 1u) & valid;
            u64 z = ~((u64)usable | ((u64)shl(usable, 1) << 16) | ((u64)shl(usable, 2) << 32)); // (past the end of the row: not usable)
            z |= z >> 1; z |= z >> 2; z |= z >> 4; z |= z >> 8; // bit i: one of i .. i + 15
            z |= z >> 15;                                       // ... i .. i + 30
            const uint32_t vb = 0xffffu & ~(uint32_t)z;
            c1 = shl(c0, 1); c2 = shl(c0, 2); c3 = shl(c0, 3);
            // ord of the canonical m-mer at each of the sixteen positions: bases 0 .. 29 reversed once (base i -> group 29 - i)
            const uint32_t r0 = skm_rev2_32((c1 << 4) | (c0 >> 28)), r1 = skm_rev2_32(c0 << 4);
            uint32_t o[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t fwd = __builtin_amdgcn_alignbit(c1, c0, 2 * j) & SKM_M30;
                const uint32_t rc = (__builtin_amdgcn_alignbit(r1, r0, 2 * (15 - j)) & SKM_M30) ^ 0x2AAAAAAAu;
                o[j] = skm_ord(fwd < rc ? fwd : rc, g);
            }
            // smallest ord of the 17 m-mers at p .. p + 16: the lane's suffix j .. 15 and the next lane's prefix 0 .. j
            uint32_t pre[16];
            pre[0] = o[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) pre[j] = umin_(pre[j - 1], o[j]);
            uint32_t suf = o[15];
#pragma unroll
            for (int j = 15; j >= 0; --j) {
                suf = umin_(suf, o[j]);
                omin[j] = umin_(suf, shl(pre[j], 1));
            }
            // run starts: a k-mer starts a run when the position before it holds none or its minimum changes
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)omin[15], 0x111, 0xf, 0xf, true); // row_shr:1
            const uint32_t vprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vb, 0x111, 0xf, 0xf, true);
            uint32_t st = ~((vb << 1) | (vprev >> 15));
#pragma unroll
            for (int j = 0; j < 16; ++j) st |= (omin[j] != (j ? omin[j - 1] : prev)) ? 1u << j : 0u;
            st &= vb;
            const uint32_t cont = vb & ~st; // positions that continue a run
            const u64 cont64 = (u64)cont | ((u64)shl(cont, 1) << 16) | ((u64)shl(cont, 2) << 32);
            // a run of more than 17 k-mers: 17 continuing positions in a row right behind one of this lane's starts
            u64 x = cont64;
            x &= x >> 1; x &= x >> 2; x &= x >> 4; x &= x >> 8; // bit i: positions i .. i + 15 continue
            x &= cont64 >> 16;                                  // ... and position i + 16
            const bool too_long = ((uint32_t)(x >> 1) & st) != 0u;
            const bool deferred = row_all_or(too_long ? 1u : 0u) != 0u;
            if (deferred) {
                if (l16 == 0 && hc.n > 0) defer[atomicAdd(defer_n, 1u)] = r_c;
                sb = 0u;
            } else {
                sb = st;
                my_total += (uint32_t)__popc(vb);
            }
            brk64 = ~cont64;
        }
        // ---- write the runs: a lane builds and stages one item per turn until its starts are used up or a bucket is full ----
        bool stuck = false;
#pragma unroll 1
        while (__any((sb != 0u || has_pend) && !stuck)) {
            if (!stuck) {
                if (!has_pend && sb) {
                    const uint32_t j = (uint32_t)__ffs((int)sb) - 1u;
                    sb &= sb - 1u;
                    const uint32_t len = (uint32_t)__ffsll((long long)(brk64 >> (j + 1u)));
                    // omin[j]: a tree of fifteen selects on the bits of j (written as instructions: as C the compiler turns the tree
                    // into an indexed load and moves omin[] to scratch memory)
                    const uint64_t m0 = __ballot(j & 1u), m1 = __ballot(j & 2u), m2 = __ballot(j & 4u), m3 = __ballot(j & 8u);
                    uint32_t t8[8], t4[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) t8[i] = sel_(omin[2 * i], omin[2 * i + 1], m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) t4[i] = sel_(t8[2 * i], t8[2 * i + 1], m1);
                    const uint32_t om = sel_(sel_(t4[0], t4[1], m2), sel_(t4[2], t4[3], m2), m3);
                    const uint32_t sh = 2u * j;
                    const uint32_t d0 = __builtin_amdgcn_alignbit(c1, c0, sh), d1 = __builtin_amdgcn_alignbit(c2, c1, sh);
                    const uint32_t d2 = __builtin_amdgcn_alignbit(c3, c2, sh) & SKM_M30;
                    pend.x = ((u64)d1 << 32) | d0;
                    pend.y = (u64)d2 | ((u64)(len - 1u) << SKM_NK_SHIFT) | ((u64)skm_part(om) << SKM_PART_SHIFT) | ((u64)run << SKM_RUN_SHIFT);
                    has_pend = true;
                }
                if (has_pend) {
                    if (S.put(skm_item_bucket(pend.y), pend)) has_pend = false;
                    else stuck = true;
                }
            }
        }
        __syncthreads();
        S.drain(wave, lane, false, G.cap1, write, slow);
        const uint32_t f = S.block_or((sb != 0u || has_pend || more_reads) ? 1u : 0u, phase, tid);
        if (!(f & 1u)) break;
    }
    S.drain(wave, lane, true, G.cap1, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur1[sub * KG_FAN + i] = S.cur[i];
    // occurrences of this launch's epoch (total_kmer of the sampling points, trim.cpp:170-176)
    const uint32_t wt = (uint32_t)wave_sum_i32((int)my_total);
    if (lane == 0 && wt) atomicAdd(s_total, wt);
    __syncthreads();
    if (tid == 0 && s_total[0]) {
        if (G.tot_by_epoch) { // (sender staging of the exchange: the owner counts what it receives)
            hist_add(G.tot_by_epoch, epoch, G.n_epochs, (long long)s_total[0]);
            atomicAdd(&T.stats[1], (u64)s_total[0]);
        }
    }
}

// ---- level 1 from items another rank extracted: the owner side of the multi-GPU exchange (faqcs_kmer_insert_device) -----------
// An item's run field holds its ABSOLUTE epoch (the group's run -> epoch table is the identity in this mode).  The item's partition
// becomes the owner's LOCAL one (KmerGroupDev::part_mul): a rank owns 1 / world of the 19-bit partitions, and mapped onto [0, 2^19) they use
// all of its level-1 buckets, level-2 regions and table slices instead of 1 / world of each.
__device__ __forceinline__ Item item_local(const KmerGroupDev &G, Item it)
{
    if (G.part_mul) it.y = skm_item_with_part(it.y, (uint32_t)(((u64)(skm_item_part(it.y) - G.part_lo) * G.part_mul) >> 32));
    return it;
}
template <int NW>
__global__ __launch_bounds__(NW * 64) void skm_items(const KmerGroupDev G, const KmerTable T, const uint32_t k, const uint32_t rot,
                                                     const Item *__restrict__ items, const u64 n_items)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    Item *s_items = reinterpret_cast<Item *>(lds);
    uint32_t *w32 = reinterpret_cast<uint32_t *>(s_items + KG_FAN * KS_STAGE);
    const Staging16<NW> S{s_items, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_tot = w32 + 2 * KG_FAN + 4; // [KG_EPOCH_SPAN] occurrences of this block by epoch
    const SkmGeom g = skm_geom(k);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN;
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    for (int i = tid; i < KG_EPOCH_SPAN; i += NW * 64) s_tot[i] = 0u;
    if (tid < 3) S.flag[tid] = 0u;
    __syncthreads();
    auto write = [&](const uint32_t b, const uint32_t pos, const Item it) { l1_region(G, b, sub)[pos] = it; };
    auto slow = [&](const uint32_t, const Item it) { skm_insert_item_atomic(T, it, skm_item_run(it.y), g, G.first_hist, G.n_epochs); };
    constexpr uint32_t PER = NW * 64;
    const u64 per_block = ((n_items + gridDim.x - 1) / gridDim.x + PER - 1) / PER * PER;
    const u64 lo = (u64)blockIdx.x * per_block < n_items ? (u64)blockIdx.x * per_block : n_items;
    const u64 hi = lo + per_block < n_items ? lo + per_block : n_items;
    uint32_t phase = 0;
    Item nx = lo + tid < hi ? item_local(G, items[lo + tid]) : make_ulonglong2(0ull, KS_NONE);
#pragma unroll 1
    for (u64 t0 = lo; t0 < hi; t0 += PER) {
        const Item cur = nx;
        nx = t0 + PER + tid < hi ? item_local(G, items[t0 + PER + tid]) : make_ulonglong2(0ull, KS_NONE);
        uint32_t pend = 0;
        if (cur.y != KS_NONE && skm_item_run(cur.y) < (uint32_t)KG_EPOCH_SPAN) {
            pend = 1u;
            atomicAdd(&s_tot[skm_item_run(cur.y)], skm_item_kmers(cur.y));
        }
#pragma unroll 1
        for (;;) {
            if (pend && S.put(skm_item_bucket(cur.y), cur)) pend = 0u;
            __syncthreads();
            S.drain(wave, lane, false, G.cap1, write, slow);
            if (!(S.block_or(pend, phase, tid) & 1u)) break;
        }
    }
    __syncthreads();
    S.drain(wave, lane, true, G.cap1, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur1[sub * KG_FAN + i] = S.cur[i];
    for (int i = tid; i < KG_EPOCH_SPAN; i += NW * 64)
        if (s_tot[i]) hist_add(G.tot_by_epoch, (uint32_t)i, G.n_epochs, (long long)s_tot[i]);
}

// ---- level 2: every bucket 256 ways; the run number becomes the epoch -------------------------------------------------------
// Block (b1, part) reads the sub-regions [part * 256 / split, (part + 1) * 256 / split) of bucket b1 and appends to sub-region
// `part` of the partitions b1 * 256 + (low 8 bits of the item's partition): again a block writes only where no other block does.
template <int NW>
__global__ __launch_bounds__(NW * 64) void skm_split(const KmerGroupDev G, const KmerTable T, const uint32_t k)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    Item *s_items = reinterpret_cast<Item *>(lds);
    uint32_t *w32 = reinterpret_cast<uint32_t *>(s_items + KG_FAN * KS_STAGE);
    const Staging16<NW> S{s_items, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_er = w32 + 2 * KG_FAN + 4;     // [KG_MAX_RUNS] epoch of run j, relative
    uint32_t *s_n = s_er + KG_MAX_RUNS;        // [256] items of the bucket's sub-regions
    const SkmGeom g = skm_geom(k);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t b1 = blockIdx.x / G.split, part = blockIdx.x % G.split;
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = 0u; s_n[i] = G.cur1[i * KG_FAN + b1]; }
    if (tid < 3) S.flag[tid] = 0u;
    for (uint32_t j = tid; j < G.n_runs; j += NW * 64) s_er[j] = G.run_epoch[j];
    __syncthreads();
    Item *const l2 = reinterpret_cast<Item *>(G.l2);
    auto write = [&](const uint32_t b, const uint32_t pos, const Item it) { l2[(((size_t)b1 * KG_FAN + b) * G.split + part) * G.cap2 + pos] = it; };
    auto slow = [&](const uint32_t, const Item it) { skm_insert_item_atomic(T, it, G.epoch_base + skm_item_run(it.y), g, G.first_hist, G.n_epochs); };
    constexpr uint32_t TILE = NW * 64 * 2;
    const uint32_t per = KG_FAN / G.split;
    uint32_t phase = 0;
#pragma unroll 1
    for (uint32_t sr = part * per; sr < (part + 1) * per; ++sr) {
        const uint32_t n_s = s_n[sr];
        if (n_s == 0) continue; // (block-uniform)
        const Item *src = l1_region(G, b1, sr);
        Item nx[2], cx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const uint32_t i = j * NW * 64 + tid; nx[j] = i < n_s ? src[i] : make_ulonglong2(0ull, KS_NONE); }
#pragma unroll
        for (int j = 0; j < 2; ++j) cx[j] = nx[j];
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < n_s; t0 += TILE) {
            Item cur[2];
            uint32_t pend = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) { const uint32_t i = t0 + TILE + j * NW * 64 + tid; nx[j] = i < n_s ? src[i] : make_ulonglong2(0ull, KS_NONE); } // next tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cur[j] = cx[j];
                if (cx[j].y != KS_NONE) {
                    cur[j].y = (cx[j].y & ~SKM_RUN_MASK) | ((u64)s_er[skm_item_run(cx[j].y)] << SKM_RUN_SHIFT);
                    pend |= 1u << j;
                }
            }
#pragma unroll 1
            for (bool fetched = false;;) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if ((pend >> j) & 1u)
                        if (S.put(skm_item_p16(cur[j].y) & 255u, cur[j])) pend &= ~(1u << j);
                __syncthreads();
                if (!fetched) {
                    asm volatile("" ::"v"(nx[0].x), "v"(nx[0].y), "v"(nx[1].x), "v"(nx[1].y));
#pragma unroll
                    for (int j = 0; j < 2; ++j) cx[j] = nx[j];
                    fetched = true;
                }
                S.drain(wave, lane, false, G.cap2, write, slow);
                if (!(S.block_or(pend ? 1u : 0u, phase, tid) & 1u)) break;
            }
        }
    }
    __syncthreads();
    S.drain(wave, lane, true, G.cap2, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur2[((size_t)b1 * KG_FAN + i) * G.split + part] = S.cur[i];
}

// ---- level 2 of a pass that is counted at its END: every bucket (256 << F) ways, by a sort of 8 192-item tiles in LDS -------------
// 2 048 staging buckets of 32 slots do not fit a CU's LDS, so the fine split does not stage per bucket: block b1 (one per bucket, one
// per CU) takes the bucket's items tile by tile, counts the tile's items per fine partition (an LDS atomic hands out the item's rank
// inside its partition), turns the counts into offsets, scatters the tile into LDS in partition order and writes it out -- an item of
// sorted position i goes to region (b1, f) at cur[f] + (i - off[f]): consecutive lanes write consecutive 16-byte words of a region, about
// 64 bytes per partition and tile.  The cursors live in the block's LDS (it is the only writer of its 256 << F regions): no atomics on
// memory.  What does not fit a region is inserted occurrence by occurrence, which also sets the fine partition's dirty bit.
template <int NT, int IPT>
__global__ __launch_bounds__(NT) void skm_split_sort(const KmerGroupDev G, const KmerTable T, const uint32_t k)
{
    constexpr uint32_t TILE = NT * IPT, NBMAX = KG_FAN << 3;
    static_assert(NBMAX == 2 * NT, "a thread scans two counters");
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    Item *s_items = reinterpret_cast<Item *>(lds);                 // [TILE] the tile in partition order
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_items + TILE); // [NBMAX] items of the tile per fine partition, then their exclusive prefix
    uint32_t *s_cur = s_cnt + NBMAX;                               // [NBMAX] items the regions hold
    uint32_t *s_pre = s_cur + NBMAX;                               // [KG_FAN + 1] exclusive prefix of the bucket's sub-region sizes
    uint32_t *s_er = s_pre + KG_FAN + 4;                           // [KG_MAX_RUNS] epoch of run j, relative
    uint32_t *s_ws = s_er + KG_MAX_RUNS;                           // [NT / 64]
    const SkmGeom g = skm_geom(k);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t NB = (uint32_t)KG_FAN << T.fine, fsh = 3u - T.fine, b1 = blockIdx.x;
    for (uint32_t i = tid; i < NBMAX; i += NT) { s_cnt[i] = 0u; s_cur[i] = 0u; }
    if (tid < KG_FAN) s_pre[tid + 1] = G.cur1[tid * KG_FAN + b1];
    for (uint32_t j = tid; j < G.n_runs; j += NT) s_er[j] = G.run_epoch[j];
    __syncthreads();
    if (tid == 0) { uint32_t run_ = 0; s_pre[0] = 0u; for (int i = 1; i <= KG_FAN; ++i) { run_ += s_pre[i]; s_pre[i] = run_; } }
    __syncthreads();
    const uint32_t N = s_pre[KG_FAN];
    Item *const l2 = reinterpret_cast<Item *>(G.l2) + (size_t)b1 * NB * G.cap2f;
    auto slow = [&](const Item it) { skm_insert_item_atomic(T, it, G.epoch_base + skm_item_run(it.y), g, G.first_hist, G.n_epochs); };
    // the items of tile t0 (item t0 + j NT + tid in register j; KS_NONE past the bucket's end): fetched one tile ahead of their use
    auto fetch = [&](const uint32_t t0, Item (&dst)[IPT]) {
        uint32_t sr = 0;
        if (t0 + (uint32_t)tid < N) { // the sub-region that holds item t0 + tid of the bucket
            const uint32_t i = t0 + (uint32_t)tid;
            uint32_t lo = 0, hi = KG_FAN - 1;
            while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= i) lo = mid; else hi = mid - 1; }
            sr = lo;
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = t0 + (uint32_t)(j * NT + tid);
            dst[j] = make_ulonglong2(0ull, KS_NONE);
            if (i < N) {
                while (i >= s_pre[sr + 1]) ++sr; // (i < N = s_pre[256]: ends)
                dst[j] = l1_region(G, b1, sr)[i - s_pre[sr]];
            }
        }
    };
    Item nxt[IPT];
    fetch(0u, nxt);
#pragma unroll 1
    for (uint32_t t0 = 0; t0 < N; t0 += TILE) {
        const uint32_t n_t = N - t0 < TILE ? N - t0 : TILE;
        // ---- count: tk = fine partition << 16 | rank of the item among the tile's items of that partition ----
        Item it[IPT];
        uint32_t tk[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) it[j] = nxt[j];
        fetch(t0 + TILE, nxt); // (in flight under the counting, the sort and the write-out of this tile)
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            tk[j] = 0xffffffffu;
            if (it[j].y != KS_NONE) {
                it[j].y = (it[j].y & ~SKM_RUN_MASK) | ((u64)s_er[skm_item_run(it[j].y)] << SKM_RUN_SHIFT);
                const uint32_t f = (skm_item_part(it[j].y) >> fsh) & (NB - 1u);
                tk[j] = (f << 16) | atomicAdd(&s_cnt[f], 1u);
            }
        }
        __syncthreads();
        // ---- counts -> exclusive offsets (a thread owns counters 2 tid, 2 tid + 1) ----
        const uint32_t a0 = s_cnt[2 * tid], a1 = s_cnt[2 * tid + 1];
        const uint32_t incl = (uint32_t)wave_incl_scan_add((int)(a0 + a1));
        if (lane == 63) s_ws[wave] = incl;
        __syncthreads();
        const uint32_t wsc = (uint32_t)wave_incl_scan_add(lane < NT / 64 ? (int)s_ws[lane] : 0);
        const uint32_t base = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wsc, wave - 1) : 0u;
        const uint32_t excl = base + incl - (a0 + a1);
        s_cnt[2 * tid] = excl; s_cnt[2 * tid + 1] = excl + a0;
        __syncthreads();
        // ---- the tile in partition order ----
#pragma unroll
        for (int j = 0; j < IPT; ++j)
            if (tk[j] != 0xffffffffu) s_items[s_cnt[tk[j] >> 16] + (tk[j] & 0xffffu)] = it[j];
        __syncthreads();
        // ---- out: sorted position -> (region, place) ----
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t pos = (uint32_t)(j * NT + tid);
            if (pos < n_t) {
                const Item x = s_items[pos];
                const uint32_t f = (skm_item_part(x.y) >> fsh) & (NB - 1u);
                const uint32_t at = s_cur[f] + (pos - s_cnt[f]);
                if (at < G.cap2f) l2[(size_t)f * G.cap2f + at] = x; else slow(x);
            }
        }
        __syncthreads();
        { const uint32_t c0 = s_cur[2 * tid] + a0, c1 = s_cur[2 * tid + 1] + a1;
          s_cur[2 * tid] = c0 < G.cap2f ? c0 : G.cap2f; s_cur[2 * tid + 1] = c1 < G.cap2f ? c1 : G.cap2f;
          s_cnt[2 * tid] = 0u; s_cnt[2 * tid + 1] = 0u; }
        __syncthreads();
    }
    for (uint32_t i = tid; i < NB; i += NT) G.cur2[(size_t)b1 * NB + i] = s_cur[i];
}
constexpr size_t KS_SORT_LDS = (size_t)1024 * 8 * 16 + (size_t)(KG_FAN << 3) * 8 + (size_t)(KG_FAN + 4 + KG_MAX_RUNS + 16) * 4;

// ---- combine + insert: one workgroup per partition --------------------------------------------------------------------------
// A lane handles one OCCURRENCE, not one item: the items of a tile (one per thread, a coalesced load) get their first occurrence
// index by a block-wide prefix sum of their k-mer counts; a bit per occurrence index marks "an item starts here" and a word per 64
// occurrences holds the index of the first item that starts in it, so lane l of the wave that takes occurrences 64 r .. 64 r + 63
// finds its item (rank of the last start bit at or before l) and its position inside it (distance to that bit) with a handful of
// bit operations, fetches the item's 16 bytes again (a gather that hits the cache: the tile was just read) and cuts its k-mer out.
// Every wave iteration has 64 busy lanes whatever the lengths of the runs are; the first form of this kernel walked an item per
// lane with rolling words and ran at 31 % lane use (profiles/r5a/pmc_skm_first.txt: 5.7 scalar + 3.2 vector instructions per
// occurrence, scalar-issue bound).
// The LDS table holds (h, count, smallest epoch), h = mix62(canonical key); LDS slot = the top 12 bits of h and the table slot inside
// the key's slice = the top bits of h too, so LDS order is table order.  When the LDS table is nearly full every wave stops
// where it is, the table is written out -- ONE update per distinct key -- and cleared, and the waves go on.
//
// MODE (round 6).  KS_GROUP: a group that is flushed while the pass goes on, one workgroup per 16-bit partition, every key through the table.
// KS_COUNT: the group is the whole pass.  A workgroup takes fine partitions p = blockIdx.x, + gridDim.x, ...; partition p's items are ALL
// the occurrences of its keys in the pass, so when they fit one round of the LDS table -- and the per-occurrence path has not put keys of
// p into the table (KmerTable::dirty) -- the keys never reach the table: count -> histogram of counts (FaQCs.cpp:518-521), first epoch ->
// keys by first epoch (the sampling points, trim.cpp:157-185), and the LDS table is cleared for the next partition.  A partition that
// does not fit, or is dirty, is put on a list (KmerGroupDev::redo) and left to KS_REDO: a launch behind this one that takes the listed
// fine partitions through their slices as KS_GROUP would, and then sweeps each slice (histogram of counts, slots back to empty).  The table
// is clean behind the pair of launches, but for keys in the overflow area (KmerTable::stats[3] counts those).  Two kernels because the
// count-only one has to keep to 64 registers (two workgroups of 1 024 threads per CU) and the table code costs twenty more.
enum { KS_GROUP = 0, KS_COUNT = 1, KS_REDO = 2 };
// an entry of the LDS table: one ds_read_b128 fetches all of it
struct __attribute__((aligned(16))) KsEnt { u64 key; uint32_t cnt, ep; }; // ep = smallest epoch << 3 | the low three bits of the key's 19-bit partition
template <int NT, bool K31, int MODE>
__global__ __launch_bounds__(NT, MODE == KS_REDO ? 4 : 8) void skm_combine(const KmerGroupDev G, const KmerTable T, const uint32_t k_arg, const uint32_t diag)
{
    constexpr bool FINAL = MODE != KS_GROUP;
    constexpr int LS = 4096, NWV = NT / 64, IPT = MODE == KS_GROUP ? 1 : 2, TILE = NT * IPT; // a tile: IPT consecutive items per thread (KS_GROUP's 8 KB claim bitmap leaves room for one)
    constexpr uint32_t LMASK = LS - 1, LIMIT = LS - 1280; // (KS_GROUP, KS_REDO) stop adding keys at 69 % (every wave may add 64 more before it sees the count)
    constexpr uint32_t PROBE_CAP = 256;                    // (KS_COUNT) a probe sequence this long: the partition does not fit the LDS table
    constexpr int MAXW = (TILE * SKM_W_MAX + 63) / 64;     // 64-occurrence words of a tile
    constexpr int CLAIM_BITS = FINAL ? KG_SLICE_MAX >> 3 : KG_SLICE_MAX; // slots of the slices this workgroup may claim in: one fine slice / a 16-bit partition's
    constexpr uint32_t CH = 256;                            // FINAL: counts below CH are added up in LDS first
    __shared__ KsEnt s_tab[LS];
    __shared__ int s_hist[KG_EPOCH_SPAN];
    __shared__ uint32_t s_claim[MODE == KS_COUNT ? 1 : CLAIM_BITS / 32]; // slots this launch has claimed
    __shared__ uint32_t s_bits[2 * MAXW];           // bit o: an item's first occurrence has index o inside the tile
    __shared__ uint32_t s_rank[MAXW];               // index (inside the tile) of the first item that starts in the word
    __shared__ uint32_t s_wsum[NWV];
    __shared__ uint32_t s_chist[FINAL ? CH : 2];
    __shared__ uint32_t s_nkeys, s_more;
    static_assert(NWV * 64 + LIMIT <= LS && LS == 4 * NT, "waves overshoot the limit by at most 64 keys each; four LDS slots per thread");
    const SkmGeom g = skm_geom(K31 ? 31u : k_arg);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t fsh = 3u - T.fine;
    const uint4 ent_empty = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0xffffffffu);
    auto clear = [&]() {
#pragma unroll
        for (int j = 0; j < LS / NT; ++j) *reinterpret_cast<uint4 *>(&s_tab[j * NT + tid]) = ent_empty;
        if (tid == 0) { s_nkeys = 0u; s_more = 0u; }
    };
    clear();
    for (int i = tid; i < KG_EPOCH_SPAN; i += NT) s_hist[i] = 0;
    if (MODE != KS_COUNT) for (int i = tid; i < CLAIM_BITS / 32; i += NT) s_claim[i] = 0u;
    if (FINAL) for (uint32_t i = tid; i < CH; i += NT) s_chist[i] = 0u;
    // one key with `c` occurrences joins the histogram of counts (FINAL)
    auto count_key = [&](const bool live, const uint32_t c) {
        const unsigned long long once = __ballot(live && c == 1u); // (most keys of a real run are seen once: sequencing errors -- one add per wave)
        if (once != 0ull && lane == __builtin_ctzll(once)) atomicAdd(&s_chist[1], (uint32_t)__popcll(once));
        if (live && c != 1u) {
            if (c < CH) atomicAdd(&s_chist[c], 1u);
            else if (c < G.dense_n) atomicAdd(&G.dense[c], 1ull);
            else { const unsigned long long at = atomicAdd(G.n_big, 1ull); if (at < G.big_cap) G.big[at] = c; }
        }
    };
    const u64 lane_lt = (1ull << lane) - 1ull;
    const uint32_t n_parts = MODE == KS_REDO ? uniu(*G.n_redo) : (FINAL ? 1u << (16u + T.fine) : (uint32_t)(KG_FAN * KG_FAN));
    const size_t pcap = FINAL ? G.cap2f : G.cap2;
    const Item none = make_ulonglong2(0ull, KS_NONE);
    // (KS_COUNT) the first tile of the NEXT partition is fetched under the last tile of this one
    uint32_t pre_p = 0xffffffffu;
    Item pre0 = none, pre1 = none;
#pragma unroll 1
    for (uint32_t pi = blockIdx.x; pi < n_parts; pi += gridDim.x) {
    const uint32_t p = MODE == KS_REDO ? uniu(G.redo[pi]) : pi;
    const uint32_t n_p = uniu(G.cur2[p]); // (split == 1)
    // the per-occurrence path has put keys of p into its slice: KS_REDO takes the partition -- also when NONE of its items reached its
    // level-2 region (a partition whose few items all overflowed their level-1 sub-regions): its slice still has to be swept
    if (MODE == KS_COUNT && T.dirty && ((uniu(T.dirty[p >> 5]) >> (p & 31u)) & 1u)) {
        if (tid == 0) G.redo[atomicAdd(G.n_redo, 1u)] = p;
        continue;
    }
    if (MODE != KS_REDO && n_p == 0) continue; // (block-uniform)
    const Item *src = reinterpret_cast<const Item *>(G.l2) + (size_t)p * pcap; // (one region per partition in this mode: kg_init)
    // the next partition this workgroup takes (KS_COUNT), for the fetch ahead
    const uint32_t p2 = pi + gridDim.x;
    const uint32_t n_p2 = MODE == KS_COUNT && p2 < n_parts ? uniu(G.cur2[p2]) : 0u;
    Item nx0, nx1;
    if (MODE == KS_COUNT && pre_p == p) { nx0 = pre0; nx1 = pre1; }
    else {
        nx0 = (uint32_t)(IPT * tid) < n_p ? src[IPT * tid] : none;
        nx1 = IPT == 2 && (uint32_t)(IPT * tid) + 1u < n_p ? src[IPT * tid + 1] : none;
    }
    bool abandon = false; // (KS_COUNT) more distinct keys than the LDS table takes

    // ONE table update per key of the LDS table, none of them a device-scope atomic: the slices belong to this workgroup for the
    // length of the launch, an empty slot is claimed through the workgroup's LDS bitmap, and no other thread holds this key.
    auto write_out = [&]() {
        if (MODE == KS_COUNT || (diag & 1u)) return; // (FAQCS_SKM_DIAG: what the table updates cost; wrong results)
        constexpr int KPT = LS / NT, KB = 2; // keys per thread, looked up KB at a time (4: spills, 27.0 instead of 23.7 ms on 16 M reads)
        static_assert(LS % NT == 0 && KPT % KB == 0, "keys per thread");
        const Slice s0 = slice_of_fine(T, FINAL ? p : p << T.fine); // (every fine slice has this size and shift)
        const u64 win = s0.mask + 1 < (u64)KS_PROBE_MAX ? s0.mask + 1 : (u64)KS_PROBE_MAX;
        auto base_of = [&](const uint32_t epf) -> u64 { return FINAL ? s0.base : s0.base + (u64)((epf & 7u) >> fsh) * (s0.mask + 1); };
#pragma unroll 1
        for (int j0 = 0; j0 < KPT; j0 += KB) {
            uint4 ew[KB];
            ulonglong2 first[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                ew[j] = *reinterpret_cast<const uint4 *>(&s_tab[(j0 + j) * NT + tid]);
                const u64 kw = ((u64)ew[j].y << 32) | ew[j].x;
                const u64 gslot = (kw >> s0.shift) & s0.mask;
                first[j] = make_ulonglong2(0ull, 0ull);
                if (kw != ~0ull) first[j] = *reinterpret_cast<const ulonglong2 *>(&T.slots[base_of(ew[j].w) + gslot]);
            }
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const u64 h = ((u64)ew[j].y << 32) | ew[j].x;
                if (h == ~0ull) continue;
                const uint32_t e_rel = ew[j].w >> 3, e = G.epoch_base + e_rel, cnt = ew[j].z;
                const u64 sbase = base_of(ew[j].w);
                u64 gslot = (h >> s0.shift) & s0.mask;
                bool placed = false;
                ulonglong2 cur = first[j];
#pragma unroll 1
                for (u64 probe = 0; probe < win; ++probe) {
                    KmerSlot *sl = &T.slots[sbase + gslot];
                    if (probe) cur = *reinterpret_cast<const ulonglong2 *>(sl);
                    if (cur.x == ~0ull) {
                        const uint32_t cb = (uint32_t)(sbase - s0.base + gslot), bit = 1u << (cb & 31u);
                        if (!(atomicOr(&s_claim[MODE == KS_COUNT ? 0 : cb >> 5], bit) & bit)) { // a new key: key, count - 1 and first epoch in one plain store
                            *reinterpret_cast<ulonglong2 *>(sl) = make_ulonglong2(h, (u64)(cnt - 1u) | ((u64)e << 32));
                            atomicAdd(&s_hist[e_rel], 1);
                            placed = true;
                            break;
                        }
                    } else if (cur.x == h) {
                        uint32_t c1 = (uint32_t)cur.y + cnt, fe = (uint32_t)(cur.y >> 32);
                        if (e < fe) { // (a key the overflow path inserted during this group, or an earlier write-out of this launch, can hold a later epoch)
                            if (fe != 0xffffffffu) {
                                if (fe >= G.epoch_base && fe - G.epoch_base < (uint32_t)KG_EPOCH_SPAN) atomicAdd(&s_hist[fe - G.epoch_base], -1);
                                else hist_add(G.first_hist, fe, G.n_epochs, -1);
                            }
                            atomicAdd(&s_hist[e_rel], 1);
                            fe = e;
                        }
                        *reinterpret_cast<u64 *>(&sl->count_m1) = (u64)c1 | ((u64)fe << 32);
                        placed = true;
                        break;
                    }
                    gslot = (gslot + 1) & s0.mask;
                }
                if (!placed) ovf_insert_atomic(T, h, e, cnt, G.first_hist, G.n_epochs); // the window is full: the key lives in the overflow area
            }
        }
    };

#pragma unroll 1
    for (uint32_t i0 = 0; i0 < n_p; i0 += TILE) {
        const Item it0 = nx0, it1 = nx1;
        {   // the next tile: of this partition, or (KS_COUNT) the first one of the next
            const uint32_t i = i0 + (uint32_t)TILE + (uint32_t)(IPT * tid);
            if (i0 + (uint32_t)TILE < n_p) { nx0 = i < n_p ? src[i] : none; nx1 = IPT == 2 && i + 1u < n_p ? src[i + 1u] : none; }
            else if (MODE == KS_COUNT && n_p2) {
                const Item *src2 = reinterpret_cast<const Item *>(G.l2) + (size_t)p2 * pcap;
                pre0 = 2u * (uint32_t)tid < n_p2 ? src2[2 * tid] : none;
                pre1 = 2u * (uint32_t)tid + 1u < n_p2 ? src2[2 * tid + 1] : none;
                pre_p = p2;
            }
        }
        const uint32_t n_t = n_p - i0 < (uint32_t)TILE ? n_p - i0 : (uint32_t)TILE; // items of this tile
        const uint32_t nk0 = it0.y != KS_NONE ? skm_item_kmers(it0.y) : 0u, nk1 = it1.y != KS_NONE ? skm_item_kmers(it1.y) : 0u;
        // first occurrence index of every item: block-wide exclusive prefix sum of the k-mer counts
        const uint32_t incl = (uint32_t)wave_incl_scan_add((int)(nk0 + nk1));
        if (lane == 63) s_wsum[wave] = incl;
        for (int i = tid; i < 2 * MAXW; i += NT) s_bits[i] = 0u;
        for (int i = tid; i < MAXW; i += NT) s_rank[i] = n_t;
        __syncthreads();
        const uint32_t wsc = (uint32_t)wave_incl_scan_add(lane < NWV ? (int)s_wsum[lane] : 0);
        const uint32_t base = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wsc, wave - 1) : 0u;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wsc, NWV - 1); // occurrences of the tile
        {
            const uint32_t st0 = base + incl - nk0 - nk1, st1 = st0 + nk0;
            if (nk0) { atomicOr(&s_bits[st0 >> 5], 1u << (st0 & 31)); atomicMin(&s_rank[st0 >> 6], (uint32_t)(IPT * tid)); }
            if (nk1) { atomicOr(&s_bits[st1 >> 5], 1u << (st1 & 31)); atomicMin(&s_rank[st1 >> 6], (uint32_t)(IPT * tid) + 1u); }
        }
        __syncthreads();
        const uint32_t n_words = (total + 63u) >> 6;
        uint32_t r = (uint32_t)wave;
        // this lane's occurrence of the 64 with indices 64 rr ..: (item, position inside it) from the start bits; the item is fetched
        // a turn ahead of its use
        uint32_t j_nx = 0;
        Item iw_nx = make_ulonglong2(0ull, 0ull);
        bool todo_nx = false;
        auto locate = [&](const uint32_t rr) {
            todo_nx = false;
            if (rr >= n_words) return;
            const uint32_t wlo = s_bits[2 * rr], whi = s_bits[2 * rr + 1]; // (the same address in every lane: a broadcast)
            const u64 word = ((u64)whi << 32) | wlo;
            todo_nx = 64u * rr + (uint32_t)lane < total;
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u)); // start bits below this lane
            const uint32_t own = (uint32_t)((word >> lane) & 1ull);
            const uint32_t idx = s_rank[rr] + before + own - 1u;
            j_nx = 0;
            if (!own) {
                const u64 wl = word & lane_lt;
                if (wl) j_nx = (uint32_t)lane - (63u - (uint32_t)__clzll((long long)wl));
                else { // the item started in the word before (a run has at most 17 k-mers)
                    const u64 prev = rr ? ((u64)s_bits[2 * rr - 1] << 32) | s_bits[2 * rr - 2] : 1ull; // (rr == 0: only lanes past the tile's last occurrence)
                    j_nx = (uint32_t)lane + 1u + (uint32_t)__clzll((long long)prev);
                }
            }
            if (todo_nx) iw_nx = src[i0 + idx];
        };
        locate(r);
#pragma unroll 1
        for (;;) {
#pragma unroll 1
            for (; r < n_words; r += NWV) {
                if (MODE != KS_COUNT && uniu(__hip_atomic_load(&s_nkeys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= LIMIT) break;
                const Item iw = iw_nx;
                const uint32_t j = j_nx;
                bool todo = todo_nx;
                locate(r + NWV);
                const uint32_t ep = (skm_item_run(iw.y) << 3) | (FINAL ? 0u : skm_item_part(iw.y) & 7u);
                u64 fwd, rc;
                if (K31) { // bases j .. j + 30: a 62-bit window at bit 2 j (<= 32) of the 94-bit string
                    const uint32_t d0 = (uint32_t)iw.x, d1 = (uint32_t)(iw.x >> 32), d2 = (uint32_t)iw.y, sh = 2u * j;
                    const uint32_t lo = sh < 32u ? __builtin_amdgcn_alignbit(d1, d0, sh) : d1;
                    const uint32_t hi = (sh < 32u ? __builtin_amdgcn_alignbit(d2, d1, sh) : d2) & SKM_M30;
                    fwd = ((u64)hi << 32) | lo;
                    rc = (skm_rev2_64(fwd) >> 2) ^ 0x2AAAAAAAAAAAAAAAull;
                } else {
                    const uint32_t sh = 2u * j;
                    const u64 hi94 = iw.y & (u64)SKM_M30;
                    fwd = (sh ? (iw.x >> sh) | (hi94 << (64u - sh)) : iw.x) & g.kmask2;
                    rc = (skm_rev2_64(fwd) >> (64u - 2u * g.k)) ^ (0xAAAAAAAAAAAAAAAAull & g.kmask2);
                }
                const u64 h = skm_mix62(fwd < rc ? fwd : rc);
                if (diag & 2u) todo = false; // (FAQCS_SKM_DIAG: what the LDS table costs; wrong results)
                // count h in the LDS table: a wave-uniform loop over the probe sequence, one slot a turn -- key, count and epoch in ONE 16-byte read;
                // a lane that finds its key (or claims an empty slot for it) adds its occurrence there and then and drops out
                uint32_t s = (uint32_t)(h >> 50);
                bool claimed = false;
                uint32_t turns = 0;
#pragma unroll 1
                for (;;) {
                    s &= LMASK;
                    const uint4 e = *reinterpret_cast<const uint4 *>(&s_tab[s]);
                    u64 kk = ((u64)e.y << 32) | e.x;
                    if (todo && kk == ~0ull) { // (an empty slot may have been taken since it was read: the compare-and-swap says by whom)
                        const u64 old = atomicCAS(&s_tab[s].key, ~0ull, h);
                        claimed = claimed || old == ~0ull;
                        kk = old == ~0ull ? h : old;
                    }
                    const bool hit = todo && kk == h;
                    if (hit) {
                        atomicAdd(&s_tab[s].cnt, 1u);
                        if (ep < e.w) atomicMin(&s_tab[s].ep, ep); // (e.w may be out of date -- it only ever falls, so a test that says "not smaller" is right)
                    }
                    todo = todo && !hit; // (slot s holds another key -- for good: keys do not leave the table inside a round)
                    ++s;
                    if (!__any(todo)) break;
                    if (MODE == KS_COUNT && ++turns >= PROBE_CAP) { if (lane == 0) s_more = 1u; break; } // (the table is as good as full)
                }
                if (MODE != KS_COUNT) {
                    const uint32_t c = (uint32_t)__popcll(__ballot(claimed));
                    if (lane == 0 && c) atomicAdd(&s_nkeys, c);
                }
            }
            if (MODE != KS_COUNT && r < n_words) s_more = 1u;
            __syncthreads();
            const bool more = __hip_atomic_load(&s_more, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
            if (!more) break; // (s_more is only written above, before the barrier, and cleared below behind one)
            if (MODE == KS_COUNT) { abandon = true; break; } // (left to KS_REDO, from its first item on)
            write_out();
            __syncthreads();
            clear();
            // (the next write-out reads slots this one stored: the waves of a workgroup share their CU's vector L1, which is write-through, so
            // the barriers' workgroup-scope fences are all it takes -- as long as the workgroup's waves run on ONE CU, which is how every launch
            // of this library is dispatched; a threadgroup-split dispatch (tgsplit: the waves of a workgroup spread over several CUs) would
            // need the device-scope fence back.  That fence, a __threadfence() here, made every round of every workgroup write back and
            // invalidate its XCD's whole L2: 3 x the time per occurrence as soon as partitions needed two rounds, 304 instead of 147 ms per
            // bench step with groups of 1.5 x 2^30 occurrences)
            __syncthreads();
        }
        if (MODE == KS_COUNT && abandon) break;
    }
    __syncthreads();
    if (MODE == KS_COUNT && abandon) {
        if (tid == 0) G.redo[atomicAdd(G.n_redo, 1u)] = p;
        clear();
    } else if (MODE != KS_COUNT) {
        write_out();
        __syncthreads();
        if (FINAL && !(diag & 1u)) { // the slice's keys (this launch's and the per-occurrence path's) join the histogram of counts; the slice is empty again
            const Slice sc = slice_of_fine(T, p);
            const ulonglong2 empty = make_ulonglong2(~0ull, 0xffffffff00000000ull);
            for (u64 i = tid; i <= (sc.mask | (u64)(NT - 1)); i += NT) { // (whole waves: count_key() votes)
                const bool in = i <= sc.mask;
                ulonglong2 v = empty;
                if (in) v = *reinterpret_cast<const ulonglong2 *>(&T.slots[sc.base + i]);
                const bool live = in && v.x != ~0ull;
                count_key(live, (uint32_t)v.y + 1u);
                if (live) *reinterpret_cast<ulonglong2 *>(&T.slots[sc.base + i]) = empty;
            }
            for (u64 i = tid; i < (sc.mask + 32) / 32; i += NT) s_claim[MODE == KS_COUNT ? 0 : i] = 0u;
        }
        clear();
    } else { // KS_COUNT, every key of the partition is in the LDS table: histograms, and the table is empty again
        uint4 ew[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ew[j] = *reinterpret_cast<const uint4 *>(&s_tab[j * NT + tid]);
        clear();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = (ew[j].x & ew[j].y) != 0xffffffffu;
            count_key(live, ew[j].z);
            if (live) atomicAdd(&s_hist[ew[j].w >> 3], 1);
        }
    }
    __syncthreads();
    } // partitions
    __syncthreads();
    for (int i = tid; i < KG_EPOCH_SPAN; i += NT)
        if (s_hist[i]) hist_add(G.first_hist, G.epoch_base + (uint32_t)i, G.n_epochs, (long long)s_hist[i]);
    if (FINAL) for (uint32_t i = tid; i < CH; i += NT) if (s_chist[i]) atomicAdd(&G.dense[i], (unsigned long long)s_chist[i]);
}

__global__ void skm_group_reset(const KmerGroupDev G)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)KG_FAN * KG_FAN) G.cur1[i] = 0u;
}

// ---- sender side of the multi-GPU exchange -----------------------------------------------------------------------------------
// Bucket b (the top 8 bits of a partition) belongs to rank (b * world) >> 8, so the level-1 buffers ARE grouped by destination:
// count[d] = items of d's buckets + d's share of the spill, then a copy of every (bucket, sub-region) to its place behind an
// exclusive scan; the spilled items of a destination follow its regions.  The run field of an item becomes its absolute epoch.
// scratch (u64): region_offset[65536], then spill_base[64], spill_cursor[64]
__global__ __launch_bounds__(256) void skm_outbox_count(const KmerGroupDev G, const uint32_t world, u64 *dest_count, u64 *scratch)
{
    __shared__ u64 s_sum[256];
    __shared__ uint32_t s_spill[64];
    u64 *region_offset = scratch, *spill_base = scratch + KG_FAN * KG_FAN, *spill_cursor = spill_base + 64;
    const uint32_t b = threadIdx.x;
    if (b < 64) s_spill[b] = 0u;
    __syncthreads();
    const uint32_t n_spill = G.spill ? (*G.spill_n < G.spill_cap ? *G.spill_n : G.spill_cap) : 0u;
    for (uint32_t i = b; i < n_spill; i += 256) atomicAdd(&s_spill[(skm_item_bucket(reinterpret_cast<const Item *>(G.spill)[i].y) * world) >> 8], 1u);
    u64 n = 0;
    for (uint32_t s = 0; s < (uint32_t)KG_FAN; ++s) n += G.cur1[s * KG_FAN + b];
    s_sum[b] = n;
    __syncthreads();
    if (b == 0) { // 256 buckets: a serial exclusive scan, the spill of a destination behind its last bucket
        u64 run_ = 0;
        uint32_t d_prev = 0;
        for (uint32_t i = 0; i < 256; ++i) {
            const uint32_t d = (i * world) >> 8;
            if (d != d_prev) { spill_base[d_prev] = run_; run_ += s_spill[d_prev]; d_prev = d; }
            const u64 v = s_sum[i]; s_sum[i] = run_; run_ += v;
        }
        spill_base[d_prev] = run_;
    }
    if (b < 64) { spill_cursor[b] = 0; if (b < world) dest_count[b] = s_spill[b]; }
    __syncthreads();
    u64 o = s_sum[b];
    for (uint32_t s = 0; s < (uint32_t)KG_FAN; ++s) { region_offset[b * KG_FAN + s] = o; o += G.cur1[s * KG_FAN + b]; }
    __threadfence();
    __syncthreads();
    atomicAdd(&dest_count[(b * world) >> 8], n);
}
__global__ __launch_bounds__(256) void skm_outbox_copy(const KmerGroupDev G, const uint32_t world, u64 *__restrict__ scratch, Item *__restrict__ out)
{
    const u64 *region_offset = scratch;
    u64 *spill_base = scratch + KG_FAN * KG_FAN, *spill_cursor = spill_base + 64;
    auto with_epoch = [&](Item it) { it.y = (it.y & ~SKM_RUN_MASK) | ((u64)(G.epoch_base + G.run_epoch[skm_item_run(it.y)]) << SKM_RUN_SHIFT); return it; };
    if (blockIdx.x < (uint32_t)KG_FAN * KG_FAN) {
        const uint32_t b = blockIdx.x >> 8, s = blockIdx.x & 255u;
        const uint32_t n = G.cur1[s * KG_FAN + b];
        const Item *src = l1_region(G, b, s);
        Item *dst = out + region_offset[b * KG_FAN + s];
        for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = with_epoch(src[i]);
        return;
    }
    // the blocks behind: the spill
    const uint32_t n_spill = G.spill ? (*G.spill_n < G.spill_cap ? *G.spill_n : G.spill_cap) : 0u;
    for (uint32_t i = (blockIdx.x - KG_FAN * KG_FAN) * 256 + threadIdx.x; i < n_spill; i += 16 * 256) {
        const Item it = reinterpret_cast<const Item *>(G.spill)[i];
        const uint32_t d = (skm_item_bucket(it.y) * world) >> 8;
        out[spill_base[d] + atomicAdd(&spill_cursor[d], 1ull)] = with_epoch(it);
    }
}

constexpr int KS_NW = 16;
constexpr size_t KS_EXTRACT_LDS = KS_STAGE_BYTES + (size_t)KS_NW * KS_ROWS * 4;

} // namespace

// blocks of an extraction launch over n_reads reads
uint32_t faqcs_skm_grid(uint32_t n_reads, int n_cu)
{
    uint32_t grid = (n_reads + 4 * KS_NW - 1) / (4 * KS_NW);
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu; // one block per CU: its staging area is most of the CU's LDS
    if (grid > (uint32_t)KG_FAN) grid = KG_FAN;       // ... and one sub-region of every bucket per block
    return grid ? grid : 1u;
}

// list != null: the reads list[0 .. *list_n) (device memory; what a faqcs_launch_skm_extract16 over [r_begin, r_end) left) instead of the range
hipError_t faqcs_launch_skm_extract(const DevParams &P, uint32_t k, const KmerGroupDev &G, const KmerTable &T, uint32_t run, uint32_t rot,
                                    uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                    uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, int n_cu, hipStream_t st,
                                    const uint32_t *list, const uint32_t *list_n, uint32_t grid_blocks)
{
    if (r_end <= r_begin) return hipSuccess;
    static unsigned long long done31 = 0, doneg = 0;
    const dim3 grid(grid_blocks ? grid_blocks : faqcs_skm_grid(r_end - r_begin, n_cu)), block(KS_NW * 64);
    if (k == 31) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_extract<KS_NW, true>), KS_EXTRACT_LDS, done31);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((skm_extract<KS_NW, true>), grid, block, KS_EXTRACT_LDS, st,
                           P, k, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results), list, list_n);
    } else {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_extract<KS_NW, false>), KS_EXTRACT_LDS, doneg);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((skm_extract<KS_NW, false>), grid, block, KS_EXTRACT_LDS, st,
                           P, k, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results), list, list_n);
    }
    return hipGetLastError();
}

// k = 31, reads of up to 256 bases: four reads per wave and round; reads it cannot take (a non-base inside the kept window, a run of more
// than 17 k-mers) are listed in defer[0 .. *defer_n) for a faqcs_launch_skm_extract(list) behind it.  *defer_n must be zero.
uint32_t faqcs_skm_grid16(uint32_t n_reads, int n_cu)
{
    uint32_t grid = (n_reads + 16 * KS_NW - 1) / (16 * KS_NW);
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu;
    if (grid > (uint32_t)KG_FAN) grid = KG_FAN;
    return grid ? grid : 1u;
}
hipError_t faqcs_launch_skm_extract16(const DevParams &P, const KmerGroupDev &G, const KmerTable &T, uint32_t run, uint32_t rot,
                                      uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                      uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, uint32_t *defer, uint32_t *defer_n,
                                      int n_cu, hipStream_t st)
{
    if (r_end <= r_begin) return hipSuccess;
    static unsigned long long done = 0;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_extract16<KS_NW>), KS_STAGE_BYTES, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((skm_extract16<KS_NW>), dim3(faqcs_skm_grid16(r_end - r_begin, n_cu)), dim3(KS_NW * 64), KS_STAGE_BYTES, st,
                       P, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results), defer, defer_n);
    return hipGetLastError();
}

// blocks of an owner-side launch over n_items received items
uint32_t faqcs_skm_items_grid(unsigned long long n_items, int n_cu)
{
    unsigned long long grid = (n_items + 16 * KS_NW * 64 - 1) / (16ull * KS_NW * 64);
    if (grid > (unsigned long long)n_cu) grid = (unsigned long long)n_cu;
    if (grid > (unsigned long long)KG_FAN) grid = KG_FAN;
    return grid ? (uint32_t)grid : 1u;
}

hipError_t faqcs_launch_skm_items(const KmerGroupDev &G, const KmerTable &T, uint32_t k, uint32_t rot, const void *items, unsigned long long n_items,
                                  int n_cu, hipStream_t st)
{
    if (!n_items) return hipSuccess;
    static unsigned long long done = 0;
    const size_t lds = KS_STAGE_BYTES + (size_t)KG_EPOCH_SPAN * 4 + 16;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_items<KS_NW>), lds, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((skm_items<KS_NW>), dim3(faqcs_skm_items_grid(n_items, n_cu)), dim3(KS_NW * 64), lds, st, G, T, k, rot,
                       reinterpret_cast<const Item *>(items), n_items);
    return hipGetLastError();
}

// stages: 1 split, 2 combine, 4 cursor reset (7: a whole flush; the steps one by one: FAQCS_KMER_DEBUG's checks between them)
hipError_t faqcs_launch_skm_flush(const KmerGroupDev &G, const KmerTable &T, uint32_t k, hipStream_t st, uint32_t stages)
{
    static unsigned long long done = 0;
    const size_t lds = KS_STAGE_BYTES + (size_t)(KG_MAX_RUNS + KG_FAN) * 4 + 16;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_split<KS_NW>), lds, done);
    if (e != hipSuccess) return e;
    if (stages & 1u) hipLaunchKernelGGL((skm_split<KS_NW>), dim3(KG_FAN * G.split), dim3(KS_NW * 64), lds, st, G, T, k);
    static const uint32_t diag = [] { const char *e = getenv("FAQCS_SKM_DIAG"); return e ? (uint32_t)atoi(e) : 0u; }(); // (diagnostics: 1 no table updates, 2 no LDS counting)
    if ((stages & 2u) && k == 31) hipLaunchKernelGGL((skm_combine<1024, true, KS_GROUP>), dim3(KG_FAN * KG_FAN), dim3(1024), 0, st, G, T, k, diag);
    else if (stages & 2u) hipLaunchKernelGGL((skm_combine<1024, false, KS_GROUP>), dim3(KG_FAN * KG_FAN), dim3(1024), 0, st, G, T, k, diag);
    if (stages & 4u) hipLaunchKernelGGL(skm_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}

// the group is the whole pass (stages: 1 fine split, 2 count, 4 cursor reset)
hipError_t faqcs_launch_skm_finish(const KmerGroupDev &G, const KmerTable &T, uint32_t k, int n_cu, hipStream_t st, uint32_t stages)
{
    static unsigned long long done = 0;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&skm_split_sort<1024, 8>), KS_SORT_LDS, done);
    if (e != hipSuccess) return e;
    if (stages & 1u) hipLaunchKernelGGL((skm_split_sort<1024, 8>), dim3(KG_FAN), dim3(1024), KS_SORT_LDS, st, G, T, k);
    static const uint32_t diag = [] { const char *e = getenv("FAQCS_SKM_DIAG"); return e ? (uint32_t)atoi(e) : 0u; }();
    // two workgroups per CU are resident; sixteen per CU in the grid even out what the partitions' sizes differ by
    uint32_t grid = (uint32_t)n_cu * 16u;
    const uint32_t n_parts = 1u << (16u + T.fine);
    if (grid > n_parts) grid = n_parts;
    if (stages & 2u) {
        hipError_t e2 = hipMemsetAsync(G.n_redo, 0, 4, st);
        if (e2 != hipSuccess) return e2;
        // (the second launch finds its partitions in the list the first one leaves: none on an even input, and its workgroups end at once)
        if (k == 31) {
            hipLaunchKernelGGL((skm_combine<1024, true, KS_COUNT>), dim3(grid), dim3(1024), 0, st, G, T, k, diag);
            hipLaunchKernelGGL((skm_combine<1024, true, KS_REDO>), dim3((uint32_t)n_cu), dim3(1024), 0, st, G, T, k, diag);
        } else {
            hipLaunchKernelGGL((skm_combine<1024, false, KS_COUNT>), dim3(grid), dim3(1024), 0, st, G, T, k, diag);
            hipLaunchKernelGGL((skm_combine<1024, false, KS_REDO>), dim3((uint32_t)n_cu), dim3(1024), 0, st, G, T, k, diag);
        }
    }
    if (stages & 4u) hipLaunchKernelGGL(skm_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}

hipError_t faqcs_launch_skm_reset(const KmerGroupDev &G, hipStream_t st)
{
    hipLaunchKernelGGL(skm_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}

hipError_t faqcs_launch_skm_outbox(const KmerGroupDev &G, uint32_t world, unsigned long long *dest_count, unsigned long long *scratch,
                                   void *out, hipStream_t st)
{
    hipLaunchKernelGGL(skm_outbox_count, dim3(1), dim3(256), 0, st, G, world, dest_count, scratch);
    hipLaunchKernelGGL(skm_outbox_copy, dim3(KG_FAN * KG_FAN + 16), dim3(256), 0, st, G, world, scratch, reinterpret_cast<Item *>(out));
    return hipGetLastError();
}
