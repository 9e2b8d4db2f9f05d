// faqcs_kmer_group_kernel.hip -- combine-before-insert k-mer counting (gfx950, wave64).
//
// update_kmer() (trim.cpp:887-931) increments a hash-map entry per k-mer occurrence, and a device table that does the same
// pays one memory-side atomic per occurrence: 13-15 G/s on MI355X whatever the table (profiles/r3c/pmc_kmer_atomics.txt), with
// 86 % of them hitting keys the same batch has already touched.  Only (distinct, total) at the sampling points
// (trim.cpp:157-185) and the final histogram of counts (FaQCs.cpp:518-521) are observable, and both are functions of
// {key -> (count, first epoch)}, where epoch = index of the first sampling point that includes the occurrence.  So:
//
//   kmer_group_extract   one wave per read as before (ballot planes, funnel-shift windows), but an occurrence becomes a 62-bit
//                        mixed key h appended to one of 256 level-1 buckets (top 8 bits of h) through an LDS staging area
//                        that leaves the CU as whole 256-byte granules.  One launch per run of segments with one epoch; the
//                        bucket cursors after the launch are the run's bounds, so an item's epoch is its POSITION.
//   kmer_group_split     per full group: every bucket is split 256 ways by the next 8 bits of h; an item becomes
//                        epoch << 46 | low 46 bits of h.  65 536 partitions, each the only holder of its keys.
//   kmer_group_combine   one workgroup per partition: counts its items in an LDS hash table (key, min epoch, count), then ONE
//                        update per DISTINCT key of the table slice the partition owns (slot = h >> shift, so the slice is
//                        contiguous): a plain 16-byte read + 8-byte write for a key the table has, a compare-and-swap only to
//                        claim the slot of a new key.  No other workgroup touches these keys in this launch, so the value
//                        word needs no atomic; slot claims still do (probe sequences of neighbouring slices may cross).
//
// Whatever does not fit -- a bucket region that is full (heavy hitters: poly-A, k = 5), an LDS table that is full -- goes
// through kmer_insert_atomic, the per-occurrence path, so exactness never depends on a capacity.
#include "faqcs_kmer.h"

#include <stdlib.h>

namespace {

typedef unsigned long long u64;
#define KG_PAD (~0ull) /* "no item" in the prefetch registers of the scatter / combine loops */

__device__ __forceinline__ void hist_add(u64 *h, uint32_t e, uint32_t n_epochs, long long v)
{
    if (e < n_epochs) atomicAdd(&h[e], (u64)v);
}

// The table is cut into 65 536 slices, one per partition (slot = h >> shift, so a partition's keys start inside its slice), and
// a probe sequence WRAPS INSIDE ITS SLICE: whatever happens to a key happens inside the slice of its partition.  That is what
// lets kmer_group_combine claim slots without a device-scope atomic (the workgroup of a partition is the only one in its
// slice during that launch).  A slice that fills up raises the "table full" flag even if other slices have room.
struct Slice { u64 base, mask; }; // first slot, slots - 1
__device__ __forceinline__ Slice slice_of(const KmerTable &T, const u64 h)
{
    const u64 size = (T.mask + 1) >> 16;
    return Slice{(h >> 46) * size, size - 1};
}

// Per-occurrence insert of a mixed key (the fallback path; also where the general first-epoch rule lives):
// old = atomic min; a successful lowering moves the key from hist[old] to hist[epoch] -- the lowerings of one key form a chain,
// so the moves telescope to exactly one count at the key's final first epoch.
__device__ void kmer_insert_atomic(const KmerTable &T, const u64 h, const uint32_t epoch, const uint32_t count, u64 *first_hist,
                                   const uint32_t n_epochs)
{
    const Slice sc = slice_of(T, h);
    u64 g = (h >> T.shift) & sc.mask;
#pragma unroll 1
    for (u64 probe = 0; probe <= sc.mask; ++probe) {
        KmerSlot *sl = &T.slots[sc.base + g];
        typedef u64 ull2_t __attribute__((ext_vector_type(2)));
        const ull2_t cur = __builtin_nontemporal_load(reinterpret_cast<const ull2_t *>(sl));
        u64 seen = cur.x;
        uint32_t add = count;
        if (seen == ~0ull) {
            seen = slot_cas(&sl->key, ~0ull, h);
            if (seen == ~0ull) { seen = h; add = count - 1u; } // claimed: count_m1 = 0 already says "seen once"
        }
        if (seen == h) {
            if (add) slot_add(&sl->count_m1, add);
            const uint32_t old = slot_min_rtn(&sl->first_epoch, epoch);
            if (epoch < old) {
                if (old != 0xffffffffu) hist_add(first_hist, old, n_epochs, -1);
                hist_add(first_hist, epoch, n_epochs, 1);
            }
            return;
        }
        g = (g + 1) & sc.mask;
    }
    atomicOr(&T.stats[2], 1ull); // slice full
}

// ---- LDS staging shared by the two scatter kernels ----------------------------------------------------------------------
// 256 buckets x 64 slots.  put(): a ticket from the bucket's LDS counter; a lane whose ticket is past the last slot keeps
// its item for the next round.  drain(): wave w owns buckets [16 w, 16 w + 16) and writes every full granule (32 items, 256
// bytes) of them -- the whole wave takes part, lane i < take holds item i -- then moves what is left to the front.
// WHERE a granule goes needs no atomic: a block appends to sub-regions that are its own (one per bucket), so the cursors
// live in its LDS.  (The first versions appended to shared regions behind a returning device-scope atomic per granule: a
// round trip of microseconds per flush, 27 us per round of 2 800 items when taken on the spot, and still a wait per round
// when reserved a round ahead -- every s_waitcnt vmcnt counts loads, stores and atomics together, in issue order.)
template <int NW> struct Staging {
    static constexpr int BPW = KG_FAN / NW;
    u64 *items;     // [KG_FAN][KG_STAGE]
    uint32_t *cnt;  // [KG_FAN]
    uint32_t *cur;  // [KG_FAN] items this block's sub-region of every bucket holds
    uint32_t *flag; // [3] block_or
    __device__ __forceinline__ bool put(const uint32_t b, const u64 item) const
    {
        const uint32_t pos = atomicAdd(&cnt[b], 1u);
        if (pos < (uint32_t)KG_STAGE) { items[b * KG_STAGE + pos] = item; return true; }
        return false;
    }
    // Two buckets per step, one per half wave: lane l of a half holds item l of its bucket's granule, so a step's instructions
    // move 64 items.  write(b, pos, item): this lane's item goes to position pos of the block's sub-region of bucket b;
    // slow(b, item): the sub-region is full, the item takes the per-occurrence path.  A bucket that still holds a granule after its
    // step (64 staged items, or the final drain's remainder) stays in the mask for another one.
    template <class Write, class Slow>
    __device__ __forceinline__ void drain(const int wave, const int lane, const bool final, const uint32_t cap, Write &&write, Slow &&slow) const
    {
        const uint32_t n_l = lane < BPW ? cnt[wave * BPW + lane] : 0u;
        uint64_t m = __ballot(final ? n_l > 0u : n_l >= (uint32_t)KG_GRAN);
        const int half = lane >> 5, l32 = lane & 31;
#pragma unroll 1
        while (m) {
            const int i0 = uni(__ffsll((long long)m) - 1);
            const uint64_t m1 = m & (m - 1);
            const int i1 = m1 ? uni(__ffsll((long long)m1) - 1) : -1;
            const int bsel = half ? i1 : i0;
            const bool act = bsel >= 0;
            const uint32_t b = (uint32_t)(wave * BPW + (act ? bsel : i0));
            uint32_t n = cnt[b];
            n = n < (uint32_t)KG_STAGE ? n : (uint32_t)KG_STAGE;
            const uint32_t pos = cur[b];
            const uint32_t take = n < (uint32_t)KG_GRAN ? n : (uint32_t)KG_GRAN;
            const bool mine = act && (uint32_t)l32 < take;
            const u64 it = items[b * KG_STAGE + (uint32_t)l32];
            const bool fits = pos + take <= cap;
            if (mine) { if (fits) write(b, pos + (uint32_t)l32, it); else slow(b, it); }
            const uint32_t left = n - take;
            const bool mv = act && (uint32_t)l32 < left;
            const u64 nx = items[b * KG_STAGE + (uint32_t)KG_GRAN + (uint32_t)l32]; // (one wave: the reads of an instruction complete before the writes of the next)
            if (mv) items[b * KG_STAGE + (uint32_t)l32] = nx;
            if (act && l32 == 0) { cnt[b] = left; cur[b] = fits ? pos + take : pos; }
            const bool again = act && l32 == 0 && (left >= (uint32_t)KG_GRAN || (final && left > 0u));
            const uint64_t keep = __ballot(again); // bit 0: bucket i0 stays, bit 32: bucket i1 stays
            m = m1 ? (m1 & (m1 - 1)) : 0ull;
            if (keep & 1ull) m |= 1ull << i0;
            if (keep >> 32) m |= 1ull << i1;
        }
    }
    // OR of `bits` over the block; one barrier.  (s_barrier behind a wait for the wave's LDS operations only: outstanding
    // global loads and stores stay in flight across it.)  Three flag words in rotation: call k writes word k % 3, reads it
    // behind the barrier, and clears the word of call k - 1, which every wave has read before it arrived here.
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
constexpr size_t KG_STAGE_BYTES = (size_t)KG_FAN * KG_STAGE * 8 + (size_t)KG_FAN * 8 + 16;

// a wave's LDS operations execute in order; this only stops the compiler from moving them across the point
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// keeps the registers of a prefetch "used" here: the compiler's wait for those loads lands at this point and not at their
// first arithmetic use
__device__ __forceinline__ void touch4(const uint32_t (&v)[4]) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
__device__ __forceinline__ void touch4(const u64 (&v)[4]) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }

// ---- level 1: reads -> 256 buckets -------------------------------------------------------------------------------------
// Block-synchronous rounds: every wave brings the k-mers of up to four 64-base chunks of ITS read (one 250-base read = one
// round), the block stages them, the bucket owners write full granules.  An item is run << 54 | low 54 bits of h (the bucket is
// h's top 8 bits).  The bytes of a wave's next piece are fetched a round ahead and waited for between the two barriers of
// the round, when everything older is long done.
template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_group_extract(
    const DevParams P, const uint32_t k, const KmerGroupDev G, const KmerTable T, const uint32_t run, const uint32_t rot, const uint32_t epoch,
    const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin,
    const uint32_t r_end, const uint2 *__restrict__ results)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    uint32_t *w32 = reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE);
    const Staging<NW> S{lds, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_total = w32 + 2 * KG_FAN + 3; // occurrences of this block
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN; // this block's sub-region of every bucket
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    if (tid < 4) S.flag[tid] = 0u; // (flags and s_total)
    const uint32_t n_waves = gridDim.x * NW;
    const bool g2n = !P.qc_only && P.replace_q > 0;
    const u64 tag = (u64)run << 54;
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) { G.l1[((size_t)b * KG_FAN + sub) * G.cap1 + pos] = it; };
    auto slow = [&](const uint32_t b, const u64 it) {
        kmer_insert_atomic(T, ((u64)b << 54) | (it & KG_M54), epoch, 1u, G.first_hist, G.n_epochs);
    };
    __syncthreads();

    struct Hdr { uint32_t o; int a, n; }; // kept window [a, a + n) of the read at byte o; n == 0: nothing to count
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
            if (h.n < (int)k) h.n = 0;
        }
        return h;
    };
    // cursor: current read r_cur (header hc, next chunk c of [c, c_end)), the read after it (header hn, fetched a round ahead)
    uint32_t r_cur = r_begin + blockIdx.x * NW + wave, r_nxt = r_cur + n_waves;
    Hdr hc = load_hdr(r_cur), hn = load_hdr(r_nxt);
    int c = hc.a >> 6, c_end = hc.n ? (hc.a + hc.n + 63) >> 6 : c;
    uint32_t nb[4], nq[4] = {0u, 0u, 0u, 0u}; // the next piece's bytes (and qualities, --replace_to_N_q)
    const uint32_t safe_o = off[r_begin];     // (a byte of the arena that is there whatever the cursor says)
    auto load_piece = [&]() { // no branches: the index is clamped into the kept window (or onto safe_o), what lies outside is zeroed
        const bool any = hc.n > 0 && c < c_end;
        const int lo_p = any ? hc.a : 0, hi_p = any ? hc.a + hc.n - 1 : 0;
        const size_t ob = any ? (size_t)hc.o : (size_t)safe_o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = (c + j) * 64 + lane;
            const bool in = any && c + j < c_end && p >= lo_p && p <= hi_p;
            const int pc = p < lo_p ? lo_p : (p > hi_p ? hi_p : p);
            const uint32_t v = seq[ob + pc];
            nb[j] = in ? v : 0u;
            if (g2n) { const uint32_t qv = qual[ob + pc]; nq[j] = in ? qv : 0u; }
        }
    };
    load_piece();
    uint32_t bb[4], bq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { bb[j] = nb[j]; bq[j] = nq[j]; }
    bool first = true; // the piece in bb is the first of its read
    KmerPlanes pl{0, 0, 0};
    const KmerWin win = kmer_win(lane, k);
    uint32_t my_total = 0, phase = 0;
#pragma unroll 1
    for (;;) {
        // advance the cursor past the piece in bb and fetch the piece after it
        const bool first_now = first;
        c += 4; first = false;
        if (c >= c_end && r_cur < r_end) {
            hc = hn; r_cur = r_nxt; r_nxt += n_waves;
            hn = load_hdr(r_nxt);
            c = hc.a >> 6; c_end = hc.n ? (hc.a + hc.n + 63) >> 6 : c; first = true;
        }
        load_piece();
        if (first_now) pl = KmerPlanes{0, 0, 0};
        u64 h[4];
        uint32_t pend = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { // (a chunk past the piece's last one is all zero bytes: no key; the planes restart with the next read)
            uint32_t b = bb[j];
            if (g2n && b == 'G') { // G -> N precedes k-mer counting (trim.cpp:390-403)
                int qv = (int)(int8_t)bq[j] - P.in_off;
                qv = qv < 0 ? 0 : qv;
                if (qv < (int)P.replace_q) b = 'N';
            }
            uint64_t key;
            const bool ok = kmer_chunk_key32(b, win, k, pl, key);
            h[j] = kmer_mix62(key);
            pend |= ok ? 1u << j : 0u;
        }
        my_total += (uint32_t)__popc(pend);
        // block-wide rounds: tickets, then the owners of the buckets write the full granules
        bool more;
#pragma unroll 1
        for (bool fetched = false;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((pend >> j) & 1u)
                    if (S.put((uint32_t)(h[j] >> 54), tag | (h[j] & KG_M54))) pend &= ~(1u << j);
            __syncthreads();
            if (!fetched) { // the prefetched piece: its loads are a round's arithmetic old, the stores of the last drain older
                touch4(nb);
                if (g2n) touch4(nq);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bb[j] = nb[j]; bq[j] = nq[j]; }
                fetched = true;
            }
            S.drain(wave, lane, false, G.cap1, write, slow);
            const uint32_t f = S.block_or((pend ? 1u : 0u) | (r_cur < r_end ? 2u : 0u), phase, tid);
            more = (f & 2u) != 0u;
            if (!(f & 1u)) break;
        }
        if (!more) break;
    }
    S.drain(wave, lane, true, G.cap1, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur1[sub * KG_FAN + i] = S.cur[i];
    // occurrences of this launch's epoch (total_kmer of the sampling points, trim.cpp:170-176)
    const uint32_t wt = (uint32_t)wave_sum_i32((int)my_total); // (a wave sees < 2^31 occurrences per launch)
    if (lane == 0 && wt) atomicAdd(s_total, wt);                // (a block sees < 2^32)
    __syncthreads();
    if (tid == 0 && s_total[0]) {
        hist_add(G.tot_by_epoch, epoch, G.n_epochs, (long long)s_total[0]);
        atomicAdd(&T.stats[1], (u64)s_total[0]);
    }
}

// ---- level 1 for reads of up to 256 bases: a lane owns FOUR consecutive positions --------------------------------------------
// kmer_group_extract spends a wave instruction on 64 positions, of which a 250-base read fills 44 on average, and builds every
// lane's window out of wave-wide ballots (2.8 vector + 2.3 scalar instructions per occurrence).  Here a lane loads the dword of
// positions 4 l ... 4 l + 3 -- the whole read is ONE load instruction --, classifies its four bases (2-bit code: A 0, C 1, T 2, G 3,
// complement = code ^ 2; any injective encoding gives the same classes, trim.cpp:904-917) and gets its eight predecessors'
// codes through the wave's LDS exchange row: nine dwords, no ballot.  The four k-mers that end in the lane's positions are then
// shifts of one 72-bit window; the reverse complement is the window with its 2-bit groups reversed, made once per lane.
template <int NW, bool K31>
__global__ __launch_bounds__(NW * 64) void kmer_group_extract4(
    const DevParams P, const uint32_t k_arg, const KmerGroupDev G, const KmerTable T, const uint32_t run, const uint32_t rot, const uint32_t epoch,
    const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin,
    const uint32_t r_end, const uint2 *__restrict__ results)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    uint32_t *w32 = reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE);
    const Staging<NW> S{lds, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_total = w32 + 2 * KG_FAN + 3;
    const uint32_t k = K31 ? 31u : k_arg; // (the reference's command line cannot set another k, SURVEY Q19: its shifts are compile-time constants)
    uint32_t *s_xch = w32 + 2 * KG_FAN + 4;                          // [NW][72]: 8 zero dwords, then the wave's 64 packed lanes
    uint8_t *s_cls = reinterpret_cast<uint8_t *>(s_xch + NW * 72);   // [256] code | valid << 2 of every byte value
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN;
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    if (tid < 4) S.flag[tid] = 0u;
    for (int i = tid; i < NW * 72; i += NW * 64) s_xch[i] = 0u;
    for (int i = tid; i < 256; i += NW * 64) {
        const uint32_t l = (uint32_t)i | 0x20u;
        s_cls[i] = (uint8_t)(l == 'a' ? 4u : l == 'c' ? 5u : l == 't' ? 6u : l == 'g' ? 7u : 0u);
    }
    const uint32_t n_waves = gridDim.x * NW;
    const bool g2n = !P.qc_only && P.replace_q > 0;
    const u64 tag = (u64)run << 54;
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) { G.l1[((size_t)b * KG_FAN + sub) * G.cap1 + pos] = it; };
    auto slow = [&](const uint32_t b, const u64 it) {
        kmer_insert_atomic(T, ((u64)b << 54) | (it & KG_M54), epoch, 1u, G.first_hist, G.n_epochs);
    };
    __syncthreads();
    struct Hdr { uint32_t o; int a, n; };
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
            if (h.n < (int)k) h.n = 0;
        }
        return h;
    };
    struct __attribute__((packed, aligned(1))) U32u { uint32_t w; };
    const uint32_t safe_o = off[r_begin];
    uint32_t r_cur = r_begin + blockIdx.x * NW + wave, r_nxt = r_cur + n_waves;
    Hdr hc = load_hdr(r_cur), hn = load_hdr(r_nxt);
    uint32_t nw = 0, nqw = 0; // the next read's four bases (qualities) of this lane
    auto load_bytes = [&]() { // (positions past the kept window are not needed: nothing behind a short last read is touched)
        const bool need = hc.n > 0 && 4 * lane < hc.a + hc.n;
        const size_t at = need ? (size_t)hc.o + 4u * (uint32_t)lane : (size_t)safe_o;
        const uint32_t v = reinterpret_cast<const U32u *>(seq + at)->w;
        nw = need ? v : 0u;
        if (g2n) { const uint32_t qv = reinterpret_cast<const U32u *>(qual + at)->w; nqw = need ? qv : 0u; }
    };
    load_bytes();
    uint32_t bw = nw, bqw = nqw;
    Hdr hb = hc; // the read whose bytes are in bw
    uint32_t my_total = 0, phase = 0;
    uint32_t *xrow = s_xch + wave * 72;
    const uint32_t kmask = (uint32_t)((1ull << k) - 1ull);
    const u64 kmask2 = (1ull << (2 * k)) - 1ull;
    const int sh0 = 2 * (33 - (int)k); // bit shift of the window for the k-mer that ends in the lane's first position
#pragma unroll 1
    for (;;) {
        // advance the cursor and fetch ahead (one read per round)
        if (r_cur < r_end) { hc = hn; r_cur = r_nxt; r_nxt += n_waves; hn = load_hdr(r_nxt); }
        load_bytes();
        // ---- this lane's four bases -> 2-bit codes + "an ACGT inside the kept window" flags ----
        uint32_t t = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = (bw >> (8 * j)) & 0xffu;
            uint32_t cls = s_cls[b];
            if (g2n && b == 'G') { // G -> N precedes k-mer counting (trim.cpp:390-403)
                int qv = (int)(int8_t)((bqw >> (8 * j)) & 0xffu) - P.in_off;
                qv = qv < 0 ? 0 : qv;
                if (qv < (int)P.replace_q) cls = 0u;
            }
            const int p = 4 * lane + j;
            const bool inw = p >= hb.a && p < hb.a + hb.n;
            t |= (cls & 3u) << (2 * j);
            t |= (inw ? (cls >> 2) : 0u) << (8 + j);
        }
        xrow[8 + lane] = t;
        lds_wave_sync();
        uint32_t x[9]; // lanes l - 8 ... l
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = xrow[lane + i];
        lds_wave_sync();
        // window of the 36 positions 4 l - 32 ... 4 l + 3: codes in wlo (32 positions) : whi (4 positions), flags in vw
        u64 wlo = 0, vw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { wlo |= (u64)(x[i] & 0xffu) << (8 * i); vw |= (u64)((x[i] >> 8) & 0xfu) << (4 * i); }
        const u64 whi = x[8] & 0xffu;
        vw |= (u64)((x[8] >> 8) & 0xfu) << 32;
        // the window with its 2-bit groups in reverse order (36 groups: group m -> group 35 - m), complemented: rev : rlo
        auto rev2 = [](u64 v) -> u64 { // reverses the 32 two-bit groups of a 64-bit word
            v = ((u64)__brev((uint32_t)v) << 32) | (u64)__brev((uint32_t)(v >> 32));
            return ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
        };
        // R = rev2 of the 72-bit window = rev2(whi, 4 groups) at the bottom, rev2(wlo) above it
        const u64 r_hi4 = rev2(whi) >> 56;       // groups 35 ... 32 -> bits 0 ... 7
        const u64 r_lo = rev2(wlo);              // group m (< 32) -> group 31 - m
        const u64 Rlo = (r_lo << 8) | r_hi4;     // bits 0 ... 63 of R
        const u64 Rhi = r_lo >> 56;              // bits 64 ... 71
        u64 h[4];
        uint32_t pend = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the k-mer that ends at window group 32 + j: groups 33 + j - k ... 32 + j
            const int s = sh0 + 2 * j;                                   // first bit of the k-mer in the window (wave-uniform)
            const u64 fwd = (s < 64 ? ((wlo >> s) | (s ? whi << (64 - s) : 0ull)) : (whi >> (s - 64))) & kmask2;
            // in R the k-mer's groups are 35 - (32 + j) = 3 - j ... 3 - j + k - 1
            const int rs = 2 * (3 - j);
            const u64 rc = (((Rlo >> rs) | (rs ? Rhi << (64 - rs) : 0ull)) & kmask2) ^ (0xAAAAAAAAAAAAAAAAull & kmask2);
            const int vs = 33 + j - (int)k;
            const bool ok = ((uint32_t)(vw >> vs) & kmask) == kmask; // k valid bases ending here (word_len >= k, trim.cpp:924)
            h[j] = kmer_mix62(fwd < rc ? fwd : rc);
            pend |= ok ? 1u << j : 0u;
        }
        my_total += (uint32_t)__popc(pend);
        bool more;
#pragma unroll 1
        for (bool fetched = false;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((pend >> j) & 1u)
                    if (S.put((uint32_t)(h[j] >> 54), tag | (h[j] & KG_M54))) pend &= ~(1u << j);
            __syncthreads();
            if (!fetched) { // the prefetched read: its loads are a round's arithmetic old, the stores of the last drain older
                asm volatile("" ::"v"(nw), "v"(nqw));
                bw = nw; bqw = nqw; hb = hc;
                fetched = true;
            }
            S.drain(wave, lane, false, G.cap1, write, slow);
            const uint32_t f = S.block_or((pend ? 1u : 0u) | (r_cur < r_end ? 2u : 0u), phase, tid);
            more = (f & 2u) != 0u;
            if (!(f & 1u)) break;
        }
        if (!more) break;
    }
    S.drain(wave, lane, true, G.cap1, write, slow);
    __syncthreads();
    for (int i = tid; i < KG_FAN; i += NW * 64) G.cur1[sub * KG_FAN + i] = S.cur[i];
    const uint32_t wt = (uint32_t)wave_sum_i32((int)my_total);
    if (lane == 0 && wt) atomicAdd(s_total, wt);
    __syncthreads();
    if (tid == 0 && s_total[0]) {
        hist_add(G.tot_by_epoch, epoch, G.n_epochs, (long long)s_total[0]);
        atomicAdd(&T.stats[1], (u64)s_total[0]);
    }
}

// ---- level 1 from (key, epoch) pairs: the owner side of the multi-GPU exchange (faqcs_kmer_insert_device) ---------------------
// The pairs other ranks extracted for the keys this rank owns join the same group buffers; a pair's epoch travels in the item's
// run field (the group's run -> epoch table is the identity in this mode: at most KG_MAX_RUNS epochs).
template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_group_items(const KmerGroupDev G, const KmerTable T, const uint32_t rot,
                                                            const ulonglong2 *__restrict__ items, const u64 n_items)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    uint32_t *w32 = reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE);
    const Staging<NW> S{lds, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_tot = w32 + 2 * KG_FAN + 4; // [KG_EPOCH_SPAN] occurrences of this block by epoch
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t sub = (blockIdx.x + rot) % KG_FAN;
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = G.cur1[sub * KG_FAN + i]; }
    for (int i = tid; i < KG_EPOCH_SPAN; i += NW * 64) s_tot[i] = 0u;
    if (tid < 3) S.flag[tid] = 0u;
    __syncthreads();
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) { G.l1[((size_t)b * KG_FAN + sub) * G.cap1 + pos] = it; };
    auto slow = [&](const uint32_t b, const u64 it) {
        kmer_insert_atomic(T, ((u64)b << 54) | (it & KG_M54), (uint32_t)(it >> 54), 1u, G.first_hist, G.n_epochs);
    };
    constexpr uint32_t PER = NW * 64;
    const u64 per_block = ((n_items + gridDim.x - 1) / gridDim.x + PER - 1) / PER * PER;
    const u64 lo = (u64)blockIdx.x * per_block < n_items ? (u64)blockIdx.x * per_block : n_items;
    const u64 hi = lo + per_block < n_items ? lo + per_block : n_items;
    uint32_t phase = 0;
    ulonglong2 nx = lo + tid < hi ? items[lo + tid] : make_ulonglong2(KG_PAD, 0ull);
#pragma unroll 1
    for (u64 t0 = lo; t0 < hi; t0 += PER) {
        const ulonglong2 cur = nx;
        nx = t0 + PER + tid < hi ? items[t0 + PER + tid] : make_ulonglong2(KG_PAD, 0ull);
        uint32_t pend = 0;
        u64 h = 0, it = 0;
        if (cur.x != KG_PAD && (uint32_t)cur.y < (uint32_t)KG_EPOCH_SPAN) {
            h = kmer_mix62(cur.x);
            it = ((u64)(uint32_t)cur.y << 54) | (h & KG_M54);
            pend = 1u;
            atomicAdd(&s_tot[(uint32_t)cur.y], 1u);
        }
#pragma unroll 1
        for (;;) {
            if (pend && S.put((uint32_t)(h >> 54), it)) pend = 0u;
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
// `part` of the partitions b1 * 256 + (next 8 bits of h): again a block writes only where no other block does.
template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_group_split(const KmerGroupDev G, const KmerTable T)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    uint32_t *w32 = reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE);
    const Staging<NW> S{lds, w32, w32 + KG_FAN, w32 + 2 * KG_FAN};
    uint32_t *s_er = w32 + 2 * KG_FAN + 4;     // [KG_MAX_RUNS] epoch of run j, relative
    uint32_t *s_n = s_er + KG_MAX_RUNS;        // [256] items of the bucket's sub-regions
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t b1 = blockIdx.x / G.split, part = blockIdx.x % G.split;
    for (int i = tid; i < KG_FAN; i += NW * 64) { S.cnt[i] = 0u; S.cur[i] = 0u; s_n[i] = G.cur1[i * KG_FAN + b1]; }
    if (tid < 3) S.flag[tid] = 0u;
    for (uint32_t j = tid; j < G.n_runs; j += NW * 64) s_er[j] = G.run_epoch[j];
    __syncthreads();
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) { G.l2[(((size_t)b1 * KG_FAN + b) * G.split + part) * G.cap2 + pos] = it; };
    auto slow = [&](const uint32_t b, const u64 item) {
        kmer_insert_atomic(T, (((u64)b1 * KG_FAN + b) << 46) | (item & KG_M46), G.epoch_base + (uint32_t)(item >> 46), 1u, G.first_hist, G.n_epochs);
    };
    constexpr uint32_t TILE = NW * 64 * 4;
    const uint32_t per = KG_FAN / G.split;
    uint32_t phase = 0;
#pragma unroll 1
    for (uint32_t sr = part * per; sr < (part + 1) * per; ++sr) {
        const uint32_t n_s = s_n[sr];
        if (n_s == 0) continue; // (block-uniform)
        const u64 *src = G.l1 + ((size_t)b1 * KG_FAN + sr) * G.cap1;
        u64 nx[4], cx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = j * NW * 64 + tid; nx[j] = i < n_s ? src[i] : KG_PAD; }
        touch4(nx);
#pragma unroll
        for (int j = 0; j < 4; ++j) cx[j] = nx[j];
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < n_s; t0 += TILE) {
            u64 cur[4];
            uint32_t pend = 0, sb = 0; // pending flags / sub-buckets of this thread's four items
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t i = t0 + TILE + j * NW * 64 + tid; nx[j] = i < n_s ? src[i] : KG_PAD; } // next tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cx[j] != KG_PAD) {
                    cur[j] = ((u64)s_er[cx[j] >> 54] << 46) | (cx[j] & KG_M46);
                    sb |= (uint32_t)((cx[j] >> 46) & 255u) << (8 * j);
                    pend |= 1u << j;
                }
            }
#pragma unroll 1
            for (bool fetched = false;;) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((pend >> j) & 1u)
                        if (S.put((sb >> (8 * j)) & 255u, cur[j])) pend &= ~(1u << j);
                __syncthreads();
                if (!fetched) {
                    touch4(nx);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cx[j] = nx[j];
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

// ---- combine + insert: one workgroup per partition --------------------------------------------------------------------------
// LDS table: word = remainder (46 bits) << 18 | epoch (relative, < 2^18 - 1): equal keys differ only in the epoch bits, so an
// atomic min on the word keeps the key and its smallest epoch; the count sits in a second array.
template <int NT, int LS>
__global__ __launch_bounds__(NT) void kmer_group_combine(const KmerGroupDev G, const KmerTable T)
{
    constexpr int KG_LDS_SLOTS = LS; // (shadows the enum: 4 096 slots, two workgroups per CU; 8 192 for groups past 2^30 items, one per CU)
    __shared__ u64 s_key[KG_LDS_SLOTS];
    __shared__ uint32_t s_cnt[KG_LDS_SLOTS];
    __shared__ int s_hist[KG_EPOCH_SPAN];
    __shared__ uint32_t s_claim[KG_SLICE_MAX / 32]; // slots of the slice this launch has claimed
    __shared__ uint32_t s_fail;
    const uint32_t p = blockIdx.x;
    // the partition's items: `split` sub-regions of cap2 items each, sub-region j holding cur2[p][j] of them
    uint32_t n_sub[8], n_p = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { n_sub[j] = (uint32_t)j < G.split ? G.cur2[(size_t)p * G.split + j] : 0u; n_p += n_sub[j]; }
    if (n_p == 0) return; // (block-uniform)
    const int tid = threadIdx.x;
    const Slice sc = slice_of(T, (u64)p << 46);
    const u64 *src = G.l2 + (size_t)p * G.split * G.cap2;
    // flat index over the sub-regions -> address (a gap of cap2 - n_sub[j] items behind sub-region j)
    auto item_at = [&](uint32_t i) -> u64 {
        uint32_t base = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i < n_sub[j]) return src[base + i];
            i -= n_sub[j]; base += G.cap2;
        }
        return KG_PAD;
    };
    u64 nx[4]; // the partition's items, four per thread in flight; the first four while the LDS tables are cleared
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t i = (uint32_t)(j * NT + tid); nx[j] = i < n_p ? item_at(i) : KG_PAD; }
    for (int i = tid; i < KG_LDS_SLOTS; i += NT) { s_key[i] = ~0ull; s_cnt[i] = 0u; }
    for (int i = tid; i < KG_EPOCH_SPAN; i += NT) s_hist[i] = 0;
    for (uint32_t i = tid; i < (uint32_t)(sc.mask + 32) / 32; i += NT) s_claim[i] = 0u;
    if (tid == 0) s_fail = 0u;
    __syncthreads();
    constexpr uint32_t LMASK = KG_LDS_SLOTS - 1;
    constexpr int LSHIFT = 46 - (LS == 4096 ? 12 : 13); // slot = top 12 (13) bits of the remainder: LDS order == table order
    static_assert(LS == 4096 || LS == 8192, "LSHIFT");
    // looks the item's key up; returns the slot, or -1 (table full and the key not in it)
    auto find_or_add = [&](const u64 rem, const u64 word, const bool add) -> int {
        uint32_t s = (uint32_t)(rem >> LSHIFT);
#pragma unroll 1
        for (uint32_t probe = 0; probe < (uint32_t)KG_LDS_SLOTS; ++probe) {
            u64 w = s_key[s];
            if (w == ~0ull) {
                if (!add) return -1;
                w = atomicCAS(&s_key[s], ~0ull, word);
                if (w == ~0ull) return (int)s;
            }
            if ((w >> 18) == rem) {
                if (add && word < w) atomicMin(&s_key[s], word);
                return (int)s;
            }
            s = (s + 1) & LMASK;
        }
        return -1;
    };
    bool failed = false;
    {
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < n_p; i0 += 4 * NT) {
            u64 it[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) it[j] = nx[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t i = i0 + (uint32_t)((4 + j) * NT + tid); nx[j] = i < n_p ? item_at(i) : KG_PAD; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (it[j] == KG_PAD) continue;
                const u64 rem = it[j] & KG_M46;
                const int sl = find_or_add(rem, (rem << 18) | (it[j] >> 46), true);
                if (sl >= 0) atomicAdd(&s_cnt[sl], 1u);
                else failed = true;
            }
        }
    }
    if (failed) s_fail = 1u;
    __syncthreads();
    // One table update per distinct key, none of them a device-scope atomic: the slice belongs to this workgroup for the length of
    // the launch, an empty slot is claimed through the workgroup's LDS bitmap, and no other thread of the launch holds this key.
    // A thread's KG_LDS_SLOTS / NT keys are looked up together (their first probes are in flight at the same time).
    constexpr int KPT = KG_LDS_SLOTS / NT;
    static_assert(KG_LDS_SLOTS % NT == 0, "keys per thread");
    u64 kw[KPT];
    ulonglong2 first[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        kw[j] = s_key[j * NT + tid];
        const u64 h = ((u64)p << 46) | (kw[j] >> 18);
        const u64 g = (h >> T.shift) & sc.mask;
        first[j] = make_ulonglong2(0ull, 0ull);
        if (kw[j] != ~0ull) first[j] = *reinterpret_cast<const ulonglong2 *>(&T.slots[sc.base + g]);
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const u64 w = kw[j];
        if (w == ~0ull) continue;
        const u64 h = ((u64)p << 46) | (w >> 18);
        const uint32_t e_rel = (uint32_t)(w & 0x3ffffu), e = G.epoch_base + e_rel, cnt = s_cnt[j * NT + tid];
        u64 g = (h >> T.shift) & sc.mask;
        bool placed = false;
        ulonglong2 cur = first[j];
#pragma unroll 1
        for (u64 probe = 0; probe <= sc.mask; ++probe) {
            KmerSlot *sl = &T.slots[sc.base + g];
            if (probe) cur = *reinterpret_cast<const ulonglong2 *>(sl);
            if (cur.x == ~0ull) {
                const uint32_t bit = 1u << (g & 31);
                if (!(atomicOr(&s_claim[g >> 5], bit) & bit)) { // a new key: key, count - 1 and first epoch in one plain store
                    *reinterpret_cast<ulonglong2 *>(sl) = make_ulonglong2(h, (u64)(cnt - 1u) | ((u64)e << 32));
                    atomicAdd(&s_hist[e_rel], 1);
                    placed = true;
                    break;
                }
            } else if (cur.x == h) {
                uint32_t c1 = (uint32_t)cur.y + cnt, fe = (uint32_t)(cur.y >> 32);
                if (e < fe) { // (only a key the fallback path inserted during this group can hold a later epoch)
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
            g = (g + 1) & sc.mask;
        }
        if (!placed) atomicOr(&T.stats[2], 1ull);
    }
    __syncthreads();
    if (s_fail) { // the LDS table filled up: every item whose key is not in it goes the per-occurrence way (a key is in the
                  // table with ALL its items or with none: the table only fills, so a key that fails once fails always)
        __threadfence(); // the plain stores above are in L2 before the atomics below
        __syncthreads();
#pragma unroll 1
        for (uint32_t i = tid; i < n_p; i += NT) {
            const u64 item = item_at(i);
            if (item == KG_PAD) continue;
            const u64 rem = item & KG_M46;
            if (find_or_add(rem, 0ull, false) < 0)
                kmer_insert_atomic(T, ((u64)p << 46) | rem, G.epoch_base + (uint32_t)(item >> 46), 1u, G.first_hist, G.n_epochs);
        }
    }
    for (int i = tid; i < KG_EPOCH_SPAN; i += NT)
        if (s_hist[i]) hist_add(G.first_hist, G.epoch_base + (uint32_t)i, G.n_epochs, (long long)s_hist[i]);
}

__global__ void kmer_group_reset(const KmerGroupDev G)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)KG_FAN * KG_FAN) G.cur1[i] = 0u;
}

constexpr int KG_NW = 16;

} // namespace

// blocks of an extraction launch over n_reads reads
uint32_t faqcs_kmer_group_grid(uint32_t n_reads, int n_cu)
{
    uint32_t grid = (n_reads + 4 * KG_NW - 1) / (4 * KG_NW);
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu; // one block per CU: its staging area is 130 KB of the CU's LDS
    if (grid > (uint32_t)KG_FAN) grid = KG_FAN;       // ... and one sub-region of every bucket per block
    return grid ? grid : 1u;
}

hipError_t faqcs_launch_kmer_group_extract(const DevParams &P, uint32_t k, const KmerGroupDev &G, const KmerTable &T, uint32_t run, uint32_t rot,
                                           uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                           uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, uint32_t max_len, int n_cu, hipStream_t st)
{
    if (r_end <= r_begin) return hipSuccess;
    const dim3 grid(faqcs_kmer_group_grid(r_end - r_begin, n_cu)), block(KG_NW * 64);
    static const bool chunked = getenv("FAQCS_KMER_EXTRACT_CHUNKED") && atoi(getenv("FAQCS_KMER_EXTRACT_CHUNKED")) != 0; // (A/B switch)
    constexpr size_t LDS4 = KG_STAGE_BYTES + 16 + (size_t)KG_NW * 72 * 4 + 256;
    if (max_len <= 256 && !chunked && k == 31) // a lane owns four positions: the whole read in one round
        hipLaunchKernelGGL((kmer_group_extract4<KG_NW, true>), grid, block, LDS4, st,
                           P, k, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results));
    else if (max_len <= 256 && !chunked)
        hipLaunchKernelGGL((kmer_group_extract4<KG_NW, false>), grid, block, LDS4, st,
                           P, k, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results));
    else
        hipLaunchKernelGGL((kmer_group_extract<KG_NW>), grid, block, KG_STAGE_BYTES + 16, st,
                           P, k, G, T, run, rot, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results));
    return hipGetLastError();
}

// blocks of an owner-side launch over n_items (key, epoch) pairs
uint32_t faqcs_kmer_group_items_grid(unsigned long long n_items, int n_cu)
{
    unsigned long long grid = (n_items + 16 * KG_NW * 64 - 1) / (16ull * KG_NW * 64);
    if (grid > (unsigned long long)n_cu) grid = (unsigned long long)n_cu;
    if (grid > (unsigned long long)KG_FAN) grid = KG_FAN;
    return grid ? (uint32_t)grid : 1u;
}

hipError_t faqcs_launch_kmer_group_items(const KmerGroupDev &G, const KmerTable &T, uint32_t rot, const void *items, unsigned long long n_items,
                                         int n_cu, hipStream_t st)
{
    if (n_items)
        hipLaunchKernelGGL((kmer_group_items<KG_NW>), dim3(faqcs_kmer_group_items_grid(n_items, n_cu)), dim3(KG_NW * 64),
                           KG_STAGE_BYTES + (size_t)KG_EPOCH_SPAN * 4 + 16, st, G, T, rot, reinterpret_cast<const ulonglong2 *>(items), n_items);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_group_flush(const KmerGroupDev &G, const KmerTable &T, hipStream_t st)
{
    hipLaunchKernelGGL((kmer_group_split<KG_NW>), dim3(KG_FAN * G.split), dim3(KG_NW * 64), KG_STAGE_BYTES + (size_t)(KG_MAX_RUNS + KG_FAN) * 4 + 16, st, G, T);
    if (G.lds_slots > 4096) hipLaunchKernelGGL((kmer_group_combine<1024, 8192>), dim3(KG_FAN * KG_FAN), dim3(1024), 0, st, G, T);
    else hipLaunchKernelGGL((kmer_group_combine<1024, 4096>), dim3(KG_FAN * KG_FAN), dim3(1024), 0, st, G, T);
    hipLaunchKernelGGL(kmer_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_group_reset(const KmerGroupDev &G, hipStream_t st)
{
    hipLaunchKernelGGL(kmer_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}
