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

namespace {

typedef unsigned long long u64;

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
// its item for the next round.  drain(): wave w owns buckets [16 w, 16 w + 16) -- lane l < 16 of the wave is the OWNER of bucket
// 16 w + l and holds, in `res`, the position of a granule (32 items, 256 bytes) it has reserved in the bucket's region ahead
// of time.  A bucket with >= 32 staged items is written to its reserved granule by the whole wave; the owners of the buckets
// that were written then reserve their next granule with ONE atomic instruction whose result is not needed before the
// bucket fills again, rounds later -- the latency of a returning device-scope atomic (microseconds under load) stays off
// the critical path (the first version took a granule's position when it needed it: one round trip per flush, serialised
// per wave, and 27 us per round of 2 800 items).  The final drain pads each bucket's last granule with KG_PAD items, so every
// reserved granule is written.
#define KG_PAD (~0ull)
template <int NW> struct Staging {
    static constexpr int BPW = KG_FAN / NW;
    u64 *items;     // [KG_FAN][KG_STAGE]
    uint32_t *cnt;  // [KG_FAN]
    __device__ __forceinline__ bool put(const uint32_t b, const u64 item) const
    {
        const uint32_t pos = atomicAdd(&cnt[b], 1u);
        if (pos < (uint32_t)KG_STAGE) { items[b * KG_STAGE + pos] = item; return true; }
        return false;
    }
    // write(b, pos, item): the wave stores a granule (lane i < 32 holds item i) at position pos of bucket b's region, or takes
    // the slow path when the reservation was refused; alloc(b): reserves a granule, returns its position
    template <class Write, class Alloc>
    __device__ __forceinline__ void drain(const int wave, const int lane, const bool final, uint32_t &res, Write &&write, Alloc &&alloc) const
    {
        const uint32_t n_l = lane < BPW ? cnt[wave * BPW + lane] : 0u;
        const uint64_t todo = __ballot(lane < BPW && (final || n_l >= (uint32_t)KG_GRAN));
        uint64_t m = todo;
#pragma unroll 1
        while (m) {
            const int i = uni(__ffsll((long long)m) - 1);
            m &= m - 1;
            const int b = wave * BPW + i;
            uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_l, i);
            n = n < (uint32_t)KG_STAGE ? n : (uint32_t)KG_STAGE;
            uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)res, i);
            uint32_t done = 0;
#pragma unroll 1
            for (;;) {
                const uint32_t take = n - done < (uint32_t)KG_GRAN ? n - done : (uint32_t)KG_GRAN;
                const u64 it = (uint32_t)lane < take ? items[b * KG_STAGE + done + lane] : KG_PAD;
                write((uint32_t)b, pos, it);
                done += take;
                if (!(n - done >= (uint32_t)KG_GRAN || (final && n > done))) break;
                uint32_t p2 = 0; // a second granule of the same bucket in one drain (64 staged items): allocated on the spot
                if (lane == 0) p2 = alloc((uint32_t)b);
                pos = uniu(p2);
            }
            const uint32_t left = n - done;
            if (left) { // (one wave: the reads of an instruction complete before the writes of the next)
                const u64 it = (uint32_t)lane < left ? items[b * KG_STAGE + done + lane] : 0ull;
                if ((uint32_t)lane < left) items[b * KG_STAGE + lane] = it;
            }
            if (lane == 0) cnt[b] = left;
        }
        if (!final && lane < BPW && ((todo >> lane) & 1ull)) res = alloc((uint32_t)(wave * BPW + lane));
    }
};

// the granule in lanes 0..31 to position pos of a bucket region; a refused reservation (pos >= cap) is remembered in lim[b] and its
// items take the slow path
template <class Slow>
__device__ __forceinline__ void granule_out(u64 *region, uint32_t *lim, const uint32_t cap, const uint32_t pos, const u64 it,
                                            const int lane, Slow &&slow)
{
    if (pos < cap) {
        if (lane < KG_GRAN) region[pos + lane] = it; // (the region is KG_GRAN items longer than cap)
    } else {
        if (lane == 0) atomicMin(lim, pos);
        if (lane < KG_GRAN && it != KG_PAD) slow(it);
    }
}

// ---- level 1: reads -> 256 buckets -------------------------------------------------------------------------------------
// Block-synchronous rounds: every wave brings the k-mers of up to four 64-base chunks of ITS read (one 250-base read = one
// round), the block stages them, the bucket owners write full granules.  The bytes of a wave's next piece and the header
// (offset, kept window) of its next read are fetched a round ahead.
template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_group_extract(
    const DevParams P, const uint32_t k, const KmerGroupDev G, const KmerTable T, const uint32_t epoch,
    const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, const uint32_t *__restrict__ off, const uint32_t r_begin,
    const uint32_t r_end, const uint2 *__restrict__ results)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const Staging<NW> S{lds, reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE)};
    uint32_t *s_total = S.cnt + KG_FAN; // occurrences of this block
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    for (int i = tid; i < KG_FAN + 2; i += NW * 64) S.cnt[i] = 0u;
    const uint32_t n_waves = gridDim.x * NW;
    const bool g2n = !P.qc_only && P.replace_q > 0;
    auto alloc = [&](const uint32_t b) { return atomicAdd(&G.cur1[b], (uint32_t)KG_GRAN); };
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) {
        granule_out(G.l1 + (size_t)b * G.stride1, &G.lim1[b], G.cap1, pos, it, lane,
                    [&](const u64 h) { kmer_insert_atomic(T, h, epoch, 1u, G.first_hist, G.n_epochs); });
    };
    uint32_t res = 0;
    if (lane < S.BPW) res = alloc((uint32_t)(wave * S.BPW + lane));
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
    bool first = true;
    uint32_t nb[4], nq[4]; // the next piece's bytes (and qualities, --replace_to_N_q)
    auto load_piece = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = (c + j) * 64 + lane;
            const bool in = c + j < c_end && p >= hc.a && p < hc.a + hc.n;
            nb[j] = in ? seq[(size_t)hc.o + p] : 0u;
            nq[j] = g2n && in ? qual[(size_t)hc.o + p] : 0u;
        }
    };
    load_piece();
    KmerPlanes pl{0, 0, 0};
    uint32_t my_total = 0;
#pragma unroll 1
    for (;;) {
        // the piece whose bytes arrived: chunks [pc, pc + pn) of the current read
        uint32_t bb[4], bq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bb[j] = nb[j]; bq[j] = nq[j]; }
        const int pn = c_end - c < 4 ? c_end - c : 4;
        if (first) pl = KmerPlanes{0, 0, 0};
        // advance the cursor and fetch ahead
        c += 4; first = false;
        if (c >= c_end && r_cur < r_end) {
            hc = hn; r_cur = r_nxt; r_nxt += n_waves;
            hn = load_hdr(r_nxt);
            c = hc.a >> 6; c_end = hc.n ? (hc.a + hc.n + 63) >> 6 : c; first = true;
        }
        load_piece();
        u64 h[4];
        uint32_t pend = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < pn) { // (wave-uniform)
                uint32_t b = bb[j];
                if (g2n && b == 'G') { // G -> N precedes k-mer counting (trim.cpp:390-403)
                    int qv = (int)(int8_t)bq[j] - P.in_off;
                    qv = qv < 0 ? 0 : qv;
                    if (qv < (int)P.replace_q) b = 'N';
                }
                uint64_t key;
                if (kmer_chunk_key(b, lane, k, pl, key)) { h[j] = kmer_mix62(key); pend |= 1u << j; }
            }
        }
        my_total += (uint32_t)__popc(pend);
        // block-wide rounds: tickets, then the owners of the buckets write the full granules
#pragma unroll 1
        for (;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((pend >> j) & 1u)
                    if (S.put((uint32_t)(h[j] >> 54), h[j])) pend &= ~(1u << j);
            __syncthreads();
            S.drain(wave, lane, false, res, write, alloc);
            if (!__syncthreads_or((int)pend)) break;
        }
        if (!__syncthreads_or(r_cur < r_end)) break;
    }
    S.drain(wave, lane, true, res, write, alloc);
    // occurrences of this launch's epoch (total_kmer of the sampling points, trim.cpp:170-176)
    const uint32_t wt = (uint32_t)wave_sum_i32((int)my_total); // (a wave sees < 2^31 occurrences per launch)
    if (lane == 0 && wt) atomicAdd(s_total, wt);                // (a block sees < 2^32)
    __syncthreads();
    if (tid == 0 && s_total[0]) {
        hist_add(G.tot_by_epoch, epoch, G.n_epochs, (long long)s_total[0]);
        atomicAdd(&T.stats[1], (u64)s_total[0]);
    }
}

// bounds[run][b] = cur1[b] after the run's launch
__global__ void kmer_group_bounds(const KmerGroupDev G, const uint32_t run)
{
    const uint32_t b = threadIdx.x;
    if (b < (uint32_t)KG_FAN) G.bounds[run * KG_FAN + b] = G.cur1[b];
}

// ---- level 2: every bucket 256 ways; the epoch moves from the position into the item ---------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void kmer_group_split(const KmerGroupDev G, const KmerTable T, const uint32_t split)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const Staging<NW> S{lds, reinterpret_cast<uint32_t *>(lds + KG_FAN * KG_STAGE)};
    uint32_t *s_eb = S.cnt + KG_FAN;          // [KG_MAX_RUNS] end of run j inside this bucket
    uint32_t *s_er = s_eb + KG_MAX_RUNS;      // [KG_MAX_RUNS] its epoch, relative
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t b1 = blockIdx.x / split, part = blockIdx.x % split;
    uint32_t n_b = G.cur1[b1];
    { const uint32_t l = G.lim1[b1]; n_b = n_b < l ? n_b : l; }
    for (int i = tid; i < KG_FAN; i += NW * 64) S.cnt[i] = 0u;
    for (uint32_t j = tid; j < G.n_runs; j += NW * 64) {
        const uint32_t e = G.bounds[j * KG_FAN + b1];
        s_eb[j] = e < n_b ? e : n_b;
        s_er[j] = G.run_epoch[j];
    }
    constexpr uint32_t TILE = NW * 64 * 4;
    const uint32_t per = ((n_b + split - 1) / split + TILE - 1) / TILE * TILE;
    const uint32_t lo = part * per < n_b ? part * per : n_b, hi = lo + per < n_b ? lo + per : n_b;
    if (lo >= hi) return; // (block-uniform: nothing reserved, nothing to pad)
    const u64 *src = G.l1 + (size_t)b1 * G.stride1;
    auto alloc = [&](const uint32_t b) { return atomicAdd(&G.cur2[b1 * KG_FAN + b], (uint32_t)KG_GRAN); };
    auto write = [&](const uint32_t b, const uint32_t pos, const u64 it) {
        const uint32_t p = b1 * KG_FAN + b;
        granule_out(G.l2 + (size_t)p * G.stride2, &G.lim2[p], G.cap2, pos, it, lane, [&](const u64 item) {
            kmer_insert_atomic(T, ((u64)p << 46) | (item & KG_M46), G.epoch_base + (uint32_t)(item >> 46), 1u, G.first_hist, G.n_epochs);
        });
    };
    uint32_t res = 0;
    if (lane < S.BPW) res = alloc((uint32_t)(wave * S.BPW + lane));
    __syncthreads();
    uint32_t jt = 0; // first run that ends behind the tile's start (block-uniform)
    u64 nx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t i = lo + j * NW * 64 + tid; nx[j] = i < hi ? src[i] : KG_PAD; }
#pragma unroll 1
    for (uint32_t t0 = lo; t0 < hi; t0 += TILE) {
        u64 cur[4];
        uint32_t pend = 0, sub = 0; // pending flags / sub-buckets of this thread's four items
        while (jt + 1 < G.n_runs && s_eb[jt] <= t0) ++jt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = t0 + j * NW * 64 + tid;
            if (nx[j] != KG_PAD) { // (padding of a bucket's last granules)
                uint32_t jr = jt;
                while (jr + 1 < G.n_runs && s_eb[jr] <= i) ++jr;
                cur[j] = ((u64)s_er[jr] << 46) | (nx[j] & KG_M46);
                sub |= (uint32_t)((nx[j] >> 46) & 255u) << (8 * j);
                pend |= 1u << j;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = t0 + TILE + j * NW * 64 + tid; nx[j] = i < hi ? src[i] : KG_PAD; } // next tile
#pragma unroll 1
        for (;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((pend >> j) & 1u)
                    if (S.put((sub >> (8 * j)) & 255u, cur[j])) pend &= ~(1u << j);
            __syncthreads();
            S.drain(wave, lane, false, res, write, alloc);
            if (!__syncthreads_or((int)pend)) break;
        }
    }
    __syncthreads();
    S.drain(wave, lane, true, res, write, alloc);
}

// ---- combine + insert: one workgroup per partition --------------------------------------------------------------------------
// LDS table: word = remainder (46 bits) << 18 | epoch (relative, < 2^18 - 1): equal keys differ only in the epoch bits, so an
// atomic min on the word keeps the key and its smallest epoch; the count sits in a second array.
template <int NT>
__global__ __launch_bounds__(NT) void kmer_group_combine(const KmerGroupDev G, const KmerTable T)
{
    __shared__ u64 s_key[KG_LDS_SLOTS];
    __shared__ uint32_t s_cnt[KG_LDS_SLOTS];
    __shared__ int s_hist[KG_EPOCH_SPAN];
    __shared__ uint32_t s_claim[KG_SLICE_MAX / 32]; // slots of the slice this launch has claimed
    __shared__ uint32_t s_fail;
    const uint32_t p = blockIdx.x;
    uint32_t n_p = G.cur2[p];
    { const uint32_t l = G.lim2[p]; n_p = n_p < l ? n_p : l; }
    if (n_p == 0) return; // (block-uniform)
    const int tid = threadIdx.x;
    const Slice sc = slice_of(T, (u64)p << 46);
    for (int i = tid; i < KG_LDS_SLOTS; i += NT) { s_key[i] = ~0ull; s_cnt[i] = 0u; }
    for (int i = tid; i < KG_EPOCH_SPAN; i += NT) s_hist[i] = 0;
    for (uint32_t i = tid; i < (uint32_t)(sc.mask + 32) / 32; i += NT) s_claim[i] = 0u;
    if (tid == 0) s_fail = 0u;
    __syncthreads();
    const u64 *src = G.l2 + (size_t)p * G.stride2;
    constexpr uint32_t LMASK = KG_LDS_SLOTS - 1;
    constexpr int LSHIFT = 46 - 12; // slot = top 12 bits of the remainder: LDS order == table order
    static_assert(KG_LDS_SLOTS == 4096, "LSHIFT");
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
    { // the partition's items, four per thread in flight
        u64 nx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = (uint32_t)(j * NT + tid); nx[j] = i < n_p ? src[i] : KG_PAD; }
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < n_p; i0 += 4 * NT) {
            u64 it[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) it[j] = nx[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t i = i0 + (uint32_t)((4 + j) * NT + tid); nx[j] = i < n_p ? src[i] : KG_PAD; }
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
            const u64 item = src[i];
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
    if (i < (uint32_t)KG_FAN) { G.cur1[i] = 0u; G.lim1[i] = 0xffffffffu; }
    if (i < (uint32_t)KG_FAN * KG_FAN) { G.cur2[i] = 0u; G.lim2[i] = 0xffffffffu; }
}

constexpr int KG_NW = 16;
constexpr size_t KG_STAGE_BYTES = (size_t)KG_FAN * KG_STAGE * 8 + (size_t)KG_FAN * 4;

} // namespace

// blocks of an extraction launch over n_reads reads (each block pads up to 256 granules: the host's item bound counts them)
uint32_t faqcs_kmer_group_grid(uint32_t n_reads, int n_cu)
{
    uint32_t grid = (n_reads + 4 * KG_NW - 1) / (4 * KG_NW);
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu; // one block per CU: its staging area is 129 KB of the CU's LDS
    return grid ? grid : 1u;
}

hipError_t faqcs_launch_kmer_group_extract(const DevParams &P, uint32_t k, const KmerGroupDev &G, const KmerTable &T, uint32_t run,
                                           uint32_t epoch, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                           uint32_t r_begin, uint32_t r_end, const faqcs_read_result *results, int n_cu, hipStream_t st)
{
    if (r_end > r_begin)
        hipLaunchKernelGGL((kmer_group_extract<KG_NW>), dim3(faqcs_kmer_group_grid(r_end - r_begin, n_cu)), dim3(KG_NW * 64), KG_STAGE_BYTES + 16, st,
                           P, k, G, T, epoch, seq, qual, off, r_begin, r_end, reinterpret_cast<const uint2 *>(results));
    hipLaunchKernelGGL(kmer_group_bounds, dim3(1), dim3(KG_FAN), 0, st, G, run);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_group_flush(const KmerGroupDev &G, const KmerTable &T, uint32_t split, hipStream_t st)
{
    hipLaunchKernelGGL((kmer_group_split<KG_NW>), dim3(KG_FAN * split), dim3(KG_NW * 64), KG_STAGE_BYTES + (size_t)KG_MAX_RUNS * 8, st, G, T, split);
    hipLaunchKernelGGL((kmer_group_combine<1024>), dim3(KG_FAN * KG_FAN), dim3(1024), 0, st, G, T);
    hipLaunchKernelGGL(kmer_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}

hipError_t faqcs_launch_kmer_group_reset(const KmerGroupDev &G, hipStream_t st)
{
    hipLaunchKernelGGL(kmer_group_reset, dim3(KG_FAN * KG_FAN / 256), dim3(256), 0, st, G);
    return hipGetLastError();
}
