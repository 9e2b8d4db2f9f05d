// faqcs_adapter_kernel.hip -- adapter_overlap: the adapter / primer / PhiX pre-pass (gfx950, wave64).
//
// Replaces trim_adapters_and_phiX(vector<Read>&,...) (trim.cpp:961-1142), SO::SeqOverlap's ungapped
// Smith-Waterman (seq_overlap.cpp:46-370, seq_overlap.h:370-411,519-549) and find_mask_range
// (trim.cpp:1144-1189).  The reference packs 8 reads into int16 SSE lanes and sweeps the DP matrix row by
// row; here ONE wavefront owns ONE read and its 64 lanes own 64 DIAGONALS of the (read x adapter) matrix.
// Ungapped local alignment decouples along diagonals: M(i,j) = max(M(i-1,j-1),0) + s(i,j) is a Kadane
// recurrence per diagonal.
//
// Two stages per (read, adapter):
//  1. bit-parallel PREFILTER (exact skip test).  match(i,j) = OR_b Qb[i] & Tb[j] over the four base bit-planes
//     b in {A,C,G,T} of the 4-bit IUPAC masks (seq_overlap.h:133-150).  The read's planes sit in LDS, the
//     adapter's 32-bit plane words come from scalar loads; a lane builds 32 match bits of its diagonal with
//     8 LDS reads + 4 v_alignbit + 7 logic ops and popcounts them.  A local alignment of score B has
//     num_match = (len + B)/2 <= (min(|read|,|adapter|) + matches_on_diagonal)/2, so when that bound is below
//     the reference's threshold for every diagonal the adapter can neither mask nor be credited
//     (trim.cpp:1024-1041) and stage 2 is skipped.  Random 150-mers almost never pass.
//  2. exact per-cell Kadane with start tracking on the surviving (read, adapter) pairs; the reference's
//     "last row-major cell among maxima wins" (seq_overlap.cpp:338-354) is a wave max-reduce on (M, i, j).
//
// Reference quirks reproduced: Q1 group-of-8 threshold from the LAST read of the
// group / tail group without the min (defined as -t 1 behaviour), Q3 start survives diagonals touching 0,
// Q4 literal find_mask_range, Q5 IUPAC bit-overlap matching, H2 stale range carried from the previous
// adapter of the same read when no cell reaches M >= 0 (the carried range is computed on demand).
#include "faqcs_dev.h"

#include <type_traits>

struct AdapterDev {
    const uint8_t *bits;     // concatenated 4-bit IUPAC masks, one byte per base
    const uint32_t *start;   // [n_adapters + 1] base offsets into bits
    const uint32_t *planes;  // per adapter word w: 4 dwords = bit-planes A,C,G,T of bases [32w, 32w+32)
    const uint32_t *wstart;  // [n_adapters + 1] word offsets into planes (in units of words)
    uint32_t n_adapters;
    float match_rate;        // float(1.0 - filterAdapterMismatchRate), trim.cpp:969
    uint32_t longest;        // (host copies) bases of the longest adapter, dwords of `planes`: what picks the kernel variant
    uint32_t plane_dwords;
};

// seq_overlap.cpp:372-411 na_to_bits() as a table over 'a'..'z' (case folded); 0 == the reference throws
// "Unknown base!".  NA bit masks: A=1 C=2 G=4 T=8 (seq_overlap.h:133-150).
__device__ const uint8_t k_iupac[26] = {
    /*a*/ 1, /*b*/ 14, /*c*/ 2, /*d*/ 13, /*e*/ 0, /*f*/ 0, /*g*/ 4, /*h*/ 11, /*i*/ 0, /*j*/ 0, /*k*/ 12, /*l*/ 0, /*m*/ 3,
    /*n*/ 15, /*o*/ 0, /*p*/ 0, /*q*/ 0, /*r*/ 5, /*s*/ 6, /*t*/ 8, /*u*/ 0, /*v*/ 7, /*w*/ 9, /*x*/ 0, /*y*/ 10, /*z*/ 0};
__device__ __forceinline__ uint32_t na_bits(uint32_t c, const uint8_t *iupac /* LDS copy of k_iupac */)
{
    if (c == '-') return 16u;
    const uint32_t l = (c | 0x20u) - 'a';
    const bool letter = ((c & 0xdfu) >= 'A') && ((c & 0xdfu) <= 'Z');
    return letter ? (uint32_t)iupac[l & 31u] : 0u;
}

// a wave's LDS operations execute in order; this only stops the compiler from moving them across the point
__device__ __forceinline__ void lds_sync_wave()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Register-blocked prefilter of one (read, adapter) pair with |read| <= 256 and |adapter| <= 128.
// A lane's diagonal in block b faces adapter word w at read bit 32 w - 64 b - lane + |read| - 1: the bit shift never
// changes and the dword index is idx00 + w - 2 b, so the lane's shifted read windows R[plane][e], e = w - 2 b + const, are
// built ONCE per read (40 or 56 v_alignbit) and every (block, word) step is 4 and/or + 1 popcount on registers -- no LDS read,
// no shift.  Returns the largest per-diagonal match count of this lane.
// NBR = blocks the window array covers (4 when |read| + |longest short adapter| - 1 <= 256, else 6): e = w - 2 b + 2 (NBR - 1)
// (x & y) | z in ONE instruction.  Written as the v_bitop3_b32 builtin (truth table 0xEA), not as inline assembly: the compiler
// pairs a plain C expression as and + and + or3, and it fenced every inline v_and_or_b32 of a dependent chain with an s_nop
// (nine per plane word in the three-block variant).
__device__ __forceinline__ uint32_t and_or_(uint32_t x, uint32_t y, uint32_t z) { return __builtin_amdgcn_bitop3_b32(x, y, z, 0xEA); }
// b in the lanes whose bit of m is set (the upper half of the wave), else a: one v_cndmask_b32
__device__ __forceinline__ uint32_t sel_half(const uint32_t a, const uint32_t b, const uint64_t m)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t plane_match(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, const uint4 &t)
{
    return and_or_(r3, t.w, and_or_(r2, t.z, and_or_(r1, t.y, r0 & t.x)));
}

template <int NB, int NBR, bool WANT_CNT = false>
__device__ __forceinline__ uint32_t prefilter_max(const uint32_t (&R)[4][2 * NBR + 2], const uint32_t *tpl, const int nw, uint32_t (&cnt_out)[8])
{
    uint32_t cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < nw) {
            const uint4 t = *reinterpret_cast<const uint4 *>(tpl + 4 * w); // the adapter's planes of bases [32w, 32w+32): LDS broadcast
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int e = w - 2 * b + 2 * (NBR - 1);
                cnt[b] += __popc(plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t));
            }
        }
    }
    uint32_t m = 0;
    if (WANT_CNT) {
#pragma unroll
        for (int b = 0; b < 8; ++b) cnt_out[b] = b < NB ? cnt[b] : 0u;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) m = umax_(m, cnt[b]);
    return m;
}

// the same with the number of plane words a compile-time constant: straight-line code (with a run-time word count the compiler
// guards every word of every block with scalar compares and branches -- 18 branches per adapter of a kernel that is bound by
// scalar issue)
template <int NB, int NBR, int NWC>
__device__ __forceinline__ uint32_t prefilter_max_words(const uint32_t (&R)[4][2 * NBR + 2], const uint32_t *tpl)
{
    uint32_t cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
#pragma unroll
    for (int w = 0; w < NWC; ++w) {
        const uint4 t = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int e = w - 2 * b + 2 * (NBR - 1);
            cnt[b] += __popc(plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t));
        }
    }
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) m = umax_(m, cnt[b]);
    return m;
}
// the same for the adapter's FIRST plane word only (the coarse test of stage 1): NB steps, no loop over words
template <int NB, int NBR>
__device__ __forceinline__ uint32_t first_word_max(const uint32_t (&R)[4][2 * NBR + 2], const uint32_t *tpl)
{
    const uint4 t = *reinterpret_cast<const uint4 *>(tpl);
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int e = -2 * b + 2 * (NBR - 1);
        m = umax_(m, (uint32_t)__popc(plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t)));
    }
    return m;
}

#ifndef FAQCS_ADAPTER_WAVES
#define FAQCS_ADAPTER_WAVES 4 /* waves per SIMD the 256-base variant is compiled for (5 = 96 VGPRs with 12 spilled: measured, no faster) */
#endif
template <int NW, int MAXLEN>
__global__ __launch_bounds__(NW * 64, MAXLEN == 320 ? 3 : (MAXLEN == 256 ? FAQCS_ADAPTER_WAVES : 4)) void adapter_overlap(
    const AdapterDev A, const uint8_t *__restrict__ seq, const uint32_t *__restrict__ off, const uint32_t n_reads,
    const uint32_t *__restrict__ seg_start, const uint32_t n_segments, uint32_t *__restrict__ ad_sl,
    uint16_t *__restrict__ ad_hit, uint64_t *__restrict__ adapter_stats, uint32_t *__restrict__ err, const uint32_t dbg)
{
    constexpr int QW = MAXLEN / 32;            // data dwords per plane
    constexpr int PADL = MAXLEN == 320 ? 14 : 12; // zero dwords on each side: the register-blocked stage 1 reads up to PADL dwords
    constexpr int PW = QW + 2 * PADL;          // before / after the data without clamping its index
    constexpr int TPL_CAP = 4096;              // adapter plane dwords cached in LDS (16 KB: every built-in set incl. PhiX)
    constexpr int NBLK = 6;                    // 64-diagonal blocks kept in registers: |read| <= 256, |adapter| <= 128
    __shared__ uint8_t s_q[NW][MAXLEN];        // the read's IUPAC masks (stage 2)
    __shared__ uint8_t s_mask[NW][MAXLEN];     // vector<bool> mask of trim.cpp:991 (1 = unmasked)
    __shared__ uint32_t s_pl[NW][4][PW];       // the read's four base bit-planes, position ordered (stage 1)
    __shared__ __attribute__((aligned(16))) uint32_t s_tpl[TPL_CAP]; // the adapters' bit-planes (4 dwords per 32 bases)
    __shared__ uint8_t s_sb[NW][FAQCS_MAX_ADAPTERS][8]; // stage 1 -> stage 2: per 64-diagonal block, an upper bound of its best score (short adapters that may pass)
    __shared__ uint32_t s_bb[NW][136];         // stage 2 on long targets: per 64-diagonal block, an upper bound of its best score
    __shared__ uint32_t s_ast[2 * FAQCS_MAX_ADAPTERS]; // (reads, bases) credited per adapter by this block
    __shared__ uint8_t s_iupac[32];
    __shared__ uint8_t s_na[256];              // na_to_bits() of every byte value (0 = the reference throws)
    __shared__ uint32_t s_start[FAQCS_MAX_ADAPTERS + 1], s_wstart[FAQCS_MAX_ADAPTERS + 1]; // adapter table of contents
    __shared__ __attribute__((aligned(16))) uint4 s_meta[FAQCS_MAX_ADAPTERS]; // {|adapter|, first plane word, int(rate * |adapter|), base planes present}
    __shared__ uint32_t s_pos[FAQCS_MAX_ADAPTERS];      // adapter -> position in the class order
    __shared__ uint32_t s_ord[FAQCS_MAX_ADAPTERS + 8];  // stage 1 visits the adapters class by class (1, 2, 3, 4 plane words; the rest): position -> adapter,
                                                        // then [64 + c] = first position of class c + 1 ... (stage 1 is order-free: it only sets flags)
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    // plain (non-volatile) pointers so the accesses stay ds_* instructions (a volatile generic pointer degrades
    // to flat_load); ordering between the wave's own LDS writes and reads is pinned by lds_sync_wave()
    uint8_t *q = s_q[wave];
    uint8_t *mk = s_mask[wave];
    uint32_t *pl = &s_pl[wave][0][0];
    const uint32_t n_waves = gridDim.x * NW;

    for (int i = lane; i < 4 * PW; i += 64) pl[i] = 0u; // pads stay zero for the whole kernel
    for (uint32_t i = threadIdx.x; i <= A.n_adapters; i += NW * 64) { s_start[i] = A.start[i]; s_wstart[i] = A.wstart[i]; }
    const uint32_t tpl_dwords = 4u * A.wstart[A.n_adapters];
    const bool tpl_cached = uni((int)(tpl_dwords <= (uint32_t)TPL_CAP)) != 0; // (uni: the branches on it stay scalar branches)
    if (tpl_cached) for (uint32_t i = threadIdx.x; i < tpl_dwords; i += NW * 64) s_tpl[i] = A.planes[i];
    if (threadIdx.x < 32) s_iupac[threadIdx.x] = threadIdx.x < 26 ? k_iupac[threadIdx.x] : (uint8_t)0;
    for (uint32_t i = threadIdx.x; i < 2 * FAQCS_MAX_ADAPTERS; i += NW * 64) s_ast[i] = 0u;
    for (uint32_t i = threadIdx.x; i < A.n_adapters; i += NW * 64) {
        const uint32_t tl = A.start[i + 1] - A.start[i];
        uint32_t amask = 0; // which of the four base planes the adapter has a bit in (any_match below)
        for (uint32_t w = A.wstart[i]; w < A.wstart[i + 1]; ++w)
            for (uint32_t b = 0; b < 4; ++b) amask |= A.planes[4 * w + b] ? 1u << b : 0u;
        s_meta[i] = make_uint4(tl, A.wstart[i], (uint32_t)(int)__fmul_rn(A.match_rate, (float)(int)tl), amask);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += NW * 64) s_na[i] = (uint8_t)na_bits(i, s_iupac);
    if (wave == 0) { // the class order (one wave: ranks from ballots)
        const bool on = (uint32_t)lane < A.n_adapters;
        const int tl = on ? (int)s_meta[on ? lane : 0].x : 0;
        // 1 ... 4 plane words; 5: long targets / uncached -- and an EMPTY adapter string (no plane word: class 0 would be placed by no pass below and
        // leave its s_ord / s_pos entries uninitialised)
        const int cls = !on ? 6 : ((MAXLEN <= 320 && tl > 0 && tl <= 128 && tpl_cached) ? (tl + 31) >> 5 : 5);
        uint32_t base = 0;
        for (int c = 1; c <= 5; ++c) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) { const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); s_ord[at] = (uint32_t)lane; s_pos[lane] = at; }
            base += (uint32_t)__popcll(m);
            if (lane == 0) s_ord[FAQCS_MAX_ADAPTERS + c] = base; // end of class c
        }
    }
    __syncthreads();
    int prev_qlen = 0; // longest span of plane words the previous read of this wave left behind
    int short_tlen_max = 0; // longest adapter the register-blocked prefilter handles (<= 128 bases)
    for (uint32_t i = 0; i < A.n_adapters; ++i) { const int tl = (int)s_meta[i].x; if (tl <= 128 && tl > short_tlen_max) short_tlen_max = tl; }
    short_tlen_max = uni(short_tlen_max);
    bool has_long = false; // some adapter takes the sliding long-target prefilter
    for (uint32_t i = 0; i < A.n_adapters; ++i) has_long = has_long || (int)s_meta[i].x > 128;
    has_long = uni((int)has_long) != 0;
    const bool prefilter_on = uni((int)((dbg & 8u) == 0u)) != 0;

    // A wave takes chunks of 64 consecutive reads: offsets load and results store as one coalesced vector per chunk,
    // per-read scalars come out of the lanes with v_readlane, and the bases of read t+1 are fetched while read t is
    // processed -- no dependent global load sits in front of a read.
    // LONG (reads of 1 025 ... 32 767 bases: one wave per block, its LDS holds the per-base arrays of ONE read): the bases are not
    // kept in registers a read ahead -- the loops over a read's 64-base pieces are run-time loops that fetch from global memory
    constexpr bool LONG = MAXLEN > 1024;
    constexpr int NCH = LONG ? 1 : MAXLEN / 64;
    const uint32_t total_chunks = (n_reads + 63u) >> 6;
#pragma unroll 1
    for (uint32_t chunk = blockIdx.x * NW + wave; chunk < total_chunks; chunk += n_waves) {
      const uint32_t base = chunk << 6;
      const uint32_t my = base + lane;
      const bool mine = my < n_reads;
      const uint32_t v_off = mine ? off[my] : 0u;
      const uint32_t v_len = mine ? off[my + 1] - v_off : 0u;
      uint32_t res_sl = 0, res_hit = 0;
      // ---- which reference segment holds the chunk's first read? (later reads advance it) ------------------
      uint32_t lo = 0, hi = n_segments; // segment s with seg_start[s] <= base < seg_start[s+1]
      {   // segments are FAQCS_SEGMENT_READS long except the tails: try that guess before searching
          const uint32_t g = base / FAQCS_SEGMENT_READS;
          if (g < n_segments && seg_start[g] <= base && base < seg_start[g + 1]) { lo = g; hi = g + 1; }
      }
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_start[mid] <= base) lo = mid; else hi = mid; }
      uint32_t s0 = seg_start[lo], s1 = seg_start[lo + 1];
      uint32_t nbyte[NCH]; // bases of the next read, one byte per lane per 64-base chunk
      if (!LONG) {
          const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)v_off, 0);
          const int l0 = __builtin_amdgcn_readlane((int)v_len, 0);
#pragma unroll
          for (int c = 0; c < NCH; ++c) nbyte[c] = (c * 64 + lane < l0) ? (uint32_t)seq[(size_t)o0 + c * 64 + lane] : 0u;
      }
#pragma unroll 1
      for (int t = 0; t < 64; ++t) {
        const uint32_t r = base + (uint32_t)t;
        if (r >= n_reads) break;
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v_off, t);
        const int qlen = __builtin_amdgcn_readlane((int)v_len, t);
        uint32_t cbyte[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) cbyte[c] = nbyte[c];
        if (!LONG && t + 1 < 64 && r + 1 < n_reads) {
            const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)v_off, t + 1);
            const int l1 = __builtin_amdgcn_readlane((int)v_len, t + 1);
#pragma unroll
            for (int c = 0; c < NCH; ++c) nbyte[c] = (c * 64 + lane < l1) ? (uint32_t)seq[(size_t)o1 + c * 64 + lane] : 0u;
        }
        // ---- which reference group of 8 is this read in? (trim.cpp:977-1071, -t 1 semantics) ----------
        while (r >= s1) { ++lo; s0 = s1; s1 = seg_start[lo + 1]; }   // (empty segments are skipped the same way)
        const uint32_t g_last = s0 + (((r - s0) >> 3) << 3) + 7; // last slot of the group
        const bool tail = g_last >= s1;
        int len8 = 0;
        if (!tail) len8 = (g_last - base < 64u) ? __builtin_amdgcn_readlane((int)v_len, (int)(g_last - base)) : (int)(off[g_last + 1] - off[g_last]);

        // ---- pack_query: bases -> the four bit-planes (the per-base arrays of stage 2 are filled on demand) --------
        bool badbase = false;
        uint64_t pm0 = 0, pm1 = 0, pm2 = 0, pm3 = 0; // OR of the read's plane words (which base planes the read has a bit in)
        const int span = qlen > prev_qlen ? qlen : prev_qlen;
        prev_qlen = qlen;
#pragma unroll(LONG ? 1 : NCH)
        for (int c = 0; c < (LONG ? MAXLEN / 64 : NCH); ++c) { // (LONG: a run-time loop, left at the first piece past the span)
            if (LONG && c * 64 >= span) break;
            if (c * 64 < span) {
                const int p = c * 64 + lane;
                // (a position past the read holds byte 0 -- the prefetch loads zeros there -- whose mask is 0: no test per lane for the short variants)
                uint32_t bits = LONG ? (p < qlen ? (uint32_t)s_na[(uint32_t)seq[(size_t)o + p]] : 0u) : (uint32_t)s_na[cbyte[LONG ? 0 : c]];
                badbase |= bits == 0u && p < qlen;
                // every chunk the previous read touched is rewritten (zeros past this read) so nothing of it survives.  The chunk's eight plane
                // dwords are gathered into lanes 0 ... 7 (two selects per plane) and stored by ONE instruction (as four lane-0 stores each plane paid an
                // EXEC region: the kernel is bound by scalar issue)
                uint32_t wv = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint64_t m = __ballot((bits >> b) & 1u);
                    if (b == 0) pm0 |= m; else if (b == 1) pm1 |= m; else if (b == 2) pm2 |= m; else pm3 |= m;
                    wv = lane == 2 * b ? (uint32_t)m : (lane == 2 * b + 1 ? (uint32_t)(m >> 32) : wv);
                }
                if (lane < 8) pl[(lane >> 1) * PW + PADL + 2 * c + (lane & 1)] = wv;
            }
        }
        const bool read_bad = __any(badbase);
        const uint32_t rmask = (pm0 ? 1u : 0u) | (pm1 ? 2u : 0u) | (pm2 ? 4u : 0u) | (pm3 ? 8u : 0u);
        lds_sync_wave();
        // exact alignment of adapter j: best (M, i, j) over all diagonals -> (score or -1, start, stop)
        auto align_exact = [&](uint32_t j, int &gM, int &gS, int &gI) {
            const uint32_t t0 = s_start[j];
            const int tlen = (int)(s_start[j + 1] - t0);
            gM = -1; gI = 0; gS = 0;
            int gJ = 0;
            const int ndiag = qlen + tlen - 1;
            // A local alignment scores at most the number of matches on its diagonal, so the largest per-diagonal match
            // count of a 64-diagonal block bounds every cell of the block (87 blocks for PhiX, 3-4 for the built-in adapters).  The block with the largest
            // bound is aligned first; a block whose bound is below the best score so far cannot win (ties are decided by
            // the explicit (M, i, j) comparison below, so the visiting order is free).
            const bool pruned = MAXLEN <= 320 && tpl_cached;
            constexpr int NBX = MAXLEN == 320 ? 7 : 6, NAX = NBX + 1; // window blocks / accumulators of the sliding bound pass
            uint32_t *bb = s_bb[wave];
            int first_block = 0;
            if (pruned) {
                uint32_t Rw[4][2 * NBX + 2];
                {
                    const int i00 = (qlen - 1) - lane;
                    const uint32_t sh = (uint32_t)i00 & 31u;
                    const uint32_t *pp = pl + (i00 >> 5) + PADL - 2 * (NBX - 1);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int e = 0; e < 2 * NBX + 2; ++e) Rw[b][e] = __builtin_amdgcn_alignbit(pp[b * PW + e + 1], pp[b * PW + e], sh);
                }
                const uint32_t *tpl = s_tpl + 4 * s_wstart[j];
                const int nw = (tlen + 31) >> 5, nb = (ndiag + 63) >> 6;
                uint32_t cnt[NAX], best = 0;
#pragma unroll
                for (int i = 0; i < NAX; ++i) cnt[i] = 0;
#pragma unroll 1
                for (int u = 0; u <= nb; ++u) {
                    uint4 t0 = make_uint4(0u, 0u, 0u, 0u), t1 = make_uint4(0u, 0u, 0u, 0u);
                    if (2 * u < nw) t0 = *reinterpret_cast<const uint4 *>(tpl + 8 * u);
                    if (2 * u + 1 < nw) t1 = *reinterpret_cast<const uint4 *>(tpl + 8 * u + 4);
#pragma unroll
                    for (int i = 0; i < NAX; ++i) {
                        const int e0 = 2 * NBX - 2 * i, e1 = 2 * NBX + 1 - 2 * i;
                        cnt[i] += __popc(plane_match(Rw[0][e0], Rw[1][e0], Rw[2][e0], Rw[3][e0], t0));
                        cnt[i] += __popc(plane_match(Rw[0][e1], Rw[1][e1], Rw[2][e1], Rw[3][e1], t1));
                    }
                    if (u >= 1) { // block u-1 is complete
                        const uint32_t bnd = wave_max_u32(cnt[0]);
                        if (lane == 0) bb[u - 1] = bnd;
                        if (bnd > best) { best = bnd; first_block = u - 1; }
                    }
#pragma unroll
                    for (int i = 0; i + 1 < NAX; ++i) cnt[i] = cnt[i + 1];
                    cnt[NAX - 1] = 0;
                }
                lds_sync_wave();
            }
            const int n_blocks = (ndiag + 63) >> 6;
#pragma unroll 1
            for (int k = 0; k < n_blocks + (pruned ? 1 : 0); ++k) {
                int blk = k;
                if (pruned) {
                    if (k == 0) blk = first_block;
                    else { blk = k - 1; if (blk == first_block) continue; }
                    const int bnd = (int)bb[blk];
                    if (bnd < 1 || bnd < gM) continue;
                }
                const int dd0 = blk << 6;
                const int d = dd0 + lane - (qlen - 1);                   // j_t - i on this lane's diagonal
                const int dlo = dd0 - (qlen - 1), dhi = dlo + 63;
                const int jt_lo = dlo > 0 ? dlo : 0;
                const int jt_hi = (dhi + qlen - 1) < (tlen - 1) ? (dhi + qlen - 1) : (tlen - 1);
                int M = -1, st = 0, bM = -1, bS = 0, bI = 0;
#pragma unroll 1
                for (int jt = jt_lo; jt <= jt_hi; ++jt) {
                    const uint32_t tb = A.bits[t0 + jt];                 // uniform -> scalar load
                    const int i = jt - d;
                    const bool valid = (unsigned)i < (unsigned)qlen;
                    const uint32_t qb = valid ? q[i] : 0u;
                    const int s = (qb & tb) ? 1 : -1;                    // seq_overlap.cpp:157-161
                    const int nS = (M < 0) ? i : st;                     // seq_overlap.cpp:255,272-275
                    const int nM = (M > 0 ? M : 0) + s;                  // seq_overlap.cpp:185-188
                    M = valid ? nM : -1;
                    st = nS;
                    const bool up = valid && nM >= 0 && nM >= bM;        // seq_overlap.cpp:338-354 (>=: later cell wins)
                    bM = up ? nM : bM; bS = up ? nS : bS; bI = up ? i : bI;
                }
                const int Mx = (int)wave_max_u32((uint32_t)(bM + 1)) - 1;
                if (Mx >= 0) {
                    const uint32_t key = (bM == Mx) ? ((((uint32_t)bI << 13) | (uint32_t)(bI + d)) + 1u) : 0u;
                    const uint32_t K = wave_max_u32(key) - 1u;
                    const int wi = (int)(K >> 13), wj = (int)(K & 8191u);
                    const int wl = wj - wi + (qlen - 1) - dd0;           // lane that owns the winning diagonal
                    const int ws = __builtin_amdgcn_readlane(bS, wl);
                    const bool better = Mx > gM || (Mx == gM && (wi > gI || (wi == gI && wj > gJ)));
                    if (better) { gM = Mx; gI = wi; gJ = wj; gS = ws; }
                }
            }
        };

        // The same alignment for a short adapter whose per-block bounds stage 1 left in s_sb: no bound pass, and the cells come
        // from MATCH BITS -- the lane's diagonal against the adapter, 32 cells per plane word (8 LDS reads + 4 v_alignbit +
        // 4 logic ops), instead of one scalar load and one LDS read per cell -- so the Kadane loop waits for nothing.
        auto align_bits = [&](uint32_t j, int &gM, int &gS, int &gI) {
            const uint4 me = s_meta[j];
            const int tlen = (int)me.x;
            const uint32_t *tpl = s_tpl + 4 * me.y;
            const int nw = (tlen + 31) >> 5;                              // <= 4
            const int ndiag = qlen + tlen - 1, n_blocks = (ndiag + 63) >> 6; // <= 8
            gM = -1; gI = 0; gS = 0;
            int gJ = 0;
            int first_block = 0, fb = -1;
            for (int b = 0; b < n_blocks; ++b) { const int v = (int)s_sb[wave][j][b]; if (v > fb) { fb = v; first_block = b; } }
#pragma unroll 1
            for (int k = 0; k <= n_blocks; ++k) {
                int blk = k - 1;
                if (k == 0) blk = first_block; else if (blk == first_block) continue;
                const int bnd = (int)s_sb[wave][j][blk];
                if (bnd < 1 || bnd < gM) continue;                         // (a local alignment scores at most the matches on its diagonal)
                const int dd0 = blk << 6;
                const int d = dd0 + lane - (qlen - 1);                     // j_t - i on this lane's diagonal
                uint32_t mb[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    mb[w] = 0u;
                    if (w < nw) {
                        const int i0 = 32 * w - d;                         // read position facing adapter base 32 w
                        int idx = i0 >> 5;
                        idx = idx < -(PADL - 1) ? -(PADL - 1) : (idx > QW + PADL - 2 ? QW + PADL - 2 : idx);
                        const uint32_t sh = (uint32_t)i0 & 31u;
                        const uint32_t *pp = pl + idx + PADL;
                        const uint4 t = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
                        mb[w] = plane_match(__builtin_amdgcn_alignbit(pp[1], pp[0], sh), __builtin_amdgcn_alignbit(pp[PW + 1], pp[PW], sh),
                                            __builtin_amdgcn_alignbit(pp[2 * PW + 1], pp[2 * PW], sh), __builtin_amdgcn_alignbit(pp[3 * PW + 1], pp[3 * PW], sh), t);
                    }
                }
                int M = -1, st = 0, bM = -1, bS = 0, bI = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (32 * w < tlen) {
                        const int kend = tlen - 32 * w < 32 ? tlen - 32 * w : 32;
                        const uint32_t bits = mb[w];
#pragma unroll 2
                        for (int kb = 0; kb < kend; ++kb) {
                            const int i = 32 * w + kb - d;
                            const bool valid = (unsigned)i < (unsigned)qlen;
                            const int sc = ((bits >> kb) & 1u) ? 1 : -1;   // seq_overlap.cpp:157-161
                            const int nS = (M < 0) ? i : st;               // seq_overlap.cpp:255,272-275
                            const int nM = (M > 0 ? M : 0) + sc;           // seq_overlap.cpp:185-188
                            M = valid ? nM : -1;
                            st = nS;
                            const bool up = valid && nM >= 0 && nM >= bM;  // seq_overlap.cpp:338-354 (>=: later cell wins)
                            bM = up ? nM : bM; bS = up ? nS : bS; bI = up ? i : bI;
                        }
                    }
                }
                const int Mx = (int)wave_max_u32((uint32_t)(bM + 1)) - 1;
                if (Mx >= 0) {
                    const uint32_t key = (bM == Mx) ? ((((uint32_t)bI << 13) | (uint32_t)(bI + d)) + 1u) : 0u;
                    const uint32_t K = wave_max_u32(key) - 1u;
                    const int wi = (int)(K >> 13), wj = (int)(K & 8191u);
                    const int wl = wj - wi + (qlen - 1) - dd0;             // lane that owns the winning diagonal
                    const int ws = __builtin_amdgcn_readlane(bS, wl);
                    const bool better = Mx > gM || (Mx == gM && (wi > gI || (wi == gI && wj > gJ)));
                    if (better) { gM = Mx; gI = wi; gJ = wj; gS = ws; }
                }
            }
        };

        int best_score = 0, best_j = -1;
        bool have = false, known = false;
        uint32_t last_j = 0;
        int rs = 0, re = 0;
        // ---- stage 1 for every adapter: two bits per adapter (any cell matches / the threshold is reachable) ---------
        uint64_t m_any = 0, m_pass = 0, m_bnd = 0; // m_bnd: stage 1 left the adapter's per-block bounds in s_sb
                                                   // (m_any, m_bnd: bit = adapter; m_pass: bit = the adapter's position in the class order)
        uint32_t pass_v = 0;                       // lane l: the adapter at position l may pass (a lane flag: no 64-bit scalar shifts per adapter)
        auto stage1 = [&](auto nbr_tag) {
            constexpr int NBR = decltype(nbr_tag)::value;
            uint32_t R[4][2 * NBR + 2];
            if (MAXLEN <= 320 && tpl_cached) {
                const int i00 = (qlen - 1) - lane;                       // read bit facing adapter base 0 on block 0's diagonal
                const uint32_t sh = (uint32_t)i00 & 31u;
                const uint32_t *pp = pl + (i00 >> 5) + PADL - 2 * (NBR - 1); // e = 0
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < 2 * NBR + 2; ++e) R[b][e] = __builtin_amdgcn_alignbit(pp[b * PW + e + 1], pp[b * PW + e], sh);
            }
            // per-adapter scalars of this read, computed once with lane = adapter (n_adapters <= 64) and read back with v_readlane inside
            // the loop (the kernel is co-bound by scalar issue: what the loop does per adapter in scalar code is kept to a few unpacks):
            //   va = |adapter| (14 bits) | first plane word << 14 (14) | plane words << 28
            //   vb = need_cnt (16) | coarse need << 16        vc = need + 32768 (16) | thr << 16
            // thr = the reference's threshold (trim.cpp:1007-1008 / :1082); need = 2 thr - min(|read|, |adapter|): what a block's best SCORE has
            // to reach ((mcap + score) / 2 >= thr); need_cnt = max(need, thr): what a diagonal's MATCH COUNT has to reach -- num_match =
            // (match_length + score) / 2 IS the number of matching positions inside the alignment (trim.cpp:1024-1027), it cannot exceed the
            // matches on the alignment's diagonal (polyA: 16 of 20, not the 12 of 20 that 15 % of random reads meet); coarse need = need_cnt -
            // (|adapter| - 32): what the diagonal has to collect on the adapter's FIRST plane word for that (the bases behind it can add at
            // most their number) -- for a two-word adapter (33 ... 64 bases) the first word alone ends the adapter for almost every random read.
            uint32_t va = 0, vb = 0, vc = 0;
            const uint32_t ja = (uint32_t)lane < A.n_adapters ? s_ord[lane] : 0u; // the adapter this lane stands for in the class order
            if ((uint32_t)lane < A.n_adapters) {
                const uint4 me = s_meta[ja];
                const int tl = (int)me.x;
                const int mm = tail ? tl : (len8 < tl ? len8 : tl);
                const int th = mm == tl ? (int)me.z : (int)__fmul_rn(A.match_rate, (float)mm); // trim.cpp:1007-1008 / :1082
                const int need = 2 * th - (qlen < tl ? qlen : tl);
                const int need_cnt = need > th ? need : th;
                const int coarse = need_cnt - (tl > 32 ? tl - 32 : 0);
                va = (uint32_t)tl | (me.y << 14);
                vb = (uint32_t)need_cnt | ((uint32_t)(coarse > 0 ? coarse : 0) << 16);
                vc = (uint32_t)(need + 32768) | ((uint32_t)th << 16);
            }
            {   // the read and the adapter share a base plane <=> some cell of the (read x adapter) matrix matches: exact; lane = adapter
                const bool share = (uint32_t)lane < A.n_adapters && (s_meta[(uint32_t)lane < A.n_adapters ? lane : 0].w & rmask) != 0u;
                m_any = prefilter_on ? __ballot(share) : ~0ull; // (prefilter off, a diagnostic: every adapter goes to stage 2)
            }
            // one adapter of class NWC (its number of plane words; 0: a long target or an uncached table) at position l of the class order
            auto one = [&](const uint32_t l, auto nw_tag) {
                constexpr int NWC = decltype(nw_tag)::value;
                const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)va, (int)l), sb = (uint32_t)__builtin_amdgcn_readlane((int)vb, (int)l);
                const int tlen = (int)(sa & 0x3fffu);
                const int need_cnt = (int)(sb & 0xffffu);
                bool may_pass = true;
                if (NWC > 0 && prefilter_on) {
                    const uint32_t *tpl = s_tpl + 4 * (sa >> 14);
                    // (Measured and rejected: fetching the next adapter's first two plane words an adapter ahead, so that no LDS broadcast
                    // sits in front of its 20 dependent instructions: 128 VGPRs with a spill, 523 -> 470 M reads/s.)
                    // (Measured and rejected: leaving a last plane word of <= 3 bases uncompared and counting those bases as matches.
                    // The weaker bound lets enough random reads through to stage 2 to cost more than the word saves: -8 %.)
                    constexpr int nw = NWC > 0 ? NWC : 1;
                    constexpr int slack = 0;
                    const int nb = (qlen + tlen - 1 + 63) >> 6;              // blocks past the last diagonal would only add zeros
                    uint32_t maxcnt;
                    constexpr int NB_LO = NBR == 4 ? 3 : (NBR == 6 ? 4 : 5), NB_HI = NBR == 4 ? 4 : (NBR == 6 ? 6 : NBR);
                    auto count = [&](auto nb_tag) {
                        constexpr int NB = decltype(nb_tag)::value;
                        if (nw == 2) { // coarse test on the first word, the full count only for what passes it
                            maxcnt = first_word_max<NB, NBR>(R, tpl);
                            may_pass = __any((int)maxcnt >= (int)(sb >> 16));
                            if (may_pass) maxcnt = prefilter_max_words<NB, NBR, 2>(R, tpl);
                        } else maxcnt = prefilter_max_words<NB, NBR, nw>(R, tpl);
                    };
                    if (nb <= NB_LO) count(std::integral_constant<int, NB_LO>{}); else count(std::integral_constant<int, NB_HI>{});
                    if (may_pass) may_pass = __any((int)maxcnt + slack >= need_cnt);
                    if (may_pass) { // rare: the per-block bounds for stage 2 (it aligns the most promising block first and prunes the rest)
                        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)ja, (int)l);
                        const int need_j = (int)((uint32_t)__builtin_amdgcn_readlane((int)vc, (int)l) & 0xffffu) - 32768;
                        uint32_t cw[8];
                        (void)prefilter_max<NBR, NBR, true>(R, tpl, nw, cw);
                        // Second filter.  The match-count bound is weak for short adapters with a low threshold (polyA: 16 of 20 lets
                        // ~15 % of random reads through).  For a block whose count bound reaches `need` the exact best score of every
                        // diagonal is cheap to get from the SAME match bits: a score-only Kadane, 4 instructions per cell, no start
                        // tracking (seq_overlap.cpp:185-188; a cell outside the read has match bit 0 and can only lower M, which is what
                        // the exact pass's reset to -1 does too).  The wave maximum is the block's exact bound: num_match =
                        // (match_length + score) / 2 <= (mcap + score) / 2, so a read passes only if some block's best score reaches `need`.
                        bool pass2 = false;
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            if (b < NBR && b < nb) {
                                uint32_t bnd = wave_max_u32(cw[b]) + (uint32_t)slack;
                                if ((int)bnd >= need_cnt && bnd > 0u) { // (wave-uniform)
                                    int M = -1, best = -1;
#pragma unroll
                                    for (int w = 0; w < 4; ++w) {
                                        if (w < nw) {
                                            const uint4 t = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
                                            const int e = w - 2 * b + 2 * (NBR - 1);
                                            const uint32_t bits = plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t);
                                            const int kend = tlen - 32 * w < 32 ? tlen - 32 * w : 32;
#pragma unroll 4
                                            for (int kb = 0; kb < kend; ++kb) {
                                                const int sc = (int)((bits >> kb) & 1u) * 2 - 1;
                                                M = (M > 0 ? M : 0) + sc;
                                                best = best > M ? best : M;
                                            }
                                        }
                                    }
                                    const int bx = (int)wave_max_u32((uint32_t)(best + 1)) - 1; // exact best score of the block (-1: no match)
                                    bnd = bx > 0 ? (uint32_t)bx : 0u;
                                    pass2 = pass2 || bx >= need_j;
                                }
                                if (lane == 0) s_sb[wave][j][b] = (uint8_t)(bnd > 255u ? 255u : bnd);
                            }
                        }
                        may_pass = pass2;
                        if (may_pass) m_bnd |= 1ull << j;
                    }
                } else if (NWC == 0 && prefilter_on && MAXLEN <= 320 && NBR >= 6 && tpl_cached) {
                    // long target (PhiX, artifact sequences): the same register windows, sliding over the target two words
                    // per step.  Words 2u and 2u+1 face exactly the NBR+1 blocks u-1 .. u+NBR-1 (window index 2 NBR - 2i and
                    // 2 NBR + 1 - 2i for block u-1+i); block u-1 has seen all of its words after step u and leaves the accumulator.
                    constexpr int NACC = NBR + 1;
                    const uint32_t *tpl = s_tpl + 4 * ((sa >> 14) & 0x3fffu);
                    const int nw = (tlen + 31) >> 5;
                    const int nb = (qlen + tlen - 1 + 63) >> 6;
                    uint32_t cnt[NACC], maxcnt = 0;
#pragma unroll
                    for (int i = 0; i < NACC; ++i) cnt[i] = 0;
#pragma unroll 1
                    for (int u = 0; u <= nb; ++u) {
                        uint4 t0 = make_uint4(0u, 0u, 0u, 0u), t1 = make_uint4(0u, 0u, 0u, 0u);
                        if (2 * u < nw) t0 = *reinterpret_cast<const uint4 *>(tpl + 8 * u);
                        if (2 * u + 1 < nw) t1 = *reinterpret_cast<const uint4 *>(tpl + 8 * u + 4);
#pragma unroll
                        for (int i = 0; i < NACC; ++i) {
                            const int e0 = 2 * NBR - 2 * i, e1 = 2 * NBR + 1 - 2 * i;
                            cnt[i] += __popc(plane_match(R[0][e0], R[1][e0], R[2][e0], R[3][e0], t0));
                            cnt[i] += __popc(plane_match(R[0][e1], R[1][e1], R[2][e1], R[3][e1], t1));
                        }
                        maxcnt = umax_(maxcnt, cnt[0]);
#pragma unroll
                        for (int i = 0; i + 1 < NACC; ++i) cnt[i] = cnt[i + 1];
                        cnt[NACC - 1] = 0;
                    }
                    may_pass = __any((int)maxcnt >= need_cnt);
                } else if (NWC == 0 && prefilter_on) {
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)ja, (int)l);
                    const uint32_t w0 = s_wstart[j];
                    const int nw = (int)(s_wstart[j + 1] - w0);
                    const int ndiag = qlen + tlen - 1;
                    uint32_t maxcnt = 0;
#pragma unroll 1
                    for (int dd0 = 0; dd0 < ndiag; dd0 += 64) {
                        const int d = dd0 + lane - (qlen - 1);
                        uint32_t cnt = 0;
#pragma unroll 1
                        for (int w = 0; w < nw; ++w) {
                            const uint32_t tA = A.planes[4 * (w0 + w) + 0], tC = A.planes[4 * (w0 + w) + 1];
                            const uint32_t tG = A.planes[4 * (w0 + w) + 2], tT = A.planes[4 * (w0 + w) + 3];
                            const int i0 = 32 * w - d;                   // read position facing adapter base 32w
                            int idx = i0 >> 5;
                            idx = idx < -2 ? -2 : (idx > QW ? QW : idx);
                            const uint32_t sh = (uint32_t)i0 & 31u;
                            const uint32_t *pp = pl + idx + PADL;
                            const uint32_t qa = __builtin_amdgcn_alignbit(pp[1], pp[0], sh);
                            const uint32_t qc = __builtin_amdgcn_alignbit(pp[PW + 1], pp[PW], sh);
                            const uint32_t qg = __builtin_amdgcn_alignbit(pp[2 * PW + 1], pp[2 * PW], sh);
                            const uint32_t qt = __builtin_amdgcn_alignbit(pp[3 * PW + 1], pp[3 * PW], sh);
                            cnt += __popc((qa & tA) | (qc & tC) | (qg & tG) | (qt & tT));
                        }
                        maxcnt = umax_(maxcnt, cnt);
                    }
                    const int bound = (int)wave_max_u32(maxcnt);             // the largest match count of a diagonal
                    may_pass = bound >= need_cnt;
                }
                pass_v = ((uint32_t)lane == l && may_pass) ? 1u : pass_v;
            };
            // class by class: inside a class the number of plane words is a compile-time constant (with the classes mixed in one loop the
            // compiler turns the choice into chains of scalar flag tests: 36 scalar instructions and 17 branches per adapter, measured)
            uint32_t l = 0;
            const uint32_t e1 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 1]), e2 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 2]), e3 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 3]),
                           e4 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 4]);
#pragma unroll 1
            for (; l < e1; ++l) one(l, std::integral_constant<int, 1>{});
#pragma unroll 1
            for (; l < e2; ++l) one(l, std::integral_constant<int, 2>{});
#pragma unroll 1
            for (; l < e3; ++l) one(l, std::integral_constant<int, 3>{});
#pragma unroll 1
            for (; l < e4; ++l) one(l, std::integral_constant<int, 4>{});
#pragma unroll 1
            for (; l < A.n_adapters; ++l) one(l, std::integral_constant<int, 0>{});
            m_pass = __ballot(pass_v != 0u);
        };
        if (!read_bad && qlen > 0) {
            if (MAXLEN == 320) stage1(std::integral_constant<int, 7>{});   // up to (320 + 128) / 64 = 7 blocks
            else if (!has_long && qlen + short_tlen_max - 1 <= 256) stage1(std::integral_constant<int, 4>{});
            else stage1(std::integral_constant<int, 6>{});
        }
        // ---- stage 2 + the reference's sequential state (stale range, mask, credit), trim.cpp:1003-1071.  With every
        // adapter matching somewhere and none able to reach its threshold (the bulk of the reads) nothing can happen.
        const uint64_t m_all = A.n_adapters >= 64 ? ~0ull : ((1ull << A.n_adapters) - 1ull);
        if (!read_bad && qlen > 0 && !(m_pass == 0 && m_any == m_all)) {
            // the per-base arrays of stage 2 are only needed here (a few percent of the reads)
            // (from the bytes this lane fetched for pack_query: a second global load here stalled every such read -- a fifth of them
            // with --polyA, whose weak threshold lets ~15 % of random reads through the prefilter -- for a memory latency)
            if (LONG) {
                for (int p = lane; p < qlen; p += 64) { q[p] = s_na[seq[(size_t)o + p]]; mk[p] = 1; }
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int p = c * 64 + lane;
                    if (p < qlen) { q[p] = s_na[cbyte[c]]; mk[p] = 1; }
                }
            }
            lds_sync_wave();
#pragma unroll 1
            for (uint32_t j = 0; j < A.n_adapters; ++j) {
                const int tlen = (int)(s_start[j + 1] - s_start[j]);
                const int m = tail ? tlen : (len8 < tlen ? len8 : tlen);
                const int thr = (int)__fmul_rn(A.match_rate, (float)m);      // trim.cpp:1007-1008 / :1082
                const bool any_match = (m_any >> j) & 1ull, may_pass = (m_pass >> uniu(s_pos[j])) & 1ull;
                int score = 0;
                if (any_match) {
                    if (!may_pass) { have = true; known = false; last_j = j; continue; } // cannot mask, cannot be credited
                    int gM, gS, gI;
                    if ((m_bnd >> j) & 1ull) align_bits(j, gM, gS, gI); else align_exact(j, gM, gS, gI);
                    if (gM >= 0) { have = true; known = true; last_j = j; rs = gS; re = gI; score = gM; }
                    else if (!have) continue;                                // (only reachable with the prefilter disabled)
                } else {
                    if (!have) continue;                                     // H2: unknown stale state -> no hit
                    if (!known) { int gM, gS, gI; align_exact(last_j, gM, gS, gI); rs = gS; re = gI; known = true; }
                }
                const int match_length = re - rs + 1;
                const int num_match = (match_length + score) / 2;            // trim.cpp:1024-1025
                if (num_match >= thr) {
                    for (int p = rs + lane; p <= re; p += 64) mk[p] = 0;     // trim.cpp:1032-1034
                    if (score > best_score) { best_score = score; best_j = (int)j; }
                }
            }
        }

        uint32_t first = 0, second = (uint32_t)qlen;
        if (best_score > 0) {
            lds_sync_wave();
            // find_mask_range, trim.cpp:1144-1189, literal -- but walked run by run instead of base by base.  The loop's state
            // only changes where the mask changes: an unmasked run [s, e) adds e - s to run_length (setting run_start when it
            // was 0), and the masked bases after it all perform the same test, so one test per gap is the whole gap.  The
            // transitions come from ballots of the mask bytes (a few scalar steps instead of |read| dependent LDS reads).
            uint32_t longest_run_start = 0, longest_run_length = 0, run_start = 0, run_length = 0;
            uint32_t pending = 0;      // start of the open unmasked run
            bool open = false;
            uint64_t carry = 0;        // mask bit of the previous position (position -1 counts as masked)
#pragma unroll(LONG ? 1 : NCH)
            for (int c = 0; c < (LONG ? MAXLEN / 64 : NCH); ++c) {
                if (LONG && c * 64 >= qlen) break;
                if (c * 64 < qlen) {
                    const int p = c * 64 + lane;
                    const uint64_t m = __ballot(p < qlen && mk[p] != 0);
                    uint64_t t = m ^ ((m << 1) | carry);   // bit b set: the mask changes between position 64c+b-1 and 64c+b
                    carry = m >> 63;
                    while (t) {
                        const uint32_t pos = (uint32_t)(c * 64) + (uint32_t)__builtin_ctzll(t);
                        t &= t - 1;
                        if (!open) { if (run_length == 0) run_start = pos; pending = pos; open = true; }
                        else {
                            run_length += pos - pending; open = false;
                            if (run_length > longest_run_length) { longest_run_length = run_length; longest_run_start = run_start; run_length = 0; }
                        }
                    }
                }
            }
            if (open) run_length += (uint32_t)qlen - pending; // the read ends inside an unmasked run
            if (run_length > longest_run_length) { longest_run_length = run_length; longest_run_start = run_start; }
            first = longest_run_length ? longest_run_start : 0u;
            second = longest_run_length;
            if (lane == 0) { // trim.cpp:1061-1064; block-local, one global atomic pair per adapter at the end of the block
                atomicAdd(&s_ast[2 * best_j], 1u);
                atomicAdd(&s_ast[2 * best_j + 1], (uint32_t)qlen - second);
            }
        }
        // 0xffff = the read holds a base na_to_bits() rejects (seq_overlap.cpp:409): the trim kernel turns it into FAQCS_F_ERR_BASE
        if (lane == t) { res_sl = first | (second << 16); res_hit = read_bad ? 0xffffu : (uint32_t)(best_score > 0 ? best_j + 1 : 0); }
        if (read_bad && lane == 0) atomicOr(err, 2u);
        lds_sync_wave();
      }
      if (mine) { ad_sl[my] = res_sl; ad_hit[my] = (uint16_t)res_hit; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * A.n_adapters; i += NW * 64)
        if (s_ast[i]) atomicAdd((unsigned long long *)&adapter_stats[i], (unsigned long long)s_ast[i]);
}

// =====================================================================================================================================
// adapter_overlap_pair: TWO reads per wave for reads of up to 160 bases against adapters of up to 128 bases (round 6).
//
// adapter_overlap spends 430 scalar instructions per read next to its 585 vector ones (profiles/r5e/pmc_adapter.txt): the per-adapter
// control -- unpacking the adapter's scalars, the LDS broadcast of its plane words and the wait for it, the wave votes and branches --
// is paid per READ although it does not depend on the read.  Here the two halves of a wave own the diagonals of two reads: lanes 0 ... 31
// read A, lanes 32 ... 63 read B, a block is 32 diagonals, and stage 1 -- the exact skip test, where nearly every read ends -- runs for both
// reads in ONE instruction stream: the adapter's plane words are fetched once, the control runs once, and the vector work per read is
// the same (a read's 150 + |adapter| - 1 diagonals fill six or seven blocks of 32 as they filled three or four of 64; the lane's windows
// are 12 registers per plane for the pair instead of 10 per read).  What a read needs beyond stage 1 -- a few per cent of the reads --
// is the code of adapter_overlap itself, run for one read at a time with the whole wave: stage 1 hands it the same facts (which
// adapters match anywhere, which may reach their threshold, per-64-diagonal bounds of the best score), all of them sound bounds, so
// every result is adapter_overlap's (trim.cpp:961-1142, seq_overlap.cpp:157-354; the quirks listed at the top of this file).
// =====================================================================================================================================
// counts of the lane's diagonal in NB blocks of 32 diagonals against NWC plane words: window index e = w - b + (NBR - 1)
template <int NB, int NBR, int NWC>
__device__ __forceinline__ void pair_counts(const uint32_t (&R)[4][NBR + 3], const uint32_t *tpl, uint32_t (&cnt)[NBR])
{
#pragma unroll
    for (int b = 0; b < NBR; ++b) cnt[b] = 0;
#pragma unroll
    for (int w = 0; w < NWC; ++w) {
        const uint4 t = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int e = w - b + (NBR - 1);
            cnt[b] += __popc(plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t));
        }
    }
}
template <int NB, int NBR>
__device__ __forceinline__ uint32_t pair_first_word_max(const uint32_t (&R)[4][NBR + 3], const uint32_t *tpl)
{
    const uint4 t = *reinterpret_cast<const uint4 *>(tpl);
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int e = -b + (NBR - 1);
        m = umax_(m, (uint32_t)__popc(plane_match(R[0][e], R[1][e], R[2][e], R[3][e], t)));
    }
    return m;
}
// the maximum over each half of the wave: lane 31 holds the one of lanes 0 ... 31, lane 63 the one of lanes 32 ... 63
__device__ __forceinline__ uint32_t half_max_u32(uint32_t v)
{
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    return v;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void adapter_overlap_pair(
    const AdapterDev A, const uint8_t *__restrict__ seq, const uint32_t *__restrict__ off, const uint32_t n_reads,
    const uint32_t *__restrict__ seg_start, const uint32_t n_segments, uint32_t *__restrict__ ad_sl,
    uint16_t *__restrict__ ad_hit, uint64_t *__restrict__ adapter_stats, uint32_t *__restrict__ err, const uint32_t dbg)
{
    constexpr int MAXLEN = 192;                // per-base arrays (reads of up to 160 bases; three 64-base pieces)
    constexpr int QW = MAXLEN / 32, PADL = 12, PW = QW + 2 * PADL;
    constexpr int TPL_CAP = 4096;
    constexpr int NBR = 9;                     // blocks of 32 diagonals: 160 + 128 - 1 <= 288
    constexpr int NCH = MAXLEN / 64;
    __shared__ uint8_t s_q[NW][2][MAXLEN];
    __shared__ uint8_t s_mask[NW][2][MAXLEN];
    __shared__ uint32_t s_pl[NW][2][4][PW];
    __shared__ __attribute__((aligned(16))) uint32_t s_tpl[TPL_CAP];
    __shared__ uint8_t s_sb[NW][2][FAQCS_MAX_ADAPTERS][8]; // stage 1 -> stage 2: per 64-diagonal block, an upper bound of its best score
    __shared__ uint32_t s_bb[NW][136];
    __shared__ uint32_t s_ast[2 * FAQCS_MAX_ADAPTERS];
    __shared__ uint8_t s_iupac[32];
    __shared__ uint8_t s_na[256];
    __shared__ uint32_t s_start[FAQCS_MAX_ADAPTERS + 1], s_wstart[FAQCS_MAX_ADAPTERS + 1];
    __shared__ __attribute__((aligned(16))) uint4 s_meta[FAQCS_MAX_ADAPTERS];
    __shared__ uint32_t s_pos[FAQCS_MAX_ADAPTERS];
    __shared__ uint32_t s_ord[FAQCS_MAX_ADAPTERS + 8];
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = uni(threadIdx.x >> 6);
    uint32_t *plw = &s_pl[wave][0][0][0];
    const uint32_t n_waves = gridDim.x * NW;

    for (int i = lane; i < 2 * 4 * PW; i += 64) plw[i] = 0u; // pads stay zero for the whole kernel
    for (uint32_t i = threadIdx.x; i <= A.n_adapters; i += NW * 64) { s_start[i] = A.start[i]; s_wstart[i] = A.wstart[i]; }
    const uint32_t tpl_dwords = 4u * A.wstart[A.n_adapters];
    for (uint32_t i = threadIdx.x; i < tpl_dwords; i += NW * 64) s_tpl[i] = A.planes[i]; // (the launcher has checked that they fit)
    if (threadIdx.x < 32) s_iupac[threadIdx.x] = threadIdx.x < 26 ? k_iupac[threadIdx.x] : (uint8_t)0;
    for (uint32_t i = threadIdx.x; i < 2 * FAQCS_MAX_ADAPTERS; i += NW * 64) s_ast[i] = 0u;
    for (uint32_t i = threadIdx.x; i < A.n_adapters; i += NW * 64) {
        const uint32_t tl = A.start[i + 1] - A.start[i];
        uint32_t amask = 0;
        for (uint32_t w = A.wstart[i]; w < A.wstart[i + 1]; ++w)
            for (uint32_t b = 0; b < 4; ++b) amask |= A.planes[4 * w + b] ? 1u << b : 0u;
        s_meta[i] = make_uint4(tl, A.wstart[i], (uint32_t)(int)__fmul_rn(A.match_rate, (float)(int)tl), amask);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += NW * 64) s_na[i] = (uint8_t)na_bits(i, s_iupac);
    if (wave == 0) { // the class order: 1 ... 4 plane words (the launcher admits no other adapter here; an empty one cannot exist: faqcs_create)
        const bool on = (uint32_t)lane < A.n_adapters;
        const int tl = on ? (int)s_meta[on ? lane : 0].x : 0;
        const int cls = !on ? 6 : (tl + 31) >> 5;
        uint32_t base = 0;
        for (int c = 1; c <= 4; ++c) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) { const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); s_ord[at] = (uint32_t)lane; s_pos[lane] = at; }
            base += (uint32_t)__popcll(m);
            if (lane == 0) s_ord[FAQCS_MAX_ADAPTERS + c] = base; // end of class c
        }
    }
    __syncthreads();
    int prev_qlen2[2] = {0, 0}; // longest span of plane words the previous read of either slot left behind
    const bool prefilter_on = uni((int)((dbg & 8u) == 0u)) != 0;
    const uint64_t m_all32 = (1ull << A.n_adapters) - 1ull; // (n_adapters <= 32)

    const uint32_t total_chunks = (n_reads + 63u) >> 6;
#pragma unroll 1
    for (uint32_t chunk = blockIdx.x * NW + wave; chunk < total_chunks; chunk += n_waves) {
      const uint32_t base = chunk << 6;
      const uint32_t my = base + lane;
      const bool mine = my < n_reads;
      const uint32_t v_off = mine ? off[my] : 0u;
      const uint32_t v_len = mine ? off[my + 1] - v_off : 0u;
      uint32_t res_sl = 0, res_hit = 0;
      uint32_t lo = 0, hi = n_segments; // segment s with seg_start[s] <= base < seg_start[s+1]
      {
          const uint32_t g = base / FAQCS_SEGMENT_READS;
          if (g < n_segments && seg_start[g] <= base && base < seg_start[g + 1]) { lo = g; hi = g + 1; }
      }
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_start[mid] <= base) lo = mid; else hi = mid; }
      uint32_t s0 = seg_start[lo], s1 = seg_start[lo + 1];
      uint32_t nbyte[2][NCH]; // bases of the next pair, one byte per lane per 64-base piece
#pragma unroll
      for (int h = 0; h < 2; ++h) {
          const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)v_off, h);
          const int l0 = __builtin_amdgcn_readlane((int)v_len, h);
#pragma unroll
          for (int c = 0; c < NCH; ++c) nbyte[h][c] = (c * 64 + lane < l0) ? (uint32_t)seq[(size_t)o0 + c * 64 + lane] : 0u;
      }
#pragma unroll 1
      for (int t = 0; t < 64; t += 2) {
        if (base + (uint32_t)t >= n_reads) break;
        uint32_t cbyte[2][NCH];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < NCH; ++c) cbyte[h][c] = nbyte[h][c];
        if (t + 2 < 64 && base + (uint32_t)t + 2u < n_reads) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)v_off, t + 2 + h);
                const int l1 = __builtin_amdgcn_readlane((int)v_len, t + 2 + h); // (a lane past the batch holds length 0)
#pragma unroll
                for (int c = 0; c < NCH; ++c) nbyte[h][c] = (c * 64 + lane < l1) ? (uint32_t)seq[(size_t)o1 + c * 64 + lane] : 0u;
            }
        }
        // ---- per read: its reference group of 8 (trim.cpp:977-1071, -t 1 semantics), its planes ----
        int qlen2[2], len8_2[2];
        bool tail2[2], bad2[2], there2[2];
        uint32_t rmask2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t r = base + (uint32_t)(t + h);
            there2[h] = r < n_reads;
            qlen2[h] = there2[h] ? __builtin_amdgcn_readlane((int)v_len, t + h) : 0;
            tail2[h] = true; len8_2[h] = 0;
            if (there2[h]) {
                while (r >= s1) { ++lo; s0 = s1; s1 = seg_start[lo + 1]; }   // (empty segments are skipped the same way)
                const uint32_t g_last = s0 + (((r - s0) >> 3) << 3) + 7; // last slot of the group
                tail2[h] = g_last >= s1;
                if (!tail2[h]) len8_2[h] = (g_last - base < 64u) ? __builtin_amdgcn_readlane((int)v_len, (int)(g_last - base)) : (int)(off[g_last + 1] - off[g_last]);
            }
            // pack_query: bases -> the four bit-planes of slot h
            uint32_t *pl = plw + h * (4 * PW);
            bool badbase = false;
            uint64_t pm0 = 0, pm1 = 0, pm2 = 0, pm3 = 0;
            const int qlen = qlen2[h];
            const int span = qlen > prev_qlen2[h] ? qlen : prev_qlen2[h];
            prev_qlen2[h] = qlen;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c * 64 < span) {
                    const int p = c * 64 + lane;
                    const uint32_t bits = (uint32_t)s_na[cbyte[h][c]];
                    badbase |= bits == 0u && p < qlen;
                    uint32_t wv = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint64_t m = __ballot((bits >> b) & 1u);
                        if (b == 0) pm0 |= m; else if (b == 1) pm1 |= m; else if (b == 2) pm2 |= m; else pm3 |= m;
                        wv = lane == 2 * b ? (uint32_t)m : (lane == 2 * b + 1 ? (uint32_t)(m >> 32) : wv);
                    }
                    if (lane < 8) pl[(lane >> 1) * PW + PADL + 2 * c + (lane & 1)] = wv;
                }
            }
            bad2[h] = __any(badbase);
            rmask2[h] = (pm0 ? 1u : 0u) | (pm1 ? 2u : 0u) | (pm2 ? 4u : 0u) | (pm3 ? 8u : 0u);
        }
        lds_sync_wave();

        // ---- stage 1 for both reads: lanes 0 ... 31 the diagonals of read A, lanes 32 ... 63 those of read B ----------------------------
        uint64_t m_any2 = 0, m_pass2 = 0;
        uint64_t m_bnd2[2] = {0, 0};
        const bool live2[2] = {there2[0] && !bad2[0] && qlen2[0] > 0, there2[1] && !bad2[1] && qlen2[1] > 0};
        if (live2[0] || live2[1]) {
            const int qlen_v = half ? qlen2[1] : qlen2[0];
            const bool live_v = half ? live2[1] : live2[0];
            const int qmax = qlen2[0] > qlen2[1] ? qlen2[0] : qlen2[1];
            uint32_t R[4][NBR + 3];
            {
                const int i00 = (qlen_v - 1) - l32;                          // read bit facing adapter base 0 on block 0's diagonal
                const uint32_t sh = (uint32_t)i00 & 31u;
                const uint32_t *pp = plw + half * (4 * PW) + (i00 >> 5) + PADL - (NBR - 1); // e = 0
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < NBR + 3; ++e) R[b][e] = __builtin_amdgcn_alignbit(pp[b * PW + e + 1], pp[b * PW + e], sh);
            }
            // per-(read, adapter) scalars with lane = (half, adapter position): as in adapter_overlap (thr, need, need_cnt, coarse need)
            uint32_t va = 0, vb = 0x7fffu | (0x7fffu << 16), vc = 0;
            const uint32_t ja = (uint32_t)l32 < A.n_adapters ? s_ord[l32] : 0u;
            if ((uint32_t)l32 < A.n_adapters) {
                const uint4 me = s_meta[ja];
                const int tl = (int)me.x;
                const bool tail_v = half ? tail2[1] : tail2[0];
                const int len8_v = half ? len8_2[1] : len8_2[0];
                const int mm = tail_v ? tl : (len8_v < tl ? len8_v : tl);
                const int th = mm == tl ? (int)me.z : (int)__fmul_rn(A.match_rate, (float)mm); // trim.cpp:1007-1008 / :1082
                const int need = 2 * th - (qlen_v < tl ? qlen_v : tl);
                const int need_cnt = need > th ? need : th;
                const int coarse = need_cnt - (tl > 32 ? tl - 32 : 0);
                va = (uint32_t)tl | (me.y << 14);
                if (live_v) vb = (uint32_t)need_cnt | ((uint32_t)(coarse > 0 ? coarse : 0) << 16); // (a read that is not there can pass nothing: 0x7fff matches)
                vc = (uint32_t)(need + 32768) | ((uint32_t)th << 16);
            }
            {
                const uint32_t rm = half ? rmask2[1] : rmask2[0];
                const bool share = (uint32_t)l32 < A.n_adapters && (s_meta[(uint32_t)l32 < A.n_adapters ? l32 : 0].w & rm) != 0u;
                m_any2 = prefilter_on ? __ballot(share) : ~0ull;
            }
            const uint64_t hmask = 0xffffffff00000000ull;
            uint32_t pass_v = 0;
            auto one = [&](const uint32_t l, auto nw_tag) {
                constexpr int NWC = decltype(nw_tag)::value;
                const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)va, (int)l);
                const uint32_t sbv = sel_half((uint32_t)__builtin_amdgcn_readlane((int)vb, (int)l), (uint32_t)__builtin_amdgcn_readlane((int)vb, (int)l + 32), hmask);
                const int tlen = (int)(sa & 0x3fffu);
                const uint32_t need_cnt_v = sbv & 0xffffu;
                uint64_t may = ~0ull; // bit = a lane of the read's half reaches the read's need
                if (prefilter_on) {
                    const uint32_t *tpl = s_tpl + 4 * (sa >> 14);
                    const int nb = (qmax + tlen - 1 + 31) >> 5;          // blocks past the last diagonal would only add zeros
                    uint32_t cnt[NBR];
                    uint32_t maxcnt = 0;
                    auto count = [&](auto nb_tag) {
                        constexpr int NB = decltype(nb_tag)::value;
                        if (NWC == 2) { // coarse test on the first word, the full count only for what passes it
                            maxcnt = pair_first_word_max<NB, NBR>(R, tpl);
                            may = __ballot(maxcnt >= (sbv >> 16));
                            if (may) {
                                pair_counts<NB, NBR, 2>(R, tpl, cnt);
                                maxcnt = 0;
#pragma unroll
                                for (int b = 0; b < NB; ++b) maxcnt = umax_(maxcnt, cnt[b]);
                            }
                        } else {
                            pair_counts<NB, NBR, NWC>(R, tpl, cnt);
#pragma unroll
                            for (int b = 0; b < NB; ++b) maxcnt = umax_(maxcnt, cnt[b]);
                        }
                    };
                    if (nb <= 6) count(std::integral_constant<int, 6>{}); else if (nb <= 7) count(std::integral_constant<int, 7>{}); else count(std::integral_constant<int, NBR>{});
                    if (may) may = __ballot(maxcnt >= need_cnt_v);
                    if (may) { // rare: the per-block bounds for stage 2, and the second filter (the exact best score of the blocks that reach the count)
                        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)ja, (int)l);
                        const uint32_t scv = sel_half((uint32_t)__builtin_amdgcn_readlane((int)vc, (int)l), (uint32_t)__builtin_amdgcn_readlane((int)vc, (int)l + 32), hmask);
                        const int need_j_v = (int)(scv & 0xffffu) - 32768;
                        pair_counts<NBR, NBR, NWC>(R, tpl, cnt); // (every block: the bounds must cover the longer read's last ones too)
                        const int needc0 = __builtin_amdgcn_readlane((int)need_cnt_v, 0), needc1 = __builtin_amdgcn_readlane((int)need_cnt_v, 32);
                        const int needj0 = __builtin_amdgcn_readlane(need_j_v, 0), needj1 = __builtin_amdgcn_readlane(need_j_v, 32);
                        uint32_t bnd64[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
                        bool pass2[2] = {false, false};
#pragma unroll
                        for (int b = 0; b < NBR; ++b) {
                            if (b < nb) {
                                const uint32_t hm = half_max_u32(cnt[b]);
                                uint32_t bh[2] = {(uint32_t)__builtin_amdgcn_readlane((int)hm, 31), (uint32_t)__builtin_amdgcn_readlane((int)hm, 63)};
                                const bool d0 = (int)bh[0] >= needc0 && bh[0] > 0u, d1 = (int)bh[1] >= needc1 && bh[1] > 0u;
                                if (d0 || d1) { // (wave-uniform) a score-only Kadane over the block's match bits: the exact best score of every diagonal
                                    int M = -1, best = -1;
#pragma unroll
                                    for (int w = 0; w < NWC; ++w) {
                                        const uint4 tw = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
                                        const int e = w - b + (NBR - 1);
                                        const uint32_t bits = plane_match(R[0][e], R[1][e], R[2][e], R[3][e], tw);
                                        const int kend = tlen - 32 * w < 32 ? tlen - 32 * w : 32;
#pragma unroll 4
                                        for (int kb = 0; kb < kend; ++kb) {
                                            const int sc = (int)((bits >> kb) & 1u) * 2 - 1;
                                            M = (M > 0 ? M : 0) + sc;
                                            best = best > M ? best : M;
                                        }
                                    }
                                    const uint32_t hb = half_max_u32((uint32_t)(best + 1));
                                    const int bx[2] = {__builtin_amdgcn_readlane((int)hb, 31) - 1, __builtin_amdgcn_readlane((int)hb, 63) - 1};
                                    if (d0) { bh[0] = bx[0] > 0 ? (uint32_t)bx[0] : 0u; pass2[0] = pass2[0] || bx[0] >= needj0; }
                                    if (d1) { bh[1] = bx[1] > 0 ? (uint32_t)bx[1] : 0u; pass2[1] = pass2[1] || bx[1] >= needj1; }
                                }
                                bnd64[0][b >> 1] = umax_(bnd64[0][b >> 1], bh[0]);
                                bnd64[1][b >> 1] = umax_(bnd64[1][b >> 1], bh[1]);
                            }
                        }
                        if (lane == 0) {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int B = 0; B < 5; ++B) s_sb[wave][h][j][B] = (uint8_t)(bnd64[h][B] > 255u ? 255u : bnd64[h][B]);
                        }
                        may = (pass2[0] ? ~hmask : 0ull) | (pass2[1] ? hmask : 0ull);
                        if (pass2[0]) m_bnd2[0] |= 1ull << j;
                        if (pass2[1]) m_bnd2[1] |= 1ull << j;
                    }
                }
                const bool p0 = (may & ~hmask) != 0ull, p1 = (may & hmask) != 0ull;
                pass_v = (((uint32_t)lane == l && p0) || ((uint32_t)lane == l + 32u && p1)) ? 1u : pass_v;
            };
            uint32_t l = 0;
            const uint32_t e1 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 1]), e2 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 2]), e3 = uniu(s_ord[FAQCS_MAX_ADAPTERS + 3]);
#pragma unroll 1
            for (; l < e1; ++l) one(l, std::integral_constant<int, 1>{});
#pragma unroll 1
            for (; l < e2; ++l) one(l, std::integral_constant<int, 2>{});
#pragma unroll 1
            for (; l < e3; ++l) one(l, std::integral_constant<int, 3>{});
#pragma unroll 1
            for (; l < A.n_adapters; ++l) one(l, std::integral_constant<int, 4>{});
            m_pass2 = __ballot(pass_v != 0u);
        }

        // ---- stage 2 + the reference's sequential state, one read at a time with the whole wave: adapter_overlap's code --------------
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            // (selects, not indexed arrays: with a run-time index the compiler keeps the arrays in scratch memory)
            if (!(h ? there2[1] : there2[0])) break;
            const int qlen = h ? qlen2[1] : qlen2[0];
            const bool tail = h ? tail2[1] : tail2[0];
            const int len8 = h ? len8_2[1] : len8_2[0];
            const bool read_bad = h ? bad2[1] : bad2[0];
            uint8_t *q = s_q[wave][h];
            uint8_t *mk = s_mask[wave][h];
            uint32_t *pl = plw + h * (4 * PW);
            const uint64_t m_any = (m_any2 >> (32 * h)) & 0xffffffffull, m_pass = (m_pass2 >> (32 * h)) & 0xffffffffull, m_bnd = h ? m_bnd2[1] : m_bnd2[0];
            uint8_t (*sb)[8] = s_sb[wave][h];

            auto align_exact = [&](uint32_t j, int &gM, int &gS, int &gI) {
                const uint32_t t0 = s_start[j];
                const int tlen = (int)(s_start[j + 1] - t0);
                gM = -1; gI = 0; gS = 0;
                int gJ = 0;
                const int ndiag = qlen + tlen - 1;
                constexpr int NBX = 6, NAX = NBX + 1; // window blocks / accumulators of the sliding bound pass
                uint32_t *bb = s_bb[wave];
                int first_block = 0;
                {
                    uint32_t Rw[4][2 * NBX + 2];
                    {
                        const int i00 = (qlen - 1) - lane;
                        const uint32_t sh = (uint32_t)i00 & 31u;
                        const uint32_t *pp = pl + (i00 >> 5) + PADL - 2 * (NBX - 1);
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int e = 0; e < 2 * NBX + 2; ++e) Rw[b][e] = __builtin_amdgcn_alignbit(pp[b * PW + e + 1], pp[b * PW + e], sh);
                    }
                    const uint32_t *tpl = s_tpl + 4 * s_wstart[j];
                    const int nw = (tlen + 31) >> 5, nb = (ndiag + 63) >> 6;
                    uint32_t cnt[NAX], best = 0;
#pragma unroll
                    for (int i = 0; i < NAX; ++i) cnt[i] = 0;
#pragma unroll 1
                    for (int u = 0; u <= nb; ++u) {
                        uint4 ta = make_uint4(0u, 0u, 0u, 0u), tb = make_uint4(0u, 0u, 0u, 0u);
                        if (2 * u < nw) ta = *reinterpret_cast<const uint4 *>(tpl + 8 * u);
                        if (2 * u + 1 < nw) tb = *reinterpret_cast<const uint4 *>(tpl + 8 * u + 4);
#pragma unroll
                        for (int i = 0; i < NAX; ++i) {
                            const int ea = 2 * NBX - 2 * i, eb = 2 * NBX + 1 - 2 * i;
                            cnt[i] += __popc(plane_match(Rw[0][ea], Rw[1][ea], Rw[2][ea], Rw[3][ea], ta));
                            cnt[i] += __popc(plane_match(Rw[0][eb], Rw[1][eb], Rw[2][eb], Rw[3][eb], tb));
                        }
                        if (u >= 1) { // block u-1 is complete
                            const uint32_t bnd = wave_max_u32(cnt[0]);
                            if (lane == 0) bb[u - 1] = bnd;
                            if (bnd > best) { best = bnd; first_block = u - 1; }
                        }
#pragma unroll
                        for (int i = 0; i + 1 < NAX; ++i) cnt[i] = cnt[i + 1];
                        cnt[NAX - 1] = 0;
                    }
                    lds_sync_wave();
                }
                const int n_blocks = (ndiag + 63) >> 6;
#pragma unroll 1
                for (int k = 0; k < n_blocks + 1; ++k) {
                    int blk = k - 1;
                    if (k == 0) blk = first_block; else if (blk == first_block) continue;
                    const int bnd = (int)bb[blk];
                    if (bnd < 1 || bnd < gM) continue;
                    const int dd0 = blk << 6;
                    const int d = dd0 + lane - (qlen - 1);                   // j_t - i on this lane's diagonal
                    const int dlo = dd0 - (qlen - 1), dhi = dlo + 63;
                    const int jt_lo = dlo > 0 ? dlo : 0;
                    const int jt_hi = (dhi + qlen - 1) < (tlen - 1) ? (dhi + qlen - 1) : (tlen - 1);
                    int M = -1, st = 0, bM = -1, bS = 0, bI = 0;
#pragma unroll 1
                    for (int jt = jt_lo; jt <= jt_hi; ++jt) {
                        const uint32_t tb = A.bits[t0 + jt];                 // uniform -> scalar load
                        const int i = jt - d;
                        const bool valid = (unsigned)i < (unsigned)qlen;
                        const uint32_t qb = valid ? q[i] : 0u;
                        const int s = (qb & tb) ? 1 : -1;                    // seq_overlap.cpp:157-161
                        const int nS = (M < 0) ? i : st;                     // seq_overlap.cpp:255,272-275
                        const int nM = (M > 0 ? M : 0) + s;                  // seq_overlap.cpp:185-188
                        M = valid ? nM : -1;
                        st = nS;
                        const bool up = valid && nM >= 0 && nM >= bM;        // seq_overlap.cpp:338-354 (>=: later cell wins)
                        bM = up ? nM : bM; bS = up ? nS : bS; bI = up ? i : bI;
                    }
                    const int Mx = (int)wave_max_u32((uint32_t)(bM + 1)) - 1;
                    if (Mx >= 0) {
                        const uint32_t key = (bM == Mx) ? ((((uint32_t)bI << 13) | (uint32_t)(bI + d)) + 1u) : 0u;
                        const uint32_t K = wave_max_u32(key) - 1u;
                        const int wi = (int)(K >> 13), wj = (int)(K & 8191u);
                        const int wl = wj - wi + (qlen - 1) - dd0;           // lane that owns the winning diagonal
                        const int ws = __builtin_amdgcn_readlane(bS, wl);
                        const bool better = Mx > gM || (Mx == gM && (wi > gI || (wi == gI && wj > gJ)));
                        if (better) { gM = Mx; gI = wi; gJ = wj; gS = ws; }
                    }
                }
            };
            auto align_bits = [&](uint32_t j, int &gM, int &gS, int &gI) {
                const uint4 me = s_meta[j];
                const int tlen = (int)me.x;
                const uint32_t *tpl = s_tpl + 4 * me.y;
                const int nw = (tlen + 31) >> 5;                              // <= 4
                const int ndiag = qlen + tlen - 1, n_blocks = (ndiag + 63) >> 6; // <= 5
                gM = -1; gI = 0; gS = 0;
                int gJ = 0;
                int first_block = 0, fb = -1;
                for (int b = 0; b < n_blocks; ++b) { const int v = (int)sb[j][b]; if (v > fb) { fb = v; first_block = b; } }
#pragma unroll 1
                for (int k = 0; k <= n_blocks; ++k) {
                    int blk = k - 1;
                    if (k == 0) blk = first_block; else if (blk == first_block) continue;
                    const int bnd = (int)sb[j][blk];
                    if (bnd < 1 || bnd < gM) continue;                         // (a local alignment scores at most the matches on its diagonal)
                    const int dd0 = blk << 6;
                    const int d = dd0 + lane - (qlen - 1);                     // j_t - i on this lane's diagonal
                    uint32_t mb[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        mb[w] = 0u;
                        if (w < nw) {
                            const int i0 = 32 * w - d;                         // read position facing adapter base 32 w
                            int idx = i0 >> 5;
                            idx = idx < -(PADL - 1) ? -(PADL - 1) : (idx > QW + PADL - 2 ? QW + PADL - 2 : idx);
                            const uint32_t sh = (uint32_t)i0 & 31u;
                            const uint32_t *pp = pl + idx + PADL;
                            const uint4 tw = *reinterpret_cast<const uint4 *>(tpl + 4 * w);
                            mb[w] = plane_match(__builtin_amdgcn_alignbit(pp[1], pp[0], sh), __builtin_amdgcn_alignbit(pp[PW + 1], pp[PW], sh),
                                                __builtin_amdgcn_alignbit(pp[2 * PW + 1], pp[2 * PW], sh), __builtin_amdgcn_alignbit(pp[3 * PW + 1], pp[3 * PW], sh), tw);
                        }
                    }
                    int M = -1, st = 0, bM = -1, bS = 0, bI = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (32 * w < tlen) {
                            const int kend = tlen - 32 * w < 32 ? tlen - 32 * w : 32;
                            const uint32_t bits = mb[w];
#pragma unroll 2
                            for (int kb = 0; kb < kend; ++kb) {
                                const int i = 32 * w + kb - d;
                                const bool valid = (unsigned)i < (unsigned)qlen;
                                const int sc = ((bits >> kb) & 1u) ? 1 : -1;   // seq_overlap.cpp:157-161
                                const int nS = (M < 0) ? i : st;               // seq_overlap.cpp:255,272-275
                                const int nM = (M > 0 ? M : 0) + sc;           // seq_overlap.cpp:185-188
                                M = valid ? nM : -1;
                                st = nS;
                                const bool up = valid && nM >= 0 && nM >= bM;  // seq_overlap.cpp:338-354 (>=: later cell wins)
                                bM = up ? nM : bM; bS = up ? nS : bS; bI = up ? i : bI;
                            }
                        }
                    }
                    const int Mx = (int)wave_max_u32((uint32_t)(bM + 1)) - 1;
                    if (Mx >= 0) {
                        const uint32_t key = (bM == Mx) ? ((((uint32_t)bI << 13) | (uint32_t)(bI + d)) + 1u) : 0u;
                        const uint32_t K = wave_max_u32(key) - 1u;
                        const int wi = (int)(K >> 13), wj = (int)(K & 8191u);
                        const int wl = wj - wi + (qlen - 1) - dd0;             // lane that owns the winning diagonal
                        const int ws = __builtin_amdgcn_readlane(bS, wl);
                        const bool better = Mx > gM || (Mx == gM && (wi > gI || (wi == gI && wj > gJ)));
                        if (better) { gM = Mx; gI = wi; gJ = wj; gS = ws; }
                    }
                }
            };

            int best_score = 0, best_j = -1;
            bool have = false, known = false;
            uint32_t last_j = 0;
            int rs = 0, re = 0;
            // With every adapter matching somewhere and none able to reach its threshold (the bulk of the reads) nothing can happen.
            if (!read_bad && qlen > 0 && !(m_pass == 0 && m_any == m_all32)) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int p = c * 64 + lane;
                    if (p < qlen) { q[p] = s_na[h ? cbyte[1][c] : cbyte[0][c]]; mk[p] = 1; }
                }
                lds_sync_wave();
#pragma unroll 1
                for (uint32_t j = 0; j < A.n_adapters; ++j) {
                    const int tlen = (int)(s_start[j + 1] - s_start[j]);
                    const int m = tail ? tlen : (len8 < tlen ? len8 : tlen);
                    const int thr = (int)__fmul_rn(A.match_rate, (float)m);      // trim.cpp:1007-1008 / :1082
                    const bool any_match = (m_any >> j) & 1ull, may_pass = (m_pass >> uniu(s_pos[j])) & 1ull; // (m_any: bit = adapter; m_pass: bit = its position in the class order)
                    int score = 0;
                    if (any_match) {
                        if (!may_pass) { have = true; known = false; last_j = j; continue; } // cannot mask, cannot be credited
                        int gM, gS, gI;
                        if ((m_bnd >> j) & 1ull) align_bits(j, gM, gS, gI); else align_exact(j, gM, gS, gI);
                        if (gM >= 0) { have = true; known = true; last_j = j; rs = gS; re = gI; score = gM; }
                        else if (!have) continue;                                // (only reachable with the prefilter disabled)
                    } else {
                        if (!have) continue;                                     // H2: unknown stale state -> no hit
                        if (!known) { int gM, gS, gI; align_exact(last_j, gM, gS, gI); rs = gS; re = gI; known = true; }
                    }
                    const int match_length = re - rs + 1;
                    const int num_match = (match_length + score) / 2;            // trim.cpp:1024-1025
                    if (num_match >= thr) {
                        for (int p = rs + lane; p <= re; p += 64) mk[p] = 0;     // trim.cpp:1032-1034
                        if (score > best_score) { best_score = score; best_j = (int)j; }
                    }
                }
            }

            uint32_t first = 0, second = (uint32_t)qlen;
            if (best_score > 0) {
                lds_sync_wave();
                // find_mask_range, trim.cpp:1144-1189, literal -- walked run by run (see adapter_overlap)
                uint32_t longest_run_start = 0, longest_run_length = 0, run_start = 0, run_length = 0;
                uint32_t pending = 0;
                bool open = false;
                uint64_t carry = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c * 64 < qlen) {
                        const int p = c * 64 + lane;
                        const uint64_t m = __ballot(p < qlen && mk[p] != 0);
                        uint64_t tt = m ^ ((m << 1) | carry);
                        carry = m >> 63;
                        while (tt) {
                            const uint32_t pos = (uint32_t)(c * 64) + (uint32_t)__builtin_ctzll(tt);
                            tt &= tt - 1;
                            if (!open) { if (run_length == 0) run_start = pos; pending = pos; open = true; }
                            else {
                                run_length += pos - pending; open = false;
                                if (run_length > longest_run_length) { longest_run_length = run_length; longest_run_start = run_start; run_length = 0; }
                            }
                        }
                    }
                }
                if (open) run_length += (uint32_t)qlen - pending;
                if (run_length > longest_run_length) { longest_run_length = run_length; longest_run_start = run_start; }
                first = longest_run_length ? longest_run_start : 0u;
                second = longest_run_length;
                if (lane == 0) { // trim.cpp:1061-1064
                    atomicAdd(&s_ast[2 * best_j], 1u);
                    atomicAdd(&s_ast[2 * best_j + 1], (uint32_t)qlen - second);
                }
            }
            if (lane == t + h) { res_sl = first | (second << 16); res_hit = read_bad ? 0xffffu : (uint32_t)(best_score > 0 ? best_j + 1 : 0); }
            if (read_bad && lane == 0) atomicOr(err, 2u);
            lds_sync_wave();
        }
      }
      if (mine) { ad_sl[my] = res_sl; ad_hit[my] = (uint16_t)res_hit; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * A.n_adapters; i += NW * 64)
        if (s_ast[i]) atomicAdd((unsigned long long *)&adapter_stats[i], (unsigned long long)s_ast[i]);
}

hipError_t faqcs_launch_adapter(const AdapterDev &A, const uint8_t *seq, const uint32_t *off, uint32_t n_reads,
                                uint32_t max_len, const uint32_t *seg_start, uint32_t n_segments, uint32_t *ad_sl,
                                uint16_t *ad_hit, uint64_t *adapter_stats, uint32_t *err, uint32_t dbg, int n_cu, hipStream_t st)
{
    if (n_reads == 0) return hipSuccess;
    // two reads per wave (round 6): reads of up to 160 bases, at most 32 adapters of at most 128 bases whose planes fit the LDS copy;
    // FAQCS_ADAPTER_PAIR=0 keeps adapter_overlap for them (A/B)
    static const bool pair_on = [] { const char *e = getenv("FAQCS_ADAPTER_PAIR"); return !e || atoi(e) != 0; }();
    if (pair_on && max_len <= 160 && A.n_adapters <= 32 && A.longest <= 128 && A.plane_dwords <= 4096 && (dbg & 8u) == 0u) {
        constexpr int NW = 4;
        uint32_t grid = (n_reads + 64 * NW - 1) / (64 * NW);
        const uint32_t cap = (uint32_t)n_cu * 8u;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL((adapter_overlap_pair<NW>), dim3(grid), dim3(NW * 64), 0, st, A, seq, off, n_reads, seg_start,
                           n_segments, ad_sl, ad_hit, adapter_stats, err, dbg);
        return hipGetLastError();
    }
    if (max_len <= 256) {
        constexpr int NW = 4; // 4 waves/SIMD either way (123 VGPRs); A/B on MI355X: 4-wave blocks +2 % over 8-wave blocks
        uint32_t grid = (n_reads + NW - 1) / NW;
        const uint32_t cap = (uint32_t)n_cu * 8u;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL((adapter_overlap<NW, 256>), dim3(grid), dim3(NW * 64), 0, st, A, seq, off, n_reads, seg_start,
                           n_segments, ad_sl, ad_hit, adapter_stats, err, dbg);
    } else if (max_len <= 320) { // MiSeq 2x300: the register-blocked prefilter with a 16-entry window per plane
        constexpr int NW = 4;
        uint32_t grid = (n_reads + NW - 1) / NW;
        const uint32_t cap = (uint32_t)n_cu * 6u;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL((adapter_overlap<NW, 320>), dim3(grid), dim3(NW * 64), 0, st, A, seq, off, n_reads, seg_start,
                           n_segments, ad_sl, ad_hit, adapter_stats, err, dbg);
    } else if (max_len <= 1024) {
        constexpr int NW = 8;
        uint32_t grid = (n_reads + NW - 1) / NW;
        const uint32_t cap = (uint32_t)n_cu * 4u;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL((adapter_overlap<NW, 1024>), dim3(grid), dim3(NW * 64), 0, st, A, seq, off, n_reads, seg_start,
                           n_segments, ad_sl, ad_hit, adapter_stats, err, dbg);
    } else if (max_len <= FAQCS_MAX_READ_LENGTH) {
        // long reads: one wave per block, its LDS holds the per-base arrays of ONE read (2.5 bytes per base + 20 KB): the variant is picked by
        // the batch's longest read so that reads of a few thousand bases still get several waves per CU (5 / 4 / 2 / 1 blocks)
#define FAQCS_ADAPTER_LONG(ML, PER_CU)                                                                                             \
        {                                                                                                                          \
            uint32_t grid = n_reads;                                                                                               \
            const uint32_t cap = (uint32_t)n_cu * PER_CU;                                                                          \
            if (grid > cap) grid = cap;                                                                                            \
            hipLaunchKernelGGL((adapter_overlap<1, ML>), dim3(grid), dim3(64), 0, st, A, seq, off, n_reads, seg_start, n_segments, \
                               ad_sl, ad_hit, adapter_stats, err, dbg);                                                            \
        }
        if (max_len <= 4096) FAQCS_ADAPTER_LONG(4096, 5u)
        else if (max_len <= 8192) FAQCS_ADAPTER_LONG(8192, 4u)
        else if (max_len <= 16384) FAQCS_ADAPTER_LONG(16384, 2u)
        else FAQCS_ADAPTER_LONG(32768, 1u)
#undef FAQCS_ADAPTER_LONG
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
