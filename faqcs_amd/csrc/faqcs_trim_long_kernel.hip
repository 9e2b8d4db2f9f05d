// faqcs_trim_long_kernel.hip -- trim_long: trim_read() (trim.cpp:225-551) for batches that hold a read of more than 1 024 bases
// (up to FAQCS_MAX_READ_LENGTH = 32 767: long-read platforms, assembled contigs run through FaQCs).  gfx950, wave64.
//
// ONE wavefront per read, lane = position modulo 64.  The chunked kernels keep a whole read (or 64 of them) in registers / LDS and
// fold the reference's float tests into per-length integer tables; neither scales to 32 767 positions, and reads of this kind are
// few and long, so this kernel trades their tricks for a plain structure:
//   * every pass streams the read from global memory, four 64-position pieces per round (a read is 2 x <= 32 KB: it stays in the L2
//     between the passes); per-read counts are per-lane 16-bit fields summed over the wave once per read (as ballots + popcounts into
//     scalar counters they made the kernel scalar-issue bound: a CU retires one scalar instruction per clock);
//   * the quality trimmers (BWA_plus / BWA / HARD, trim.cpp:629-793) run as wave-parallel scans: a lane stands for one STEP of the
//     reference's walk, the running area is a DPP prefix sum, the step that ends the walk and the step of the cut come out of
//     ballots and a wave maximum (64 steps per ~40 vector and ~25 scalar instructions; rounds 3's form walked the steps one by
//     one in scalar code over v_readlane, which made the kernel scalar-issue bound);
//   * the float expressions of the average-quality and low-complexity tests and of the composition bins are evaluated as the
//     reference writes them (IEEE single / double operations, no contraction), not through tables;
//   * the per-position matrices (position x quality, position x base, before and after trimming: four counters per base) are NOT
//     added by this kernel: it leaves the read's verdict in the result array and its terminal-N runs in a scratch word, and
//     long_accumulate (below) adds the matrices tile by tile through LDS.  (The first form of this kernel added them itself, one
//     64-bit global atomic per counter and base: 13 G atomics/s, the chip's memory-side atomic rate, = 3.3 G bases/s.)
// Results are bit-identical to the chunked kernels on reads both can take (tests/test_gpu_parity.py runs short batches through
// this kernel with FAQCS_TRIM_LONG=1).
#include "faqcs_trim_common.h"

namespace {

__device__ __forceinline__ int q_score(const uint32_t raw, const int off) // fastq.h:quality_score without its throw: > 41 is the caller's error
{
    const int v = (int)(int8_t)raw - off;
    return v < 0 ? 0 : v;
}
// column of update_base_statistics (trim.cpp:810-875): A 0, T 1, C 2, G 3, N 4 (either case), 5 = not counted
__device__ __forceinline__ uint32_t base_col(const uint32_t b)
{
    const uint32_t l = b | 0x20u;
    return l == 'a' ? 0u : l == 't' ? 1u : l == 'c' ? 2u : l == 'g' ? 3u : l == 'n' ? 4u : 5u;
}
__device__ __forceinline__ void add64(uint64_t *p, const uint64_t v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
__device__ __forceinline__ uint32_t popc64(const uint64_t m) { return (uint32_t)__builtin_popcountll(m); }

// the six composition bins of one read (trim.cpp:860-874)
__device__ __forceinline__ void composition_bins(uint64_t *comp, const uint32_t len, const uint32_t nA, const uint32_t nT, const uint32_t nC,
                                                 const uint32_t nG, const uint32_t nN)
{
    const float norm = len > 0 ? __fdiv_rn((float)(FAQCS_NCOMP_BIN - 1), (float)len) : 0.0f;
    const uint32_t iA = (uint32_t)__fmul_rn(norm, (float)nA), iT = (uint32_t)__fmul_rn(norm, (float)nT);
    const uint32_t iC = (uint32_t)__fmul_rn(norm, (float)nC), iG = (uint32_t)__fmul_rn(norm, (float)nG), iN = (uint32_t)__fmul_rn(norm, (float)nN);
    add64(comp + (size_t)iA * FAQCS_NCOMP_KIND + 0, 1); add64(comp + (size_t)iT * FAQCS_NCOMP_KIND + 1, 1);
    add64(comp + (size_t)iC * FAQCS_NCOMP_KIND + 2, 1); add64(comp + (size_t)iG * FAQCS_NCOMP_KIND + 3, 1);
    add64(comp + (size_t)iN * FAQCS_NCOMP_KIND + 4, 1); add64(comp + (size_t)(iG + iC) * FAQCS_NCOMP_KIND + 5, 1);
}
// int(average_quality()) and the value itself (trim.cpp:553-576)
__device__ __forceinline__ float average_q(const int total, const uint32_t len, const int in_off)
{
    if (!len) return 0.0f;
    const float v = __fsub_rn(__fdiv_rn((float)total, (float)len), (float)in_off);
    return v > 0.0f ? v : 0.0f;
}

} // namespace

template <int NW>
__global__ __launch_bounds__(NW * 64) void trim_long(const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                    const uint32_t *__restrict__ off, const uint32_t n_reads, const uint32_t *__restrict__ ad_sl,
                                                    const uint16_t *__restrict__ ad_hit, uint2 *__restrict__ out, uint64_t *__restrict__ counters,
                                                    uint32_t *__restrict__ err, uint32_t *__restrict__ lead_trail)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)uni((int)(threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * NW;
    const faqcs_layout &L = P.lay;
    uint64_t *fs = counters + L.filter_stats;
    const int in_off = P.in_off;
    bool any_err = false;
    // Counters every read adds to (FilterStat, the 42-bin histograms, the length histograms of equal-length reads) are hot ADDRESSES: as
    // one global atomic per read and counter they serialise at the memory side (2 M reads x 12 same-address atomics = 130 ms, measured).
    // FilterStat: sums in wave-uniform registers, added once per wave at the end; quality-bin histograms: the block's LDS, flushed at the
    // end; length histograms: consecutive reads of one length are counted up and added when the length changes.
    __shared__ uint32_t s_qh[4][FAQCS_NQ]; // pre reads, pre bases, post reads, post bases per int(average quality)
    for (int i = threadIdx.x; i < 4 * FAQCS_NQ; i += NW * 64) (&s_qh[0][0])[i] = 0u;
    __syncthreads();
    uint64_t w_fs[FAQCS_NUM_STAT];
#pragma unroll
    for (int k = 0; k < FAQCS_NUM_STAT; ++k) w_fs[k] = 0;
    uint32_t pre_key = 0xffffffffu, pre_run = 0, post_key = 0xffffffffu, post_run = 0;
#pragma unroll 1
    for (uint32_t r = blockIdx.x * NW + wave; r < n_reads; r += n_waves) {
        const uint32_t o = uniu(off[r]);                       // (wave-uniform: everything derived from these stays in scalar registers)
        const int len0 = uni((int)(off[r + 1] - o));
        const uint8_t *s = seq + o, *q = qual + o;
        // ---- mask_quality_terminal_N (trim.cpp:1191-1216): the runs of upper-case N at the two ends read as quality == offset ----
        int lead = len0, trail = len0;
#pragma unroll 1
        for (int c = 0; c < len0; c += 64) {
            const int p = c + lane;
            const uint64_t m = __ballot(p < len0 && s[p < len0 ? p : len0 - 1] != 'N'); // (clamped index, unconditional load: a per-lane `cond ? load : 0` costs an EXEC region)
            if (m) { lead = c + (int)__builtin_ctzll(m); break; }
        }
        if (lead < len0) {
#pragma unroll 1
            for (int c = 0; c < len0; c += 64) {
                const int p = len0 - 1 - c - lane; // descending
                const uint64_t m = __ballot(p >= 0 && s[p >= 0 ? p : 0] != 'N');
                if (m) { trail = c + (int)__builtin_ctzll(m); break; }
            }
        }
        lead = uni(lead); trail = uni(trail);
        const int tail_from = len0 - trail; // positions >= tail_from are patched
        // the quality byte of position p as trim_read() sees it; p is clamped into the read so that the load needs no EXEC region (the callers
        // mask what lies outside): the branchy form made the kernel scalar-bound -- 335 SALU instructions per 64 positions, measured
        const int last_p = len0 > 0 ? len0 - 1 : 0;
        auto raw_q = [&](const int p) -> uint32_t {
            const uint32_t v = (uint32_t)q[p < 0 ? 0 : (p > last_p ? last_p : p)];
            return (p < lead || p >= tail_from) ? (uint32_t)(in_off & 0xff) : v;
        };
        auto base_at = [&](const int p) -> uint32_t { return (uint32_t)s[p < 0 ? 0 : (p > last_p ? last_p : p)]; };

        // ---- pass 1 over the whole read: range check, the sum of the raw bytes, base counts (trim.cpp:247-258) ----
        int total0 = 0;
        uint32_t nAT = 0, nCG = 0, nNN = 0; // per-lane counts, two 16-bit fields each (a lane sees <= 512 positions of a read)
        bool bad_q = false;
#pragma unroll 1
        for (int c = 0; c < len0; c += 256) { // four 64-position pieces per round: their loads are in flight together
            uint32_t rqv[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = c + 64 * u + lane;
                rqv[u] = raw_q(p);
                bv[u] = base_at(p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = c + 64 * u + lane < len0;
                bad_q = bad_q || (in && q_score(rqv[u], in_off) > 41);
                total0 += in ? (int)(int8_t)rqv[u] : 0;
                const uint32_t col = in ? base_col(bv[u]) : 5u;
                nAT += (col == 0u ? 1u : 0u) + (col == 1u ? 0x10000u : 0u);
                nCG += (col == 2u ? 1u : 0u) + (col == 3u ? 0x10000u : 0u);
                nNN += col == 4u ? 1u : 0u;
            }
        }
        nAT = (uint32_t)wave_sum_i32((int)nAT); nCG = (uint32_t)wave_sum_i32((int)nCG); // (a read has <= 32 767 bases: the low field cannot carry)
        const uint32_t pA = nAT & 0xffffu, pT = nAT >> 16, pC = nCG & 0xffffu, pG = nCG >> 16, pN = (uint32_t)wave_sum_i32((int)nNN);
        const bool read_err = __any(bad_q);
        total0 = wave_sum_i32(total0);

        // ---- the window trim_read() works on: adapter mask (trim.cpp:270-277,934-954), --5end / --3end (:279-314) ----
        uint32_t flags = 0, filt = 0;
        bool ret = true;
        if (read_err) { ret = false; flags |= FAQCS_F_ERR_QUALITY; any_err = true; } // (fastq.h:31-33 throws: the run ends, this read counts nothing)
        int w0 = 0, len = len0;      // current substring [w0, w0 + len) of the read
        uint32_t offset_5 = 0;       // the reference's bookkeeping of the 5' offset (differs from w0 for a read an adapter wiped out)
        if (P.has_adapters) {
            const uint32_t sl = uniu(ad_sl[r]);
            const int first = (int)(sl & 0xffffu), second = (int)(sl >> 16);
            if (len != second) { w0 += first; offset_5 += second == 0 ? (uint32_t)len : (uint32_t)first; len = second; flags |= FAQCS_F_ADAPTER; }
        }
        if (P.trim5 && !P.qc_only) {
            if ((int)P.trim5 > len) len = 0;
            else { w0 += (int)P.trim5; len -= (int)P.trim5; offset_5 += P.trim5; }
        }
        if (P.trim3 && !P.qc_only) len = (int)P.trim3 > len ? 0 : len - (int)P.trim3;
        uint64_t f_len_reads = 0, f_len_bases = 0;
        if (len < (int)P.min_len || len == 0) { f_len_bases += (uint64_t)len; ++f_len_reads; ret = false; filt = FAQCS_FILT_LENGTH_PRE; }

        // ---- quality trimming (hard_trim / BWA_trim / BWA_plus_trim, trim.cpp:629-793) as wave-parallel scans over 64-position
        // pieces of the window: lane l of piece k stands for STEP t = 64 k + l of the reference's walk -- window position len - 1 - t
        // of a 3' walk, position t of a 5' walk.  What a walk carries from step to step is a running sum (the "area") and a small
        // counter that a step re-arms; both have closed forms over a piece:
        //   area before step t        A(t) = sum over the earlier steps of (Q - score): an inclusive prefix sum over the lanes (DPP) plus
        //                             the total of the earlier pieces;
        //   re-arm R(t)               BWA_plus: the position is more than 2 away from the far end and A(t) >= 0 (trim.cpp:739-741,767-769);
        //   where the walk ends       BWA_plus: the counter starts at min(len, 5) and a re-arming step sets it to min(len, 2), so the walk
        //                             ends after step min(len, 5) - 1 when none of those steps re-arms, else after the first step t at least
        //                             two past the first re-arming one with !R(t - 1) and !R(t): two ballots and a count-trailing-zeros;
        //                             BWA: before the first step with A(t) < 0 or position 0 (:690); HARD: at the first score above Q (:640-668);
        //   the cut                   the step with the largest area after it among the visited ones, the EARLIEST on ties, if that
        //                             area is positive (`if (area > maxArea)` is strict, :745,:773): a wave maximum and a ballot.
        // A good read ends its walks inside the first piece; a long low-quality tail costs one piece per 64 positions.
        uint32_t qt_bases = 0;
        if (!P.qc_only && ret && !read_err) {
            const int Q = P.Q;
            // scores of the 64 steps of piece k (desc: from the window's 3' end), 0 and `valid` false past the window
            auto piece = [&](const int k, const bool desc, bool &valid) -> int {
                const int t = 64 * k + lane, i = desc ? len - 1 - t : t;
                valid = t < len;
                return valid ? q_score(raw_q(w0 + i), in_off) : 0;
            };
            const uint64_t lane_lt = (1ull << lane) - 1ull; // lanes below this one
            // One area walk (BWA_plus 3' / 5', BWA): returns the step after which the area was largest and positive (-1: none).
            // far_lim: BWA_plus re-arms while (distance to the far end, in steps) ... see the call sites; plus = false: the BWA rule.
            auto area_walk = [&](const bool desc, const bool plus, const int rearm_below) -> int {
                const int a0 = len < 5 ? len : 5, nan2 = len < 2 ? len : 2; // at_least_scan / num_after_neg of trim.cpp:723-724
                (void)nan2;
                int carry = 0, best = 0, best_t = -1;
                bool seen = false, prev_nr = false;
#pragma unroll 1
                for (int k = 0; 64 * k < len; ++k) {
                    bool valid;
                    const int sc = piece(k, desc, valid);
                    const int d = valid ? Q - sc : 0;
                    const int incl = wave_incl_scan_add(d);
                    const int a_before = carry + incl - d, a_after = carry + incl;
                    const int t = 64 * k + lane;
                    int stop_lane = 64; // last visited lane of this piece when the walk ends in it
                    uint64_t visited;
                    if (plus) {
                        // re-arm: 3' walk: position > num_after_neg (:739); 5' walk: position < final_pos_3 - num_after_neg (:767) -- both
                        // "step index below rearm_below"
                        const uint64_t rm = __ballot(valid && t < rearm_below && a_before >= 0);
                        const uint64_t nr = ~rm;
                        uint64_t cand;
                        if (!seen) { // (piece 0) the initial count of min(len, 5)
                            const int f = rm ? (int)__builtin_ctzll(rm) : 64;
                            if (f >= a0) { stop_lane = a0 - 1; cand = 0; }
                            else { seen = true; cand = nr & (nr << 1) & ~((1ull << (f + 2)) - 1ull); }
                        } else {
                            cand = nr & ((nr << 1) | (prev_nr ? 1ull : 0ull));
                        }
                        if (cand) stop_lane = (int)__builtin_ctzll(cand);
                        prev_nr = (nr >> 63) != 0;
                        visited = stop_lane < 64 ? ((2ull << stop_lane) - 1ull) : ~0ull;
                    } else { // BWA: `while (pos_3 > 0 && area >= 0)`: the steps before the first one that fails the test
                        const uint64_t go = __ballot(valid && t < len - 1 && a_before >= 0);
                        const int first_fail = ~go ? (int)__builtin_ctzll(~go) : 64;
                        visited = first_fail < 64 ? ((1ull << first_fail) - 1ull) : ~0ull;
                        if (first_fail < 64) stop_lane = first_fail; // (ends here; lane first_fail itself is not visited)
                    }
                    const bool mine = valid && ((visited >> lane) & 1ull);
                    const uint32_t key = mine ? (uint32_t)a_after ^ 0x80000000u : 0u; // signed order as unsigned
                    const uint32_t mx = wave_max_u32(key);
                    const int m = (int)(mx ^ 0x80000000u);
                    if (mx != 0u && m > best) { // strictly larger than every earlier piece's: the earliest step wins ties
                        best = m;
                        best_t = 64 * k + (int)__builtin_ctzll(__ballot(mine && key == mx));
                    }
                    if (stop_lane < 64) break;
                    carry += __builtin_amdgcn_readlane(incl, 63);
                }
                (void)lane_lt;
                return best_t;
            };
            int cut5 = 0, kept = len;
            if (P.mode == FAQCS_MODE_HARD) {
                // 3': the last window position >= 1 whose score exceeds Q (none: the window's end stays, and the 5' scan has no room)
                int pos_3 = 0, final_pos_3 = len - 1, final_pos_5 = 0;
#pragma unroll 1
                for (int k = 0; 64 * k < len; ++k) {
                    bool valid;
                    const int sc = piece(k, true, valid);
                    const uint64_t hit = __ballot(valid && 64 * k + lane < len - 1 && Q < sc); // (position 0 is not looked at, :640)
                    if (hit) { pos_3 = len - 1 - (64 * k + (int)__builtin_ctzll(hit)); final_pos_3 = pos_3; break; }
                }
                if (!P.protect5) { // 5': the first position below pos_3 whose score exceeds Q (:655-668)
#pragma unroll 1
                    for (int k = 0; 64 * k < pos_3; ++k) {
                        bool valid;
                        const int sc = piece(k, false, valid);
                        const uint64_t hit = __ballot(valid && 64 * k + lane < pos_3 && Q < sc);
                        if (hit) { final_pos_5 = 64 * k + (int)__builtin_ctzll(hit); break; }
                    }
                }
                kept = final_pos_3 - final_pos_5 + 1; cut5 = final_pos_5;
            } else if (P.mode == FAQCS_MODE_BWA) {
                const int bt = area_walk(true, false, 0);
                const int final_pos_3 = bt >= 0 ? (len - 1 - bt) - 1 : len - 1;
                kept = final_pos_3 + 1; cut5 = 0;
            } else {
                const int nan2 = len < 2 ? len : 2;
                // 3' walk: step t is at position len - 1 - t; it re-arms while that is > num_after_neg, i.e. t < len - 1 - num_after_neg
                const int bt3 = area_walk(true, true, len - 1 - nan2);
                const int final_pos_3 = bt3 >= 0 ? (len - 1 - bt3) - 1 : len - 1;
                int final_pos_5 = 0;
                if (!P.protect5) { // 5' walk: step t is at position t; it re-arms while t < final_pos_3 - num_after_neg
                    const int bt5 = area_walk(false, true, final_pos_3 - nan2);
                    final_pos_5 = bt5 >= 0 ? bt5 + 1 : 0;
                }
                kept = final_pos_3 <= final_pos_5 ? 0 : final_pos_3 - final_pos_5 + 1;
                cut5 = final_pos_5;
            }
            cut5 = uni(cut5); kept = uni(kept);
            const int init_len = len;
            offset_5 += (uint32_t)cut5; w0 += cut5; len = kept;
            if (init_len != len) { qt_bases = (uint32_t)(init_len - len); flags |= FAQCS_F_QUAL_TRIMMED; }
            if (len < (int)P.min_len || len == 0) { f_len_bases += (uint64_t)len; ++f_len_reads; ret = false; filt = FAQCS_FILT_LENGTH_POST; }
        }

        // ---- pass 2 over the window: poly-N (before G -> N, trim.cpp:363-371,578-597), the quality sum (:374), then with
        // --replace_to_N_q applied (:390-403) the base counts and the dinucleotide transitions of the low-complexity test (:405-513) ----
        int totalw = 0;
        // per-lane counters (16-bit fields: a lane sees <= 512 positions of a read), summed over the wave after the loop: as ballots + popcounts
        // into scalar counters these 17 counts were 34 scalar instructions per 64 positions of a kernel that is bound by scalar issue
        uint32_t wAT = 0, wCG = 0, wN = 0, dcw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dcw[k] = 0;
        uint32_t run_max = 0, run_carry = 0;
        uint32_t prev_cls = 4; // class of the position in front of the chunk (4 = none / not a base)
        if (!read_err) {
#pragma unroll 1
            for (int cc = 0; cc < len; cc += 256) { // the loads of four 64-position pieces are issued together; the pieces are then judged in order
              uint32_t b0v[4], rqv[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                  const int i = cc + 64 * u + lane;
                  b0v[u] = base_at(w0 + i);
                  rqv[u] = raw_q(w0 + i);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int c = cc + 64 * u;
                if (c >= len) break; // (wave-uniform)
                const int i = c + lane;
                const bool in = i < len;
                const uint32_t b0 = b0v[u];
                const uint32_t rq = rqv[u];
                totalw += in ? (int)(int8_t)rq : 0;
                // count_poly_n: the longest run of upper-case N
                const uint64_t mN = __ballot(in && b0 == 'N');
                const int wv = len - c < 64 ? len - c : 64; // valid width of this chunk
                if (mN) {
                    const uint32_t lead1 = ~mN ? (uint32_t)__builtin_ctzll(~mN) : 64u;
                    run_max = umax_(run_max, run_carry + lead1);
                    uint64_t t = mN; uint32_t inner = 0;
                    while (t) { t &= t << 1; ++inner; }
                    run_max = umax_(run_max, inner);
                    if ((int)lead1 >= wv) run_carry += (uint32_t)wv; // the whole chunk is N
                    else { const uint64_t top = mN << (64 - wv); run_carry = ~top ? (uint32_t)__builtin_clzll(~top) : 64u; }
                } else run_carry = 0;
                // the bases the rest of trim_read() sees
                const uint32_t b = (P.replace_q > 0 && b0 == 'G' && q_score(rq, in_off) < (int)P.replace_q) ? (uint32_t)'N' : b0;
                const uint32_t col = in ? base_col(b) : 5u;
                wAT += (col == 0u ? 1u : 0u) + (col == 1u ? 0x10000u : 0u);
                wCG += (col == 2u ? 1u : 0u) + (col == 3u ? 0x10000u : 0u);
                wN += col == 4u ? 1u : 0u;
                const uint32_t cls = col < 4u ? col : 4u; // A T C G / none (the codes differ from trim.cpp:447-470; only equality matters)
                uint32_t pv = (uint32_t)__shfl((int)cls, (lane + 63) & 63);
                pv = lane == 0 ? prev_cls : pv;
                prev_cls = (uint32_t)__builtin_amdgcn_readlane((int)cls, 63);
                const uint32_t pair = (in && cls != 4u && pv != 4u && cls != pv) ? (pv << 2 | cls) : 16u;
                const uint32_t inc = 1u << ((pair & 1u) * 16u), slot = pair >> 1; // (pair 16 = none: slot 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) dcw[k] += slot == (uint32_t)k ? inc : 0u;
              }
            }
        }
        totalw = wave_sum_i32(totalw);
        wAT = (uint32_t)wave_sum_i32((int)wAT); wCG = (uint32_t)wave_sum_i32((int)wCG); // (<= 32 767 per field: no carry between the fields)
        const uint32_t cA = wAT & 0xffffu, cT = wAT >> 16, cC = wCG & 0xffffu, cG = wCG >> 16, cN = (uint32_t)wave_sum_i32((int)wN);
        uint32_t dc[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const uint32_t v = (uint32_t)wave_sum_i32((int)dcw[k]); dc[2 * k] = v & 0xffffu; dc[2 * k + 1] = v >> 16; }
        uint64_t f_nn_reads = 0, f_nn_bases = 0, f_avg_reads = 0, f_avg_bases = 0, f_lc_reads = 0, f_lc_bases = 0;
        if (ret && !read_err && run_max >= P.max_poly_n) {
            f_nn_bases += (uint64_t)len; ++f_nn_reads; flags |= FAQCS_F_POLY_N_SEEN;
            if (!P.qc_only) { ret = false; filt = FAQCS_FILT_POLY_N; }
        }
        const float ave_Q = average_q(totalw, (uint32_t)len, in_off);
        if (ret && ave_Q < P.avg_q) { f_avg_bases += (uint64_t)len; ++f_avg_reads; ret = false; filt = FAQCS_FILT_AVG_Q; }
        if (ret && len != 0 && !read_err) {
            float norm = (float)(1.0 / (double)len); // trim.cpp:483
            const float lc = P.lc_ratio;
            bool trip = __fmul_rn((float)cA, norm) > lc || __fmul_rn((float)cT, norm) > lc || __fmul_rn((float)cG, norm) > lc || __fmul_rn((float)cC, norm) > lc;
            if (!trip) {
                norm = (float)((double)norm * 2.0); // trim.cpp:499
#pragma unroll
                for (int k = 0; k < 16; ++k) trip = trip || __fmul_rn((float)dc[k], norm) > lc;
            }
            if (trip) { f_lc_bases += (uint64_t)len; ++f_lc_reads; ret = false; filt = FAQCS_FILT_LOW_COMPLEXITY; }
        }
        // (pass 3, the per-position matrices: long_accumulate, from the verdict below and these two run lengths)
        // ---- per-read scalars: one lane ----
        if (lane == 0) {
            const uint32_t hit = P.has_adapters ? (uint32_t)ad_hit[r] : 0u;
            const bool bad_base = hit == 0xffffu;
            lead_trail[r] = (uint32_t)lead | ((uint32_t)trail << 16);
            out[r] = make_uint2(ret ? ((offset_5 & 0xffffu) | ((uint32_t)len << 16)) : 0u,
                                ((flags | (ret ? (uint32_t)FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT)) & 0x3ffu) |
                                    (bad_base ? (uint32_t)FAQCS_F_ERR_BASE : (hit << 16)));
        }
        // (wave-uniform values: every lane keeps the same sums, lane 0 adds them at the end)
        w_fs[FAQCS_TOTAL_COUNT] += 1; w_fs[FAQCS_TOTAL_NUMBER] += 1; w_fs[FAQCS_TOTAL_LENGTH] += (uint64_t)len0;
        w_fs[FAQCS_READ_LENGTH] += f_len_reads; w_fs[FAQCS_BASE_LENGTH] += f_len_bases;
        if (flags & FAQCS_F_QUAL_TRIMMED) { w_fs[FAQCS_READ_QUAL_TRIM] += 1; w_fs[FAQCS_BASE_QUAL_TRIM] += qt_bases; }
        w_fs[FAQCS_READ_NN] += f_nn_reads; w_fs[FAQCS_BASE_NN] += f_nn_bases;
        w_fs[FAQCS_READ_AVG_Q] += f_avg_reads; w_fs[FAQCS_BASE_AVG_Q] += f_avg_bases;
        w_fs[FAQCS_READ_LOW_COMPLEXITY] += f_lc_reads; w_fs[FAQCS_BASE_LOW_COMPLEXITY] += f_lc_bases;
        if (ret) { w_fs[FAQCS_TOTAL_TRIMMED_NUMBER] += 1; w_fs[FAQCS_TOTAL_TRIMMED_LENGTH] += (uint64_t)len; }
        if (!read_err) {
            if ((uint32_t)len0 != pre_key) {
                if (pre_run && lane == 0) add64(counters + L.pre_len_hist + pre_key, pre_run);
                pre_key = (uint32_t)len0; pre_run = 0;
            }
            ++pre_run;
        }
        if (ret) {
            if ((uint32_t)len != post_key) {
                if (post_run && lane == 0) add64(counters + L.post_len_hist + post_key, post_run);
                post_key = (uint32_t)len; post_run = 0;
            }
            ++post_run;
        }
        if (lane == 0) {
            if (!read_err) {
                const int qb = (int)average_q(total0, (uint32_t)len0, in_off);
                atomicAdd(&s_qh[0][qb], 1u); atomicAdd(&s_qh[1][qb], (uint32_t)len0);
                composition_bins(counters + L.pre_comp, (uint32_t)len0, pA, pT, pC, pG, pN);
            }
            if (ret) {
                const int qb = (int)ave_Q;
                atomicAdd(&s_qh[2][qb], 1u); atomicAdd(&s_qh[3][qb], (uint32_t)len);
                composition_bins(counters + L.post_comp, (uint32_t)len, cA, cT, cC, cG, cN);
            }
        }
    }
    if (lane == 0) {
        if (pre_run) add64(counters + L.pre_len_hist + pre_key, pre_run);
        if (post_run) add64(counters + L.post_len_hist + post_key, post_run);
#pragma unroll
        for (int k = 0; k < FAQCS_NUM_STAT; ++k) if (w_fs[k]) add64(fs + k, w_fs[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * FAQCS_NQ; i += NW * 64) {
        const uint32_t v = (&s_qh[0][0])[i];
        const int which = i / FAQCS_NQ, qb = i % FAQCS_NQ;
        if (v) add64(counters + (which == 0 ? L.pre_read_qhist : which == 1 ? L.pre_base_qhist : which == 2 ? L.post_read_qhist : L.post_base_qhist) + qb, v);
    }
    if (__any(any_err) && lane == 0) atomicOr(err, 1u);
}

// ---- long_accumulate: the per-position matrices of a trim_long batch (update_quality_matrix / update_base_statistics,
// trim.cpp:795-875), tile by tile.  A block walks the positions in tiles of 512: for one tile its LDS holds position x quality
// [42][512] and position x base [5][512] as dwords (pre count low, post count high half-word), its sixteen waves take the block's reads one
// each -- lane = position inside the tile, so the 64 adds of an instruction fall on 64 consecutive cells: no bank conflicts -- and
// at the end of the tile the non-zero cells go to the u64 counter block.  A base costs two LDS adds (quality, class) instead of four
// memory-side atomics; the global atomics left are one per non-zero cell, tile and block.
constexpr int LA_TILE = 512, LA_NW = 16, LA_UNR = LA_TILE / 64; // (16 waves per CU, a tile's 16 loads per read in flight: the loop waits on global loads)
__global__ __launch_bounds__(LA_NW * 64) void long_accumulate(const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                             const uint32_t *__restrict__ off, const uint32_t n_reads, const uint2 *__restrict__ out,
                                                             const uint32_t *__restrict__ lead_trail, uint64_t *__restrict__ counters, const uint32_t max_len)
{
    __shared__ uint32_t s_q[FAQCS_NQ][LA_TILE];
    __shared__ uint32_t s_b[FAQCS_NBASE][LA_TILE];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)uni((int)(threadIdx.x >> 6));
    const faqcs_layout &L = P.lay;
    const int in_off = P.in_off;
    const uint32_t limit = max_len < P.R ? max_len : P.R; // rows past the matrices' capacity are not counted (trim.cpp:797-805 grows them: R is that capacity)
    for (int i = threadIdx.x; i < FAQCS_NQ * LA_TILE; i += LA_NW * 64) (&s_q[0][0])[i] = 0u;
    for (int i = threadIdx.x; i < FAQCS_NBASE * LA_TILE; i += LA_NW * 64) (&s_b[0][0])[i] = 0u;
    __syncthreads();
    auto flush = [&](const uint32_t t0) {
        __syncthreads();
        for (int i = threadIdx.x; i < FAQCS_NQ * LA_TILE; i += LA_NW * 64) {
            const uint32_t v = (&s_q[0][0])[i];
            if (v) {
                (&s_q[0][0])[i] = 0u;
                const uint32_t q = (uint32_t)i / LA_TILE, p = t0 + (uint32_t)i % LA_TILE;
                if (v & 0xffffu) add64(counters + L.pre_qual + (size_t)p * FAQCS_NQ + q, v & 0xffffu);
                if (v >> 16) add64(counters + L.post_qual + (size_t)p * FAQCS_NQ + q, v >> 16);
            }
        }
        for (int i = threadIdx.x; i < FAQCS_NBASE * LA_TILE; i += LA_NW * 64) {
            const uint32_t v = (&s_b[0][0])[i];
            if (v) {
                (&s_b[0][0])[i] = 0u;
                const uint32_t c = (uint32_t)i / LA_TILE, p = t0 + (uint32_t)i % LA_TILE;
                if (v & 0xffffu) add64(counters + L.pre_base + (size_t)p * FAQCS_NBASE + c, v & 0xffffu);
                if (v >> 16) add64(counters + L.post_base + (size_t)p * FAQCS_NBASE + c, v >> 16);
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (uint32_t t0 = 0; t0 < limit; t0 += LA_TILE) {
        const uint32_t t1 = t0 + LA_TILE < limit ? t0 + LA_TILE : limit;
        uint32_t since = 0; // reads of this block since the tile's last flush: a 16-bit half takes 65 535 increments
#pragma unroll 1
        for (uint32_t k = 0;; ++k) { // (every wave runs the same number of rounds: the flush inside is a block barrier)
            const uint32_t r0 = blockIdx.x + k * LA_NW * gridDim.x;
            if (r0 >= n_reads) break; // (block-uniform: r0 belongs to wave 0)
            const uint32_t r = r0 + wave * gridDim.x;
            if (r < n_reads) {
                const uint32_t o = uniu(off[r]);
                const uint32_t len0 = uniu(off[r + 1]) - o;
                const uint2 res = out[r];
                if (len0 > t0 && !(uniu(res.y) & FAQCS_F_ERR_QUALITY)) {
                    const uint32_t lt = uniu(lead_trail[r]);
                    const uint32_t lead = lt & 0xffffu, tail_from = len0 - (lt >> 16);
                    const bool kept = (uniu(res.y) & FAQCS_F_VALID) != 0u;
                    const uint32_t k0 = kept ? (uniu(res.x) & 0xffffu) : 0u, k1 = kept ? k0 + (uniu(res.x) >> 16) : 0u;
                    const uint32_t e = len0 < t1 ? len0 : t1;
#pragma unroll 1
                    for (uint32_t pb = t0; pb < e; pb += 64u * LA_UNR) { // the whole tile of a read in one round: its 2 x LA_UNR loads are in flight together (the loop waits on memory)
                        uint32_t rq[LA_UNR], b0v[LA_UNR];
#pragma unroll
                        for (int u = 0; u < LA_UNR; ++u) {
                            const uint32_t p = pb + 64u * (uint32_t)u + (uint32_t)lane;
                            const uint32_t pc = p < e ? p : e - 1u; // (clamped: unconditional loads, no EXEC region per load)
                            const uint32_t qv = (uint32_t)qual[(size_t)o + pc];
                            rq[u] = (p < lead || p >= tail_from) ? (uint32_t)(in_off & 0xff) : qv;
                            b0v[u] = (uint32_t)seq[(size_t)o + pc];
                        }
#pragma unroll
                        for (int u = 0; u < LA_UNR; ++u) {
                            // (no EXEC regions: a lane past the read's end, or on a byte that is no base, adds 0 -- the kernel is bound by scalar issue)
                            const uint32_t p = pb + 64u * (uint32_t)u + (uint32_t)lane;
                            const bool ok = p < e;
                            int sc = q_score(rq[u], in_off);
                            sc = sc > 41 ? 41 : sc; // (cannot happen: a read with such a score carries FAQCS_F_ERR_QUALITY)
                            const uint32_t b0 = b0v[u];
                            const bool in = ok && p >= k0 && p < k1;
                            const uint32_t x = ok ? p - t0 : 0u;
                            atomicAdd(&s_q[sc][x], ok ? (in ? 0x10001u : 1u) : 0u);
                            const uint32_t col0 = base_col(b0);
                            const uint32_t b = (P.replace_q > 0 && b0 == 'G' && sc < (int)P.replace_q) ? (uint32_t)'N' : b0;
                            const uint32_t col = in ? base_col(b) : 5u;
                            atomicAdd(&s_b[col0 < 5u ? col0 : 0u][x], (ok && col0 < 5u) ? (col == col0 ? 0x10001u : 1u) : 0u);
                            const bool other = col < 5u && col != col0; // (a G that --replace_to_N_q turned into N inside the window)
                            if (__any(other)) { if (other) atomicAdd(&s_b[col][x], 0x10000u); }
                        }
                    }
                }
            }
            since += LA_NW;
            if (since + LA_NW > 65535u) { flush(t0); since = 0; }
        }
        flush(t0);
    }
}

hipError_t faqcs_launch_trim_long(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off, uint32_t n_reads,
                                  const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out, uint64_t *counters, uint32_t *err,
                                  int n_cu, hipStream_t st, uint32_t *lead_trail, uint32_t max_len)
{
    if (n_reads == 0) return hipSuccess;
    constexpr int NW = 4;
    uint32_t grid = (n_reads + NW - 1) / NW;
    const uint32_t cap = (uint32_t)n_cu * 8u; // 32 waves per CU: the passes wait on memory, not on issue slots
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((trim_long<NW>), dim3(grid), dim3(NW * 64), 0, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), counters, err, lead_trail);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    uint32_t g2 = (n_reads + LA_NW - 1) / LA_NW;
    if (g2 > (uint32_t)n_cu) g2 = (uint32_t)n_cu; // one block per CU (its 94 KB of LDS)
    hipLaunchKernelGGL(long_accumulate, dim3(g2), dim3(LA_NW * 64), 0, st, P, seq, qual, off, n_reads, reinterpret_cast<const uint2 *>(out), lead_trail,
                       counters, max_len);
    return hipGetLastError();
}
