// faqcs_trim_long_kernel.hip -- trim_long: trim_read() (trim.cpp:225-551) for batches that hold a read of more than 1 024 bases
// (up to FAQCS_MAX_READ_LENGTH = 32 767: long-read platforms, assembled contigs run through FaQCs).  gfx950, wave64.
//
// ONE wavefront per read, lane = position modulo 64.  The chunked kernels keep a whole read (or 64 of them) in registers / LDS and
// fold the reference's float tests into per-length integer tables; neither scales to 32 767 positions, and reads of this kind are
// few and long, so this kernel trades their tricks for a plain structure:
//   * every pass streams the read from global memory 64 positions at a time (a read is 2 x <= 32 KB: it stays in the L2 between
//     the passes), per-read sums are ballots + popcounts into wave-uniform (scalar) counters;
//   * the quality trimmers (BWA_plus / BWA / HARD, trim.cpp:629-793) are WALKED exactly as the reference walks them, as
//     wave-uniform scalar code over a 64-score register chunk read with v_readlane: a walk ends two positions after the area
//     turns negative, i.e. after a handful of steps on a good read, and costs |tail| steps on a bad one;
//   * the float expressions of the average-quality and low-complexity tests and of the composition bins are evaluated as the
//     reference writes them (IEEE single / double operations, no contraction), not through tables;
//   * accumulators are global u64 atomics on the counter block (position x quality, position x base: one per base and table).
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
                                                    uint32_t *__restrict__ err)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)uni((int)(threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * NW;
    const faqcs_layout &L = P.lay;
    uint64_t *fs = counters + L.filter_stats;
    const int in_off = P.in_off;
    bool any_err = false;
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
            const uint64_t m = __ballot(p < len0 && s[p] != 'N');
            if (m) { lead = c + (int)__builtin_ctzll(m); break; }
        }
        if (lead < len0) {
#pragma unroll 1
            for (int c = 0; c < len0; c += 64) {
                const int p = len0 - 1 - c - lane; // descending
                const uint64_t m = __ballot(p >= 0 && s[p] != 'N');
                if (m) { trail = c + (int)__builtin_ctzll(m); break; }
            }
        }
        lead = uni(lead); trail = uni(trail);
        const int tail_from = len0 - trail; // positions >= tail_from are patched
        auto raw_q = [&](const int p) -> uint32_t { return (p < lead || p >= tail_from) ? (uint32_t)(in_off & 0xff) : (uint32_t)q[p]; };

        // ---- pass 1 over the whole read: range check, the sum of the raw bytes, base counts (trim.cpp:247-258) ----
        int total0 = 0;
        uint32_t pA = 0, pT = 0, pC = 0, pG = 0, pN = 0;
        bool bad_q = false;
#pragma unroll 1
        for (int c = 0; c < len0; c += 64) {
            const int p = c + lane;
            const bool in = p < len0;
            const uint32_t rq = in ? raw_q(p) : 0u;
            bad_q = bad_q || (in && q_score(rq, in_off) > 41);
            total0 += in ? (int)(int8_t)rq : 0;
            const uint32_t col = in ? base_col(s[p]) : 5u;
            pA += popc64(__ballot(col == 0u)); pT += popc64(__ballot(col == 1u)); pC += popc64(__ballot(col == 2u));
            pG += popc64(__ballot(col == 3u)); pN += popc64(__ballot(col == 4u));
        }
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

        // ---- quality trimming, walked as written (trim.cpp:629-793).  score(i) = quality of window position i, from a 64-score
        // register chunk (cq holds window positions [cbase, cbase + 64)) ----
        uint32_t qt_bases = 0;
        if (!P.qc_only && ret && !read_err) {
            int cbase = -64;
            int cq = 0;
            auto score = [&](const int i) -> int {
                const int b = i & ~63;
                if (b != cbase) {
                    cbase = b;
                    const int p = w0 + b + lane;
                    cq = (b + lane < len) ? q_score(raw_q(p), in_off) : 0;
                }
                return __builtin_amdgcn_readlane(cq, i & 63);
            };
            const int Q = P.Q;
            int cut5 = 0, kept = len;
            if (P.mode == FAQCS_MODE_HARD) {
                int pos_3 = len - 1, final_pos_5 = 0, final_pos_3 = pos_3;
                while (pos_3 > 0) { if (Q < score(pos_3)) { final_pos_3 = pos_3; break; } --pos_3; }
                if (!P.protect5) {
                    int pos_5 = final_pos_5;
                    while (pos_5 < pos_3) { if (Q < score(pos_5)) { final_pos_5 = pos_5; break; } ++pos_5; }
                }
                kept = final_pos_3 - final_pos_5 + 1; cut5 = final_pos_5;
            } else if (P.mode == FAQCS_MODE_BWA) {
                int pos_3 = len - 1, final_pos_3 = pos_3, area = 0, maxArea = 0;
                while (pos_3 > 0 && area >= 0) {
                    area += Q - score(pos_3);
                    if (area > maxArea) { maxArea = area; final_pos_3 = pos_3 - 1; }
                    --pos_3;
                }
                kept = final_pos_3 + 1; cut5 = 0;
            } else {
                int at_least_scan = len < 5 ? len : 5;
                const int num_after_neg = len < 2 ? len : 2;
                int pos_3 = len - 1, final_pos_5 = 0, final_pos_3 = pos_3, area = 0, maxArea = 0;
                while (at_least_scan) {
                    --at_least_scan;
                    if (pos_3 > num_after_neg && area >= 0) at_least_scan = num_after_neg;
                    area += Q - score(pos_3);
                    if (area > maxArea) { maxArea = area; final_pos_3 = pos_3 - 1; }
                    --pos_3;
                }
                if (!P.protect5) {
                    int pos_5 = 0;
                    maxArea = 0; area = 0;
                    at_least_scan = len < 5 ? len : 5;
                    while (at_least_scan) {
                        --at_least_scan;
                        if (pos_5 < final_pos_3 - num_after_neg && area >= 0) at_least_scan = num_after_neg;
                        area += Q - score(pos_5);
                        if (area > maxArea) { maxArea = area; final_pos_5 = pos_5 + 1; }
                        ++pos_5;
                    }
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
        uint32_t cA = 0, cT = 0, cC = 0, cG = 0, cN = 0, dc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) dc[k] = 0;
        uint32_t run_max = 0, run_carry = 0;
        uint32_t prev_cls = 4; // class of the position in front of the chunk (4 = none / not a base)
        if (!read_err) {
#pragma unroll 1
            for (int c = 0; c < len; c += 64) {
                const int i = c + lane, p = w0 + i;
                const bool in = i < len;
                const uint32_t b0 = in ? (uint32_t)s[p] : 0u;
                const uint32_t rq = in ? raw_q(p) : 0u;
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
                cA += popc64(__ballot(col == 0u)); cT += popc64(__ballot(col == 1u)); cC += popc64(__ballot(col == 2u));
                cG += popc64(__ballot(col == 3u)); cN += popc64(__ballot(col == 4u));
                const uint32_t cls = col < 4u ? col : 4u; // A T C G / none (the codes differ from trim.cpp:447-470; only equality matters)
                uint32_t pv = (uint32_t)__shfl((int)cls, (lane + 63) & 63);
                pv = lane == 0 ? prev_cls : pv;
                prev_cls = (uint32_t)__builtin_amdgcn_readlane((int)cls, 63);
                const uint32_t pair = (in && cls != 4u && pv != 4u && cls != pv) ? (pv << 2 | cls) : 16u;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if ((k >> 2) != (k & 3)) dc[k] += popc64(__ballot(pair == (uint32_t)k));
            }
        }
        totalw = wave_sum_i32(totalw);
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
        // ---- pass 3: the per-position matrices (trim.cpp:795-875; a read with a quality error counts nothing: the run ends there) ----
        if (!read_err) {
            uint64_t *pre_q = counters + L.pre_qual, *post_q = counters + L.post_qual, *pre_b = counters + L.pre_base, *post_b = counters + L.post_base;
            const int R = (int)P.R;
            const int k0 = ret ? w0 : 0, k1 = ret ? w0 + len : 0; // kept window in read coordinates
            // post-trim rows are i + offset_5 (trim.cpp:533-535); for a kept read offset_5 == w0, so row == read position
#pragma unroll 1
            for (int c = 0; c < len0; c += 64) {
                const int p = c + lane;
                if (p < len0 && p < R) {
                    const uint32_t rq = raw_q(p);
                    const int sc = q_score(rq, in_off);
                    const uint32_t b0 = (uint32_t)s[p];
                    add64(pre_q + (size_t)p * FAQCS_NQ + sc, 1);
                    const uint32_t col0 = base_col(b0);
                    if (col0 < 5u) add64(pre_b + (size_t)p * FAQCS_NBASE + col0, 1);
                    if (p >= k0 && p < k1) {
                        add64(post_q + (size_t)p * FAQCS_NQ + sc, 1);
                        const uint32_t b = (P.replace_q > 0 && b0 == 'G' && sc < (int)P.replace_q) ? (uint32_t)'N' : b0;
                        const uint32_t col = base_col(b);
                        if (col < 5u) add64(post_b + (size_t)p * FAQCS_NBASE + col, 1);
                    }
                }
            }
        }
        // ---- per-read scalars: one lane ----
        if (lane == 0) {
            const uint32_t hit = P.has_adapters ? (uint32_t)ad_hit[r] : 0u;
            const bool bad_base = hit == 0xffffu;
            out[r] = make_uint2(ret ? ((offset_5 & 0xffffu) | ((uint32_t)len << 16)) : 0u,
                                ((flags | (ret ? (uint32_t)FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT)) & 0x3ffu) |
                                    (bad_base ? (uint32_t)FAQCS_F_ERR_BASE : (hit << 16)));
            add64(fs + FAQCS_TOTAL_COUNT, 1); add64(fs + FAQCS_TOTAL_NUMBER, 1); add64(fs + FAQCS_TOTAL_LENGTH, (uint64_t)len0);
            if (!read_err) {
                add64(counters + L.pre_len_hist + len0, 1);
                const int qb = (int)average_q(total0, (uint32_t)len0, in_off);
                add64(counters + L.pre_read_qhist + qb, 1); add64(counters + L.pre_base_qhist + qb, (uint64_t)len0);
                composition_bins(counters + L.pre_comp, (uint32_t)len0, pA, pT, pC, pG, pN);
            }
            if (f_len_reads) { add64(fs + FAQCS_READ_LENGTH, f_len_reads); add64(fs + FAQCS_BASE_LENGTH, f_len_bases); }
            if (flags & FAQCS_F_QUAL_TRIMMED) { add64(fs + FAQCS_READ_QUAL_TRIM, 1); add64(fs + FAQCS_BASE_QUAL_TRIM, qt_bases); }
            if (f_nn_reads) { add64(fs + FAQCS_READ_NN, 1); add64(fs + FAQCS_BASE_NN, f_nn_bases); }
            if (f_avg_reads) { add64(fs + FAQCS_READ_AVG_Q, 1); add64(fs + FAQCS_BASE_AVG_Q, f_avg_bases); }
            if (f_lc_reads) { add64(fs + FAQCS_READ_LOW_COMPLEXITY, 1); add64(fs + FAQCS_BASE_LOW_COMPLEXITY, f_lc_bases); }
            if (ret) {
                add64(fs + FAQCS_TOTAL_TRIMMED_NUMBER, 1); add64(fs + FAQCS_TOTAL_TRIMMED_LENGTH, (uint64_t)len);
                add64(counters + L.post_len_hist + len, 1);
                const int qb = (int)ave_Q;
                add64(counters + L.post_read_qhist + qb, 1); add64(counters + L.post_base_qhist + qb, (uint64_t)len);
                composition_bins(counters + L.post_comp, (uint32_t)len, cA, cT, cC, cG, cN);
            }
        }
    }
    if (__any(any_err) && lane == 0) atomicOr(err, 1u);
}

hipError_t faqcs_launch_trim_long(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off, uint32_t n_reads,
                                  const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out, uint64_t *counters, uint32_t *err,
                                  int n_cu, hipStream_t st)
{
    if (n_reads == 0) return hipSuccess;
    constexpr int NW = 4;
    uint32_t grid = (n_reads + NW - 1) / NW;
    const uint32_t cap = (uint32_t)n_cu * 8u; // 32 waves per CU: the passes wait on memory, not on issue slots
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((trim_long<NW>), dim3(grid), dim3(NW * 64), 0, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), counters, err);
    return hipGetLastError();
}
