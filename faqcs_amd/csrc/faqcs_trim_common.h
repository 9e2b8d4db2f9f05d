// faqcs_trim_common.h -- pieces shared by the trim kernels (faqcs_trim_kernel.hip, faqcs_trim_lds_kernel.hip):
// the LDS accumulator layout (RowCfg), byte / bit helpers, the block flush and the per-chunk epilogue.
#pragma once
#include "faqcs_dev.h"

#include <stdlib.h>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting of a kernel: a process with contexts on several devices
// (faqcs_mi --gpus N) has to make it on each of them once.  `done` = one bit per device ordinal, owned by the caller (one per kernel).
static inline hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes, unsigned long long &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
    return e;
}

namespace {

template <int C, int LPR, int WQ_ = 0> struct RowCfg {
    static constexpr int D = (C + 3) / 4;          // dwords per lane per arena
    static constexpr int W = LPR * C;              // positions covered by a row == columns of the LDS matrices
    static constexpr int WQ = WQ_ ? WQ_ : W;       // columns of the position x quality matrix (trim_lds: a multiple of 32, see its Q-B pass)
    // position x quality in LDS: one dword per cell (pre count lo16 / post count hi16) while that fits next to the
    // other tables (W <= 768); wider rows pack two cells per dword as 8-bit pre/post counters and the block flushes
    // them every 16 reads per wave (HQ8_EVERY x NW <= 255 increments per cell between flushes)
    static constexpr bool HQ8 = W > 768;           // (768 wide: 158 KB of the CU's 160 KB LDS, one block per CU)
    static constexpr int HQ8_EVERY = 16;
    static constexpr int HQ = HQ8 ? FAQCS_NQ * W / 2 : FAQCS_NQ * WQ;
    // |sum of (Q - q)| <= W * 168: key bias and the bit width of a position field inside the argmax keys
    static constexpr int KEY_BIAS = LPR <= 16 ? (1 << 16) : (1 << 18);
    static constexpr int PB = LPR <= 16 ? 9 : 11;
    static constexpr int FK = 2 * W;               // "first position" keys are FK - p (0 == none)
    static constexpr int HB = FAQCS_NBASE * W;
    static constexpr int O_HQ = 0;
    static constexpr int O_HB = O_HQ + HQ;
    static constexpr int O_LEN = O_HB + HB;        // [W+1] lo16 pre / hi16 post
    static constexpr int O_RQ = O_LEN + W + 2;     // [42]  lo16 pre / hi16 post
    static constexpr int O_BQPRE = O_RQ + 42;      // [42]
    static constexpr int O_BQPOST = O_BQPRE + 42;  // [42]
    static constexpr int O_FS = O_BQPOST + 42;     // [32]
    static constexpr int N_ZERO = O_FS + FS_SLOTS; // everything above is zero-initialised and flushed
    static constexpr int O_TBASE = N_ZERO;         // [256] base table
    static constexpr int O_TLC = O_TBASE + 256;    // [W+1]
    static constexpr int O_TAVGQ = O_TLC + W + 1;  // [W+1]
    static constexpr int O_TMAGIC = O_TAVGQ + W + 1;
    static constexpr int BMW = D <= 4 ? 4 : 8;     // dwords per byte-mask row (one or two ds_read_b128)
    static constexpr int O_TBM = (O_TMAGIC + W + 1 + 3) & ~3; // [C+2][BMW] byte masks "first vb bytes of the lane's dwords" (trim_lds: vb <= C + 1)
    static constexpr int LDS_DWORDS = O_TBM + BMW * (C + 2);
    static constexpr int JB = C > 16 ? 5 : 4;      // bits of a position index inside the lane-local argmax keys
};

template <int D> struct __attribute__((packed, aligned(1))) PackedBytes { uint32_t w[D]; };

__device__ __forceinline__ int med3i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// bits j in [0, C) with lo <= pbase + j < hi
template <int C> __device__ __forceinline__ uint32_t range_mask(int lo, int hi, int pbase)
{
    const int s = med3i(lo - pbase, 0, C);
    int e = med3i(hi - pbase, 0, C);
    e = e > s ? e : s;
    uint32_t m; // ((1 << (e - s)) - 1) << s in one instruction
    asm("v_bfm_b32 %0, %1, %2" : "=v"(m) : "v"(e - s), "v"(s));
    return m;
}
// 0 / -1 from bit j of mask.  Pinned to ONE v_bfe_i32: left to itself the compiler rewrites `x & -(bit)` into
// and + compare + select (3 instructions per use, ~40 uses per read in a VALU-bound kernel).
// 4 * byte K of w in ONE instruction (SDWA source select): the LDS byte offset of a base-table lookup
template <int K> __device__ __forceinline__ uint32_t byte_times4(uint32_t w, uint32_t two)
{
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    else if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
    else if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
    return r;
}
// t_base_lds = LDS byte address of the table (the kernel's dynamic LDS starts at address 0, checked at kernel start): the
// lookup is then ds_read_b32 v, <4 * byte> offset:<table> with no address add at all.
typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
typedef __attribute__((address_space(3))) uint32_t *lds_u32_mut;
// ds_add_u32 at an LDS BYTE offset computed in 32 bits (a generic pointer makes the compiler form the address with
// v_mad_u64_u32, a multi-pass instruction, once per base)
__device__ __forceinline__ void lds_add_u32(uint32_t byte_offset, uint32_t v)
{
    __hip_atomic_fetch_add((lds_u32_mut)(size_t)byte_offset, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int C, int J> struct BaseLookup {
    static __device__ __forceinline__ void run(const uint32_t t_base_lds, const uint32_t *ws, uint32_t two, uint32_t *inc)
    {
        inc[J] = *(lds_u32_ptr)(size_t)(byte_times4<J & 3>(ws[J >> 2], two) + t_base_lds);
        BaseLookup<C, J + 1>::run(t_base_lds, ws, two, inc);
    }
};
template <int C> struct BaseLookup<C, C> {
    static __device__ __forceinline__ void run(const uint32_t, const uint32_t *, uint32_t, uint32_t *) {}
};
__device__ __forceinline__ uint32_t umax3_(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int bit_m1(uint32_t mask, int j)
{
    int r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(mask), "n"(j));
    return r;
}

} // namespace

// LDS -> global u64 block.  Deliberately NOT inlined: the counter-block layout (a dozen 64-bit offsets) would
// otherwise stay live in SGPRs across the whole read loop and push the kernel into scalar-register spills.
// PAST_END: the read loop adds the "pre" 1 for every position slot of a counted read, also past its end (corrected here);
// false: the loop only counts positions inside the read (trim_lds)
template <int C, int LPR, int NW, bool PAST_END = true>
__device__ __noinline__ void flush_block(uint32_t *smem, uint64_t *counters, const uint32_t R, const int tid)
{
    using Cfg = RowCfg<C, LPR>;
    constexpr int W = Cfg::W;
    uint32_t *hq = smem + Cfg::O_HQ, *hb = smem + Cfg::O_HB, *hlen = smem + Cfg::O_LEN, *hrq = smem + Cfg::O_RQ;
    uint32_t *hbqpre = smem + Cfg::O_BQPRE, *hbqpost = smem + Cfg::O_BQPOST, *lfs = smem + Cfg::O_FS;
    // faqcs_counters_layout() restated (include/faqcs_mi.h): only R is needed for the offsets used here
    faqcs_layout L;
    {
        uint64_t o = 0;
        L.filter_stats = o;    o += 32;
        L.pre_read_qhist = o;  o += FAQCS_NQ;
        L.pre_base_qhist = o;  o += FAQCS_NQ;
        L.post_read_qhist = o; o += FAQCS_NQ;
        L.post_base_qhist = o; o += FAQCS_NQ;
        L.pre_len_hist = o;    o += (uint64_t)R + 1;
        L.post_len_hist = o;   o += (uint64_t)R + 1;
        L.pre_qual = o;        o += (uint64_t)R * FAQCS_NQ;
        L.post_qual = o;       o += (uint64_t)R * FAQCS_NQ;
        L.pre_base = o;        o += (uint64_t)R * FAQCS_NBASE;
        L.post_base = o;
    }
    __syncthreads();
    for (int i = tid; i < (Cfg::HQ8 ? 0 : Cfg::HQ); i += NW * 64) {
        const uint32_t v = hq[i];
        if (v) {
            hq[i] = 0;
            const uint32_t q = i / W, p = i % W;
            uint32_t pre = v & 0xffffu;
            if (PAST_END && q == 0) {
                // The read loop adds the "pre" 1 for EVERY position slot of a counted read, also past its end (where the
                // masked quality byte is 0): column 0 of position p is over-counted once per read with len <= p, and the
                // interval's length histogram says how many those are.
                uint32_t shorter = 0;
                for (uint32_t l = 0; l <= p; ++l) shorter += hlen[l] & 0xffffu;
                pre -= shorter;
            }
            if (p < R) {
                if (pre) atomicAdd((unsigned long long *)(counters + L.pre_qual + (uint64_t)p * FAQCS_NQ + q), (unsigned long long)pre);
                if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_qual + (uint64_t)p * FAQCS_NQ + q), (unsigned long long)(v >> 16));
            }
        }
    }
    __syncthreads(); // (the correction above reads hlen, which the loop below clears)
    for (int i = tid; i < Cfg::HB; i += NW * 64) {
        const uint32_t v = hb[i];
        if (v) {
            hb[i] = 0;
            const uint32_t c = i / W, p = i % W;
            if (p < R) {
                if (v & 0xffffu) atomicAdd((unsigned long long *)(counters + L.pre_base + (uint64_t)p * FAQCS_NBASE + c), (unsigned long long)(v & 0xffffu));
                if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_base + (uint64_t)p * FAQCS_NBASE + c), (unsigned long long)(v >> 16));
            }
        }
    }
    for (int i = tid; i <= W; i += NW * 64) {
        const uint32_t v = hlen[i];
        if (v && (uint32_t)i <= R) {
            hlen[i] = 0;
            if (v & 0xffffu) atomicAdd((unsigned long long *)(counters + L.pre_len_hist + i), (unsigned long long)(v & 0xffffu));
            if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_len_hist + i), (unsigned long long)(v >> 16));
        }
    }
    if (tid < FAQCS_NQ) {
        const uint32_t v = hrq[tid], x = hbqpre[tid], y = hbqpost[tid];
        hrq[tid] = 0; hbqpre[tid] = 0; hbqpost[tid] = 0;
        if (v & 0xffffu) atomicAdd((unsigned long long *)(counters + L.pre_read_qhist + tid), (unsigned long long)(v & 0xffffu));
        if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_read_qhist + tid), (unsigned long long)(v >> 16));
        if (x) atomicAdd((unsigned long long *)(counters + L.pre_base_qhist + tid), (unsigned long long)x);
        if (y) atomicAdd((unsigned long long *)(counters + L.post_base_qhist + tid), (unsigned long long)y);
    }
    if (tid >= 64 && tid < 64 + FAQCS_NUM_STAT) {
        const int k = tid - 64;
        const uint32_t v = lfs[k];
        if (v) { atomicAdd((unsigned long long *)(counters + L.filter_stats + k), (unsigned long long)v); lfs[k] = 0; }
    }
    __syncthreads();
}

// 8-bit packed position x quality cells (rows wider than 512) -> global u64 block; called by the whole block
template <int C, int LPR, int NW>
__device__ __noinline__ void flush_hq8(uint32_t *smem, uint64_t *counters, const uint32_t R, const int tid)
{
    using Cfg = RowCfg<C, LPR>;
    constexpr int HW = Cfg::W / 2;
    uint32_t *hq = smem + Cfg::O_HQ;
    const uint64_t pre_qual = 32 + 4 * FAQCS_NQ + 2 * ((uint64_t)R + 1), post_qual = pre_qual + (uint64_t)R * FAQCS_NQ;
    __syncthreads();
    for (int i = tid; i < Cfg::HQ; i += NW * 64) {
        const uint32_t v = hq[i];
        if (v) {
            hq[i] = 0;
            const uint32_t q = i / HW, p = 2u * (uint32_t)(i % HW);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t a = (v >> (16 * h)) & 0xffu, b = (v >> (16 * h + 8)) & 0xffu;
                if (p + h < R) {
                    if (a) atomicAdd((unsigned long long *)(counters + pre_qual + (uint64_t)(p + h) * FAQCS_NQ + q), (unsigned long long)a);
                    if (b) atomicAdd((unsigned long long *)(counters + post_qual + (uint64_t)(p + h) * FAQCS_NQ + q), (unsigned long long)b);
                }
            }
        }
    }
    __syncthreads();
}

// What depends on a read's scalars alone, done once per 64-read chunk with one read per lane: result word, composition
// records, length / average-quality histograms, FilterStat sums.
struct ReadOutcome {
    uint32_t an, fl, pAT, pCG, cAT, cCG, N; // start | kept << 16 ; flags ; base counts before / after (A|T<<16, C|G<<16, N pre | post << 16)
    int Vpre, Vpost;                       // sum(raw - offset) over the read / over the kept window
};
// FilterStat sums kept per lane over several chunks (trim_lds: folded into the block's cells every few chunks, next to the register
// spill, instead of seven wave reductions per chunk): read count << 20 | base count, at most 2^12 reads and 2^20 bases per wave
struct FsAcc { uint32_t tot = 0, trim = 0, len = 0, qt = 0; }; // (poly-N, low complexity, average quality: rare, added per chunk when they occur)
__device__ __forceinline__ void fs_acc_flush(FsAcc &f, const int lane, uint32_t *lfs)
{
    const uint32_t m = (1u << 20) - 1u;
    const uint32_t s_tot = (uint32_t)wave_sum_i32((int)f.tot), s_trim = (uint32_t)wave_sum_i32((int)f.trim), s_len = (uint32_t)wave_sum_i32((int)f.len);
    const uint32_t s_qt = (uint32_t)wave_sum_i32((int)f.qt);
    const uint32_t s_nn = 0, s_lc = 0, s_avg = 0;
    if (lane == 0) {
        if (s_tot) { atomicAdd(&lfs[FAQCS_TOTAL_COUNT], s_tot >> 20); atomicAdd(&lfs[FAQCS_TOTAL_NUMBER], s_tot >> 20); atomicAdd(&lfs[FAQCS_TOTAL_LENGTH], s_tot & m); }
        if (s_trim) { atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_NUMBER], s_trim >> 20); atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_LENGTH], s_trim & m); }
        if (s_len) { atomicAdd(&lfs[FAQCS_READ_LENGTH], s_len >> 20); atomicAdd(&lfs[FAQCS_BASE_LENGTH], s_len & m); }
        if (s_nn) { atomicAdd(&lfs[FAQCS_READ_NN], s_nn >> 20); atomicAdd(&lfs[FAQCS_BASE_NN], s_nn & m); }
        if (s_avg) { atomicAdd(&lfs[FAQCS_READ_AVG_Q], s_avg >> 20); atomicAdd(&lfs[FAQCS_BASE_AVG_Q], s_avg & m); }
        if (s_qt) { atomicAdd(&lfs[FAQCS_READ_QUAL_TRIM], s_qt >> 20); atomicAdd(&lfs[FAQCS_BASE_QUAL_TRIM], s_qt & m); }
        if (s_lc) { atomicAdd(&lfs[FAQCS_READ_LOW_COMPLEXITY], s_lc >> 20); atomicAdd(&lfs[FAQCS_BASE_LOW_COMPLEXITY], s_lc & m); }
    }
    f = FsAcc();
}

template <int LPR>
__device__ __forceinline__ void chunk_epilogue(const ReadOutcome &o, const bool mine, const uint32_t my, const uint32_t v_len,
                                               const uint32_t v_hit, const int lane, uint32_t *hlen, uint32_t *hrq, uint32_t *hbqpre,
                                               uint32_t *hbqpost, uint32_t *lfs, const uint32_t *t_magic, uint2 *__restrict__ out,
                                               unsigned long long *__restrict__ rec_pre, unsigned long long *__restrict__ rec_post,
                                               const bool o_avgq_on, const uint32_t o_dbg, FsAcc *defer = nullptr, const bool wide_rt = false)
{
        const bool e_ret = (o.fl & FAQCS_F_VALID) != 0, e_err = (o.fl & FAQCS_F_ERR_QUALITY) != 0;
        const uint32_t e_len = v_len, e_n = o.an >> 16, e_filt = (o.fl & FAQCS_F_FILTER_MASK) >> FAQCS_F_FILTER_SHIFT;
        // int(ave_Q) == max(0, floor(V / len)), V = sum(raw - offset); floor via mulhi with a host magic
        int qb_pre = 0, qb_post = 0;
        if (mine && e_len > 0 && o.Vpre > 0) qb_pre = e_len == 1 ? o.Vpre : (int)__umulhi((uint32_t)o.Vpre, t_magic[e_len]);
        if (e_ret && o.Vpost > 0) qb_post = e_n == 1 ? o.Vpost : (int)__umulhi((uint32_t)o.Vpost, t_magic[e_n]);
        qb_pre = qb_pre > 41 ? 41 : qb_pre;
        qb_post = qb_post > 41 ? 41 : qb_post;
        if (defer) {
            // (trim_lds) the same six histogram cells, with the lanes that share the first lane's (length, quality bin) pair folded
            // into ONE add per cell by that lane: 64 reads of one length are 64 adds to one LDS address otherwise, which the LDS
            // serialises (~110 clocks per instruction, measured; equal-length reads with a narrow quality spread are the common case)
            const bool act = mine && !e_err;
            if (act) {
                const uint32_t key = e_len | ((uint32_t)qb_pre << 16);
                const bool same = key == (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(same));
                if (same) {
                    if (lane == __ffsll((unsigned long long)__ballot(true)) - 1) {
                        atomicAdd(hlen + e_len, cnt);
                        atomicAdd(hrq + qb_pre, cnt);
                        if (e_len) atomicAdd(hbqpre + qb_pre, cnt * e_len);
                    }
                } else {
                    atomicAdd(hlen + e_len, 1u);
                    atomicAdd(hrq + qb_pre, 1u);
                    if (e_len) atomicAdd(hbqpre + qb_pre, e_len);
                }
            }
            if (act && e_ret) {
                const uint32_t key = e_n | ((uint32_t)qb_post << 16);
                const bool same = key == (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(same));
                if (same) {
                    if (lane == __ffsll((unsigned long long)__ballot(true)) - 1) {
                        atomicAdd(hlen + e_n, cnt << 16);
                        atomicAdd(hrq + qb_post, cnt << 16);
                        atomicAdd(hbqpost + qb_post, cnt * e_n);
                    }
                } else {
                    atomicAdd(hlen + e_n, 0x10000u);
                    atomicAdd(hrq + qb_post, 0x10000u);
                    atomicAdd(hbqpost + qb_post, e_n);
                }
            }
        } else if (!(o_dbg & 2u) && mine && !e_err) { // length and int(average quality) histograms (trim.cpp:254-258,539-543,877-885)
            atomicAdd(hlen + e_len, 1u);
            atomicAdd(hrq + qb_pre, 1u);
            if (e_len) atomicAdd(hbqpre + qb_pre, e_len);
            if (e_ret) {
                atomicAdd(hlen + e_n, 0x10000u);
                atomicAdd(hrq + qb_post, 0x10000u);
                atomicAdd(hbqpost + qb_post, e_n);
            }
        }
        if (mine) {
            const bool bad_base = v_hit == 0xffffu; // set by adapter_overlap
            out[my] = make_uint2(e_ret ? o.an : 0u, (o.fl & 0x3ffu) | (bad_base ? (uint32_t)FAQCS_F_ERR_BASE : (v_hit << 16)));
            const bool pre_on = !e_err, post_on = e_ret && !e_err;
            const unsigned long long pA = o.pAT & 0xffffu, pT = o.pAT >> 16, pC = o.pCG & 0xffffu, pG = o.pCG >> 16, pn = o.N & 0xffffu;
            const unsigned long long cA = o.cAT & 0xffffu, cT = o.cAT >> 16, cC = o.cCG & 0xffffu, cG = o.cCG >> 16, cn = o.N >> 16;
            if (LPR <= 16 && !wide_rt) { // (wide_rt, wave-uniform: a 16-lane kernel whose batch holds a read of more than 256 bases)
                rec_pre[my] = pre_on ? (CR_VALID | e_len | (pA << 9) | (pT << 18) | (pC << 27) | (pG << 36) | (pn << 45)) : 0ull;
                rec_post[my] = post_on ? (CR_VALID | e_n | (cA << 9) | (cT << 18) | (cC << 27) | (cG << 36) | (cn << 45)) : 0ull;
            } else { // reads past 511 bases do not fit 9-bit fields: 11-bit fields over two words
                reinterpret_cast<ulonglong2 *>(rec_pre)[my] =
                    pre_on ? make_ulonglong2(CR_VALID | e_len | (pA << 11) | (pT << 22) | (pC << 33), pG | (pn << 11)) : make_ulonglong2(0ull, 0ull);
                reinterpret_cast<ulonglong2 *>(rec_post)[my] =
                    post_on ? make_ulonglong2(CR_VALID | e_n | (cA << 11) | (cT << 22) | (cC << 33), cG | (cn << 11)) : make_ulonglong2(0ull, 0ull);
            }
        }
        // FilterStat (trim.cpp:238-240,317-323,325-387,505-513,527-531): a read count in the high and a base
        // count in the low 20 bits, summed over the 64 reads of the chunk (64 x 1024 bases < 2^20)
        const bool e_rlen = e_filt == FAQCS_FILT_LENGTH_PRE || e_filt == FAQCS_FILT_LENGTH_POST;
        const uint32_t one = 1u << 20;
        if (defer) { // (a compile-time choice: the pointer is a constant of the inlined call)
            defer->tot += mine ? one | e_len : 0u;
            defer->trim += e_ret ? one | e_n : 0u;
            defer->len += e_rlen ? one | e_n : 0u;
            defer->qt += (o.fl & FAQCS_F_QUAL_TRIMMED) ? one | (o.fl >> 20) : 0u;
            // the rare ones at once (a wave reduction only in a chunk that has such a read)
            const uint32_t m = one - 1u;
            const bool is_nn = (o.fl & FAQCS_F_POLY_N_SEEN) != 0, is_lc = e_filt == FAQCS_FILT_LOW_COMPLEXITY, is_avg = o_avgq_on && e_filt == FAQCS_FILT_AVG_Q;
            if (__any(is_nn)) { const uint32_t v = (uint32_t)wave_sum_i32((int)(is_nn ? one | e_n : 0u)); if (lane == 0) { atomicAdd(&lfs[FAQCS_READ_NN], v >> 20); atomicAdd(&lfs[FAQCS_BASE_NN], v & m); } }
            if (__any(is_lc)) { const uint32_t v = (uint32_t)wave_sum_i32((int)(is_lc ? one | e_n : 0u)); if (lane == 0) { atomicAdd(&lfs[FAQCS_READ_LOW_COMPLEXITY], v >> 20); atomicAdd(&lfs[FAQCS_BASE_LOW_COMPLEXITY], v & m); } }
            if (__any(is_avg)) { const uint32_t v = (uint32_t)wave_sum_i32((int)(is_avg ? one | e_n : 0u)); if (lane == 0) { atomicAdd(&lfs[FAQCS_READ_AVG_Q], v >> 20); atomicAdd(&lfs[FAQCS_BASE_AVG_Q], v & m); } }
            return;
        }
        const uint32_t s_tot = (uint32_t)wave_sum_i32((int)(mine ? one | e_len : 0u));
        const uint32_t s_trim = (uint32_t)wave_sum_i32((int)(e_ret ? one | e_n : 0u));
        const uint32_t s_len = (uint32_t)wave_sum_i32((int)(e_rlen ? one | e_n : 0u));
        const uint32_t s_nn = (uint32_t)wave_sum_i32((int)((o.fl & FAQCS_F_POLY_N_SEEN) ? one | e_n : 0u));
        const uint32_t s_qt = (uint32_t)wave_sum_i32((int)((o.fl & FAQCS_F_QUAL_TRIMMED) ? one | (o.fl >> 20) : 0u));
        const uint32_t s_lc = (uint32_t)wave_sum_i32((int)(e_filt == FAQCS_FILT_LOW_COMPLEXITY ? one | e_n : 0u));
        uint32_t s_avg = 0;
        if (o_avgq_on) s_avg = (uint32_t)wave_sum_i32((int)(e_filt == FAQCS_FILT_AVG_Q ? one | e_n : 0u));
        if (lane == 0) {
            const uint32_t m = one - 1u;
            if (s_tot) { atomicAdd(&lfs[FAQCS_TOTAL_COUNT], s_tot >> 20); atomicAdd(&lfs[FAQCS_TOTAL_NUMBER], s_tot >> 20); atomicAdd(&lfs[FAQCS_TOTAL_LENGTH], s_tot & m); }
            if (s_trim) { atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_NUMBER], s_trim >> 20); atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_LENGTH], s_trim & m); }
            if (s_len) { atomicAdd(&lfs[FAQCS_READ_LENGTH], s_len >> 20); atomicAdd(&lfs[FAQCS_BASE_LENGTH], s_len & m); }
            if (s_nn) { atomicAdd(&lfs[FAQCS_READ_NN], s_nn >> 20); atomicAdd(&lfs[FAQCS_BASE_NN], s_nn & m); }
            if (s_avg) { atomicAdd(&lfs[FAQCS_READ_AVG_Q], s_avg >> 20); atomicAdd(&lfs[FAQCS_BASE_AVG_Q], s_avg & m); }
            if (s_qt) { atomicAdd(&lfs[FAQCS_READ_QUAL_TRIM], s_qt >> 20); atomicAdd(&lfs[FAQCS_BASE_QUAL_TRIM], s_qt & m); }
            if (s_lc) { atomicAdd(&lfs[FAQCS_READ_LOW_COMPLEXITY], s_lc >> 20); atomicAdd(&lfs[FAQCS_BASE_LOW_COMPLEXITY], s_lc & m); }
        }
}

#ifdef __HIPCC__
// ---- composition records -> the 10 001 x 6 composition tables (trim.cpp:860-874), by a block that owns >= 122 KB of LDS at address `tab` ----------
// The arithmetic of composition_histogram (faqcs_trim_kernel.hip) with the records handed out dynamically: chunks of NT x 4 records from a
// global counter, five chunks per claim.  Used by the blocks of a trim_lds launch that have run out of reads (round 6): `which` = 0 / 1 takes
// the pre- / post-trim records first and the other array behind it, so that the two halves of the grid meet in the middle of the work.
// One-word records only (reads of up to 256 bases).  Every block returns when both counters are exhausted: when the LAST block of the launch
// returns, every chunk has been claimed, and the kernel does not end before its claimer is done -- the fold is complete behind the launch.
template <int NT>
__device__ __forceinline__ void comp_fold_tail(uint32_t *tab, const DevParams &P, const int tid, const uint32_t which)
{
    constexpr int NE = FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND, ND = (NE + 1) / 2, U = 4;
    constexpr uint32_t CHUNK = NT * U, PER_CLAIM = 5, FLUSH_CLAIMS = 65535u / (CHUNK * PER_CLAIM);
    static_assert(FLUSH_CLAIMS >= 1, "a claim must fit the 16-bit counters");
    float *normt = reinterpret_cast<float *>(tab + ND);
    uint32_t *s_next = tab + ND + 512;
    const uint32_t n = P.fold_n, n_chunks = (n + CHUNK - 1) / CHUNK;
    // nothing left in either array: no table to set up
    if (uniu(__hip_atomic_load(&P.fold_claim[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= n_chunks &&
        uniu(__hip_atomic_load(&P.fold_claim[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= n_chunks) return;
    for (int i = tid; i < ND; i += NT) tab[i] = 0;
    for (int i = tid; i < 512; i += NT) normt[i] = i <= FAQCS_TAB_LEN ? P.comp_norm[i] : 0.0f;
    __syncthreads();
#pragma unroll 1
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const uint32_t a = which ^ pass;
        const unsigned long long *__restrict__ rec = a ? P.fold_post : P.fold_pre;
        uint64_t *__restrict__ dst = a ? P.fold_dst_post : P.fold_dst_pre;
        uint32_t since_flush = 0;
        bool dirty = false;
#pragma unroll 1
        for (;;) {
            if (tid == 0) *s_next = atomicAdd(&P.fold_claim[a], PER_CLAIM);
            __syncthreads();
            const uint32_t c0 = *s_next;
            __syncthreads();
            if (c0 >= n_chunks) break;
            dirty = true;
#pragma unroll 1
            for (uint32_t c = c0; c < c0 + PER_CLAIM && c < n_chunks; ++c) {
                unsigned long long x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const uint32_t i = c * CHUNK + (uint32_t)(u * NT + tid); x[u] = i < n ? rec[i] : 0ull; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t nbin = 0xffffffffu;
                    if (x[u] & CR_VALID) {
                        const uint32_t len = (uint32_t)(x[u] & 511u);
                        uint32_t cnt[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) cnt[k] = (uint32_t)(x[u] >> (9 + 9 * k)) & 511u;
                        const float norm = normt[len];
                        uint32_t idx[6];
#pragma unroll
                        for (int k = 0; k < 5; ++k) idx[k] = (uint32_t)__fmul_rn(norm, (float)cnt[k]); // trim.cpp:862-872
                        idx[5] = idx[3] + idx[2];                                                      // :874 (G + C)
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            if (k == 4) continue;
                            const uint32_t e = idx[k] * FAQCS_NCOMP_KIND + k;
                            atomicAdd(&tab[e >> 1], 1u << (16 * (e & 1u)));
                        }
                        nbin = idx[4] * FAQCS_NCOMP_KIND + 4;
                    }
                    constexpr uint32_t bin0 = 4u; // (no N at all: nearly every read -- one add per wave, as in composition_histogram)
                    const unsigned long long zero = __ballot(nbin == bin0);
                    if (zero != 0ull && (tid & 63) == __builtin_ctzll(zero)) atomicAdd(&tab[bin0 >> 1], (uint32_t)__popcll(zero) << (16 * (bin0 & 1u)));
                    if (nbin != 0xffffffffu && nbin != bin0) atomicAdd(&tab[nbin >> 1], 1u << (16 * (nbin & 1u)));
                }
            }
            if (++since_flush == FLUSH_CLAIMS) {
                __syncthreads();
                for (int d = tid; d < ND; d += NT) {
                    const uint32_t v = tab[d];
                    if (v) {
                        tab[d] = 0;
                        if (v & 0xffffu) atomicAdd((unsigned long long *)(dst + 2 * d), (unsigned long long)(v & 0xffffu));
                        if (v >> 16) atomicAdd((unsigned long long *)(dst + 2 * d + 1), (unsigned long long)(v >> 16));
                    }
                }
                __syncthreads();
                since_flush = 0; dirty = false;
            }
        }
        if (dirty) { // (block-uniform)
            __syncthreads();
            for (int d = tid; d < ND; d += NT) {
                const uint32_t v = tab[d];
                if (v) {
                    tab[d] = 0;
                    if (v & 0xffffu) atomicAdd((unsigned long long *)(dst + 2 * d), (unsigned long long)(v & 0xffffu));
                    if (v >> 16) atomicAdd((unsigned long long *)(dst + 2 * d + 1), (unsigned long long)(v >> 16));
                }
            }
            __syncthreads();
        }
    }
}
#endif
