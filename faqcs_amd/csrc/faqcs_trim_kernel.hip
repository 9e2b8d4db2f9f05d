// faqcs_trim_kernel.hip -- trim_filter_accumulate: the fused per-read kernel (gfx950, wave64).
//
// Replaces trim_read() and its helpers (trim.cpp:225-551, :553-597, :629-885, :1191-1216) for every read
// of a batch.  One wavefront owns one read at a time; lane l owns the C consecutive positions
// [l*C, l*C+C) (C = ceil(max_len/64) <= 4), fetched with ONE unaligned dword load per arena.  Everything
// per-read is then either
//   * a per-lane byte op (C-way unrolled),
//   * a 64-bit ballot (v_cmp -> SGPR pair) + scalar bit logic / popcount, or
//   * one DPP prefix scan / max-reduce (6 v_*_dpp, no LDS).
// The BWA_plus sequential state machine (trim.cpp:714-793) is evaluated in closed form from ONE prefix
// sum of (Q - q[i]) -- see bwa_plus() below and SURVEY.md section 8 a-5.
//
// Accumulators: position x quality and position x base matrices live in LDS, pre- and post-trim counts
// packed in the low / high 16 bits of one dword so a position costs ONE ds_add for both (the post-trim
// quality of a kept base equals its pre-trim quality: trim.cpp:516-533).  Column = j*64 + lane, so the 32
// lanes of a half-wave always hit 32 different banks.  Sparse accumulators (composition bins, length and
// average-quality histograms) go through a small LDS hash table.  A block flushes to the global u64 block
// with atomics before any 16-bit field can overflow (every <= 65535 reads per block).
//
// Float semantics of the reference are folded into integer lookup tables built on the host
// (DevParams); the only float op left is the composition-bin multiply, an exact IEEE v_mul_f32.
#include "faqcs_dev.h"

namespace {

constexpr int KEY_BIAS = 1 << 16; // |sum of (Q - q)| <= 256 * 168 < 2^16 for C <= 4

template <int C> struct TrimCfg {
    static constexpr int W = 64 * C;          // columns of the LDS matrices
    static constexpr int HQ = FAQCS_NQ * W;   // dwords
    static constexpr int HB = FAQCS_NBASE * W;
};

// position p+m seen from slot j of the same lane: returns the ballot word whose bit l answers "mask at
// position (l*C+j)+m".  m in {1,2} (next) ; implemented per call site with compile-time j.
template <int C, int M> __device__ __forceinline__ uint64_t next_mask(const uint64_t (&R)[C], int j)
{
    const int jj = (j + M) % C, sh = (j + M) / C;
    return R[jj] >> sh;
}
template <int C, int M> __device__ __forceinline__ uint64_t prev_mask(const uint64_t (&R)[C], int j)
{
    // position p-M: slot (j-M) mod C, lane shift = ceil((M-j)/C) when j < M
    const int t = j - M;
    const int jj = ((t % C) + C) % C;
    const int sh = t >= 0 ? 0 : (-t + C - 1) / C;
    return R[jj] << sh;
}

template <int C> __device__ __forceinline__ int hi_pos(const uint64_t (&R)[C])
{ // highest position with a set bit, -1 if none
    int best = -1;
#pragma unroll
    for (int j = 0; j < C; ++j)
        if (R[j]) { const int p = (63 - __builtin_clzll(R[j])) * C + j; best = p > best ? p : best; }
    return best;
}
template <int C> __device__ __forceinline__ int lo_pos(const uint64_t (&R)[C])
{ // lowest position with a set bit, INT_MAX if none
    int best = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < C; ++j)
        if (R[j]) { const int p = __builtin_ctzll(R[j]) * C + j; best = p < best ? p : best; }
    return best;
}
template <int C> __device__ __forceinline__ bool any_mask(const uint64_t (&R)[C])
{
    uint64_t o = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) o |= R[j];
    return o != 0;
}
template <int C> __device__ __forceinline__ int pop_mask(const uint64_t (&R)[C])
{
    int c = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) c += __popcll(R[j]);
    return c;
}

__device__ __forceinline__ void hash_add(uint32_t *hkey, uint32_t *hval, int hsize_mask, uint32_t key, uint32_t val,
                                         const DevParams &P, uint64_t *counters);

__device__ __forceinline__ uint64_t *hs_dest(const DevParams &P, uint64_t *ctr, uint32_t slot, uint32_t bin)
{
    const faqcs_layout &L = P.lay;
    if (slot < 6) return ctr + L.pre_comp + (uint64_t)bin * FAQCS_NCOMP_KIND + slot;
    if (slot < 12) return ctr + L.post_comp + (uint64_t)bin * FAQCS_NCOMP_KIND + (slot - 6);
    switch (slot) {
    case HS_PRE_LEN: return ctr + L.pre_len_hist + bin;
    case HS_POST_LEN: return ctr + L.post_len_hist + bin;
    case HS_PRE_RQ: return ctr + L.pre_read_qhist + bin;
    case HS_POST_RQ: return ctr + L.post_read_qhist + bin;
    case HS_PRE_BQ: return ctr + L.pre_base_qhist + bin;
    default: return ctr + L.post_base_qhist + bin;
    }
}

// insert (key -> += val) into the block's LDS hash table; falls through to a global atomic when the
// probe sequence is exhausted (table full of other keys)
__device__ __forceinline__ void hash_add(uint32_t *hkey, uint32_t *hval, int hmask, uint32_t key, uint32_t val,
                                         const DevParams &P, uint64_t *counters)
{
    uint32_t h = (key * 2654435761u) >> 16;
    bool done = false;
#pragma unroll 1
    for (int probe = 0; probe < 8 && !done; ++probe) {
        const uint32_t s = (h + probe) & hmask;
        const uint32_t old = atomicCAS(&hkey[s], HS_EMPTY, key);
        if (old == HS_EMPTY || old == key) { atomicAdd(&hval[s], val); done = true; }
    }
    if (!done) atomicAdd((unsigned long long *)hs_dest(P, counters, key & 31u, key >> 5), (unsigned long long)val);
}

} // namespace

template <int C, int NW, int HSIZE>
__global__ __launch_bounds__(NW * 64) void trim_filter_accumulate(
    const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
    const uint32_t *__restrict__ off, const uint32_t n_reads, const uint32_t *__restrict__ ad_sl,
    const uint16_t *__restrict__ ad_hit, uint2 *__restrict__ out, uint64_t *__restrict__ counters,
    uint32_t *__restrict__ err)
{
    using Cfg = TrimCfg<C>;
    constexpr int W = Cfg::W;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hq = smem;                 // [42][W]  lo16 = pre, hi16 = post
    uint32_t *hb = hq + Cfg::HQ;         // [5][W]
    uint32_t *hkey = hb + Cfg::HB;       // [HSIZE]
    uint32_t *hval = hkey + HSIZE;       // [HSIZE]
    uint32_t *lfs = hval + HSIZE;        // [FS_SLOTS]
    constexpr int LDS_DWORDS = Cfg::HQ + Cfg::HB + 2 * HSIZE + FS_SLOTS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uni(tid >> 6);

    for (int i = tid; i < LDS_DWORDS; i += NW * 64) smem[i] = (i >= Cfg::HQ + Cfg::HB && i < Cfg::HQ + Cfg::HB + HSIZE) ? HS_EMPTY : 0u;
    __syncthreads();

    int pos[C];
#pragma unroll
    for (int j = 0; j < C; ++j) pos[j] = lane * C + j;

    const uint32_t total_chunks = (n_reads + 63) >> 6;
    const uint32_t chunks_per_iter = gridDim.x * NW;
    const uint32_t n_iter = (total_chunks + chunks_per_iter - 1) / chunks_per_iter;
    constexpr uint32_t FLUSH_EVERY = 65535u / (NW * 64) > 0 ? 65535u / (NW * 64) : 1;

    const int in_off = P.in_off, Q = P.Q;

#pragma unroll 1
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t chunk = (it * gridDim.x + blockIdx.x) * NW + wave;
        if (chunk < total_chunks) {
            const uint32_t base = chunk << 6;
            const uint32_t cnt = uniu(n_reads - base < 64u ? n_reads - base : 64u);
            const uint32_t my = base + lane < n_reads ? base + lane : n_reads;
            const uint32_t v_off = off[my];
            const uint32_t v_len = off[my < n_reads ? my + 1 : n_reads] - v_off;
            const uint32_t v_sl = (ad_sl && base + lane < n_reads) ? ad_sl[base + lane] : 0u;
            const uint32_t v_hit = (ad_hit && base + lane < n_reads) ? ad_hit[base + lane] : 0u;
            uint32_t res_lo = 0, res_hi = 0;

            // per-wave FilterStat accumulators (wave-uniform -> SGPRs)
            uint32_t fs_total_len = 0, fs_trim_num = 0, fs_trim_len = 0, fs_rlen = 0, fs_blen = 0, fs_rnn = 0, fs_bnn = 0,
                     fs_ravg = 0, fs_bavg = 0, fs_rqt = 0, fs_bqt = 0, fs_rlc = 0, fs_blc = 0;
            uint32_t any_err = 0;

            // software prefetch of read 0
            // (a lane only loads when its first position is inside the read: the over-read is <= 3 bytes)
            uint32_t o_cur = uniu(__builtin_amdgcn_readlane((int)v_off, 0));
            const int len0 = (int)uniu((uint32_t)__builtin_amdgcn_readlane((int)v_len, 0));
            typedef uint32_t __attribute__((aligned(1))) u32u;
            uint32_t wseq = 0, wqual = 0;
            if (lane * C < len0) {
                wseq = *(const u32u *)(seq + (size_t)o_cur + lane * C);
                wqual = *(const u32u *)(qual + (size_t)o_cur + lane * C);
            }

#pragma unroll 1
            for (uint32_t r = 0; r < cnt; ++r) {
                const int len = (int)uniu((uint32_t)__builtin_amdgcn_readlane((int)v_len, r));
                const uint32_t cseq = wseq, cqual = wqual;
                if (r + 1 < cnt) {
                    const uint32_t o_n = uniu((uint32_t)__builtin_amdgcn_readlane((int)v_off, r + 1));
                    const int len_n = (int)uniu((uint32_t)__builtin_amdgcn_readlane((int)v_len, r + 1));
                    wseq = 0; wqual = 0;
                    if (lane * C < len_n) {
                        wseq = *(const u32u *)(seq + (size_t)o_n + lane * C);
                        wqual = *(const u32u *)(qual + (size_t)o_n + lane * C);
                    }
                }

                // ---- unpack this lane's C bases / quality bytes ------------------------------------
                uint32_t b[C];
                int rq[C], qs[C];
                bool inr[C];
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    inr[j] = pos[j] < len;
                    b[j] = inr[j] ? ((cseq >> (8 * j)) & 0xffu) : 0u;
                    rq[j] = inr[j] ? (int)(int8_t)((cqual >> (8 * j)) & 0xffu) : in_off;
                }

                // ---- mask_quality_terminal_N (trim.cpp:1191-1216): upper-case 'N' runs at either end ---
                uint64_t NU[C];
                bool term = false;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    NU[j] = __ballot(b[j] == 'N');
                    term |= (b[j] == 'N') && (pos[j] == 0 || pos[j] == len - 1);
                }
                if (__any(term)) {
                    uint64_t NON[C];
#pragma unroll
                    for (int j = 0; j < C; ++j) NON[j] = __ballot(inr[j] && b[j] != 'N');
                    const int fn = lo_pos<C>(NON), ln = hi_pos<C>(NON);
                    const int lead = fn == 0x7fffffff ? len : fn;
                    const int trail_start = ln + 1;
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (inr[j] && (pos[j] < lead || pos[j] >= trail_start)) rq[j] = in_off;
                }

                // ---- quality_score (fastq.h:17-36) -----------------------------------------------------
                bool bad = false;
                int sum_lane = 0;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int v = rq[j] - in_off;
                    qs[j] = v < 0 ? 0 : v;
                    bad |= qs[j] > 41;
                    qs[j] = qs[j] > 41 ? 41 : qs[j];
                    sum_lane += inr[j] ? rq[j] + 128 : 0;
                }
                const bool read_err = __any(bad);
                const uint32_t S_pre = (uint32_t)wave_sum_i32(sum_lane);

                // ---- base classes (case-insensitive A,T,C,G,N -> 0..4, else 5) ---------------------------
                // perfect hash on the lower-cased byte: h = (c>>1)&7 : a->0 c->1 t->2 g->3 n->7
                uint32_t code[C];
                uint64_t MA[C], MT[C], MC[C], MG[C], MN[C];
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const uint32_t lc = b[j] | 0x20u;
                    const uint32_t h = (lc >> 1) & 7u;
                    const uint32_t want = (uint32_t)((0x6e00000067746361ull >> (8 * h)) & 0xffu);
                    const uint32_t cd = (0x40003120u >> (4 * h)) & 7u; // a->0 c->2 t->1 g->3 n->4
                    code[j] = (want == lc) ? cd : 5u;
                    MA[j] = __ballot(code[j] == 0u);
                    MT[j] = __ballot(code[j] == 1u);
                    MC[j] = __ballot(code[j] == 2u);
                    MG[j] = __ballot(code[j] == 3u);
                    MN[j] = __ballot(code[j] == 4u);
                }
                const uint32_t pA = pop_mask<C>(MA), pT = pop_mask<C>(MT), pC = pop_mask<C>(MC), pG = pop_mask<C>(MG),
                               pN = pop_mask<C>(MN);

                // ---- window after the adapter pre-pass and --5end/--3end (trim.cpp:270-314) --------------
                int a = 0, n = len;
                uint32_t flags = 0, filt = 0;
                bool ret = true;
                if (P.has_adapters) {
                    const uint32_t sl = uniu((uint32_t)__builtin_amdgcn_readlane((int)v_sl, r));
                    const int first = (int)(sl & 0xffffu), second = (int)(sl >> 16);
                    if (len != second) { a = first; n = second; flags |= FAQCS_F_ADAPTER; }
                }
                if (P.trim5 && !P.qc_only) {
                    if ((int)P.trim5 > n) n = 0; else { a += (int)P.trim5; n -= (int)P.trim5; }
                }
                if (P.trim3 && !P.qc_only) {
                    if ((int)P.trim3 > n) n = 0; else n -= (int)P.trim3;
                }
                if (n < (int)P.min_len || n == 0) { fs_blen += n; ++fs_rlen; ret = false; filt = FAQCS_FILT_LENGTH_PRE; }

                // ---- quality trim (trim.cpp:325-360) -----------------------------------------------------
                if (!P.qc_only && ret) {
                    bool inw[C];
                    int d[C], Pin[C], Pex[C];
                    int run = 0;
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        inw[j] = (unsigned)(pos[j] - a) < (unsigned)n;
                        d[j] = inw[j] ? Q - qs[j] : 0;
                        run += d[j];
                        Pin[j] = run;
                    }
                    const int incl = wave_incl_scan_add(run);
                    const int T = __builtin_amdgcn_readlane(incl, 63);
                    const int E = incl - run;
#pragma unroll
                    for (int j = 0; j < C; ++j) { Pin[j] += E; Pex[j] = Pin[j] - d[j]; }

                    int fp3, fp5 = 0; // window-local indices
                    if (P.mode == FAQCS_MODE_BWA_PLUS) {
                        const int a5 = n < 5 ? n : 5, nan2 = n < 2 ? n : 2;
                        // 3' pass: reset[p] = (i > nan) && area_before >= 0, area_before(i) = S[i+1] = T - Pin[i]
                        uint64_t R3[C], F5[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            R3[j] = __ballot(inw[j] && (pos[j] - a > nan2) && (T - Pin[j] >= 0));
                            F5[j] = R3[j] & __ballot(pos[j] >= a + n - a5);
                        }
                        int pstar;
                        if (any_mask<C>(F5)) {
                            uint64_t C3[C];
#pragma unroll
                            for (int j = 0; j < C; ++j) C3[j] = ~R3[j] & ~next_mask<C, 1>(R3, j) & next_mask<C, 2>(R3, j);
                            pstar = hi_pos<C>(C3);
                        } else {
                            pstar = a + n - a5;
                        }
                        uint32_t key = 0;
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const uint32_t k = ((uint32_t)(T - Pex[j] + KEY_BIAS) << 9) | (uint32_t)(pos[j] - a);
                            key = umax_(key, (inw[j] && pos[j] >= pstar) ? k : 0u);
                        }
                        const uint32_t K3 = wave_max_u32(key);
                        fp3 = ((int)(K3 >> 9) - KEY_BIAS > 0) ? (int)(K3 & 511u) - 1 : n - 1;
                        if (!P.protect5) {
                            const int lim = fp3 - nan2;
                            uint64_t R5[C], G5[C];
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                R5[j] = __ballot(inw[j] && (pos[j] - a < lim) && (Pex[j] >= 0));
                                G5[j] = R5[j] & __ballot(pos[j] - a < a5);
                            }
                            int pstar5;
                            if (any_mask<C>(G5)) {
                                uint64_t C5[C];
#pragma unroll
                                for (int j = 0; j < C; ++j) C5[j] = ~R5[j] & ~prev_mask<C, 1>(R5, j) & prev_mask<C, 2>(R5, j);
                                pstar5 = lo_pos<C>(C5);
                            } else {
                                pstar5 = a + a5 - 1;
                            }
                            uint32_t key5 = 0;
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const uint32_t k = ((uint32_t)(Pin[j] + KEY_BIAS) << 9) | (uint32_t)(511 - (pos[j] - a));
                                key5 = umax_(key5, (inw[j] && pos[j] <= pstar5) ? k : 0u);
                            }
                            const uint32_t K5 = wave_max_u32(key5);
                            fp5 = ((int)(K5 >> 9) - KEY_BIAS > 0) ? 511 - (int)(K5 & 511u) + 1 : 0;
                        }
                    } else if (P.mode == FAQCS_MODE_BWA) { // trim.cpp:675-709
                        uint64_t NEG[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) NEG[j] = __ballot(inw[j] && (T - Pin[j] < 0));
                        const int pf = hi_pos<C>(NEG); // absolute position of the first failing step, or -1
                        const int lo = (pf < a ? a : pf) + 1;
                        uint32_t key = 0;
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const uint32_t k = ((uint32_t)(T - Pex[j] + KEY_BIAS) << 9) | (uint32_t)(pos[j] - a);
                            key = umax_(key, (inw[j] && pos[j] >= lo) ? k : 0u);
                        }
                        const uint32_t K3 = wave_max_u32(key);
                        fp3 = ((int)(K3 >> 9) - KEY_BIAS > 0) ? (int)(K3 & 511u) - 1 : n - 1;
                    } else { // HARD, trim.cpp:629-672
                        uint64_t H1[C], H0[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            H0[j] = __ballot(inw[j] && Q < qs[j]);
                            H1[j] = H0[j] & __ballot(pos[j] - a >= 1);
                        }
                        const int h = hi_pos<C>(H1);
                        int pos3 = 0;
                        fp3 = n - 1;
                        if (h >= 0) { fp3 = h - a; pos3 = fp3; }
                        if (!P.protect5) {
                            const int l = lo_pos<C>(H0);
                            if (l != 0x7fffffff && l - a < pos3) fp5 = l - a;
                        }
                    }
                    int kept = (P.mode == FAQCS_MODE_BWA_PLUS && fp3 <= fp5) ? 0 : fp3 - fp5 + 1;
                    if (kept != n) { fs_bqt += (uint32_t)(n - kept); ++fs_rqt; flags |= FAQCS_F_QUAL_TRIMMED; }
                    a += fp5;
                    n = kept;
                    if (n < (int)P.min_len || n == 0) { fs_blen += n; ++fs_rlen; ret = false; filt = FAQCS_FILT_LENGTH_POST; }
                }

                // ---- final window predicate --------------------------------------------------------------
                bool inw2[C];
                uint64_t W2[C];
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    inw2[j] = (unsigned)(pos[j] - a) < (unsigned)n;
                    W2[j] = __ballot(inw2[j]);
                }
                const bool whole = (a == 0 && n == len);

                // ---- poly-N filter (trim.cpp:363-371, :578-597) -- upper-case 'N' only --------------------
                if (ret) {
                    uint64_t NW2[C];
#pragma unroll
                    for (int j = 0; j < C; ++j) NW2[j] = NU[j] & W2[j];
                    bool trip;
                    const uint32_t K = P.max_poly_n;
                    if (K == 0) trip = true;
                    else if (!any_mask<C>(NW2)) trip = false;
                    else if (K == 1) trip = true;
                    else {
                        uint64_t X[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) X[j] = NW2[j] & next_mask<C, 1>(NW2, j);
                        const bool any2 = any_mask<C>(X);
                        if (K == 2) trip = any2;
                        else if (!any2) trip = false;
                        else { // exact longest run: run ending at p = p - (last non-N position <= p)
                            uint32_t loc[C], m = 0;
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const bool isn = inw2[j] && b[j] == 'N';
                                m = umax_(m, (inw2[j] && !isn) ? (uint32_t)pos[j] + 1u : 0u);
                                loc[j] = m;
                            }
                            uint32_t s = m;
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false));
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false));
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false));
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xf, false));
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false));
                            s = umax_(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false));
                            const uint32_t excl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x138, 0xf, 0xf, false); // wave_shr:1
                            uint32_t best = 0;
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const bool isn = inw2[j] && b[j] == 'N';
                                const uint32_t lastnon = umax_(umax_(excl, loc[j]), (uint32_t)a);
                                best = umax_(best, isn ? (uint32_t)pos[j] + 1u - lastnon : 0u);
                            }
                            trip = wave_max_u32(best) >= K;
                        }
                    }
                    if (trip) {
                        fs_bnn += n; ++fs_rnn; flags |= FAQCS_F_POLY_N_SEEN;
                        if (!P.qc_only) { ret = false; filt = FAQCS_FILT_POLY_N; }
                    }
                }

                // ---- average quality (trim.cpp:374-382, :553-576) -----------------------------------------
                uint32_t S_post = S_pre;
                if (ret && !whole) {
                    int sl = 0;
#pragma unroll
                    for (int j = 0; j < C; ++j) sl += inw2[j] ? rq[j] + 128 : 0;
                    S_post = (uint32_t)wave_sum_i32(sl);
                }
                if (ret && P.avgq_on && S_post < P.avgq_min_sum[n]) {
                    fs_bavg += n; ++fs_ravg; ret = false; filt = FAQCS_FILT_AVG_Q;
                }

                // ---- G -> N (trim.cpp:390-403) and the low-complexity filter (trim.cpp:405-513) -----------
                uint64_t REP[C];
#pragma unroll
                for (int j = 0; j < C; ++j) REP[j] = 0;
                uint32_t cA = pA, cT = pT, cC = pC, cG = pG, cN = pN;
                if (ret) {
                    if (P.replace_q > 0) {
#pragma unroll
                        for (int j = 0; j < C; ++j) REP[j] = __ballot(inw2[j] && b[j] == 'G' && qs[j] < (int)P.replace_q);
                    }
                    uint64_t XA[C], XT[C], XC[C], XG[C];
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        XA[j] = MA[j] & W2[j]; XT[j] = MT[j] & W2[j]; XC[j] = MC[j] & W2[j];
                        XG[j] = MG[j] & W2[j] & ~REP[j];
                    }
                    if (!whole || P.replace_q > 0) {
                        cA = pop_mask<C>(XA); cT = pop_mask<C>(XT); cC = pop_mask<C>(XC); cG = pop_mask<C>(XG);
                        uint64_t XN[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) XN[j] = (MN[j] & W2[j]) | REP[j];
                        cN = pop_mask<C>(XN);
                    }
                    const uint32_t mthr = P.mono_thr[n];
                    bool trip = cA >= mthr || cT >= mthr || cG >= mthr || cC >= mthr;
                    if (!trip) {
                        const uint32_t dthr = P.di_thr[n];
                        // dc[X->Y] <= min(count X, count Y): only pairs whose two counts both reach dthr can trip
                        const uint32_t cc[4] = {cA, cT, cC, cG};
                        const uint32_t nbig = (cA >= dthr) + (cT >= dthr) + (cC >= dthr) + (cG >= dthr);
                        if (nbig >= 2) {
#define FAQCS_PAIR(X, Y, cx, cy)                                                                          \
    if (!trip && cx >= dthr && cy >= dthr) {                                                              \
        int dc = 0;                                                                                       \
        _Pragma("unroll") for (int j = 0; j < C; ++j) dc += __popcll(prev_mask<C, 1>(X, j) & Y[j]);       \
        trip = (uint32_t)dc >= dthr;                                                                      \
    }
                            FAQCS_PAIR(XA, XT, cA, cT) FAQCS_PAIR(XT, XA, cT, cA) FAQCS_PAIR(XA, XC, cA, cC)
                            FAQCS_PAIR(XC, XA, cC, cA) FAQCS_PAIR(XA, XG, cA, cG) FAQCS_PAIR(XG, XA, cG, cA)
                            FAQCS_PAIR(XT, XC, cT, cC) FAQCS_PAIR(XC, XT, cC, cT) FAQCS_PAIR(XT, XG, cT, cG)
                            FAQCS_PAIR(XG, XT, cG, cT) FAQCS_PAIR(XC, XG, cC, cG) FAQCS_PAIR(XG, XC, cG, cC)
#undef FAQCS_PAIR
                        }
                        (void)cc;
                    }
                    if (trip) { fs_blc += n; ++fs_rlc; ret = false; filt = FAQCS_FILT_LOW_COMPLEXITY; }
                }

                if (ret) { fs_trim_len += n; ++fs_trim_num; }
                fs_total_len += len;

                // ---- position matrices: one LDS atomic per base for pre+post ------------------------------
                if (!read_err) {
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        if (inr[j]) {
                            const uint32_t post = (ret && inw2[j]) ? 0x10000u : 0u;
                            const int col = j * 64 + lane;
                            atomicAdd(&hq[qs[j] * W + col], 1u | post);
                            const bool rep = (REP[j] >> lane) & 1ull;
                            if (code[j] < 5u) atomicAdd(&hb[code[j] * W + col], 1u | (rep ? 0u : post));
                            if (rep && post) atomicAdd(&hb[4 * W + col], post);
                        }
                    }
                } else {
                    any_err = 1;
                    flags |= FAQCS_F_ERR_QUALITY;
                }

                // ---- composition bins + sparse histograms through the LDS hash (lanes 0..17) --------------
                {
                    // int(average_quality) == max(0, floor(S/len) - offset): floor via host magic (exact)
                    const uint32_t qdiv_pre = len ? (uint32_t)(((uint64_t)S_pre * P.div_magic[len]) >> 44) : 0u;
                    const uint32_t qdiv_post = (ret && n) ? (uint32_t)(((uint64_t)S_post * P.div_magic[n]) >> 44) : 0u;
                    int qb_pre = (int)qdiv_pre - 128 - in_off, qb_post = (int)qdiv_post - 128 - in_off;
                    qb_pre = (len == 0 || qb_pre < 0) ? 0 : (qb_pre > 41 ? 41 : qb_pre);
                    qb_post = qb_post < 0 ? 0 : (qb_post > 41 ? 41 : qb_post);
                    const float norm_pre = P.comp_norm[len], norm_post = P.comp_norm[ret ? n : 0];
                    uint32_t cv = 0;
                    cv = lane == 0 ? pA : cv; cv = lane == 1 ? pT : cv; cv = lane == 2 ? pC : cv;
                    cv = lane == 3 ? pG : cv; cv = lane == 4 ? pN : cv;
                    cv = lane == 6 ? cA : cv; cv = lane == 7 ? cT : cv; cv = lane == 8 ? cC : cv;
                    cv = lane == 9 ? cG : cv; cv = lane == 10 ? cN : cv;
                    const float nv = lane < 6 ? norm_pre : norm_post;
                    uint32_t bin = (uint32_t)__fmul_rn(nv, (float)cv);
                    const uint32_t gc_pre = (uint32_t)__builtin_amdgcn_readlane((int)bin, 3) + (uint32_t)__builtin_amdgcn_readlane((int)bin, 2);
                    const uint32_t gc_post = (uint32_t)__builtin_amdgcn_readlane((int)bin, 9) + (uint32_t)__builtin_amdgcn_readlane((int)bin, 8);
                    bin = lane == 5 ? gc_pre : bin;
                    bin = lane == 11 ? gc_post : bin;
                    bin = lane == HS_PRE_LEN ? (uint32_t)len : bin;
                    bin = lane == HS_POST_LEN ? (uint32_t)n : bin;
                    bin = (lane == HS_PRE_RQ || lane == HS_PRE_BQ) ? (uint32_t)qb_pre : bin;
                    bin = (lane == HS_POST_RQ || lane == HS_POST_BQ) ? (uint32_t)qb_post : bin;
                    uint32_t val = 1;
                    val = lane == HS_PRE_BQ ? (uint32_t)len : val;
                    val = lane == HS_POST_BQ ? (uint32_t)n : val;
                    const bool is_post = (lane >= 6 && lane < 12) || lane == HS_POST_LEN || lane == HS_POST_RQ || lane == HS_POST_BQ;
                    if (lane < HS_NSLOT && !read_err && (!is_post || ret) && val != 0)
                        hash_add(hkey, hval, HSIZE - 1, (uint32_t)lane | (bin << 5), val, P, counters);
                }

                // ---- per-read result ----------------------------------------------------------------------
                const uint32_t hit = uniu((uint32_t)__builtin_amdgcn_readlane((int)v_hit, r));
                const uint32_t lo = ret ? ((uint32_t)a | ((uint32_t)n << 16)) : 0u;
                const uint32_t hi = flags | (ret ? FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT) | (hit << 16);
                res_lo = lane == (int)r ? lo : res_lo;
                res_hi = lane == (int)r ? hi : res_hi;
            }

            if ((uint32_t)lane < cnt) out[base + lane] = make_uint2(res_lo, res_hi);
            if (lane == 0) {
                atomicAdd(&lfs[FAQCS_TOTAL_COUNT], cnt);
                atomicAdd(&lfs[FAQCS_TOTAL_NUMBER], cnt);
                atomicAdd(&lfs[FAQCS_TOTAL_LENGTH], fs_total_len);
                if (fs_trim_num) { atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_NUMBER], fs_trim_num); atomicAdd(&lfs[FAQCS_TOTAL_TRIMMED_LENGTH], fs_trim_len); }
                if (fs_rlen) { atomicAdd(&lfs[FAQCS_READ_LENGTH], fs_rlen); atomicAdd(&lfs[FAQCS_BASE_LENGTH], fs_blen); }
                if (fs_rnn) { atomicAdd(&lfs[FAQCS_READ_NN], fs_rnn); atomicAdd(&lfs[FAQCS_BASE_NN], fs_bnn); }
                if (fs_ravg) { atomicAdd(&lfs[FAQCS_READ_AVG_Q], fs_ravg); atomicAdd(&lfs[FAQCS_BASE_AVG_Q], fs_bavg); }
                if (fs_rqt) { atomicAdd(&lfs[FAQCS_READ_QUAL_TRIM], fs_rqt); atomicAdd(&lfs[FAQCS_BASE_QUAL_TRIM], fs_bqt); }
                if (fs_rlc) { atomicAdd(&lfs[FAQCS_READ_LOW_COMPLEXITY], fs_rlc); atomicAdd(&lfs[FAQCS_BASE_LOW_COMPLEXITY], fs_blc); }
                if (any_err) atomicOr(err, 1u);
            }
        }

        // ---- flush before a 16-bit field can overflow, and at the end -----------------------------------
        if (((it + 1) % FLUSH_EVERY) == 0 || it + 1 == n_iter) {
            __syncthreads();
            const faqcs_layout &L = P.lay;
            for (int i = tid; i < Cfg::HQ; i += NW * 64) {
                const uint32_t v = hq[i];
                if (v) {
                    hq[i] = 0;
                    const int q = i / W, col = i % W, p = (col & 63) * C + (col >> 6);
                    if ((uint32_t)p < P.R) {
                        if (v & 0xffffu) atomicAdd((unsigned long long *)(counters + L.pre_qual + (uint64_t)p * FAQCS_NQ + q), (unsigned long long)(v & 0xffffu));
                        if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_qual + (uint64_t)p * FAQCS_NQ + q), (unsigned long long)(v >> 16));
                    }
                }
            }
            for (int i = tid; i < Cfg::HB; i += NW * 64) {
                const uint32_t v = hb[i];
                if (v) {
                    hb[i] = 0;
                    const int c = i / W, col = i % W, p = (col & 63) * C + (col >> 6);
                    if ((uint32_t)p < P.R) {
                        if (v & 0xffffu) atomicAdd((unsigned long long *)(counters + L.pre_base + (uint64_t)p * FAQCS_NBASE + c), (unsigned long long)(v & 0xffffu));
                        if (v >> 16) atomicAdd((unsigned long long *)(counters + L.post_base + (uint64_t)p * FAQCS_NBASE + c), (unsigned long long)(v >> 16));
                    }
                }
            }
            for (int i = tid; i < HSIZE; i += NW * 64) {
                const uint32_t k = hkey[i];
                if (k != HS_EMPTY) {
                    atomicAdd((unsigned long long *)hs_dest(P, counters, k & 31u, k >> 5), (unsigned long long)hval[i]);
                    hkey[i] = HS_EMPTY;
                    hval[i] = 0;
                }
            }
            if (tid < FAQCS_NUM_STAT) {
                const uint32_t v = lfs[tid];
                if (v) { atomicAdd((unsigned long long *)(counters + L.filter_stats + tid), (unsigned long long)v); lfs[tid] = 0; }
            }
            __syncthreads();
        }
    }
}

// ---- launch wrapper -------------------------------------------------------------------------------------
template <int C, int NW, int HSIZE>
static hipError_t launch_trim_t(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                uint32_t n_reads, const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out,
                                uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st)
{
    constexpr size_t lds = (size_t)(TrimCfg<C>::HQ + TrimCfg<C>::HB + 2 * HSIZE + FS_SLOTS) * 4;
    static bool attr_set = false;
    auto kern = trim_filter_accumulate<C, NW, HSIZE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint32_t chunks = (n_reads + 63) / 64;
    const int blocks_per_cu = (lds * 2 <= 160 * 1024) ? 2 : 1;
    uint32_t grid = (chunks + NW - 1) / NW;
    const uint32_t cap = (uint32_t)(n_cu * blocks_per_cu);
    if (grid > cap) grid = cap;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), counters, err);
    return hipGetLastError();
}

hipError_t faqcs_launch_trim(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                             uint32_t n_reads, uint32_t max_len, const uint32_t *ad_sl, const uint16_t *ad_hit,
                             faqcs_read_result *out, uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st)
{
    if (max_len <= 64) return launch_trim_t<1, 16, 4096>(P, seq, qual, off, n_reads, ad_sl, ad_hit, out, counters, err, n_cu, st);
    if (max_len <= 128) return launch_trim_t<2, 16, 4096>(P, seq, qual, off, n_reads, ad_sl, ad_hit, out, counters, err, n_cu, st);
    if (max_len <= 192) return launch_trim_t<3, 16, 4096>(P, seq, qual, off, n_reads, ad_sl, ad_hit, out, counters, err, n_cu, st);
    if (max_len <= 256) return launch_trim_t<4, 16, 2048>(P, seq, qual, off, n_reads, ad_sl, ad_hit, out, counters, err, n_cu, st);
    return hipErrorInvalidValue;
}
