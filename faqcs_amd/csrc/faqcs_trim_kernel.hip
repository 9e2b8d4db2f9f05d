// faqcs_trim_kernel.hip -- trim_tpr, trim_filter_accumulate + composition_histogram (gfx950, wave64).
//
// Replaces trim_read() and its helpers (trim.cpp:225-551, :553-597, :629-885, :1191-1216) for every read
// of a batch.  Three kernels share the accumulators, the flush and the per-chunk epilogue (faqcs_trim_common.h);
// faqcs_launch_trim at the end of this file picks one per submission:
//   trim_lds (faqcs_trim_lds_kernel.hip)   reads of 77 ... 304 bases, every option set but --replace_to_N_q: the headline shapes
//   trim_tpr                 the default-like option sets on reads of up to 76 bases: two phases per 64-read chunk, described at
//                            its definition (its wider instantiations are kept for A/B runs: FAQCS_TRIM_LDS=0)
//   trim_filter_accumulate   every other option set and read length up to 1 024 bases: one pass, described here
//
// Mapping (trim_filter_accumulate).  LPR lanes share one read and lane l of the group owns the C consecutive positions [l*C, l*C+C), fetched
// with ONE unaligned global_load_dwordx{D} per arena; a wave takes chunks of 64 reads.
//   LPR =  4  reads <= 76 bases (C = 16 / 19):      sixteen reads per wave, one per DPP quad
//   LPR =  8  reads <= 160 bases (C = 8 ... 20):    eight reads per wave, two per 16-lane DPP row
//   LPR = 16  reads <= 256 bases (C = 4 ... 16):    four reads per wave, one per DPP row
//   LPR = 32  reads <= 512 bases (C = 10 / 16):     two reads per wave
//   LPR = 64  reads <= 1024 bases (C = 16; 5 / 8 as the A/B fallback of LPR = 32): the whole wave on one read
// Every per-read scalar (length, window, cut points, filter decision) is a group-uniform VGPR value: there are no
// ballots and no scalar-ALU bit logic in the loop (round-1 profiling showed the ballot formulation was SALU-bound at
// ~900 scalar instructions per read).  Cross-lane work is DPP only (RowOps<LPR> in faqcs_dev.h): prefix scans and
// butterflies of 3-4 instructions, shared by all reads of the wave.  What depends on a read's scalars alone (FilterStat,
// small histograms, composition records, the result word) is parked in the read's owner lane and done once per
// chunk, 64 reads wide.
//
// BWA_plus (trim.cpp:714-793) in closed form from ONE prefix sum P of (Q - q[i]) over the window
// (SURVEY.md section 8 a-5):
//   3' pass: reset[i] = (i > 2) & (T - Pin[i] >= 0); the walk stops at the first i (descending) with
//            !reset[i] & !reset[i+1] & reset[i+2] (or after min(5,n) steps if no reset occurs in them);
//            final_pos_3 = argmax_{visited i} (T - Pex[i]) (largest i on ties, only if > 0) - 1
//   5' pass: mirror image with reset[i] = (i < final_pos_3 - 2) & (Pex[i] >= 0) and argmax of Pin[i].
//
// Accumulators.  position x quality: LDS [42][W] dwords, pre-trim count in the low and post-trim count in
// the high 16 bits so a base costs ONE ds_add for both (the post-trim quality of a kept base equals its
// pre-trim quality, trim.cpp:516-533).  position x base: a lane always owns the same positions, so the
// matrix is privatised in REGISTERS (6-bit fields A,T,C,G,N per position, one v_add per base) and spilled
// to LDS every <= 56 reads.  Length / average-quality histograms are small dense LDS arrays.  Composition
// bins (10 001 x 6, sparse and data dependent) are NOT accumulated here: the kernel emits one 8-byte record
// per read (length + 5 base counts, pre and post) and composition_histogram folds the records with the
// whole LDS as a 16-bit table.  A block flushes LDS to the global u64 block with atomics before a 16-bit
// field can overflow (every <= 65 535 reads per block).
//
// Float semantics of the reference are folded into integer lookup tables built on the host
// (DevParams); the only float op left is the composition-bin multiply, an exact IEEE v_mul_f32.
#include "faqcs_trim_common.h"

#ifndef FAQCS_TRIM_NW
#define FAQCS_TRIM_NW 4        /* waves per block (A/B on MI355X: 4 waves x 3 blocks/CU beat 8 x 1 by 9 %) */
#endif
// trim_tpr: ONE block per CU (its LDS holds 10 KB of prefix snapshots per wave): 12 waves = 3 per SIMD while the registers
// allow it (C = 19, the 2x150 shape: 168 VGPRs without a spill), 8 waves otherwise
constexpr int tpr_waves_per_simd(int C, int LPR = 8, bool ext = false) { return (LPR == 4 || (C == 19 && !ext) || C == 13) ? 3 : 2; } // what the hardware gets to run
// C = 19: one block of 12 waves, compiled for 3 waves per SIMD (168 VGPRs, no spill).  C = 13 needs 152 registers when the
// compiler is asked for 2 waves per SIMD but spills under a 168 cap, so it is compiled for 2 and launched as three
// blocks of 4 waves (the hardware co-schedules them: 152 <= 168).  C = 20: one block of 8.
// 4 lanes per read in phase B (reads <= 76 bases): like C = 13.
// EXT (--5trim_off, --avg_q, -n 0/1 on top of the default set): C = 19 then runs one block of 8 like C = 20.
constexpr int tpr_nw(int C, int LPR = 8, bool ext = false) { return LPR == 4 ? 4 : (C == 19 ? (ext ? 8 : 12) : (C == 13 ? 4 : 8)); }
constexpr int tpr_bounds_waves(int C, int LPR = 8, bool ext = false) { return (LPR == 8 && C == 19 && !ext) ? 3 : 2; }
#ifndef FAQCS_TRIM_MINWAVES
#define FAQCS_TRIM_MINWAVES 3  /* __launch_bounds__ 2nd argument: waves per SIMD the register allocator must allow */
#endif

// WINDOWED: an adapter pre-pass or --5end/--3end can move the window off [0, len); when false (the headline
// configuration) the prefix sum runs over all positions without per-position window tests.
// GENERIC: false = the headline option set (BWA_plus, 5' trimming on, not --qc_only, no --replace_to_N_q, no
// --avg_q, -n 2) is compiled in, so those tests and their live scalars disappear from the loop.
template <int C, int LPR, int NW, bool WINDOWED, bool GENERIC>
__global__ __launch_bounds__(NW * 64, (LPR == 8 || C > 10) ? 2 : FAQCS_TRIM_MINWAVES) void trim_filter_accumulate(
    const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
    const uint32_t *__restrict__ off, const uint32_t n_reads, const uint32_t *__restrict__ ad_sl,
    const uint16_t *__restrict__ ad_hit, uint2 *__restrict__ out, unsigned long long *__restrict__ rec_pre,
    unsigned long long *__restrict__ rec_post, uint64_t *__restrict__ counters, uint32_t *__restrict__ err)
{
    using Cfg = RowCfg<C, LPR>;
    using RW = RowOps<LPR>;
    constexpr int D = Cfg::D, W = Cfg::W, KEY_BIAS = Cfg::KEY_BIAS, PB = Cfg::PB, FK = Cfg::FK;
    constexpr uint32_t PMX = (1u << PB) - 1u;
    constexpr int JB = Cfg::JB;
    constexpr uint32_t JM = (1u << JB) - 1u;
    static_assert(!Cfg::HQ8 || NW * Cfg::HQ8_EVERY <= 255, "an 8-bit cell must not overflow between two flushes");
    constexpr uint32_t CMASK = (1u << C) - 1u;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hq = smem + Cfg::O_HQ, *hb = smem + Cfg::O_HB, *hlen = smem + Cfg::O_LEN, *hrq = smem + Cfg::O_RQ;
    uint32_t *hbqpre = smem + Cfg::O_BQPRE, *hbqpost = smem + Cfg::O_BQPOST, *lfs = smem + Cfg::O_FS;
    const uint32_t *t_base = smem + Cfg::O_TBASE, *t_lc = smem + Cfg::O_TLC, *t_magic = smem + Cfg::O_TMAGIC;
    const int32_t *t_avgq = (const int32_t *)(smem + Cfg::O_TAVGQ);
    const uint32_t *t_bm = smem + Cfg::O_TBM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int rl = lane & (LPR - 1);    // lane inside the row
    const int rowb = lane & (64 - LPR); // first lane of the row
    const int wave = uni(tid >> 6);
    const int pbase = rl * C;

    for (int i = tid; i < Cfg::N_ZERO; i += NW * 64) smem[i] = 0u;
    for (int i = tid; i < 256; i += NW * 64) smem[Cfg::O_TBASE + i] = P.base_tab[i];
    for (int i = tid; i <= W; i += NW * 64) {
        smem[Cfg::O_TLC + i] = P.lc_thr[i];
        smem[Cfg::O_TAVGQ + i] = (uint32_t)P.avgq_min_v[i];
        smem[Cfg::O_TMAGIC + i] = P.div_magic[i];
    }
    for (int i = tid; i < Cfg::BMW * (C + 1); i += NW * 64) {
        const int nb = med3i((i / Cfg::BMW) - 4 * (i % Cfg::BMW), 0, 4);
        smem[Cfg::O_TBM + i] = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
    }
    uint32_t two = 2u;
    asm volatile("" : "+v"(two)); // a VGPR operand for the SDWA shifts
    // the base-table lookups address LDS by offset: the kernel's only LDS object must start at LDS address 0
    if (tid == 0 && blockIdx.x == 0 && (uint32_t)(size_t)((lds_u32_ptr)smem) != 0u) atomicOr(err, 4u);
    __syncthreads();

    const uint32_t total_chunks = (n_reads + 63) >> 6;
    const uint32_t chunks_per_iter = gridDim.x * NW;
    const uint32_t n_iter = (total_chunks + chunks_per_iter - 1) / chunks_per_iter;
    constexpr uint32_t FLUSH_EVERY = 65535u / (NW * 64) > 0 ? 65535u / (NW * 64) : 1;
    // 6-bit fields: 3 chunks x 16 reads per row = 48 <= 63; a 64-lane row sees 64 reads per chunk and spills mid-chunk too
    constexpr uint32_t REG_FLUSH_EVERY = LPR == 16 ? 3 : (LPR == 8 ? 7 : (LPR == 4 ? 15 : 1)); // 32 lanes: 32 reads per chunk; 64: spills mid-chunk too

    const int in_off = P.in_off, Q = P.Q;
    const int o_mode = GENERIC ? P.mode : (int)FAQCS_MODE_BWA_PLUS;
    const bool o_protect5 = GENERIC ? P.protect5 != 0 : false;
    const bool o_qc_only = GENERIC ? P.qc_only != 0 : false;
    const uint32_t o_replace_q = GENERIC ? P.replace_q : 0u;
    const bool o_avgq_on = GENERIC ? P.avgq_on != 0 : false;
    const uint32_t o_dbg = GENERIC ? P.dbg : 0u;
    const uint32_t o_max_poly_n = GENERIC ? P.max_poly_n : 2u;
    const bool do_trim = !o_qc_only && !(o_dbg & 4u);
    uint32_t bpre[C], bpost[C];
#pragma unroll
    for (int j = 0; j < C; ++j) { bpre[j] = 0; bpost[j] = 0; }
    uint32_t any_err = 0;
    // spill of the register-privatised base matrix to LDS (before a 6-bit field can overflow)
    auto spill_base_regs = [&]() {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t x = bpre[j], y = bpost[j];
            if (x) {
#pragma unroll
                for (int c = 0; c < FAQCS_NBASE; ++c) {
                    const uint32_t v = ((x >> BT_SHIFT(c)) & 63u) | (((y >> BT_SHIFT(c)) & 63u) << 16);
                    if (v) atomicAdd(&hb[c * W + pbase + j], v);
                }
            }
            bpre[j] = 0; bpost[j] = 0;
        }
    };

#pragma unroll 1
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t chunk = (it * gridDim.x + blockIdx.x) * NW + wave;
        // (the 8-bit variant has block barriers inside the read loop: a wave without a chunk runs it with no reads)
        if (Cfg::HQ8 || chunk < total_chunks) {
            const uint32_t base = chunk << 6;
            const uint32_t my = base + lane;
            const bool mine = my < n_reads;
            const uint32_t v_off = mine ? off[my] : 0u;
            const uint32_t v_len = mine ? off[my + 1] - v_off : 0u;
            const uint32_t v_sl = (WINDOWED && ad_sl && mine) ? ad_sl[my] : (v_len << 16);
            const uint32_t v_hit = (ad_hit && mine) ? ad_hit[my] : 0u;
            // Per-read outcome, parked in the lane that owns the read (lane rowb + t): everything that is a function of
            // these scalars alone -- FilterStat sums, the small histograms, composition records, the result word -- is
            // done ONCE per chunk after the read loop, 64 reads wide, instead of 16 times per chunk on row-uniform values.
            uint32_t st_an = 0, st_fl = 0, st_pAT = 0, st_pCG = 0, st_cAT = 0, st_cCG = 0, st_N = 0;
            int st_Vpre = 0, st_Vpost = 0;

            // ---- software prefetch of the row's read 0 ---------------------------------------------------
            PackedBytes<D> nseq, nqual;
            int n_len = __shfl((int)v_len, rowb);
            {
                const uint32_t o = (uint32_t)__shfl((int)v_off, rowb);
#pragma unroll
                for (int k = 0; k < D; ++k) { nseq.w[k] = 0; nqual.w[k] = 0; }
                if (pbase < n_len) {
                    nseq = *(const PackedBytes<D> *)(seq + (size_t)o + pbase);
                    nqual = *(const PackedBytes<D> *)(qual + (size_t)o + pbase);
                }
            }

#pragma unroll 1
            for (int t = 0; t < LPR; ++t) {
                if (!Cfg::HQ8 && base + (uint32_t)t >= n_reads) break; // wave-uniform: no row has a read left
                if (LPR == 64 && t == 32) spill_base_regs();
                const int len = n_len;
                const bool act = base + (uint32_t)(rowb + t) < n_reads;
                uint32_t ws[D], wq[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { ws[k] = nseq.w[k]; wq[k] = nqual.w[k]; }
                const uint32_t sl = WINDOWED ? (uint32_t)__shfl((int)v_sl, rowb + t) : 0u;
                if (t + 1 < LPR) {
                    n_len = __shfl((int)v_len, rowb + t + 1);
                    const uint32_t o = (uint32_t)__shfl((int)v_off, rowb + t + 1);
#pragma unroll
                    for (int k = 0; k < D; ++k) { nseq.w[k] = 0; nqual.w[k] = 0; }
                    if (pbase < n_len) {
                        nseq = *(const PackedBytes<D> *)(seq + (size_t)o + pbase);
                        nqual = *(const PackedBytes<D> *)(qual + (size_t)o + pbase);
                    }
                }
                // zero the bytes past the end of the read (the last dword of a lane may over-read 1..3 bytes)
                {
                    const int vb = med3i(len - pbase, 0, C);
                    const uint4 bm = *reinterpret_cast<const uint4 *>(t_bm + Cfg::BMW * vb); // one ds_read_b128
                    uint32_t m[8] = {bm.x, bm.y, bm.z, bm.w, 0u, 0u, 0u, 0u};
                    if (D > 4) {
                        const uint4 bm2 = *reinterpret_cast<const uint4 *>(t_bm + Cfg::BMW * vb + 4);
                        m[4] = bm2.x; m[5] = bm2.y; m[6] = bm2.z; m[7] = bm2.w;
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) { ws[k] &= m[k]; wq[k] &= m[k]; }
                }

                // ---- window after the adapter pre-pass and --5end/--3end (trim.cpp:270-314) --------------
                int a = 0, n = len;
                uint32_t flags = 0, filt = 0;
                if (WINDOWED && P.has_adapters) {
                    const int first = (int)(sl & 0xffffu), second = (int)(sl >> 16);
                    const bool mod = len != second;
                    a = mod ? first : 0; n = mod ? second : len;
                    flags = mod ? FAQCS_F_ADAPTER : 0u;
                }
                if (WINDOWED && P.trim5 && !o_qc_only) {
                    const bool over = (int)P.trim5 > n;
                    a = over ? a : a + (int)P.trim5;
                    n = over ? 0 : n - (int)P.trim5;
                }
                if (WINDOWED && P.trim3 && !o_qc_only) n = (int)P.trim3 > n ? 0 : n - (int)P.trim3;
                bool ret = act;
                if (ret && (n < (int)P.min_len || n == 0)) { ret = false; filt = FAQCS_FILT_LENGTH_PRE; }
                uint32_t qt_removed = 0;        // bases removed by the quality trim (BASE_QUAL_TRIM)

                // ---- pass 1 over the lane's C positions ---------------------------------------------------
                uint32_t incf[C];          // class word of the base: 6-bit count fields A,T,C,G,N; the two flag bits above them
                                           // ride along (sums only ever carry them out of the word; every reader masks)
                int q[C], Pin[C];          // clamped quality; inclusive prefix sum of (Q - q) up to this position
                int run, sumv, T, E;
                uint32_t cntpack, nubits, gubits, maxq;
                const int pa = pbase - a;
#pragma unroll 1
                for (int attempt = 0; attempt < 2; ++attempt) {
                    uint32_t inc[C];
                    BaseLookup<C, 0>::run((uint32_t)(Cfg::O_TBASE * 4), ws, two, inc);
                    run = 0; sumv = 0; cntpack = 0; nubits = 0; gubits = 0; maxq = 0;
#pragma unroll
                    for (int j = C - 1; j >= 0; --j) nubits = __builtin_amdgcn_alignbit(nubits, inc[j], 31); // bit j = BT_IS_NU of j
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const int sq = (int)(int8_t)((wq[j >> 2] >> (8 * (j & 3))) & 0xffu);
                        const int v = sq - in_off;                              // quality_score() before the clamp
                        q[j] = v < 0 ? 0 : v;                                   // fastq.h:29
                        sumv += v;
                        maxq = umax_(maxq, (uint32_t)q[j]);
                        int dq = Q - q[j];
                        if (WINDOWED) dq = ((unsigned)(pa + j) < (unsigned)n) ? dq : 0;
                        run += dq;
                        Pin[j] = run;
                        cntpack += inc[j];
                        incf[j] = inc[j];
                        if (o_replace_q > 0) gubits |= ((inc[j] >> 30) & 1u) << j;
                    }
                    if (attempt == 1) break;
                    // mask_quality_terminal_N (trim.cpp:1191-1216): upper-case 'N' runs at either end get Q0.
                    // Rare: detect after the fact, patch the quality bytes and redo the pass.
                    const int lastj = len - 1 - pbase;
                    const bool term = ((rl == 0) && (nubits & 1u)) || ((unsigned)lastj < (unsigned)C && ((nubits >> lastj) & 1u));
                    if (!__any(term)) break;
                    const uint32_t inr = range_mask<C>(0, len, pbase);
                    const uint32_t non = inr & ~nubits;
                    const uint32_t fn_l = non ? (uint32_t)(FK - (pbase + __builtin_ctz(non))) : 0u;
                    const uint32_t ln_l = non ? (uint32_t)(pbase + (31 - __builtin_clz(non)) + 1) : 0u;
                    const uint32_t fn_m = RW::all_umax(fn_l), ln_m = RW::all_umax(ln_l);
                    const int lead = fn_m ? FK - (int)fn_m : len;   // first non-N position (len if the read is all N)
                    const int trail_start = (int)ln_m;                // 1 + last non-N position (0 if none)
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const int p = pbase + j;
                        if (p < len && (p < lead || p >= trail_start)) {
                            const uint32_t sh = 8 * (j & 3);
                            wq[j >> 2] = (wq[j >> 2] & ~(0xffu << sh)) | (((uint32_t)in_off & 0xffu) << sh);
                        }
                    }
                }
                {
                    const int incl = RW::incl_scan_add(run);
                    E = incl - run;                                  // prefix before this lane's first position
                    T = RW::all_sum(run);
                    // without window masks the zero bytes past the read contributed (Q - 0) each
                    if (!WINDOWED) T -= Q * (LPR * C - len);
                }
                // Pin[] stays LANE-LOCAL (prefix inside the lane); E is folded into whoever needs the row-wide prefix
                const int TE = T - E;
                // whole-read sums: base counts (A,T | C,G as 16-bit pairs), N count and V = sum(raw - offset)
                uint32_t pAT, pCG, pN;
                int V_pre;
                {
                    const uint32_t c = cntpack & BT_FIELDS;
                    const uint32_t at = (c & 63u) | (((c >> 6) & 63u) << 16), cg = ((c >> 12) & 63u) | (((c >> 18) & 63u) << 16);
                    pAT = (uint32_t)RW::all_sum((int)at);
                    pCG = (uint32_t)RW::all_sum((int)cg);
                    // positions past the read contributed (0 - in_off) each: add them back
                    const int vb = med3i(len - pbase, 0, C);
                    const int both = RW::all_sum((((int)((c >> 24) & 63u)) << 20) + (sumv + in_off * (C - vb) + (1 << 12)));
                    pN = (uint32_t)both >> 20;
                    V_pre = (int)((uint32_t)both & 0xfffffu) - (LPR << 12);
                }
                const bool read_err = RW::all_umax(maxq) > 41u;

                // ---- quality trim (trim.cpp:325-360) -----------------------------------------------------
                int hi_sum = 0, lo_sum = 0;     // BWA_plus by-products: prefix sums of (Q - q) at the two cut points
                bool have_sums = false;
                if (do_trim) {
                    int fp3 = n - 1, fp5 = 0;
                    const int a5 = n < 5 ? n : 5, nan2 = n < 2 ? n : 2;
                    hi_sum = 0; lo_sum = 0;
                    if (o_mode == FAQCS_MODE_BWA_PLUS) {
                        // Pex[j] (prefix before position j) is Pin[j-1], or E for the lane's first position
                        // D[j] = T - Pin[j] = suffix sum after position j: its sign is the 3' reset flag, and D[j-1] the
                        // argmax value of position j.  Sign bits are shifted in with one v_alignbit each.
                        int Dv[C];
                        uint32_t nn = 0;
#pragma unroll
                        for (int j = C - 1; j >= 0; --j) {
                            Dv[j] = TE - Pin[j];
                            nn = __builtin_amdgcn_alignbit(nn, (uint32_t)Dv[j], 31); // (nn << 1) | (D < 0)
                        }
                        nn = ~nn;
                        const uint32_t r3 = nn & range_mask<C>(a + nan2 + 1, a + n, pbase);
                        const uint32_t f5 = r3 & range_mask<C>(a + n - a5, a + n, pbase);
                        const uint32_t ext = r3 | (RW::next(r3) << C);
                        const uint32_t c3 = ~ext & ~(ext >> 1) & (ext >> 2) & CMASK;
                        const uint32_t red = RW::all_umax((c3 ? (uint32_t)(pbase + (31 - __builtin_clz(c3)) + 1) : 0u));
                        const bool early = RW::all_or(f5) != 0u;
                        const int pstar = early ? (int)red - 1 : a + n - a5;
                        const uint32_t vis = range_mask<C>(pstar > a ? pstar : a, a + n, pbase);
                        // lane-local argmax of S = T - Pex (largest position on ties), then one row max
                        uint32_t kl = 0, kx[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const uint32_t k = ((uint32_t)(j ? Dv[j - 1] : TE) << JB) + (uint32_t)((KEY_BIAS << JB) | j);
                            kx[j] = k & (uint32_t)bit_m1(vis, j);
                        }
#pragma unroll
                        for (int j = 0; j + 1 < C; j += 2) kl = umax3_(kl, kx[j], kx[j + 1]);
                        if (C & 1) kl = umax_(kl, kx[C - 1]);
                        const uint32_t K3 = RW::all_umax(kl ? (((kl >> JB) << PB) + (uint32_t)(pa + (int)(kl & JM))) : 0u);
                        const int S3 = (int)(K3 >> PB) - KEY_BIAS;
                        fp3 = (S3 > 0) ? (int)(K3 & PMX) - 1 : n - 1;
                        hi_sum = (S3 > 0) ? T - S3 : T;                       // sum of (Q-q) over window positions <= fp3
                        // 5' pass.  Its cut is the argmax of the prefix sums over the visited positions and only counts when that
                        // maximum is positive (trim.cpp:760-790): with no positive prefix sum anywhere in any of the wave's
                        // reads -- the usual case, a read that starts at Q >= -q -- the whole pass is skipped.
                        int pmax = Pin[0];
#pragma unroll
                        for (int j = 1; j < C; ++j) pmax = pmax > Pin[j] ? pmax : Pin[j];
                        if (!o_protect5 && __any(pmax + E > 0)) {
#pragma unroll
                            for (int j = 0; j < C; ++j) Pin[j] += E; // (rare path: row-wide prefixes from here on)
                            uint32_t np = 0;
#pragma unroll
                            for (int j = C - 1; j >= 0; --j) np = __builtin_amdgcn_alignbit(np, (uint32_t)(j ? Pin[j - 1] : E), 31);
                            np = ~np; // bit j = (prefix before position j >= 0)
                            const uint32_t r5 = np & range_mask<C>(a, a + fp3 - nan2, pbase);
                            const uint32_t g5 = r5 & range_mask<C>(a, a + a5, pbase);
                            const uint32_t ext5 = (r5 << 2) | ((RW::prev(r5) >> (C - 2)) & 3u); // bit k <-> position pbase + k - 2
                            const uint32_t c5 = ~(ext5 >> 2) & ~(ext5 >> 1) & ext5 & CMASK;
                            const uint32_t red5 = RW::all_umax(c5 ? (uint32_t)(FK - (pbase + __builtin_ctz(c5))) : 0u);
                            const bool early5 = RW::all_or(g5) != 0u;
                            const int pstar5 = early5 ? FK - (int)red5 : a + a5 - 1;
                            const uint32_t vis5 = range_mask<C>(a, (pstar5 + 1 < a + n) ? pstar5 + 1 : a + n, pbase);
                            uint32_t kl5 = 0, ky[C];
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const uint32_t k = ((uint32_t)Pin[j] << JB) + (uint32_t)((KEY_BIAS << JB) | ((int)JM - j));
                                ky[j] = k & (uint32_t)bit_m1(vis5, j);
                            }
#pragma unroll
                            for (int j = 0; j + 1 < C; j += 2) kl5 = umax3_(kl5, ky[j], ky[j + 1]);
                            if (C & 1) kl5 = umax_(kl5, ky[C - 1]);
                            const uint32_t K5 = RW::all_umax(kl5 ? (((kl5 >> JB) << PB) + (uint32_t)((int)PMX - (pa + (int)JM - (int)(kl5 & JM)))) : 0u);
                            const int S5 = (int)(K5 >> PB) - KEY_BIAS;
                            fp5 = (S5 > 0) ? (int)PMX - (int)(K5 & PMX) + 1 : 0;
                            lo_sum = (S5 > 0) ? S5 : 0;                       // sum of (Q-q) over window positions < fp5
                        }
                        have_sums = true;
                    } else if (o_mode == FAQCS_MODE_BWA) { // trim.cpp:675-709
                        uint32_t neg = 0;
#pragma unroll
                        for (int j = C - 1; j >= 0; --j) neg = (neg << 1) | (uint32_t)(Pin[j] > TE);
                        neg &= range_mask<C>(a, a + n, pbase);
                        const int pf = (int)RW::all_umax(neg ? (uint32_t)(pbase + (31 - __builtin_clz(neg)) + 1) : 0u) - 1; // -1: none
                        const int lo = (pf < a ? a : pf) + 1;
                        const uint32_t vis = range_mask<C>(lo, a + n, pbase);
                        uint32_t key = 0;
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const uint32_t k = ((uint32_t)(TE - (j ? Pin[j - 1] : 0) + KEY_BIAS) << PB) + (uint32_t)(pa + j);
                            key = umax_(key, k & (uint32_t)bit_m1(vis, j));
                        }
                        const uint32_t K3 = RW::all_umax(key);
                        fp3 = ((int)(K3 >> PB) - KEY_BIAS > 0) ? (int)(K3 & PMX) - 1 : n - 1;
                    } else { // HARD, trim.cpp:629-672
                        uint32_t h0 = 0;
#pragma unroll
                        for (int j = C - 1; j >= 0; --j) h0 = (h0 << 1) | (uint32_t)(Q < q[j]);
                        h0 &= range_mask<C>(a, a + n, pbase);
                        const uint32_t h1 = h0 & range_mask<C>(a + 1, a + n, pbase);
                        const int h = (int)RW::all_umax(h1 ? (uint32_t)(pbase + (31 - __builtin_clz(h1)) + 1) : 0u) - 1;
                        int pos3 = 0;
                        if (h >= 0) { fp3 = h - a; pos3 = fp3; }
                        if (!o_protect5) {
                            const uint32_t lm = RW::all_umax(h0 ? (uint32_t)(FK - (pbase + __builtin_ctz(h0))) : 0u);
                            const int l = lm ? FK - (int)lm - a : 0x7fffffff;
                            if (l < pos3) fp5 = l;
                        }
                    }
                    if (ret) {
                        const int kept = (o_mode == FAQCS_MODE_BWA_PLUS && fp3 <= fp5) ? 0 : fp3 - fp5 + 1;
                        if (kept != n) { qt_removed = (uint32_t)(n - kept); flags |= FAQCS_F_QUAL_TRIMMED; }
                        a += fp5;
                        n = kept;
                        if (n < (int)P.min_len || n == 0) { ret = false; filt = FAQCS_FILT_LENGTH_POST; }
                    }
                }

                // ---- final window: poly-N, counts, sum(raw - offset) ---------------------------------------
                const uint32_t win2 = range_mask<C>(a, a + n, pbase);
                const bool whole = (a == 0 && n == len);
                if (ret && !(o_dbg & 16u)) { // poly-N filter (trim.cpp:363-371, :578-597): upper-case 'N' runs only
                    const uint32_t K = o_max_poly_n;
                    const uint32_t nw = nubits & win2;
                    bool trip;
                    if (K == 0) trip = true;
                    else {
                        const uint32_t e2 = nw | (RW::next(nw) << C);
                        const uint32_t pairs = e2 & (e2 >> 1) & CMASK;
                        const uint32_t red = RW::all_or((nw ? 1u : 0u) | (pairs ? 2u : 0u));
                        if (K == 1) trip = (red & 1u) != 0;
                        else if (K == 2) trip = (red & 2u) != 0;
                        else if (!(red & 2u)) trip = false;
                        else { // exact longest run (rare): run ending at p = p - (last non-N position <= p)
                            uint32_t loc[C], m = 0;
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const bool isn = (nw >> j) & 1u, in = (win2 >> j) & 1u;
                                m = umax_(m, (in && !isn) ? (uint32_t)(pbase + j) + 1u : 0u);
                                loc[j] = m;
                            }
                            const uint32_t excl = RW::prev(RW::incl_scan_umax(m)); // max-scan over the lanes before this one
                            uint32_t best = 0;
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const uint32_t lastnon = umax_(umax_(excl, loc[j]), (uint32_t)a);
                                best = umax_(best, ((nw >> j) & 1u) ? (uint32_t)(pbase + j) + 1u - lastnon : 0u);
                            }
                            trip = RW::all_umax(best) >= K;
                        }
                    }
                    if (trip) {
                        flags |= FAQCS_F_POLY_N_SEEN;
                        if (!o_qc_only) { ret = false; filt = FAQCS_FILT_POLY_N; }
                    }
                }

                // counts inside the final window, after G -> N (trim.cpp:390-403)
                uint32_t cAT = pAT, cCG = pCG, cN = pN;
                int V_post = V_pre;
                uint32_t repbits = 0;           // positions whose 'G' becomes 'N'
                if (o_replace_q > 0) {
#pragma unroll
                    for (int j = 0; j < C; ++j) repbits |= (uint32_t)(q[j] < (int)o_replace_q) << j;
                    repbits &= gubits & win2;
                }
                if (!(o_dbg & 16u) && __any(ret && (!whole || repbits))) {
                    uint32_t cp = 0;
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const uint32_t w = ((repbits >> j) & 1u) ? (1u << BT_SHIFT(4)) : incf[j];
                        cp += w & (uint32_t)bit_m1(win2, j);
                    }
                    const uint32_t at = (cp & 63u) | (((cp >> 6) & 63u) << 16), cg = ((cp >> 12) & 63u) | (((cp >> 18) & 63u) << 16);
                    cAT = (uint32_t)RW::all_sum((int)at);
                    cCG = (uint32_t)RW::all_sum((int)cg);
                    cN = (uint32_t)RW::all_sum((int)((cp >> 24) & 63u));
                    // V_post = sum over the final window of (raw - offset).  With no raw byte below the offset it is
                    // n*Q - sum(Q - q), and BWA_plus already produced both partial sums; otherwise re-add per position.
                    const bool clean = V_pre == len * Q - (WINDOWED ? 0 : T) && !WINDOWED; // all v == q over the whole read
                    if (__all(!ret || (clean && have_sums))) {
                        V_post = n * Q - (hi_sum - lo_sum);
                    } else {
                        int sv = 0;
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const int sq = (int)(int8_t)((wq[j >> 2] >> (8 * (j & 3))) & 0xffu);
                            sv += (sq - in_off) & bit_m1(win2, j);
                        }
                        V_post = RW::all_sum(sv);
                    }
                }
                const uint32_t cA = cAT & 0xffffu, cT = cAT >> 16, cC = cCG & 0xffffu, cG = cCG >> 16;

                // ---- average quality (trim.cpp:374-382) ----------------------------------------------------
                if (ret && o_avgq_on && V_post < t_avgq[n]) { ret = false; filt = FAQCS_FILT_AVG_Q; }

                // ---- low-complexity filter (trim.cpp:405-513) ----------------------------------------------
                if (ret && !(o_dbg & 16u)) {
                    const uint32_t thr = t_lc[n];
                    const uint32_t mthr = thr & 0xffffu, dthr = thr >> 16;
                    bool trip = cA >= mthr || cT >= mthr || cG >= mthr || cC >= mthr;
                    // dc[X->Y] <= min(count X, count Y): only pairs whose two counts both reach dthr can trip
                    const uint32_t nbig = (cA >= dthr) + (cT >= dthr) + (cC >= dthr) + (cG >= dthr);
                    if (__any(!trip && nbig >= 2)) {
                        // class index per position (0..3 = A,T,C,G inside the window, 7 = anything else)
                        uint32_t cls[C];
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const uint32_t w = (((repbits >> j) & 1u) ? 0u : incf[j] & 0xffffffu) & (uint32_t)bit_m1(win2, j);
                            cls[j] = w ? (uint32_t)(__builtin_ctz(w) / 6) : 7u;
                        }
                        const uint32_t prev_last = RW::prev(cls[C - 1] + 1u); // 0 at the row edge
                        const uint32_t cnts[4] = {cA, cT, cC, cG};
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int y = 0; y < 4; ++y)
                                if (x != y) {
                                    int dc = 0;
#pragma unroll
                                    for (int j = 0; j < C; ++j) {
                                        const uint32_t pv = j ? cls[j - 1] : prev_last - 1u; // row edge: 0xffffffff
                                        dc += (pv == (uint32_t)x && cls[j] == (uint32_t)y) ? 1 : 0;
                                    }
                                    dc = RW::all_sum(dc);
                                    trip = trip || (cnts[x] >= dthr && cnts[y] >= dthr && (uint32_t)dc >= dthr);
                                }
                    }
                    if (trip) { ret = false; filt = FAQCS_FILT_LOW_COMPLEXITY; }
                }

                // ---- accumulate: position x quality (LDS) and position x base (registers) -----------------
                if (read_err) { any_err = 1; flags |= FAQCS_F_ERR_QUALITY; }
                if (!(o_dbg & 2u)) {
                    // branch-free: a position outside the read adds 0 to a valid address
                    // (bytes past the read are zero -> class word 0 and quality column 0 with increment 0; a read with
                    //  Q > 41 aborts the whole run, fastq.h:31-33, so its row only has to stay inside the tables)
                    if (__any(read_err)) { // rare: keep the row's table indices in range, count nothing
#pragma unroll
                        for (int j = 0; j < C; ++j) { q[j] = read_err ? 0 : q[j]; incf[j] = read_err ? 0u : incf[j]; }
                    }
                    // 1024-wide rows keep exact per-position "pre" bits; the others add 1 for every slot of a counted read and
                    // let flush_block subtract the slots past the read's end
                    const uint32_t inr = Cfg::HQ8 ? ((act && !read_err) ? range_mask<C>(0, len, pbase) : 0u) : 0u;
                    const uint32_t counted = (act && !read_err) ? 1u : 0u;
                    const uint32_t pb4 = 4u * (uint32_t)pbase;
                    const uint32_t postm = ret ? (Cfg::HQ8 ? (win2 & inr) : win2) : 0u;
                    if (Cfg::HQ8) {
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const int qq = q[j];
                            const uint32_t x = ((inr >> j) & 1u) | ((uint32_t)bit_m1(postm, j) & 0x100u); // pre -> byte 0, post -> byte 1
                            lds_add_u32(__umul24((uint32_t)qq, (uint32_t)(W / 2 * 4)) + (uint32_t)(Cfg::O_HQ * 4) + 4u * (uint32_t)((pbase + j) >> 1),
                                        x << (16 * ((pbase + j) & 1)));
                        }
                    }
                    if (o_replace_q > 0) {
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const int qq = q[j];
                            if (!Cfg::HQ8) lds_add_u32(__umul24((uint32_t)qq, (uint32_t)(W * 4)) + pb4 + (uint32_t)(Cfg::O_HQ * 4 + 4 * j),
                                                       counted | ((uint32_t)bit_m1(postm, j) & 0x10000u));
                            bpre[j] += incf[j];
                            const uint32_t w = ((repbits >> j) & 1u) ? (1u << BT_SHIFT(4)) : incf[j];
                            bpost[j] += w & (uint32_t)bit_m1(postm, j);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const int qq = q[j];
                            if (!Cfg::HQ8) lds_add_u32(__umul24((uint32_t)qq, (uint32_t)(W * 4)) + pb4 + (uint32_t)(Cfg::O_HQ * 4 + 4 * j),
                                                       counted | ((uint32_t)bit_m1(postm, j) & 0x10000u));
                            bpre[j] += incf[j];
                            bpost[j] += incf[j] & (uint32_t)bit_m1(postm, j);
                        }
                    }
                }

                // ---- park the read's outcome in its owner lane (see the chunk epilogue) ----------------------
                if (rl == t) {
                    st_an = (uint32_t)a | ((uint32_t)n << 16);
                    st_fl = flags | (ret ? FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT) | (qt_removed << 20);
                    st_pAT = pAT; st_pCG = pCG; st_cAT = cAT; st_cCG = cCG; st_N = pN | (cN << 16);
                    st_Vpre = V_pre; st_Vpost = V_post;
                }
                if (Cfg::HQ8 && (t % Cfg::HQ8_EVERY) == Cfg::HQ8_EVERY - 1) flush_hq8<C, LPR, NW>(smem, counters, P.R, tid);
            }

            // ---- chunk epilogue: one read per lane ----------------------------------------------------------
            if (!(o_dbg & 32u)) {
                const ReadOutcome oc{st_an, st_fl, st_pAT, st_pCG, st_cAT, st_cCG, st_N, st_Vpre, st_Vpost};
                chunk_epilogue<LPR>(oc, mine, my, v_len, v_hit, lane, smem + Cfg::O_LEN, smem + Cfg::O_RQ, smem + Cfg::O_BQPRE,
                                    smem + Cfg::O_BQPOST, smem + Cfg::O_FS, smem + Cfg::O_TMAGIC, out, rec_pre, rec_post, o_avgq_on, o_dbg);
            }
        }

        // ---- spill the register-privatised base matrix to LDS before a 6-bit field can overflow ----------
        const bool block_flush = ((it + 1) % FLUSH_EVERY) == 0 || it + 1 == n_iter;
        if (((it + 1) % REG_FLUSH_EVERY) == 0 || block_flush) spill_base_regs();

        // ---- flush LDS -> global before a 16-bit field can overflow, and at the end ----------------------
        if (block_flush && !(o_dbg & 64u)) flush_block<C, LPR, NW>(smem, counters, P.R, tid);
    }
    if (__any(any_err != 0) && lane == 0) atomicOr(err, 1u);
}

// ---------------------------------------------------------------------------------------------------------
// trim_tpr: the headline option set (BWA_plus, 5' trimming on, -n 2, no --qc_only / --replace_to_N_q / --avg_q)
// for reads of at most 160 bases, in two phases per 64-read chunk.
//
//   phase A, "thread per read": the lane that OWNS a read holds its bytes in registers (ten 16-byte loads per
//     arena) and takes every per-read decision alone: class counts (an LDS table with 8-bit A,T,C,G fields, a prefix
//     snapshot per dword parked in LDS), upper-case N bits, quality sum and range check on four bytes per
//     instruction, the two BWA_plus walks exactly as trim.cpp:714-793 states them (serially, position by position,
//     all 64 reads in lockstep; a dword no lane still needs is skipped), length / poly-N / low-complexity filters.
//     No cross-lane reduction and no row-uniform control flow is left: what used to be ~300 instructions per 8 reads
//     is now spent once per 64.
//   phase B, position-parallel (8 lanes per read, as trim_filter_accumulate): only applies the window -- position x
//     quality and position x base accumulation.
//   Rare inputs (a raw quality outside [offset, offset + 41], letters other than ACGTN, dinucleotide candidates)
//     take exact per-position passes over the registers, entered only by chunks that contain such a read.
// ---------------------------------------------------------------------------------------------------------
template <int C, int LPR = 8> struct TprCfg {
    using Row = RowCfg<C, LPR>;
    static constexpr int NP = (Row::W + 15) / 16;  // 16-byte pieces per read and arena (the out-of-line exact passes)
    static constexpr int ND = (Row::W + 3) / 4;    // dwords per read and arena held by the owner lane
    static constexpr int NF = ND / 4, NR = ND % 4; // ... fetched as NF 16-byte pieces and one of NR dwords
    static constexpr int SROW = 65;                // dwords per snapshot row: 64 lanes + 1, so that a column walk changes bank
    static constexpr int NWORD = (ND * 4 + 31) / 32;
    static constexpr int O_T2 = (Row::LDS_DWORDS + 3) & ~3;   // [256][2]: A,T,C,G one-hot in 8-bit fields ; isN(upper) | isN(any) << 1
    static constexpr int O_SNAP = O_T2 + 512;                 // [NW][ND][SROW] class counts before dword k; after phase A: the qualities
    static constexpr int SNAP_WAVE = ND * SROW;
    static constexpr int lds_dwords(int nw) { return O_SNAP + nw * SNAP_WAVE; }
};

namespace {
constexpr int TPR_SROW_BYTES = 65 * 4; // == TprCfg::SROW * 4
typedef uint32_t LdsPair __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) LdsPair *lds_u2_ptr;
__device__ __forceinline__ void lds_store_u32(uint32_t byte_offset, uint32_t v) { *(lds_u32_mut)(size_t)byte_offset = v; }
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t byte_offset) { return *(lds_u32_ptr)(size_t)byte_offset; }
template <int K> __device__ __forceinline__ uint32_t byte_times8(uint32_t w, uint32_t three)
{
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
    else if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
    else if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
    return r;
}
// mask of the bytes [0, nb) of a dword, nb in 0..4
__device__ __forceinline__ uint32_t low_bytes(int nb) { return nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u); }
// bits [s, e) of a 32-bit word, 0 <= s, e <= 32
__device__ __forceinline__ uint32_t bit_range(int s, int e)
{
    const uint32_t hi = e >= 32 ? 0xffffffffu : ((1u << e) - 1u), lo = s >= 32 ? 0xffffffffu : ((1u << s) - 1u);
    return hi & ~lo;
}
} // namespace

// ---- exact per-position passes of phase A, entered only by a chunk that holds such a read.  Out of line on purpose: their
// register needs must not weigh on the main path (a call saves what it clobbers only when it is taken).
struct ExactQuality { int sv, svp, mq; };   // sum(raw - offset) over the read / over the kept window, max(raw - offset)
// patch = lead | trail << 8: terminal-N positions (< lead or >= trail) read as the offset (mask_quality_terminal_N)
template <int NP>
__device__ __noinline__ ExactQuality exact_quality_pass(const uint8_t *__restrict__ qual, const uint32_t v_off, const int len, const uint32_t patch,
                                                        const int a, const int n, const int in_off)
{
    const int lead = (int)(patch & 0xffu), trail = (int)(patch >> 8);
    ExactQuality r{0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < NP; ++k) {
        if (!__any(16 * k < len)) break;
        PackedBytes<4> tq;
#pragma unroll
        for (int i = 0; i < 4; ++i) tq.w[i] = 0u;
        if (16 * k < len) tq = *(const PackedBytes<4> *)(qual + (size_t)v_off + 16 * k);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = 16 * k + i;
            int v = (int)(int8_t)((tq.w[i >> 2] >> (8 * (i & 3))) & 0xffu) - in_off;
            if (p < lead || p >= trail) v = 0;
            if (p < len) {
                r.sv += v;
                r.mq = r.mq > v ? r.mq : v;
                if ((unsigned)(p - a) < (unsigned)n) r.svp += v;
            }
        }
    }
    return r;
}
struct ExactBases { uint32_t npre, npost; bool trip; }; // N (any case) in the read / in the kept window; dinucleotide filter
template <int NP>
__device__ __noinline__ ExactBases exact_base_pass(const uint8_t *__restrict__ seq, const uint32_t v_off, const int len, const int a, const int n,
                                                   const bool dinuc, const uint32_t dthr, const uint32_t cpk, const uint32_t snap_base,
                                                   const uint32_t t2_lds)
{
    ExactBases r{0u, 0u, false};
    uint32_t prev = 8u; // class 0..3 of the previous position if it is ACGT inside the window
    // (the lane's snapshot column is free by now and holds the 16 transition counters)
#pragma unroll
    for (int i = 0; i < 16; ++i) lds_store_u32(snap_base + (uint32_t)i * (uint32_t)(TPR_SROW_BYTES), 0u);
#pragma unroll 1
    for (int k = 0; k < NP; ++k) {
        if (!__any(16 * k < len)) break;
        PackedBytes<4> ts;
#pragma unroll
        for (int i = 0; i < 4; ++i) ts.w[i] = 0u;
        if (16 * k < len) ts = *(const PackedBytes<4> *)(seq + (size_t)v_off + 16 * k);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = 16 * k + i;
            const uint32_t byte = p < len ? (ts.w[i >> 2] >> (8 * (i & 3))) & 0xffu : 0u;
            const LdsPair e = *(lds_u2_ptr)(size_t)(byte * 8u + t2_lds);
            const bool inw = (unsigned)(p - a) < (unsigned)n;
            const uint32_t isn = (e.y >> 1) & 1u;
            r.npre += isn;
            r.npost += inw ? isn : 0u;
            const uint32_t cur = (e.x != 0u && inw) ? (uint32_t)__builtin_ctz(e.x) >> 3 : 8u;
            if (dinuc && cur < 4u && prev < 4u && cur != prev)
                __hip_atomic_fetch_add((lds_u32_mut)(size_t)(snap_base + (prev * 4u + cur) * (uint32_t)(TPR_SROW_BYTES)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = cur;
        }
    }
    if (dinuc) { // dc[X->Y] <= min(count X, count Y): only pairs whose two counts both reach dthr can trip
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (x != y) {
                    const uint32_t dc = lds_load_u32(snap_base + (uint32_t)(x * 4 + y) * (uint32_t)(TPR_SROW_BYTES));
                    r.trip = r.trip || (((cpk >> (8 * x)) & 0xffu) >= dthr && ((cpk >> (8 * y)) & 0xffu) >= dthr && dc >= dthr);
                }
    }
    return r;
}

// The two BWA_plus walks of phase A, one 4-position step per template instance so that the recursion ends as soon as no
// lane needs another step.  K = area * 256 + <position byte> is carried along (no per-position constants in registers);
// the bounds are kept relative to the current step (r* = bound - 4 * g) and re-based once per step.  A step is
// branch-free: "this lane visits position j" is the sign bit of a difference and gates the updates arithmetically, so
// the four positions form one basic block with no exec-mask round trips through the scalar unit.
template <int G, int ND, bool WINDOWED> struct Walk3 {
    // rlow: the walk's last position (at_least_scan == 0 after it); rend: end of the window; rthr: window start + n2
    template <bool ENDS>
    static __device__ __forceinline__ void step(const uint32_t w, const int qoff_v, const int q_v, int &K, int &best, int &rlow, const int rend, const int rthr)
    {
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            K -= 1;
            int live = rlow - (j + 1);                       // < 0: j >= rlow
            if (ENDS) live &= j - rend;                      // < 0: j < rend
            int rs = live & ~K;                              // < 0: visited and area >= 0 before this position
            if (WINDOWED || G == 0) rs &= rthr - j;          // < 0: j > rthr (always true above position 2 of an unwindowed read)
            rlow = rs < 0 ? j - 2 : rlow;
            const int t = qoff_v - (int)(int8_t)((w >> (8 * j)) & 0xffu);
            const int dq = t < q_v ? t : q_v;                // Q - quality_score() = min(Q, Q + offset - (signed char)raw)
            K += (dq & (live >> 31)) * 256;
            best = best > K ? best : K;                      // (a lane that is not visiting only counts K down: never a new maximum)
        }
    }
    static __device__ __forceinline__ void run(const uint32_t (&qd)[ND], const int qoff_v, const int q_v, int &K, int &best, int rlow, int rend, int rthr)
    {
        if (!__any(rlow <= 3)) return; // every lane is done (a lane whose window still ends below keeps rlow <= 3)
        if (__any(rend < 4)) step<true>(qd[G], qoff_v, q_v, K, best, rlow, rend, rthr); // some window ends inside or below this step
        else step<false>(qd[G], qoff_v, q_v, K, best, rlow, rend, rthr);
        Walk3<G - 1, ND, WINDOWED>::run(qd, qoff_v, q_v, K, best, rlow + 4, rend + 4, rthr + 4);
    }
};
template <int ND, bool WINDOWED> struct Walk3<-1, ND, WINDOWED> {
    static __device__ __forceinline__ void run(const uint32_t (&)[ND], int, int, int &, int &, int, int, int) {}
};
template <int G, int ND, bool WINDOWED> struct Walk5 {
    // rhigh: the walk's last position; rthr: final_pos_3 - n2 (resets need a position below it); rwa: window start
    static __device__ __forceinline__ void run(const uint32_t (&qd)[ND], const int qoff_v, const int q_v, int &K, int &best, int rhigh, int rthr, int rwa)
    {
        if (!__any(rhigh >= 0)) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            K -= 1;
            int live = (j - 1) - rhigh;                      // < 0: j <= rhigh
            if (WINDOWED) live &= rwa - (j + 1);             // < 0: j >= rwa
            const int rs = live & ~K & (j - rthr);           // < 0: visited, area >= 0 before, j < rthr
            rhigh = rs < 0 ? j + 2 : rhigh;
            const int t = qoff_v - (int)(int8_t)((qd[G] >> (8 * j)) & 0xffu);
            const int dq = t < q_v ? t : q_v;
            K += (dq & (live >> 31)) * 256;
            best = best > K ? best : K;
        }
        Walk5<G + 1, ND, WINDOWED>::run(qd, qoff_v, q_v, K, best, rhigh - 4, rthr - 4, rwa - 4);
    }
};
template <int ND, bool WINDOWED> struct Walk5<ND, ND, WINDOWED> {
    static __device__ __forceinline__ void run(const uint32_t (&)[ND], int, int, int &, int &, int, int, int) {}
};

template <int C, int NW, bool WINDOWED, int LPR = 8, bool EXT = false>
__global__ __launch_bounds__(NW * 64, tpr_bounds_waves(C, LPR, EXT)) void trim_tpr(
    const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
    const uint32_t *__restrict__ off, const uint32_t n_reads, const uint32_t *__restrict__ ad_sl,
    const uint16_t *__restrict__ ad_hit, uint2 *__restrict__ out, unsigned long long *__restrict__ rec_pre,
    unsigned long long *__restrict__ rec_post, uint64_t *__restrict__ counters, uint32_t *__restrict__ err)
{
    static_assert(LPR == 8 || LPR == 4, "phase B runs 8 or 4 lanes per read");
    using Cfg = RowCfg<C, LPR>;
    using T = TprCfg<C, LPR>;
    constexpr int D = Cfg::D, W = Cfg::W, NP = T::NP, ND = T::ND, NF = T::NF, NR = T::NR, NRX = NR ? NR : 1, NWORD = T::NWORD, NPOS = ND * 4;
    static_assert(!Cfg::HQ8 && NPOS <= 255, "positions must fit the low byte of the argmax keys");
    static_assert(ND >= 16, "the snapshot column doubles as the 16 transition counters of exact_base_pass");
    static_assert(T::SROW * 4 == TPR_SROW_BYTES, "row stride");
    static_assert(T::lds_dwords(NW) * 4 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hb = smem + Cfg::O_HB;
    const uint32_t *t_lc = smem + Cfg::O_TLC;
    const uint32_t *t_bm = smem + Cfg::O_TBM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int rl = lane & (LPR - 1);
    const int rowb = lane & (64 - LPR);
    const int wave = uni(tid >> 6);
    const int pbase = rl * C;

    for (int i = tid; i < Cfg::N_ZERO; i += NW * 64) smem[i] = 0u;
    for (int i = tid; i < 256; i += NW * 64) {
        const uint32_t w = P.base_tab[i];
        smem[Cfg::O_TBASE + i] = w;
        smem[T::O_T2 + 2 * i] = ((w >> BT_SHIFT(0)) & 1u) | (((w >> BT_SHIFT(1)) & 1u) << 8) | (((w >> BT_SHIFT(2)) & 1u) << 16) | (((w >> BT_SHIFT(3)) & 1u) << 24);
        smem[T::O_T2 + 2 * i + 1] = (w >> 31) | (((w >> BT_SHIFT(4)) & 1u) << 1);
    }
    for (int i = tid; i <= W; i += NW * 64) {
        smem[Cfg::O_TLC + i] = P.lc_thr[i];
        smem[Cfg::O_TAVGQ + i] = (uint32_t)P.avgq_min_v[i];
        smem[Cfg::O_TMAGIC + i] = P.div_magic[i];
    }
    for (int i = tid; i < Cfg::BMW * (C + 1); i += NW * 64) {
        const int nb = med3i((i / Cfg::BMW) - 4 * (i % Cfg::BMW), 0, 4);
        smem[Cfg::O_TBM + i] = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
    }
    uint32_t two = 2u, three = 3u;
    asm volatile("" : "+v"(two), "+v"(three)); // VGPR operands for the SDWA shifts
    if (tid == 0 && blockIdx.x == 0 && (uint32_t)(size_t)((lds_u32_ptr)smem) != 0u) atomicOr(err, 4u);
    __syncthreads();

    const uint32_t total_chunks = (n_reads + 63) >> 6;
    const uint32_t chunks_per_iter = gridDim.x * NW;
    const uint32_t n_iter = (total_chunks + chunks_per_iter - 1) / chunks_per_iter;
    constexpr uint32_t FLUSH_EVERY = 65535u / (NW * 64) > 0 ? 65535u / (NW * 64) : 1;
    constexpr uint32_t REG_FLUSH_EVERY = LPR == 8 ? 7 : 15; // 6-bit fields: 7 chunks x 8 (15 x 4) reads per row <= 63

    const int in_off = P.in_off, Q = P.Q;
    const uint32_t snap_base = (uint32_t)(T::O_SNAP + wave * T::SNAP_WAVE + lane) * 4u; // LDS byte address of this lane's column
    const uint32_t offb = ((uint32_t)in_off & 0xffu) * 0x01010101u;
    const bool swar_ok = in_off >= 0 && in_off <= 86; // else every read takes the exact quality pass
    uint32_t bpre[C], bpost[C];
#pragma unroll
    for (int j = 0; j < C; ++j) { bpre[j] = 0; bpost[j] = 0; }
    uint32_t any_err = 0;
    auto spill_base_regs = [&]() {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t x = bpre[j], y = bpost[j];
            if (x) {
#pragma unroll
                for (int c = 0; c < FAQCS_NBASE; ++c) {
                    const uint32_t v = ((x >> BT_SHIFT(c)) & 63u) | (((y >> BT_SHIFT(c)) & 63u) << 16);
                    if (v) atomicAdd(&hb[c * W + pbase + j], v);
                }
            }
            bpre[j] = 0; bpost[j] = 0;
        }
    };

#pragma unroll 1
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t chunk = (it * gridDim.x + blockIdx.x) * NW + wave;
        if (chunk < total_chunks) {
            const uint32_t base = chunk << 6;
            const uint32_t my = base + lane;
            const bool mine = my < n_reads;
            const uint32_t v_off = mine ? off[my] : 0u;
            const uint32_t v_len = mine ? off[my + 1] - v_off : 0u;
            const uint32_t v_sl = (WINDOWED && ad_sl && mine) ? ad_sl[my] : (v_len << 16);
            const uint32_t v_hit = (ad_hit && mine) ? ad_hit[my] : 0u;

            // ---- software prefetch of the row's read 0 for phase B ----------------------------------------
            PackedBytes<D> nseq; // (bases only: the qualities reach phase B through LDS)
            int n_len = __shfl((int)v_len, rowb);
            {
                const uint32_t o = (uint32_t)__shfl((int)v_off, rowb);
#pragma unroll
                for (int k = 0; k < D; ++k) nseq.w[k] = 0;
                if (pbase < n_len) nseq = *(const PackedBytes<D> *)(seq + (size_t)o + pbase);
            }

            // ================= phase A: one read per lane =====================================================
            ReadOutcome oc;
            uint32_t v_info, v_patch = 0; // what phase B needs: start | kept << 8 | valid << 16 | error << 17 | patch << 18 ; lead | trail << 8
            {
                const int len = (int)v_len;
                uint32_t sd[ND], qd[ND];
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    PackedBytes<4> ts, tq;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ts.w[i] = 0u; tq.w[i] = offb; }
                    if (16 * k < len) {
                        ts = *(const PackedBytes<4> *)(seq + (size_t)v_off + 16 * k);
                        tq = *(const PackedBytes<4> *)(qual + (size_t)v_off + 16 * k);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sd[4 * k + i] = ts.w[i]; qd[4 * k + i] = tq.w[i]; }
                }
                if (NR) {
                    PackedBytes<NRX> ts, tq;
#pragma unroll
                    for (int i = 0; i < NR; ++i) { ts.w[i] = 0u; tq.w[i] = offb; }
                    if (16 * NF < len) {
                        ts = *(const PackedBytes<NRX> *)(seq + (size_t)v_off + 16 * NF);
                        tq = *(const PackedBytes<NRX> *)(qual + (size_t)v_off + 16 * NF);
                    }
#pragma unroll
                    for (int i = 0; i < NR; ++i) { sd[4 * NF + i] = ts.w[i]; qd[4 * NF + i] = tq.w[i]; }
                }
                uint32_t blast = 0; // last base (a lane without a read must not touch the arena: its offset is not one)
                if (len) blast = (uint32_t)seq[(size_t)v_off + len - 1];
                // bytes past the read inside its last 16-byte piece: base 0 (class "none"), quality == offset (q = 0, adds 0 to the sums)
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    if (__any(len < 4 * k + 4 && 16 * (k >> 2) < len)) {
                        const uint32_t m = low_bytes(med3i(len - 4 * k, 0, 4));
                        sd[k] &= m;
                        qd[k] = (qd[k] & m) | (offb & ~m);
                    }
                }

                // ---- classes: A,T,C,G counts (8-bit fields) with a prefix snapshot per dword, upper-case N bits ----
                uint32_t cnt4 = 0, nub[NWORD];
#pragma unroll
                for (int w = 0; w < NWORD; ++w) nub[w] = 0;
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    lds_store_u32(snap_base + (uint32_t)k * (uint32_t)(TPR_SROW_BYTES), cnt4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ad = (j == 0 ? byte_times8<0>(sd[k], three) : j == 1 ? byte_times8<1>(sd[k], three)
                                             : j == 2 ? byte_times8<2>(sd[k], three) : byte_times8<3>(sd[k], three)) + (uint32_t)(T::O_T2 * 4);
                        const LdsPair e = *(lds_u2_ptr)(size_t)ad;
                        cnt4 += e.x;
                        nub[(4 * k + j) >> 5] = __builtin_amdgcn_alignbit(e.y, nub[(4 * k + j) >> 5], 1); // bit (p & 31) = upper-case N at p
                    }
                }
                if ((NPOS & 31) != 0) nub[NWORD - 1] >>= (32 - (NPOS & 31));
                const int nACGT = (int)__builtin_amdgcn_sad_u8(cnt4, 0u, 0u);
                int nup = 0;
#pragma unroll
                for (int w = 0; w < NWORD; ++w) nup += __builtin_popcount(nub[w]);
                const bool abn_seq = nACGT + nup != len; // a letter that is neither ACGT (any case) nor 'N'

                // ---- mask_quality_terminal_N (trim.cpp:1191-1216): upper-case 'N' runs at either end get Q0 -----
                const bool tn = len > 0 && ((nub[0] & 1u) != 0u || blast == 'N');
                if (__any(tn)) {
                    int lead = NPOS, trail = 0;
#pragma unroll
                    for (int w = NWORD - 1; w >= 0; --w) { // first position that is not N
                        const uint32_t x = ~nub[w];
                        lead = x ? 32 * w + __builtin_ctz(x) : lead;
                    }
                    lead = lead < len ? lead : len;
#pragma unroll
                    for (int w = 0; w < NWORD; ++w) { // 1 + last position that is not N
                        const uint32_t x = ~nub[w] & bit_range(0, med3i(len - 32 * w, 0, 32));
                        trail = x ? 32 * w + 32 - __builtin_clz(x) : trail;
                    }
                    if (!tn) { lead = 0; trail = len; }
                    v_patch = (uint32_t)lead | ((uint32_t)trail << 8);
#pragma unroll
                    for (int k = 0; k < ND; ++k) {
                        if (!__any(lead > 4 * k || (trail < 4 * k + 4 && 4 * k < len))) continue;
                        const uint32_t m1 = low_bytes(med3i(lead - 4 * k, 0, 4));   // positions < lead
                        const uint32_t m2 = ~low_bytes(med3i(trail - 4 * k, 0, 4)); // positions >= trail (past the read: already == offset)
                        const uint32_t m = m1 | m2;
                        qd[k] = (qd[k] & ~m) | (offb & m);
                    }
                }

                // ---- quality: sum and range check, four bytes per instruction --------------------------------
                // V = sum(raw - offset); a byte outside [offset, offset + 41] (or >= 128) sends the read to the exact pass
                uint32_t qsum = 0, qbad = 0, qor = 0;
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    qsum = __builtin_amdgcn_sad_u8(qd[k], 0u, qsum);
                    const uint32_t t = (qd[k] | 0x80808080u) - offb;                      // byte: 128 + raw - offset (raw < 128)
                    const uint32_t s = (t & 0x7f7f7f7fu) + 0x56565656u;                   // bit 7: raw - offset > 41
                    qbad |= ~t | s;                                                       // bit 7 clear in t: raw < offset
                    qor |= qd[k];
                }
                const bool badq = !swar_ok || (((qbad | qor) & 0x80808080u) != 0u);
                int V_pre = (int)qsum - NPOS * in_off;
                bool read_err = false;

                // ---- the window the reference trims: after the adapter pre-pass and --5end/--3end (trim.cpp:270-314) ----
                int wa = 0, wn = len;
                uint32_t flags = 0, filt = 0;
                if (WINDOWED && P.has_adapters) {
                    const int first = (int)(v_sl & 0xffffu), second = (int)(v_sl >> 16);
                    const bool mod = len != second;
                    wa = mod ? first : 0; wn = mod ? second : len;
                    flags = mod ? FAQCS_F_ADAPTER : 0u;
                }
                if (WINDOWED && P.trim5) {
                    const bool over = (int)P.trim5 > wn;
                    wa = over ? wa : wa + (int)P.trim5;
                    wn = over ? 0 : wn - (int)P.trim5;
                }
                if (WINDOWED && P.trim3) wn = (int)P.trim3 > wn ? 0 : wn - (int)P.trim3;

                // ---- BWA_plus (trim.cpp:714-793), walked as the reference walks it -----------------------------
                const int a5 = wn < 5 ? wn : 5, nn2 = wn < 2 ? wn : 2, wend = wa + wn;
                int qoff_v = Q + in_off, q_v = Q;
                asm volatile("" : "+v"(qoff_v), "+v"(q_v));
                // Q - quality_score(p) = min(Q, Q + offset - (signed char)raw)
                // 3' walk.  at_least_scan == 0 after position p  <=>  p == lowp: the walk covers min(5, n) positions, and a
                // reset at p (p > n2 and area >= 0 before p) moves its end to p - 2 (n2 == 2 whenever a reset can fire).
                // best = maxArea << 8 | position of the first maximum.
                int best = 255, K3 = NPOS;
                Walk3<ND - 1, ND, WINDOWED>::run(qd, qoff_v, q_v, K3, best, (wn > 0 ? wend - a5 : NPOS) - 4 * (ND - 1), wend - 4 * (ND - 1),
                                                 wa + nn2 - 4 * (ND - 1));
                const int S3 = best >> 8;
                const int fp3 = S3 > 0 ? (best & 255) - 1 - wa : wn - 1;
                // 5' walk (trim.cpp:752-779): resets need pos_5 < final_pos_3 - n2; the FIRST maximum wins (low byte = 255 - p)
                int best5 = 255, K5 = 256;
                if (!(EXT && P.protect5)) // --5trim_off (trim.cpp:752)
                    Walk5<0, ND, WINDOWED>::run(qd, qoff_v, q_v, K5, best5, wn > 0 ? wa + a5 - 1 : -1, wa + fp3 - nn2, wa);
                const int S5 = best5 >> 8;
                const int fp5 = S5 > 0 ? (255 - (best5 & 255)) + 1 - wa : 0;

                // ---- length filters and the kept window (trim.cpp:317-360) -------------------------------------
                int a = wa, n = wn;
                bool ret = mine;
                uint32_t qt_removed = 0;
                if (ret && (n < (int)P.min_len || n == 0)) { ret = false; filt = FAQCS_FILT_LENGTH_PRE; }
                if (ret) {
                    const int kept = fp3 <= fp5 ? 0 : fp3 - fp5 + 1;
                    if (kept != n) { qt_removed = (uint32_t)(n - kept); flags |= FAQCS_F_QUAL_TRIMMED; }
                    a += fp5;
                    n = kept;
                    if (n < (int)P.min_len || n == 0) { ret = false; filt = FAQCS_FILT_LENGTH_POST; }
                }

                // ---- poly-N (trim.cpp:363-371, :578-597): -n 2 = two adjacent upper-case N inside the kept window ----
                if (EXT && P.max_poly_n != 2u) { // -n 0: every read trips; -n 1: any upper-case N inside the kept window
                    uint32_t hit = 0;
#pragma unroll
                    for (int w = 0; w < NWORD; ++w) hit |= nub[w] & bit_range(med3i(a - 32 * w, 0, 32), med3i(a + n - 32 * w, 0, 32));
                    if (ret && (P.max_poly_n == 0u || hit != 0u)) { flags |= FAQCS_F_POLY_N_SEEN; ret = false; filt = FAQCS_FILT_POLY_N; }
                } else {
                    uint32_t pr[NWORD], anyp = 0; // bit e: N at e - 1 and at e
#pragma unroll
                    for (int w = 0; w < NWORD; ++w) {
                        pr[w] = nub[w] & ((nub[w] << 1) | (w ? nub[w - 1] >> 31 : 0u));
                        anyp |= pr[w];
                    }
                    if (__any(ret && anyp != 0u)) {
                        uint32_t hit = 0;
#pragma unroll
                        for (int w = 0; w < NWORD; ++w) hit |= pr[w] & bit_range(med3i(a + 1 - 32 * w, 0, 32), med3i(a + n - 32 * w, 0, 32));
                        if (ret && hit != 0u) { flags |= FAQCS_F_POLY_N_SEEN; ret = false; filt = FAQCS_FILT_POLY_N; }
                    }
                }

                // ---- base counts before / inside the kept window (trim.cpp:390-403, :810-875) ---------------------
                // prefix(x) = snapshot of dword x >> 2 plus the x & 3 bytes in front of x (one 4-byte load from the arena)
                auto prefix4 = [&](int x) -> uint32_t {
                    uint32_t c = lds_load_u32(snap_base + (uint32_t)(x < NPOS ? x >> 2 : 0) * (uint32_t)(TPR_SROW_BYTES));
                    c = x < NPOS ? c : cnt4; // (the whole read)
                    if (__any((x & 3) != 0)) {
                        uint32_t w = 0;
                        if (x & 3) w = ((const PackedBytes<1> *)(seq + (size_t)v_off + (x & ~3)))->w[0] & low_bytes(x & 3);
                        c += ((lds_u2_ptr)(size_t)(byte_times8<0>(w, three) + (uint32_t)(T::O_T2 * 4)))->x;
                        c += ((lds_u2_ptr)(size_t)(byte_times8<1>(w, three) + (uint32_t)(T::O_T2 * 4)))->x;
                        c += ((lds_u2_ptr)(size_t)(byte_times8<2>(w, three) + (uint32_t)(T::O_T2 * 4)))->x;
                    }
                    return c;
                };
                uint32_t c4post = cnt4;
                if (__any(ret && (a != 0 || n != len))) {
                    c4post = prefix4(a + n);
                    if (__any(a != 0)) c4post -= prefix4(a);
                }
                const uint32_t pA = cnt4 & 0xffu, pT = (cnt4 >> 8) & 0xffu, pC = (cnt4 >> 16) & 0xffu, pG = cnt4 >> 24;
                const uint32_t cA = c4post & 0xffu, cT = (c4post >> 8) & 0xffu, cC = (c4post >> 16) & 0xffu, cG = c4post >> 24;
                uint32_t pN = (uint32_t)(len - nACGT), cN = (uint32_t)n - (cA + cT + cC + cG); // (exact pass below when abn_seq)

                // ---- sum(raw - offset) over the kept window (trim.cpp:374, :553-576) ---------------------------
                int V_post;
                if (!WINDOWED) {
                    // all v == q here (no byte below the offset): sum q over the kept window from the two walks' areas
                    const int Tsum = len * Q - V_pre; // sum of (Q - q) over the read
                    V_post = n * Q - (Tsum - (S3 > 0 ? S3 : 0) - (S5 > 0 ? S5 : 0));
                } else {
                    uint32_t s = 0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) {
                        const uint32_t m = low_bytes(med3i(a + n - 4 * k, 0, 4)) & ~low_bytes(med3i(a - 4 * k, 0, 4));
                        s = __builtin_amdgcn_sad_u8(qd[k] & m, 0u, s);
                    }
                    V_post = (int)s - n * in_off;
                }
                // exact pass for a read with a raw quality outside [offset, offset + 41] (fastq.h:17-36: negative scores clamp
                // to 0 in the trimmers but not in the averages; > 41 aborts the run)
                if (__any(badq)) {
                    const ExactQuality xq = exact_quality_pass<NP>(qual, v_off, len, tn ? v_patch : ((uint32_t)len << 8), a, n, in_off);
                    if (badq) { V_pre = xq.sv; V_post = xq.svp; read_err = xq.mq > 41; }
                }

                // ---- average quality (trim.cpp:374-382) -------------------------------------------------------------
                if (EXT && P.avgq_on && ret && V_post < ((const int32_t *)(smem + Cfg::O_TAVGQ))[n]) { ret = false; filt = FAQCS_FILT_AVG_Q; }

                // ---- low-complexity filter (trim.cpp:405-513) ---------------------------------------------------
                bool lc_trip = false, dinuc = false;
                uint32_t dthr = 0;
                if (ret) {
                    const uint32_t thr = t_lc[n];
                    const uint32_t mthr = thr & 0xffffu;
                    dthr = thr >> 16;
                    lc_trip = cA >= mthr || cT >= mthr || cG >= mthr || cC >= mthr;
                    // dc[X->Y] <= min(count X, count Y): only pairs whose two counts both reach dthr can trip
                    dinuc = !lc_trip && ((cA >= dthr) + (cT >= dthr) + (cC >= dthr) + (cG >= dthr)) >= 2;
                }
                // exact per-position pass over the bases: N counts for reads with other letters, transition counts for
                // dinucleotide candidates (both rare; the lane's snapshot column is free by now and holds the 16 counters)
                if (__any(abn_seq || dinuc)) {
                    const uint32_t cpk = cA | (cT << 8) | (cC << 16) | (cG << 24);
                    const ExactBases xb = exact_base_pass<NP>(seq, v_off, len, a, n, dinuc, dthr, cpk, snap_base, (uint32_t)(T::O_T2 * 4));
                    if (abn_seq) { pN = xb.npre; cN = xb.npost; }
                    if (dinuc) lc_trip = lc_trip || xb.trip;
                }
                if (ret && lc_trip) { ret = false; filt = FAQCS_FILT_LOW_COMPLEXITY; }

                if (read_err) { any_err = 1; flags |= FAQCS_F_ERR_QUALITY; }
                oc.an = (uint32_t)a | ((uint32_t)n << 16);
                oc.fl = flags | (ret ? FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT) | (qt_removed << 20);
                oc.pAT = pA | (pT << 16); oc.pCG = pC | (pG << 16);
                oc.cAT = cA | (cT << 16); oc.cCG = cC | (cG << 16);
                oc.N = pN | (cN << 16);
                oc.Vpre = V_pre; oc.Vpost = V_post;
                v_info = (uint32_t)a | ((uint32_t)n << 8) | (ret ? 1u << 16 : 0u) | (read_err ? 1u << 17 : 0u);
                // the snapshots are spent: the column now carries this read's quality bytes (terminal-N runs already at the
                // offset, bytes past the read too) to the 8 lanes that accumulate it in phase B
#pragma unroll
                for (int k = 0; k < ND; ++k) lds_store_u32(snap_base + (uint32_t)k * (uint32_t)(TPR_SROW_BYTES), qd[k]);
            }

            // ================= phase B: 8 lanes per read, accumulate only ====================================
            const uint32_t qcol = (uint32_t)(T::O_SNAP + wave * T::SNAP_WAVE) * 4u; // LDS byte address of the wave's column block
#pragma unroll 1
            for (int t = 0; t < LPR; ++t) {
                if (base + (uint32_t)t >= n_reads) break; // wave-uniform: no row has a read left
                const int len = n_len;
                const bool act = base + (uint32_t)(rowb + t) < n_reads;
                uint32_t ws[D], wq[D];
#pragma unroll
                for (int k = 0; k < D; ++k) ws[k] = nseq.w[k];
                const uint32_t info = (uint32_t)__shfl((int)v_info, rowb + t);
                if (t + 1 < LPR) {
                    n_len = __shfl((int)v_len, rowb + t + 1);
                    const uint32_t o = (uint32_t)__shfl((int)v_off, rowb + t + 1);
#pragma unroll
                    for (int k = 0; k < D; ++k) nseq.w[k] = 0;
                    if (pbase < n_len) nseq = *(const PackedBytes<D> *)(seq + (size_t)o + pbase);
                }
                { // zero the base bytes past the end of the read (the last dword of a lane may over-read 1..3 bytes)
                    const int vb = med3i(len - pbase, 0, C);
                    const uint4 bm = *reinterpret_cast<const uint4 *>(t_bm + Cfg::BMW * vb);
                    uint32_t m[8] = {bm.x, bm.y, bm.z, bm.w, 0u, 0u, 0u, 0u};
                    if (D > 4) {
                        const uint4 bm2 = *reinterpret_cast<const uint4 *>(t_bm + Cfg::BMW * vb + 4);
                        m[4] = bm2.x; m[5] = bm2.y; m[6] = bm2.z; m[7] = bm2.w;
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) ws[k] &= m[k];
                }
                { // the lane's C quality bytes out of the owner's column: dwords pbase / 4 ..., shifted into place
                    const uint32_t col = qcol + (uint32_t)(rowb + t) * 4u;
                    uint32_t rq[D + 1];
#pragma unroll
                    for (int i = 0; i <= D; ++i) {
                        const int row = (pbase >> 2) + i;
                        rq[i] = lds_load_u32(col + (uint32_t)(row < ND ? row : ND - 1) * (uint32_t)(TPR_SROW_BYTES));
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) wq[i] = __builtin_amdgcn_alignbyte(rq[i + 1], rq[i], (uint32_t)(pbase & 3));
                }
                const int a = (int)(info & 0xffu), n = (int)((info >> 8) & 0xffu);
                const bool ret = ((info >> 16) & 1u) != 0u, read_err = ((info >> 17) & 1u) != 0u;
                uint32_t incf[C];
                BaseLookup<C, 0>::run((uint32_t)(Cfg::O_TBASE * 4), ws, two, incf);
                int q[C];
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int v = (int)(int8_t)((wq[j >> 2] >> (8 * (j & 3))) & 0xffu) - in_off;
                    q[j] = v < 0 ? 0 : v;
                }
                if (__any(read_err)) { // rare: keep the row's table indices in range, count nothing (fastq.h:31-33 aborts the run)
#pragma unroll
                    for (int j = 0; j < C; ++j) { q[j] = read_err ? 0 : q[j]; incf[j] = read_err ? 0u : incf[j]; }
                }
                // a position outside the read adds to quality column 0 (flush_block subtracts those) and class "none"
                const uint32_t counted = (act && !read_err) ? 1u : 0u;
                const uint32_t pb4 = 4u * (uint32_t)pbase;
                const uint32_t postm = ret ? range_mask<C>(a, a + n, pbase) : 0u;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    lds_add_u32(__umul24((uint32_t)q[j], (uint32_t)(W * 4)) + pb4 + (uint32_t)(Cfg::O_HQ * 4 + 4 * j),
                                counted | ((uint32_t)bit_m1(postm, j) & 0x10000u));
                    bpre[j] += incf[j];
                    bpost[j] += incf[j] & (uint32_t)bit_m1(postm, j);
                }
            }

            // ---- chunk epilogue: one read per lane ----------------------------------------------------------
            chunk_epilogue<LPR>(oc, mine, my, v_len, v_hit, lane, smem + Cfg::O_LEN, smem + Cfg::O_RQ, smem + Cfg::O_BQPRE,
                                smem + Cfg::O_BQPOST, smem + Cfg::O_FS, smem + Cfg::O_TMAGIC, out, rec_pre, rec_post, EXT && P.avgq_on != 0, 0u);
        }

        const bool block_flush = ((it + 1) % FLUSH_EVERY) == 0 || it + 1 == n_iter;
        if (((it + 1) % REG_FLUSH_EVERY) == 0 || block_flush) spill_base_regs();
        if (block_flush) flush_block<C, LPR, NW>(smem, counters, P.R, tid);
    }
    if (__any(any_err != 0) && lane == 0) atomicOr(err, 1u);
}

// ---------------------------------------------------------------------------------------------------------
// composition_histogram: update_base_statistics()'s composition part (trim.cpp:860-874) from the per-read
// records.  One thread per record; the block's LDS holds the whole 10 001 x 6 table as 16-bit counters
// (two per dword), flushed to the global u64 block before any of them can overflow.
// ---------------------------------------------------------------------------------------------------------
#ifndef FAQCS_COMP_U
#define FAQCS_COMP_U 4 /* records per thread and round */
#endif
template <int NT, bool WIDE>
__global__ __launch_bounds__(NT) void composition_histogram(const unsigned long long *__restrict__ rec_a, const unsigned long long *__restrict__ rec_b,
                                                            const uint32_t n, const float *__restrict__ comp_norm,
                                                            uint64_t *__restrict__ dst_a, uint64_t *__restrict__ dst_b /* counters + L.{pre,post}_comp */)
{
    constexpr int NE = FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND; // 60 006 16-bit counters
    constexpr int ND = (NE + 1) / 2;
    constexpr int U = FAQCS_COMP_U; // records per thread and round, fetched together: a round is one memory latency, not U
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    const int tid = threadIdx.x;
    // ONE launch folds both record arrays (pre- and post-trim): even blocks take the first, odd blocks the second (the table
    // fills the LDS of a CU, so the two cannot share one)
    const bool second = (blockIdx.x & 1u) != 0u;
    const unsigned long long *__restrict__ rec = second ? rec_b : rec_a;
    uint64_t *__restrict__ dst = second ? dst_b : dst_a;
    const uint32_t bid = blockIdx.x >> 1, nblk = gridDim.x >> 1; // (the grid is even)
    // the per-length factors behind the table, in LDS too: as a global load the factor sat between a record and its atomics,
    // a second memory latency per round (measured: 5 % of the trim launch the fold shares the GPU with)
    float *normt = reinterpret_cast<float *>(tab + ND);
    constexpr int NNORM = WIDE ? FAQCS_TAB_LEN + 1 : 512;
    for (int i = tid; i < ND; i += NT) tab[i] = 0;
    for (int i = tid; i < NNORM; i += NT) normt[i] = i <= FAQCS_TAB_LEN ? comp_norm[i] : 0.0f;
    __syncthreads();
    const uint32_t per_round = nblk * NT * U;
    const uint32_t rounds = (n + per_round - 1) / per_round;
    constexpr uint32_t FLUSH_EVERY = 65535u / (NT * U);
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i0 = (r * nblk + bid) * (NT * U) + tid;
        unsigned long long x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * NT;
            x[u] = 0; y[u] = 0;
            if (i < n) {
                if (WIDE) { const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(rec)[i]; x[u] = v.x; y[u] = v.y; }
                else x[u] = rec[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t nbin = 0xffffffffu; // this thread's N bin (none)
            if (x[u] & CR_VALID) {
                uint32_t len, cnt[5];
                if (WIDE) {
                    len = (uint32_t)(x[u] & 2047u);
                    cnt[0] = (uint32_t)(x[u] >> 11) & 2047u; cnt[1] = (uint32_t)(x[u] >> 22) & 2047u; cnt[2] = (uint32_t)(x[u] >> 33) & 2047u;
                    cnt[3] = (uint32_t)y[u] & 2047u; cnt[4] = (uint32_t)(y[u] >> 11) & 2047u;
                } else {
                    len = (uint32_t)(x[u] & 511u);
#pragma unroll
                    for (int k = 0; k < 5; ++k) cnt[k] = (uint32_t)(x[u] >> (9 + 9 * k)) & 511u;
                }
                const float norm = normt[len];
                uint32_t idx[6];
#pragma unroll
                for (int k = 0; k < 5; ++k) idx[k] = (uint32_t)__fmul_rn(norm, (float)cnt[k]); // :862-872
                idx[5] = idx[3] + idx[2];                                                                                  // :874 (G + C)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (k == 4) continue; // N: below
                    const uint32_t e = idx[k] * FAQCS_NCOMP_KIND + k;
                    atomicAdd(&tab[e >> 1], 1u << (16 * (e & 1u)));
                }
                nbin = idx[4] * FAQCS_NCOMP_KIND + 4;
            }
            // The N bin is the same for nearly every read (no N at all: bin 0): 64 lanes adding to ONE LDS address serialise, and
            // that one kind cost more than the other five together.  The lanes of a wave that hit bin 0 add their count once; a read
            // with N in it adds for itself (a loop over the distinct bins of the wave was measured: 3 % of the co-running trim launch).
            {
                constexpr uint32_t bin0 = 4u; // idx 0, kind 4
                const unsigned long long zero = __ballot(nbin == bin0);
                if (zero != 0ull && (int)(threadIdx.x & 63u) == __builtin_ctzll(zero)) atomicAdd(&tab[bin0 >> 1], (uint32_t)__popcll(zero) << (16 * (bin0 & 1u)));
                if (nbin != 0xffffffffu && nbin != bin0) atomicAdd(&tab[nbin >> 1], 1u << (16 * (nbin & 1u)));
            }
        }
        if (((r + 1) % FLUSH_EVERY) == 0 || r + 1 == rounds) {
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

// ---- launch wrappers ---------------------------------------------------------------------------------------
template <int C, int LPR, int NW, bool WINDOWED, bool GENERIC>
static hipError_t launch_trim_t(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                uint32_t n_reads, const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out,
                                unsigned long long *rec_pre, unsigned long long *rec_post, uint64_t *counters, uint32_t *err,
                                int n_cu, hipStream_t st)
{
    constexpr size_t lds = (size_t)RowCfg<C, LPR>::LDS_DWORDS * 4;
    static unsigned long long attr_done = 0;
    auto kern = trim_filter_accumulate<C, LPR, NW, WINDOWED, GENERIC>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const uint32_t chunks = (n_reads + 63) / 64;
    int blocks_per_cu = (int)((160 * 1024) / lds);
    // (variants whose per-position arrays do not fit 168 VGPRs run at 2 waves/SIMD rather than spill: the kernel is issue-bound)
    constexpr int minwaves = (LPR == 8 || C > 10) ? 2 : (FAQCS_TRIM_MINWAVES > 2 ? FAQCS_TRIM_MINWAVES : 2);
    const int by_waves = (4 * minwaves + NW - 1) / NW; // resident waves per CU the registers allow
    if (blocks_per_cu > by_waves) blocks_per_cu = by_waves;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint32_t grid = (chunks + NW - 1) / NW;
    const uint32_t cap = (uint32_t)(n_cu * blocks_per_cu);
    if (grid > cap) grid = cap;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), rec_pre, rec_post, counters, err);
    return hipGetLastError();
}

template <int C, int NW, bool WINDOWED, int LPR = 8, bool EXT = false>
static hipError_t launch_trim_tpr(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                  uint32_t n_reads, const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out,
                                  unsigned long long *rec_pre, unsigned long long *rec_post, uint64_t *counters, uint32_t *err,
                                  int n_cu, hipStream_t st)
{
    constexpr size_t lds = (size_t)TprCfg<C, LPR>::lds_dwords(NW) * 4;
    static unsigned long long attr_done = 0;
    auto kern = trim_tpr<C, NW, WINDOWED, LPR, EXT>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const uint32_t chunks = (n_reads + 63) / 64;
    int blocks_per_cu = (int)((160 * 1024) / lds);
    const int by_waves = (4 * tpr_waves_per_simd(C, LPR, EXT) + NW - 1) / NW;
    if (blocks_per_cu > by_waves) blocks_per_cu = by_waves;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint32_t grid = (chunks + NW - 1) / NW;
    const uint32_t cap = (uint32_t)(n_cu * blocks_per_cu);
    if (grid > cap) grid = cap;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), rec_pre, rec_post, counters, err);
    return hipGetLastError();
}

hipError_t faqcs_launch_trim_long(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off, uint32_t n_reads,
                                  const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out, uint64_t *counters, uint32_t *err,
                                  int n_cu, hipStream_t st, uint32_t *lead_trail, uint32_t max_len); // faqcs_trim_long_kernel.hip
hipError_t faqcs_launch_trim_lds(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                 uint32_t n_reads, uint32_t max_len, const uint32_t *ad_sl, const uint16_t *ad_hit,
                                 faqcs_read_result *out, unsigned long long *rec_pre, unsigned long long *rec_post,
                                 uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st, const uint8_t *tn_flags);

static thread_local const char *g_last_trim_kernel = "";
const char *faqcs_last_trim_kernel() { return g_last_trim_kernel; }
bool faqcs_trim_lds_tail_folded();
static thread_local bool g_last_trim_folded = false;
// the last faqcs_launch_trim() of this thread folded the composition records DevParams::fold_* named (a trim_lds variant with room for the table)
bool faqcs_last_trim_folded() { return g_last_trim_folded; }

hipError_t faqcs_launch_trim(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                             uint32_t n_reads, uint32_t max_len, const uint32_t *ad_sl, const uint16_t *ad_hit,
                             faqcs_read_result *out, unsigned long long *rec_pre, unsigned long long *rec_post,
                             uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st, const uint8_t *tn_flags)
{
    g_last_trim_folded = false;
    {   // trim_long (faqcs_trim_long_kernel.hip): a batch that holds a read of more than 1 024 bases; FAQCS_TRIM_LONG=1 sends every batch there (tests)
        const char *e_long = getenv("FAQCS_TRIM_LONG"); // (read per launch: the tests switch it inside one process)
        const bool force_long = e_long && atoi(e_long) != 0;
        if (max_len > FAQCS_FAST_READ_LENGTH || force_long) {
            g_last_trim_kernel = "trim_long";
            // (scratch: a word per read for the terminal-N runs; the composition record array is free, trim_long writes no records)
            return faqcs_launch_trim_long(P, seq, qual, off, n_reads, ad_sl, ad_hit, out, counters, err, n_cu, st, reinterpret_cast<uint32_t *>(rec_pre), max_len);
        }
    }
    {   // trim_lds (faqcs_trim_lds_kernel.hip): every byte from HBM once, through LDS; FAQCS_TRIM_LDS=0 switches it off
        static const bool lds_on = [] { const char *e = getenv("FAQCS_TRIM_LDS"); return !e || atoi(e) != 0; }();
        if (lds_on) {
            const hipError_t e = faqcs_launch_trim_lds(P, seq, qual, off, n_reads, max_len, ad_sl, ad_hit, out, rec_pre, rec_post, counters, err, n_cu, st, tn_flags);
            if (e != hipErrorNotSupported) { g_last_trim_kernel = "trim_lds"; g_last_trim_folded = e == hipSuccess && faqcs_trim_lds_tail_folded(); return e; }
        }
    }
    g_last_trim_kernel = "trim_filter_accumulate";
    const bool windowed = P.has_adapters || ((P.trim5 || P.trim3) && !P.qc_only);
    const bool generic = !(P.mode == FAQCS_MODE_BWA_PLUS && !P.protect5 && !P.qc_only && P.replace_q == 0 && !P.avgq_on &&
                           P.max_poly_n == 2 && P.dbg == 0);
#define FAQCS_TRIM_ARGS P, seq, qual, off, n_reads, ad_sl, ad_hit, out, rec_pre, rec_post, counters, err, n_cu, st
#define FAQCS_TRIM_CASE(C, NW)                                                                              \
    return windowed ? (generic ? launch_trim_t<C, 16, NW, true, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 16, NW, true, false>(FAQCS_TRIM_ARGS)) \
                    : (generic ? launch_trim_t<C, 16, NW, false, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 16, NW, false, false>(FAQCS_TRIM_ARGS))
    {   // 8 lanes per read: the headline shape (reads <= 160 bases, default option set); FAQCS_TRIM_LPR8=0 switches it off
        static const bool lpr8 = [] { const char *e = getenv("FAQCS_TRIM_LPR8"); return !e || atoi(e) != 0; }();
#define FAQCS_TRIM_CASE8(C)                                                                                 \
    return windowed ? (generic ? launch_trim_t<C, 8, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 8, FAQCS_TRIM_NW, true, false>(FAQCS_TRIM_ARGS)) \
                    : (generic ? launch_trim_t<C, 8, FAQCS_TRIM_NW, false, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 8, FAQCS_TRIM_NW, false, false>(FAQCS_TRIM_ARGS))
        // trim_tpr (round 1's two-phase kernel) is NOT part of the shipped library since round 6: every batch it took runs trim_lds, its
        // instantiations were 2.8 MB of a 4.8 MB library and most of every rebuild.  -DFAQCS_WITH_TRIM_TPR compiles them in again for A/B
        // runs (FAQCS_TRIM_LDS=0 then reaches them as in rounds 4-5); without it FAQCS_TRIM_LDS=0 goes straight to trim_filter_accumulate.
#ifdef FAQCS_WITH_TRIM_TPR
        {   // the two-phase kernel for the headline option set; FAQCS_TRIM_TPR=0 switches it off
            static const bool tpr = [] { const char *e = getenv("FAQCS_TRIM_TPR"); return !e || atoi(e) != 0; }();
            // EXT: the default set plus --5trim_off / --avg_q / -n 0 or 1 (compiled apart so that the default variants stay as they are)
            const bool ext = generic && P.mode == FAQCS_MODE_BWA_PLUS && !P.qc_only && P.replace_q == 0 && P.max_poly_n <= 2 && P.dbg == 0;
#define FAQCS_TRIM_CASE_TPR(C) \
    { g_last_trim_kernel = "trim_tpr"; \
    return ext ? (windowed ? launch_trim_tpr<C, tpr_nw(C, 8, true), true, 8, true>(FAQCS_TRIM_ARGS) : launch_trim_tpr<C, tpr_nw(C, 8, true), false, 8, true>(FAQCS_TRIM_ARGS)) \
               : (windowed ? launch_trim_tpr<C, tpr_nw(C), true>(FAQCS_TRIM_ARGS) : launch_trim_tpr<C, tpr_nw(C), false>(FAQCS_TRIM_ARGS)); }
            if (lpr8 && tpr && (!generic || ext) && max_len > 76 && max_len <= 104) FAQCS_TRIM_CASE_TPR(13);   // 2x100
            if (lpr8 && tpr && (!generic || ext) && max_len > 104 && max_len <= 152) FAQCS_TRIM_CASE_TPR(19);  // 2x150
            if (lpr8 && tpr && !generic && max_len > 152 && max_len <= 160) FAQCS_TRIM_CASE_TPR(20);
#undef FAQCS_TRIM_CASE_TPR
#define FAQCS_TRIM_CASE_TPR4(C) \
    { g_last_trim_kernel = "trim_tpr"; \
    return ext ? (windowed ? launch_trim_tpr<C, tpr_nw(C, 4), true, 4, true>(FAQCS_TRIM_ARGS) : launch_trim_tpr<C, tpr_nw(C, 4), false, 4, true>(FAQCS_TRIM_ARGS)) \
               : (windowed ? launch_trim_tpr<C, tpr_nw(C, 4), true, 4>(FAQCS_TRIM_ARGS) : launch_trim_tpr<C, tpr_nw(C, 4), false, 4>(FAQCS_TRIM_ARGS)); }
            {   // reads <= 76 bases: 4 lanes per read in phase B (2x75, 2x50)
                static const bool lpr4t = [] { const char *e = getenv("FAQCS_TRIM_LPR4"); return !e || atoi(e) != 0; }();
                if (lpr4t && tpr && (!generic || ext) && max_len > 0 && max_len <= 64) FAQCS_TRIM_CASE_TPR4(16);
                if (lpr4t && tpr && (!generic || ext) && max_len > 64 && max_len <= 76) FAQCS_TRIM_CASE_TPR4(19);
            }
#undef FAQCS_TRIM_CASE_TPR4
        }
#endif
        {   // 4 lanes per read (16 reads per wave): reads <= 76 bases (2x75, 2x50); FAQCS_TRIM_LPR4=0 switches it off
            static const bool lpr4 = [] { const char *e = getenv("FAQCS_TRIM_LPR4"); return !e || atoi(e) != 0; }();
#define FAQCS_TRIM_CASE4(C)                                                                                 \
    return windowed ? (generic ? launch_trim_t<C, 4, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 4, FAQCS_TRIM_NW, true, false>(FAQCS_TRIM_ARGS)) \
                    : (generic ? launch_trim_t<C, 4, FAQCS_TRIM_NW, false, true>(FAQCS_TRIM_ARGS) : launch_trim_t<C, 4, FAQCS_TRIM_NW, false, false>(FAQCS_TRIM_ARGS))
            if (lpr4 && max_len <= 64) FAQCS_TRIM_CASE4(16);
            if (lpr4 && max_len <= 76) FAQCS_TRIM_CASE4(19);
#undef FAQCS_TRIM_CASE4
        }
        if (lpr8 && max_len <= 64) FAQCS_TRIM_CASE8(8);
        if (lpr8 && max_len <= 104) FAQCS_TRIM_CASE8(13);   // 2x100
        if (lpr8 && max_len <= 128) FAQCS_TRIM_CASE8(16);
        if (lpr8 && max_len <= 152) FAQCS_TRIM_CASE8(19);   // 2x150: 152 position slots instead of 160
        if (lpr8 && max_len <= 160) FAQCS_TRIM_CASE8(20);
#undef FAQCS_TRIM_CASE8
    }
    // 16 lanes per read: 161..256 bases (and the A/B fallback of the 8-lane variants)
    if (max_len <= 208) FAQCS_TRIM_CASE(13, FAQCS_TRIM_NW);
    if (max_len <= 256) FAQCS_TRIM_CASE(16, FAQCS_TRIM_NW);
    // long reads: the whole wave on one read, one superset variant per width (MiSeq 2x300 -> C = 5)
    {   // two reads per wave (32 lanes each) up to 512 bases; FAQCS_TRIM_LPR32=0 falls back to the whole wave per read
        static const bool lpr32 = [] { const char *e = getenv("FAQCS_TRIM_LPR32"); return !e || atoi(e) != 0; }();
        if (lpr32 && max_len <= 320) // MiSeq 2x300: all four option variants like the short-read kernels
            return windowed ? (generic ? launch_trim_t<10, 32, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS) : launch_trim_t<10, 32, FAQCS_TRIM_NW, true, false>(FAQCS_TRIM_ARGS))
                            : (generic ? launch_trim_t<10, 32, FAQCS_TRIM_NW, false, true>(FAQCS_TRIM_ARGS) : launch_trim_t<10, 32, FAQCS_TRIM_NW, false, false>(FAQCS_TRIM_ARGS));
        if (lpr32 && max_len <= 512)
            return (windowed || generic) ? launch_trim_t<16, 32, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS)
                                         : launch_trim_t<16, 32, FAQCS_TRIM_NW, false, false>(FAQCS_TRIM_ARGS);
    }
    if (max_len <= 320) return launch_trim_t<5, 64, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS);
    if (max_len <= 512) return launch_trim_t<8, 64, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS);
    if (max_len <= 768) return launch_trim_t<12, 64, FAQCS_TRIM_NW, true, true>(FAQCS_TRIM_ARGS);
    if (max_len <= 1024) return launch_trim_t<16, 64, 8, true, true>(FAQCS_TRIM_ARGS); // 8 x 16 reads <= 255 per 8-bit cell
#undef FAQCS_TRIM_CASE
#undef FAQCS_TRIM_ARGS
    return hipErrorInvalidValue;
}

// wide: the two-word records of the long-read kernels (max_len > 256).  One launch for the pre- and the post-trim records.
hipError_t faqcs_launch_composition(const unsigned long long *rec_pre, const unsigned long long *rec_post, uint32_t n, bool wide,
                                    const float *comp_norm, uint64_t *dst_pre, uint64_t *dst_post, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    constexpr int NT = 1024;
    constexpr size_t lds = (size_t)((FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND + 1) / 2) * 4 + (size_t)(FAQCS_TAB_LEN + 1) * 4; // table + per-length factors
    static unsigned long long attr_done_w = 0, attr_done_n = 0;
    auto kern = wide ? composition_histogram<NT, true> : composition_histogram<NT, false>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, wide ? attr_done_w : attr_done_n); e != hipSuccess) return e;
    uint32_t per_array = (n + NT * FAQCS_COMP_U - 1) / (NT * FAQCS_COMP_U); // blocks one array can use
    if (per_array > (uint32_t)(n_cu / 2)) per_array = (uint32_t)(n_cu / 2); // (fewer, longer blocks are slower: 64 per array -6 %)
    if (per_array < 1) per_array = 1;
    hipLaunchKernelGGL(kern, dim3(2 * per_array), dim3(NT), lds, st, rec_pre, rec_post, n, comp_norm, dst_pre, dst_post);
    return hipGetLastError();
}
