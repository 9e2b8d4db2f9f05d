// faqcs_trim_lds_kernel.hip -- trim_lds: the trim / filter / accumulate pass with every byte read from HBM ONCE, as
// coalesced 16-byte LDS-DMA loads (global_load_lds_dwordx4), for reads of 77 ... 304 bases (gfx950, wave64).
//
// Replaces trim_read() and its helpers (trim.cpp:225-551, :553-597, :629-885, :1191-1216) like the other trim kernels;
// the accumulators, the block flush and the chunk epilogue are shared with them (faqcs_trim_common.h).
//
// (What follows describes the variant of the headline shape, 77 ... 152 bases: 8 lanes per read in the position-parallel passes, chunks of
// 64 reads.  Reads of 153 ... 252 / 253 ... 304 bases run the same kernel text with 16 lanes per read and chunks of 32 / 20 reads --
// template parameters LPR and RPC, see trim_lds's definition and DESIGN.md section 4.1e.)
// A wave owns a chunk of 64 consecutive reads.  Reads are packed back to back in the arenas, so the chunk is ONE
// contiguous span of <= 64 x W bytes per arena: the wave copies the span of the QUALITY arena into its 9.8 KB slot of LDS
// with <= 10 wave-wide DMA instructions (1 KB each, no VGPR round trip), works on it, then copies the span of the BASE
// arena into the same slot and works on that.  Inside a slot read i still starts at (offset[i] - span start): a lane reads
// "its" read with per-lane LDS addresses (dynamic indexing that a register-resident copy cannot give) -- as two ALIGNED
// dwords and one v_alignbyte per four bytes: a misaligned ds_read_b32 is legal on gfx950 but serialises the wave.
//
//   Q, one read per lane        terminal-N patch (in place, rare; whether a read starts / ends with N comes from the batch's
//                               terminal_n flags, or from two byte loads per read when the caller has none), a flat range check of the
//                               whole slot 16 bytes per lane, the two BWA_plus walks exactly as trim.cpp:714-793 states them -- every
//                               lane starts at ITS window end and reads the dword under its own cursor --, length filters
//   Q-B, 8 lanes per read       position x quality accumulation: one ds_add per base (20 positions per lane, the rows of a half wave
//                               rotated against each other: conflict-free, below), address = raw byte x row stride (v_mul_u32_u24 with
//                               an SDWA byte select) + lane base, data = one v_perm_b32 of two byte masks (position inside the read ->
//                               pre count, inside the kept window -> post count); the reads' quality sums fall out of the same bytes
//   S, 8 lanes per read         ONE pass over the staged bases: an 8-byte table entry per base carries the pre and the post increment
//                               of the position x base registers (6-bit fields per class); a base outside the kept window is looked up
//                               at (byte | 0x80), whose entry has the pre increment only.  The same increments, summed over the lane's
//                               19 positions and then over the 8 lanes of the read (DPP), ARE the read's base counts before / inside
//                               the window -- what a separate lane-per-read class pass (S-A, round 2) used to compute.  Two adjacent
//                               upper-case N (the default -n 2) are found by a SWAR test on the same registers, only in reads with >= 2 N
//   verdicts, one read per lane poly-N, average quality, low complexity (dinucleotide counts only for reads whose two commonest bases
//                               both reach the threshold) -> the read's verdict
//   epilogue, one read per lane result word, composition records, small histograms (equal cells of a chunk combined by one lane),
//                               FilterStat sums kept in registers across chunks (chunk_epilogue)
//
// The post-trim cells are added before a read can be vetoed (poly-N / low complexity / average quality): a vetoed read is rare,
// and its post cells are taken back by corrective passes over the chunk (S and Q-B with negative increments) afterwards.
//
// Work distribution: blocks claim GROUPS of NW consecutive chunks from a global counter, waves claim chunks of the block's groups
// from an LDS counter (a 4-entry ring of group ids), so a block that falls behind simply takes fewer chunks -- the static
// chunk -> wave map of round 2 ended every launch with a tail of idle CUs.  The offsets / DMA of a wave's NEXT chunk are issued
// under the epilogue of the current one.  Every FLUSH_CHUNKS chunks the block adds its LDS accumulators to ITS row of a partial-sum
// array in global memory (plain adds, no atomics: the row is private to the block) and fold_partials, launched behind the kernel,
// adds the rows to the u64 counter block.
//
// THE SLOT'S VALIDITY RULE (round 5: decided to keep it a rule, state its keepers here and CHECK it in the test suite, rather than pay two
// VALU per position in Q-B for an address that is forced into the table).  The LDS address of a Q-B add is (quality byte) x (row stride) +
// lane base even when the increment is zero -- a position behind the read or outside the kept window -- and an LDS add of zero is still a
// read-modify-write: with a byte that is not a quality (a base letter is "row" 64 ... 83 of a 42-row table) the add lands in another wave's
// slot and can put stale bytes back over that wave's LDS-DMA.  So EVERY byte a position-parallel quality pass can read -- the staged span,
// the <= 15 bytes of alignment slack in front of it, W + 20 bytes behind it, the 16-byte tails of padded rows -- holds a byte inside
// [offset, offset + 41] whenever such a pass runs.  The writes that keep it so, all of them:
//   1. stage(): the DMA of the QUALITY arena (bytes of the batch: the flat range check of Q-A flags anything outside [offset, offset + 41]
//      as FAQCS_E_QUALITY -- such a batch ends in that error and its counters are never used);
//   2. pad_behind_span(), called behind EVERY staging of qualities -- the first one of a chunk and the second one of the take-back pass (the
//      missing second call was round 4's race) -- writes the offset byte into the slack, the W + 20 bytes behind the span and, for padded
//      rows, behind the last row;
//   3. the terminal-N patch of Q-A writes the offset byte over quality bytes of the span (never outside it);
//   4. nothing else writes to a slot while it holds qualities: the BASE arena is staged into the slot only after the last quality pass of
//      the chunk, and a slot belongs to one wave.
// tests/test_gpu_slot_rule.py runs batches of every lane geometry through a build that counts the adds outside the matrix
// (-DFAQCS_LDS_DIAG_CHECK_QB_ADDR, built by __graft_entry__.build() as tests/_diag/libfaqcs_mi_qbchk.so): the count must be zero.
//
// Dispatch (faqcs_launch_trim_lds at the end of the file): every option set except --replace_to_N_q.  A chunk whose reads all have the
// same length, a multiple of 32 bases, is staged as padded rows instead of one span (dma_rows: LDS bank stride of the lane-per-read passes).
#include "faqcs_trim_common.h"

#include <stdlib.h>
#include <type_traits>

// dwords in flight per LDS wait of the two lane-per-read passes (tuned on MI355X)
#ifndef FAQCS_LDS_SUM_UNROLL
#define FAQCS_LDS_SUM_UNROLL 4
#endif
// reads per chunk of the 16-lanes-per-read variants (A/B on MI355X: see faqcs_launch_trim_lds)
#ifndef FAQCS_LDS16_RPC
#define FAQCS_LDS16_RPC 32
#endif
#ifndef FAQCS_LDS16_NW
#define FAQCS_LDS16_NW 12
#endif
#ifndef FAQCS_LDS_SA_UNROLL
#define FAQCS_LDS_SA_UNROLL 2
#endif

namespace {

// Q-B's own partition of the positions (8 lanes per read, C = 19): 20 per lane instead of 19, quality rows of 160 cells.  With rows that are a
// multiple of 32 cells the bank of a cell is its position mod 32 whatever the quality; lane rl of a read starts at position 20 rl
// (banks 0, 20, 8, 28, 16, 4, 24, 12: the multiples of 4) and the four reads of a half wave walk their 20 positions ROTATED by
// 0, 1, 2, 3 bytes, so the 32 lanes of one ds_add hit 32 different banks -- and four different positions: no two adds of an
// instruction meet on a bank or on a cell (the round-2 kernel lost half of its LDS-atomic time to such conflicts).
// 16 lanes per read (C = 16, reads of 161 ... 252 bases): a lane's 16 cells are followed by one cell of padding, so lane rl starts on
// cell 17 rl -- 16 different banks in rows of 288 cells -- and the two reads of a half wave are rotated by 0 and 1 bytes
// (17 rl + 1 = 17 rl' has no solution with both lanes below 16): conflict-free as well.
// 16 lanes per read, C = 19 (reads of 253 ... 304 bases: 2x300): 20 positions per lane in Q-B as in the 8-lane variant, 21 cells per lane, rows of 352.
constexpr int lds_cq(int C, int LPR = 8) { return C == 19 ? 20 : C; }
constexpr int lds_qstride(int C, int LPR = 8) { return LPR == 16 ? lds_cq(C, LPR) + 1 : lds_cq(C, LPR); } // cells from one lane's first position to the next lane's
constexpr int lds_wq(int C, int LPR = 8) { return (LPR == 16 || C == 19) ? (LPR * lds_qstride(C, LPR) + 31) / 32 * 32 : 0; } // (8 lanes, C = 19: 160; 4 lanes: 96)
constexpr int lds_nrot(int C, int LPR = 8) { return LPR == 16 ? 2 : (C == 19 ? 4 : 1); }   // reads of a half wave = byte rotations in use
// The longest read a variant takes.  Up to 252 bases the step index of a walk lives in the low byte of the argmax keys (codes 254 - step), a
// window's start and length in a byte each, a read's N count in 8 bits; the 304-base variant (lds_wide) has nine-bit codes and fields, the
// N counts in a register of their own and two-word composition records.
constexpr int lds_maxlen(int C, int LPR = 8) { return LPR * C <= 252 ? LPR * C : (LPR * C == 256 ? 252 : LPR * C); }
constexpr bool lds_wide(int C, int LPR = 8) { return lds_maxlen(C, LPR) > 252; }

template <int C, int NW, int LPR = 8, int RPC = 64> struct LdsCfg {
    using Row = RowCfg<C, LPR, lds_wq(C, LPR)>;
    static constexpr int CQ = lds_cq(C, LPR);                  // positions per lane in Q-B
    static constexpr int QSTRIDE = lds_qstride(C, LPR);        // cells per lane in a quality row
    static constexpr int NROT = lds_nrot(C, LPR);
    static constexpr bool ROT = NROT > 1;
    static constexpr int W = Row::W;
    static constexpr int MAXLEN = lds_maxlen(C, LPR);
    static constexpr int ND = (W + 3) / 4;                     // dwords of the longest read
    static constexpr int NP = (W + 15) / 16;                   // 16-byte pieces (the out-of-line exact passes)
    static constexpr int NWORD = (ND * 4 + 31) / 32;
    static constexpr int O_T2 = (Row::LDS_DWORDS + 3) & ~3;    // [256][2] (exact passes of rare reads) A,T,C,G one-hot in 8-bit fields ; isN(upper) | isN(any) << 1
    static constexpr int O_T3 = O_T2 + 512;                    // [256][2] S: 6-bit count fields, pre ; post (entry[b | 0x80]: b outside the kept window, pre only)
    static constexpr int O_CTR = O_T3 + 512;                   // [8] the block's chunk queue: [0] next unclaimed chunk number, [1] the block's chunk
                                                               // count once known, [4..7] ring: group number << 20 | group id
    static constexpr int O_TBQ = O_CTR + 8;                    // (ROT) [NROT][CQ + 1][8] rotated byte masks "positions < vb" of Q-B
    static constexpr int O_STG = O_TBQ + (ROT ? NROT * (CQ + 1) * 8 : 0);
    static constexpr int PADLEN = MAXLEN / 32 * 32;            // the longest read of a chunk staged as padded rows (dma_rows): L + 16 bytes each
    static constexpr int STG_BYTES = RPC * (PADLEN + 16 > MAXLEN ? PADLEN + 16 : MAXLEN) + 32; // one arena's span of a chunk (RPC reads) + 16-byte alignment slack
    static constexpr int STG_DW = (STG_BYTES + 15) / 16 * 4;
    static constexpr int TAIL_PAD = W + 64 > 256 ? (W + 64) / 4 : 64; // dwords: a lane may read W + 20 bytes from the start of the span's last read
    static constexpr int lds_dwords() { return O_STG + NW * STG_DW + TAIL_PAD; }
};

typedef uint32_t LdsPair2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) LdsPair2 *lds_u2c_ptr;
typedef __attribute__((address_space(3))) uint8_t *lds_u8_mut;
typedef const __attribute__((address_space(3))) uint8_t *lds_u8_ptr;
struct __attribute__((packed, aligned(1))) U32u { uint32_t w; };
typedef const __attribute__((address_space(3))) U32u *lds_u32u_ptr;

__device__ __forceinline__ uint32_t lds_ld(uint32_t byte_offset) { return *(lds_u32_ptr)(size_t)byte_offset; }
// four bytes at ANY byte address: two aligned dwords and a funnel shift (a misaligned ds_read_b32 is legal on gfx950 but
// serialises the wave: measured 64 LDS clocks per instruction)
__device__ __forceinline__ uint32_t lds_ld_any(uint32_t byte_offset)
{
    const uint32_t a4 = byte_offset & ~3u;
    return __builtin_amdgcn_alignbyte(lds_ld(a4 + 4u), lds_ld(a4), byte_offset & 3u);
}
__device__ __forceinline__ uint32_t lds_ld_u8(uint32_t byte_offset) { return (uint32_t) * (lds_u8_ptr)(size_t)byte_offset; }
__device__ __forceinline__ void lds_st_u8(uint32_t byte_offset, uint32_t v) { *(lds_u8_mut)(size_t)byte_offset = (uint8_t)v; }

template <int K> __device__ __forceinline__ uint32_t byte_x8(uint32_t w, uint32_t three)
{
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
    else if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
    else if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
    return r;
}
// byte K of w times m (m < 2^24), one instruction
template <int K> __device__ __forceinline__ uint32_t byte_mul(uint32_t w, uint32_t m)
{
    uint32_t r;
    if (K == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(m), "v"(w));
    else if (K == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(m), "v"(w));
    else if (K == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(m), "v"(w));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(m), "v"(w));
    return r;
}
__device__ __forceinline__ uint32_t low_bytes_(int nb) { return nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u); }
__device__ __forceinline__ uint32_t bit_range_(int s, int e)
{
    const uint32_t hi = e >= 32 ? 0xffffffffu : ((1u << e) - 1u), lo = s >= 32 ? 0xffffffffu : ((1u << s) - 1u);
    return hi & ~lo;
}

// One dword (four positions) of a BWA_plus walk, for the lanes that are still walking; the others are switched off in EXEC
// and never come back.
//   it     step index of the dword's first position (wave-uniform)
//   c0     (it + 1) * (Q + offset) * 256 + 254 - it ; dc = (Q + offset) * 256 - 1: key = (area << 8) + code with
//          area = a + (steps so far) * (Q + offset) (a = minus the sum of the raw bytes), code = 254 - step index
//   rb     (last step index of the walk) - it: position j of the dword is visited while rb >= j; a reset at j (area >= 0
//          before the step, step index < rlim) makes it j + 2; the caller's view moves on by 4 per dword
//   DESC   the walk visits the dword's bytes 3, 2, 1, 0 (3' walk) or 0, 1, 2, 3 (5' walk)
// Per position: v_cmpx (alive), 2 x v_cmp + s_or (no reset), v_cndmask (rb), v_sub_sdwa (area), v_lshl_add (key), v_max.
template <bool DESC, int CB>
__device__ __forceinline__ void walk_step(const uint32_t w, const int it, const int c0, const int dc, int &rb, const int rlim, int &a, int &K, int &best)
{
    unsigned long long sv, t;
    int s = uni(it), c = uni(c0);
#define FAQCS_WALK_POS(J, J2, B)                                                                                   \
    "v_cmpx_le_i32 vcc, " #J ", %[rb]\n\t"                                                                         \
    "v_cmp_gt_i32 vcc, 0, %[K]\n\t"                                                                                \
    "v_cmp_le_i32 %[t], %[rlim], %[s]\n\t"                                                                         \
    "s_or_b64 vcc, vcc, %[t]\n\t"                                                                                  \
    "v_cndmask_b32 %[rb], " #J2 ", %[rb], vcc\n\t"                                                                 \
    "v_sub_u32_sdwa %[a], %[a], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B "\n\t"  \
    "v_lshl_add_u32 %[K], %[a], %[cb], %[c]\n\t"                                                                   \
    "v_max_i32 %[best], %[best], %[K]\n\t"                                                                         \
    "s_add_i32 %[s], %[s], 1\n\t"                                                                                  \
    "s_add_i32 %[c], %[c], %[dc]\n\t"
    if (DESC)
        asm volatile("s_mov_b64 %[sv], exec\n\t" FAQCS_WALK_POS(0, 2, 3) FAQCS_WALK_POS(1, 3, 2) FAQCS_WALK_POS(2, 4, 1) FAQCS_WALK_POS(3, 5, 0)
                     "s_mov_b64 exec, %[sv]\n\t"
                     "v_add_u32 %[rb], -4, %[rb]"
                     : [rb] "+v"(rb), [a] "+v"(a), [K] "+v"(K), [best] "+v"(best), [s] "+s"(s), [c] "+s"(c), [sv] "=&s"(sv), [t] "=&s"(t)
                     : [w] "v"(w), [rlim] "v"(rlim), [dc] "s"(dc), [cb] "n"(CB)
                     : "vcc");
    else
        asm volatile("s_mov_b64 %[sv], exec\n\t" FAQCS_WALK_POS(0, 2, 0) FAQCS_WALK_POS(1, 3, 1) FAQCS_WALK_POS(2, 4, 2) FAQCS_WALK_POS(3, 5, 3)
                     "s_mov_b64 exec, %[sv]\n\t"
                     "v_add_u32 %[rb], -4, %[rb]"
                     : [rb] "+v"(rb), [a] "+v"(a), [K] "+v"(K), [best] "+v"(best), [s] "+s"(s), [c] "+s"(c), [sv] "=&s"(sv), [t] "=&s"(t)
                     : [w] "v"(w), [rlim] "v"(rlim), [dc] "s"(dc), [cb] "n"(CB)
                     : "vcc");
#undef FAQCS_WALK_POS
}

// One arena's span of the chunk -> the wave's LDS slot: <= NI wave-wide 1 KB DMA loads from 16-byte aligned addresses.
// g = arena + (span start rounded down to 16 bytes), nbytes = bytes from there to the span's end.
template <int NI>
__device__ __forceinline__ void dma_span(const uint8_t *g, const uint32_t nbytes, uint32_t *slot, const int lane)
{
    // (written so that nothing per-lane depends on i: the lane's address once, i x 1 024 in the instruction's offset field or a scalar
    // operand -- otherwise the compiler keeps NI lane-dependent values alive across the whole kernel and spills them)
    const uint8_t *gl = g + lane * 16;
    const uint32_t l16 = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if ((uint32_t)(i * 1024) >= nbytes) break; // wave-uniform
        if (l16 < nbytes - (uint32_t)(i * 1024))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gl + i * 1024),
                                             (__attribute__((address_space(3))) void *)(slot + i * 256), 16, 0, 0);
    }
}

// The same for a chunk of n_rows reads that all have L bases, L a multiple of 32: read r goes to row r of L + 16 bytes.  With the reads
// back to back, equal lengths of 4 x 32 k bytes put "dword d of my read" of EVERY lane on one LDS bank (2x128: each ds_read of the
// lane-per-read walks took 32 passes); rows of L / 4 + 4 dwords spread them over 8 banks.  Each lane of a DMA instruction names its own
// global address, so the padded layout costs one instruction more per span: unit u = 64 i + lane of the slot is unit u mod U of row u / U
// (U = L / 16 + 1 units of 16 bytes per row; the last unit of a row holds the 16 bytes behind the read -- the head of the next one).
template <int NI>
__device__ __forceinline__ void dma_rows(const uint8_t *g, const uint32_t L, const uint32_t n_rows, uint32_t *slot, const int lane)
{
    const uint32_t U = L / 16u + 1u, total = n_rows * U;
    const uint32_t M = (65536u + U - 1u) / U; // u / U == (u * M) >> 16 for u < 4 096 (U <= 16)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if ((uint32_t)(i * 64) >= total) break; // wave-uniform
        const uint32_t u = (uint32_t)(i * 64 + lane);
        if (u < total) {
            const uint32_t r = (u * M) >> 16, col = u - r * U;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)(r * L + col * 16u)),
                                             (__attribute__((address_space(3))) void *)(slot + i * 256), 16, 0, 0);
        }
    }
}

// in_off bytes into the 16-byte alignment slack in front of a staged span and into the W + 20 bytes behind it (see where it is called)
template <int W, int STG_BYTES>
__device__ __forceinline__ void pad_behind_span(const uint32_t slot_b, const uint32_t span, const uint32_t slack, const int lane, const uint32_t in_off)
{
    if ((uint32_t)lane < slack) lds_st_u8(slot_b + (uint32_t)lane, in_off); // (the first read's first dword starts in the slack)
    const uint32_t o0 = span + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < (W + 20 + 63) / 64; ++i)
        if (lane < W + 20 - 64 * i && o0 < (uint32_t)(STG_BYTES - 64 * i)) lds_st_u8((slot_b + o0) + (uint32_t)(64 * i), in_off);
}

// ---- rare exact passes over the GLOBAL arenas (entered only by a chunk that holds such a read; out of line) -----------
struct ExactQ { int sv, svp, mq; };   // sum(raw - offset) over the read / over the kept window, max(raw - offset)
// patch = lead | trail << 16: terminal-N positions (< lead or >= trail) read as the offset (mask_quality_terminal_N)
__device__ __noinline__ ExactQ exact_quality(const uint8_t *__restrict__ qual, const uint32_t v_off, const int len, const uint32_t patch,
                                             const int a, const int n, const int in_off, const bool need)
{
    const int lead = (int)(patch & 0xffffu), trail = (int)(patch >> 16);
    ExactQ r{0, 0, 0};
#pragma unroll 1
    for (int p = 0; __any(need && p < len); ++p) {
        if (need && p < len) {
            int v = (int)(int8_t)qual[(size_t)v_off + p] - in_off;
            if (p < lead || p >= trail) v = 0;
            r.sv += v;
            r.mq = r.mq > v ? r.mq : v;
            if ((unsigned)(p - a) < (unsigned)n) r.svp += v;
        }
    }
    return r;
}
struct ExactB { uint32_t npre, npost; bool trip; }; // N (any case) in the read / in the kept window; dinucleotide filter
__device__ __noinline__ ExactB exact_bases(const uint8_t *__restrict__ seq, const uint32_t v_off, const int len, const int a, const int n,
                                           const bool need, const bool dinuc, const uint32_t dthr)
{
    ExactB r{0u, 0u, false};
    uint32_t prev = 4u, dmax = 0; // class 0..3 of the previous window position if it is ACGT
    uint32_t dc0 = 0, dc1 = 0, dc2 = 0, dc3 = 0; // 16 transition counters, 8 bits each (a window holds <= 160 transitions... saturating is enough: see below)
#pragma unroll 1
    for (int p = 0; __any(need && p < len); ++p) {
        if (need && p < len) {
            const uint32_t b = (uint32_t)seq[(size_t)v_off + p];
            const bool inw = (unsigned)(p - a) < (unsigned)n;
            const uint32_t u = b & 0xdfu;
            const uint32_t isn = u == 'N' ? 1u : 0u;
            r.npre += isn;
            r.npost += inw ? isn : 0u;
            uint32_t cur = u == 'A' ? 0u : u == 'T' ? 1u : u == 'C' ? 2u : u == 'G' ? 3u : 4u;
            if (!inw) cur = 4u;
            if (dinuc && cur < 4u && prev < 4u && cur != prev) {
                const uint32_t k = prev * 4u + cur, sh = 8u * (k & 3u);
                uint32_t &d = (k >> 2) == 0 ? dc0 : (k >> 2) == 1 ? dc1 : (k >> 2) == 2 ? dc2 : dc3;
                uint32_t c = (d >> sh) & 0xffu;
                c = c < 255u ? c + 1u : c; // (a count of 255 already exceeds every threshold: dthr <= 161)
                d = (d & ~(0xffu << sh)) | (c << sh);
                dmax = dmax > c ? dmax : c;
            }
            prev = cur;
        }
    }
    r.trip = dinuc && dmax >= dthr;
    return r;
}

} // namespace

// waves per block = slots of 64 x MAXLEN bytes next to the accumulators in 160 KB: 12 x 9.8 KB + 37 KB (8 lanes per read), 6 x 16.2 KB + 63 KB (16)
constexpr int lds_waves(int C, int LPR = 8, int RPC = 64) { return LPR == 4 ? 12 : LPR == 16 ? (RPC == 64 ? 6 : (C == 19 ? 12 : FAQCS_LDS16_NW)) : (C <= 19 ? 12 : 8); }

// LDS accumulators -> a row of global memory that belongs to THIS block and THIS flush, as plain coalesced 16-byte stores of the
// cells as they are (pre count in the low, post count in the high half-word); fold_partials, launched behind the trim kernel, adds
// the rows of all blocks to the u64 counter block.  History: the direct flush of faqcs_trim_common.h costs one 64-bit global atomic
// per non-zero counter and block -- 256 blocks x ~14 000 counters, each on a sector of its own, all at the end of the launch: 15 % of
// the kernel time; one row per block with no-return atomic adds (first form of this function) still sent 2 atomics per non-zero
// cell and block to the memory side, all blocks at once at the end of a launch -- the chip retires 13.5 G of them per second.
template <int C, int LPR, int NW>
__device__ __noinline__ void flush_block_partial(uint32_t *smem, uint32_t *__restrict__ row, const int tid)
{
    using Cfg = RowCfg<C, LPR, lds_wq(C, LPR)>;
    constexpr int N4 = (Cfg::N_ZERO + 3) / 4; // (the dwords behind N_ZERO belong to the base table: copied along, never read back)
    static_assert(N4 * 4 <= FAQCS_PARTIAL_ROW, "partial-sum row");
    __syncthreads();
    for (int i = tid; i < N4; i += NW * 64) {
        const uint4 v = reinterpret_cast<const uint4 *>(smem)[i];
        reinterpret_cast<uint4 *>(row)[i] = v;
        if (4 * i + 3 < Cfg::N_ZERO) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
        else for (int k = 4 * i; k < Cfg::N_ZERO; ++k) smem[k] = 0u;
    }
    __syncthreads();
}

// One thread per LDS accumulator cell i and sixteenth of the rows: the sum over the rows the blocks flushed -> the counter block (the index
// mapping of flush_block in faqcs_trim_common.h).  Only one thread adds to a given counter and kernels on one stream do not overlap,
// but the add stays atomic: another context's kernels may share the block in a caller's design.
template <int C, int LPR>
__global__ __launch_bounds__(1024) void fold_partials(const uint32_t *__restrict__ partials, const uint32_t *__restrict__ rows_used, const uint32_t n_blocks,
                                                      uint64_t *__restrict__ counters, const faqcs_layout lay, uint32_t *__restrict__ g_next)
{
    using Cfg = RowCfg<C, LPR, lds_wq(C, LPR)>;
    constexpr int W = Cfg::W;
    constexpr int CQ = lds_cq(C, LPR), QSTRIDE = lds_qstride(C, LPR);
    if (blockIdx.x == 0 && threadIdx.x == 0) *g_next = 0u; // (the trim kernel's group counter: zero between launches)
    __shared__ unsigned long long part[2][16][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63); // 64 cells x 16 interleaved sets of rows per block
    unsigned long long lo = 0, hi = 0;
    if (i < Cfg::N_ZERO) {
        const bool halves = i < Cfg::O_BQPRE; // two 16-bit counters per cell (the other cells are one 32-bit number)
        for (uint32_t b = threadIdx.x >> 6; b < n_blocks; b += 16) { // (a block's rows: one or two per launch, 8 at most)
            const uint32_t cnt = rows_used[b];
            const uint32_t *row = partials + (size_t)b * FAQCS_PARTIAL_FLUSHES * FAQCS_PARTIAL_ROW + i;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t x = row[(size_t)k * FAQCS_PARTIAL_ROW];
                if (halves) { lo += x & 0xffffu; hi += x >> 16; } else lo += x;
            }
        }
    }
    part[0][threadIdx.x >> 6][threadIdx.x & 63] = lo;
    part[1][threadIdx.x >> 6][threadIdx.x & 63] = hi;
    __syncthreads();
    if (threadIdx.x >= 64 || i >= Cfg::N_ZERO) return;
    lo = 0; hi = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { lo += part[0][g][threadIdx.x]; hi += part[1][g][threadIdx.x]; }
    if (!(lo | hi)) return;
    // the counter block's layout: the one faqcs_counters_layout() made (DevParams::lay), not a restatement of it
    const uint64_t filter_stats = lay.filter_stats, pre_read_qhist = lay.pre_read_qhist, pre_base_qhist = lay.pre_base_qhist,
                   post_read_qhist = lay.post_read_qhist, post_base_qhist = lay.post_base_qhist, pre_len_hist = lay.pre_len_hist,
                   post_len_hist = lay.post_len_hist, pre_qual = lay.pre_qual, post_qual = lay.post_qual, pre_base = lay.pre_base,
                   post_base = lay.post_base;
    const uint32_t R = lay.max_read_length;
    auto add = [&](uint64_t idx, unsigned long long v) { if (v) atomicAdd((unsigned long long *)(counters + idx), v); };
    if (i < Cfg::O_HB) { // position x quality: pre in the low, post in the high half-word
        const uint32_t q = (uint32_t)(i - Cfg::O_HQ) / Cfg::WQ, cell = (uint32_t)(i - Cfg::O_HQ) % Cfg::WQ;
        // (a lane's CQ cells may be followed by padding, see lds_qstride: such cells and the ones past the last lane's stay zero)
        const uint32_t p = QSTRIDE == CQ ? cell : ((cell % QSTRIDE < (uint32_t)CQ && cell / QSTRIDE < (uint32_t)LPR) ? cell / QSTRIDE * CQ + cell % QSTRIDE : R);
        if (p < R) { add(pre_qual + (uint64_t)p * FAQCS_NQ + q, lo); add(post_qual + (uint64_t)p * FAQCS_NQ + q, hi); }
    } else if (i < Cfg::O_LEN) {
        const uint32_t c = (uint32_t)(i - Cfg::O_HB) / W, p = (uint32_t)(i - Cfg::O_HB) % W;
        if (p < R) { add(pre_base + (uint64_t)p * FAQCS_NBASE + c, lo); add(post_base + (uint64_t)p * FAQCS_NBASE + c, hi); }
    } else if (i < Cfg::O_RQ) {
        const uint32_t l = (uint32_t)(i - Cfg::O_LEN);
        if (l <= R) { add(pre_len_hist + l, lo); add(post_len_hist + l, hi); }
    } else if (i < Cfg::O_BQPRE) {
        const uint32_t k = (uint32_t)(i - Cfg::O_RQ);
        if (k < FAQCS_NQ) { add(pre_read_qhist + k, lo); add(post_read_qhist + k, hi); }
    } else if (i < Cfg::O_BQPOST) { // 32-bit cells: the two halves of one number
        add(pre_base_qhist + (uint32_t)(i - Cfg::O_BQPRE), lo);
    } else if (i < Cfg::O_FS) {
        add(post_base_qhist + (uint32_t)(i - Cfg::O_BQPOST), lo);
    } else {
        const uint32_t k = (uint32_t)(i - Cfg::O_FS);
        if (k < FAQCS_NUM_STAT) add(filter_stats + k, lo);
    }
}

// RPC = reads per chunk: 64 (one per lane in the lane-per-read passes) or 32 -- half the lanes idle there, but a slot of half the size: reads of
// 161 ... 252 bases get 12 waves per CU instead of 6.  The position-parallel passes run TPR = RPC x LPR / 64 steps per chunk; the lane that owns a
// read in the lane-per-read passes is lane t of the row of LPR lanes that works on it in step t.
// a variant whose block owns at least the 122 KB the composition fold needs (the table of 16-bit counters, the per-length factors)
template <int C, int NW, int LPR, int RPC> constexpr bool lds_tail_folds_v =
    LdsCfg<C, NW, LPR, RPC>::lds_dwords() >= (FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND + 1) / 2 + 512 + 8 && lds_maxlen(C, LPR) <= 256; // (one-word records)
template <int C, int NW, bool WINDOWED, bool EXT, int LPR, int RPC>
__global__ __launch_bounds__(NW * 64, NW / 4) void trim_lds(
    const DevParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
    const uint32_t *__restrict__ off, const uint32_t n_reads, const uint32_t *__restrict__ ad_sl,
    const uint16_t *__restrict__ ad_hit, uint2 *__restrict__ out, unsigned long long *__restrict__ rec_pre,
    unsigned long long *__restrict__ rec_post, uint64_t *__restrict__ counters, uint32_t *__restrict__ err,
    const uint8_t *__restrict__ tn_flags)
{
    static_assert(LPR == 8 || (LPR == 16 && (C == 16 || C == 19)) || (LPR == 4 && (C == 13 || C == 19)), "lanes per read");
    using Cfg = RowCfg<C, LPR, lds_wq(C, LPR)>;
    static_assert(RPC <= 64 && (RPC * LPR) % 64 == 0, "reads per chunk: the same number for every row of LPR lanes");
    using T = LdsCfg<C, NW, LPR, RPC>;
    using RW = RowOps<LPR>;
    constexpr int TPR = RPC * LPR / 64;               // reads a row of LPR lanes works on per chunk
    constexpr int D = Cfg::D, W = Cfg::W, ND = T::ND, NWORD = T::NWORD, NPOS = ND * 4, BMW = Cfg::BMW;
    constexpr int NI = (T::STG_BYTES + 1023) / 1024;
    constexpr bool WIDE = lds_wide(C, LPR);           // reads past 252 bases
    constexpr int AB = WIDE ? 9 : 8;                  // bits of a window start / length in the words handed to the position-parallel passes
    constexpr uint32_t AM = (1u << AB) - 1u;
    constexpr int FB = 2 * AB;                        // ... their flags: post << FB | counted << (FB + 1) | chk << (FB + 2)
    constexpr int CB = WIDE ? 9 : 8, CMAX = (1 << CB) - 1; // bits of a walk's step code inside the argmax keys: code = CMAX - 1 - step, CMAX = no step yet
    static_assert(!Cfg::HQ8 && T::MAXLEN <= CMAX - 1, "step indices must fit the code field of the argmax keys");
    static_assert(T::lds_dwords() * 4 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hb = smem + Cfg::O_HB;
    const uint32_t *t_lc = smem + Cfg::O_TLC;
    const uint32_t *t_bm = smem + Cfg::O_TBM;

#ifdef FAQCS_LDS_STAMPS
    const unsigned long long clk_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int rl = lane & (LPR - 1);
    const int rowb = lane & (64 - LPR);
    const int wave = uni(tid >> 6);
    const int pbase = rl * C;
    const bool owner = RPC == 64 || rl < TPR;                                            // this lane stands for a read in the lane-per-read passes
    const uint32_t ridx = RPC == 64 ? (uint32_t)lane : (uint32_t)((lane / LPR) * TPR + rl); // ... for read `ridx` of the chunk

    for (int i = tid; i < Cfg::N_ZERO; i += NW * 64) smem[i] = 0u;
    for (int i = tid; i < 256; i += NW * 64) {
        const uint32_t w = P.base_tab[i];
        smem[T::O_T2 + 2 * i] = ((w >> BT_SHIFT(0)) & 1u) | (((w >> BT_SHIFT(1)) & 1u) << 8) | (((w >> BT_SHIFT(2)) & 1u) << 16) | (((w >> BT_SHIFT(3)) & 1u) << 24);
        smem[T::O_T2 + 2 * i + 1] = (w >> 31) | (((w >> BT_SHIFT(4)) & 1u) << 1);
    }
    for (int i = tid; i < 256; i += NW * 64) { // S: entry[b] = (pre, post) increments of the ASCII byte b; entry[b | 0x80] = (pre, 0): b outside the kept window
        // entries 128..255: a base OUTSIDE the kept window, looked up at (byte ^ 0x88) -- bit 7 says "outside", bit 3 moves the entry 64 bytes on:
        // the in-window entries of A, C, G, T, N sit on banks 2, 6, 14, 8, 28 (8-byte entries), their out-of-window twins 1 024 bytes further on
        // would sit on the SAME banks (a half wave that mixes both -- the lanes at a trimmed 3' end -- pays a two-way conflict per look-up);
        // with the extra 64 bytes they sit on banks 18, 22, 30, 24, 12.
        const uint32_t f = P.base_tab[(i & 128) ? ((i ^ 0x88) & 127) : i] & BT_FIELDS; // (a byte >= 0x80 in the INPUT is no base: such a chunk is patched and redone, see the S pass)
        smem[T::O_T3 + 2 * i] = f;
        smem[T::O_T3 + 2 * i + 1] = i < 128 ? f : 0u;
    }
    for (int i = tid; i <= W; i += NW * 64) {
        smem[Cfg::O_TLC + i] = P.lc_thr[i];
        smem[Cfg::O_TAVGQ + i] = (uint32_t)P.avgq_min_v[i];
        smem[Cfg::O_TMAGIC + i] = P.div_magic[i];
    }
    for (int i = tid; i < BMW * (C + 2); i += NW * 64) {
        const int nb = med3i((i / BMW) - 4 * (i % BMW), 0, 4);
        smem[Cfg::O_TBM + i] = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
    }
    constexpr int CQ = T::CQ;                         // Q-B: positions per lane (LdsCfg)
    constexpr bool ROT = T::ROT;
    static_assert((CQ + 3) / 4 == D && (!ROT || CQ % 4 == 0), "Q-B works on the same number of dwords per lane");
    static_assert(T::QSTRIDE * LPR <= Cfg::WQ, "quality rows");
    const int pbase_q = rl * CQ;
    const uint32_t rot = ROT ? (uint32_t)((lane / LPR) & (T::NROT - 1)) : 0u; // this read's byte rotation inside a half wave
    if (ROT) { // rotated byte masks: byte i of row (rot, vb) <-> position (i + rot) mod CQ of the lane, 0xff when that is < vb
        for (int i = tid; i < T::NROT * (CQ + 1) * 8; i += NW * 64) {
            const int r_ = i / ((CQ + 1) * 8), vb_ = (i / 8) % (CQ + 1), k_ = i % 8;
            uint32_t w_ = 0;
            for (int b_ = 0; b_ < 4; ++b_)
                if (4 * k_ + b_ < CQ && (4 * k_ + b_ + r_) % CQ < vb_) w_ |= 0xffu << (8 * b_);
            smem[T::O_TBQ + i] = w_;
        }
    }
    uint32_t three = 3u, wx4 = (uint32_t)(Cfg::WQ * 4);
    asm volatile("" : "+v"(three), "+v"(wx4)); // VGPR operands for the SDWA instructions
    if (tid == 0 && blockIdx.x == 0 && (uint32_t)(size_t)((lds_u32_ptr)smem) != 0u) atomicOr(err, 4u);
    const uint32_t total_chunks = (n_reads + RPC - 1) / RPC;
    const uint32_t n_groups = (total_chunks + NW - 1) / NW;
    uint32_t *g_next = err + 8; // groups handed out beyond the first gridDim.x (zero between launches: fold_partials resets it)
    if (tid == 0) {
        smem[T::O_CTR] = (uint32_t)NW; // the first NW chunks of the block are taken by wave number
        smem[T::O_CTR + 1] = 0xffffffffu;
        smem[T::O_CTR + 4] = blockIdx.x; // group number 0 of the block: group blockIdx.x
        smem[T::O_CTR + 5] = 0xffffffffu; smem[T::O_CTR + 6] = 0xffffffffu; smem[T::O_CTR + 7] = 0xffffffffu;
        const uint32_t g1 = gridDim.x + atomicAdd(g_next, 1u);
        if (g1 < n_groups) smem[T::O_CTR + 5] = (1u << 20) | g1; else smem[T::O_CTR + 1] = (uint32_t)NW;
    }
    __syncthreads();

    // ---- which chunks.  The reads are cut into groups of NW consecutive 64-read chunks.  A block starts with group blockIdx.x and
    // takes further groups from a counter in global memory (one atomic per NW chunks): a block that starts late -- the previous
    // launch's composition fold still holds its CU -- or runs on a slower CU simply takes fewer groups (with a fixed share per block
    // the last block finished 16 % after the first, and the fold beside the next launch cost 8 % of it).  INSIDE a block the waves
    // claim chunks one at a time from a counter in LDS: chunk number c of the block = chunk c % NW of the block's group number c / NW.
    // The id of group number L + 1 is fetched by the wave that claims the first chunk of group number L, a whole group ahead of its
    // first use, and published in a four-entry ring in LDS (tagged with the group number).  When the global counter runs out, the
    // block's chunk count becomes known instead.
    // flush k (k = 1, 2, ...) comes before the block's chunk k * FLUSH_CHUNKS, the last one after its last chunk: the 16-bit halves of
    // the LDS cells take FLUSH_CHUNKS x 64 <= 65535 increments in between (chunks are claimed in order, so a wave that holds a chunk
    // >= k * FLUSH_CHUNKS waits at flush k while exactly the chunks below it are being finished)
#ifdef FAQCS_LDS_TEST_FLUSH_CHUNKS // (test build: flush every few chunks, so that a small launch goes through many flushes and fills every block's rows)
    constexpr uint32_t FLUSH_CHUNKS = (uint32_t)(FAQCS_LDS_TEST_FLUSH_CHUNKS) / NW * NW;
#else
    constexpr uint32_t FLUSH_CHUNKS = 65535u / RPC / NW * NW;
#endif
    constexpr uint32_t MAX_GROUPS = (uint32_t)FAQCS_PARTIAL_FLUSHES * FLUSH_CHUNKS / NW; // groups a block takes at most: one flush row per FLUSH_CHUNKS chunks
    constexpr uint32_t REG_FLUSH_EVERY = 63 / TPR; // 6-bit fields: 7 chunks x 8 reads (3 x 16) per row <= 63
    constexpr uint32_t NO_CHUNK = 0xffffffffu;
    auto lds_word = [&](const int i) { return uniu(*(volatile const __attribute__((address_space(3))) uint32_t *)(size_t)(uint32_t)((T::O_CTR + i) * 4)); };
    // the global chunk of the block's chunk number c_ (NO_CHUNK: past the block's last chunk); waits for the group's id if need be
    auto chunk_of = [&](const uint32_t c_) -> uint32_t {
        const uint32_t L = c_ / NW;
        for (;;) {
            if (c_ >= lds_word(1)) return NO_CHUNK;
            const uint32_t e = lds_word(4 + (int)(L & 3u));
            if ((e >> 20) == (L & 0xfffu)) return (e & 0xfffffu) * NW + c_ % NW;
            __builtin_amdgcn_s_sleep(8);
        }
    };

    const int in_off = P.in_off, Q = P.Q;
    uint32_t *slot = smem + T::O_STG + wave * T::STG_DW;
    const uint32_t slot_b = (uint32_t)(T::O_STG + wave * T::STG_DW) * 4u; // LDS byte address of the wave's slot
    const uint32_t offb = ((uint32_t)in_off & 0xffu) * 0x01010101u;
    const bool swar_ok = in_off >= 0 && in_off <= 86; // else every read takes the exact quality pass
    const uint32_t hq_lane = (uint32_t)(Cfg::O_HQ * 4 + rl * T::QSTRIDE * 4) + rot * 4u - (uint32_t)in_off * (uint32_t)(Cfg::WQ * 4); // + raw byte * WQ * 4 = the cell
    // (ROT) rotated byte i stands for position (i + rot) mod CQ: the last three wrap around for some rotations
    const uint32_t *t_bmq = ROT ? smem + T::O_TBQ + rot * (uint32_t)((CQ + 1) * 8) : t_bm; // Q-B's mask rows of this lane
    constexpr int BMQ = ROT ? 8 : BMW, VQ = ROT ? CQ : C + 1;
    uint32_t bpre[C], bpost[C];
#pragma unroll
    for (int j = 0; j < C; ++j) { bpre[j] = 0; bpost[j] = 0; }
    uint32_t any_err = 0;
    // the register fields -> the block's position x base cells.  Straight-line: one LDS add per field, no test per lane (the
    // branchy form -- skip a zero field -- cost 23 000 clocks per call: 95 divergent regions with an LDS round trip each)
    FsAcc fs_acc; // FilterStat sums of the chunks since the last spill (7 chunks x 64 reads x 152 bases per wave: inside the packed fields)
    auto spill_base_regs = [&]() {
        // the 8 (4) rows of the wave hold the same positions: each row starts with another class, so that an add meets at most one other
        // row on its cell instead of seven
        // (from a lane number the compiler cannot see through: these ten values are needed once in seven chunks -- hoisted out of the chunk loop
        // they would occupy ten registers for the whole kernel)
        uint32_t sh_c[FAQCS_NBASE], ad_c[FAQCS_NBASE], lane_o = (uint32_t)lane;
        asm volatile("" : "+v"(lane_o));
        const uint32_t hb_lane_o = (uint32_t)(Cfg::O_HB * 4) + (lane_o & (uint32_t)(LPR - 1)) * (uint32_t)(C * 4);
#pragma unroll
        for (int c = 0; c < FAQCS_NBASE; ++c) {
            uint32_t cp = (uint32_t)c + (lane_o / (uint32_t)LPR) % FAQCS_NBASE;
            cp = cp >= FAQCS_NBASE ? cp - FAQCS_NBASE : cp;
            sh_c[c] = cp * 6u;
            ad_c[c] = hb_lane_o + cp * (uint32_t)(W * 4);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t x = bpre[j], y = bpost[j];
#pragma unroll
            for (int c = 0; c < FAQCS_NBASE; ++c) {
                const uint32_t v = ((x >> sh_c[c]) & 63u) | (((y >> sh_c[c]) & 63u) << 16);
                lds_add_u32(ad_c[c] + (uint32_t)(j * 4), v);
            }
            bpre[j] = 0; bpost[j] = 0;
        }
        fs_acc_flush(fs_acc, lane, smem + Cfg::O_FS);
    };

    // ---- the position-parallel passes (8 lanes per read).  What a lane fetches from LDS for one read: six aligned dwords
    // that hold its C bytes and three rows of the byte-mask table (positions inside the read / in front of the kept
    // window's end / in front of its start).  Fetched one read AHEAD of its use so that the LDS latency is covered.
    // i0 = slot offset of the read | len << 16 ; i1 = a | n << AB | post << FB | counted << (FB + 1)
    struct RawB { uint32_t r[D + 1], mv[D], mh[D], ml[D]; };
    auto load_b = [&](const uint32_t i0, const uint32_t i1, RawB &x) {
        const int len = (int)(i0 >> 16), a = (int)(i1 & AM), n = (int)((i1 >> AB) & AM);
        const bool post = ((i1 >> FB) & 1u) != 0u, counted = ((i1 >> (FB + 1)) & 1u) != 0u;
        const int vb = counted ? med3i(len - pbase_q, 0, VQ) : 0; // (a read that is not counted: no byte of it is)
        const int lo = post ? med3i(a - pbase_q, 0, VQ) : 0, hi = post ? med3i(a + n - pbase_q, 0, VQ) : 0;
#ifdef FAQCS_LDS_DIAG_LINEAR_LOADS // (diagnostic build, wrong results: what the bank conflicts of these per-lane loads cost)
        const uint32_t qa = slot_b + (uint32_t)lane * 4u;
#pragma unroll
        for (int k = 0; k <= D; ++k) x.r[k] = lds_ld(qa + 256u * (uint32_t)k);
#else
        const uint32_t qa = (slot_b + (i0 & 0xffffu) + (uint32_t)pbase_q) & ~3u;
#pragma unroll
        for (int k = 0; k <= D; ++k) x.r[k] = lds_ld(qa + 4u * (uint32_t)k);
#endif
#pragma unroll
#ifdef FAQCS_LDS_DIAG_UNIFORM_MASKROWS // (diagnostic build, wrong results: every lane fetches the same mask row -- a broadcast, no conflicts)
        for (int k = 0; k < D; ++k) { x.mv[k] = t_bmq[BMQ * VQ + k]; x.mh[k] = t_bmq[BMQ * VQ + k]; }
#else
        for (int k = 0; k < D; ++k) { x.mv[k] = t_bmq[BMQ * vb + k]; x.mh[k] = t_bmq[BMQ * hi + k]; }
#endif
        // row 0 of the mask table is all zeros, and a kept window that starts at the read's first base is the common case: the third
        // row is fetched only when some read of this step is trimmed at its 5' end
        if (__any(lo != 0)) {
#pragma unroll
            for (int k = 0; k < D; ++k) x.ml[k] = t_bmq[BMQ * lo + k];
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) x.ml[k] = 0u;
        }
    };
    // position x quality cells: DATA = +1 pre / +1 post per base (undo: -1 post, for a read S-A vetoed afterwards).
    // A byte past the read belongs to the next read of the span or to the pad behind it (valid quality bytes both: its
    // zero increment stays inside the table).
#ifdef FAQCS_LDS_DIAG_CHECK_QB_ADDR
    unsigned long long *dbg_qb_bad = reinterpret_cast<unsigned long long *>(err + 16) + 3; // debug words 3 (adds outside the matrix) and 4 (steps checked)
#endif
    uint32_t qb_sum = 0; // (Q-B, `sum` steps) this lane's read: the sum of its raw quality bytes
    auto quality_cells = [&](const RawB &x, const uint32_t i0, const uint32_t i1, auto undo_t, const int t, const bool sum) {
        constexpr bool undo = decltype(undo_t)::value;
        const uint32_t sh = ((i0 & 0xffffu) + (uint32_t)pbase_q + slot_b) & 3u;
        uint32_t cm[D], im[D], wq[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            cm[k] = !undo ? x.mv[k] & 0x01010101u : 0u; // (load_b: no byte of a read that is not counted is inside it)
            im[k] = (x.mh[k] ^ x.ml[k]) & (undo ? 0xffffffffu : 0x01010101u);
            wq[k] = __builtin_amdgcn_alignbyte(x.r[k + 1], x.r[k], sh);
        }
        if (ROT) { // the lane's CQ bytes rotated by `rot` (the masks come rotated from their table)
            uint32_t w0 = wq[0];
#pragma unroll
            for (int k = 0; k < D; ++k) wq[k] = __builtin_amdgcn_alignbyte(k + 1 < D ? wq[k + 1] : w0, wq[k], rot);
        }
        if (!undo && sum) { // (wave-uniform) the read's quality sum: the lane's bytes inside the read, then the 8 lanes of the read
            uint32_t qs = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                uint32_t m = x.mv[k];
                if (4 * k + 4 > CQ) m &= low_bytes_(CQ - 4 * k); // (the unrotated masks cover C + 1 positions)
                qs = __builtin_amdgcn_sad_u8(wq[k] & m, 0u, qs);
            }
            qs = (uint32_t)RW::all_sum((int)qs);
            qb_sum = (rl == t) ? qs : qb_sum;
        }
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const uint32_t b = (uint32_t)(j & 3);
            // byte 0 = cm.byte[b] (pre, lo16), bytes 2 (and 3 when undoing: 0xffff = -1 in the hi16) = im.byte[b] (post)
            const uint32_t sel = undo ? (((b) << 24) | ((b) << 16) | 0x0c0cu) : ((0x0cu << 24) | (b << 16) | (0x0cu << 8) | (4u + b));
            const uint32_t data = __builtin_amdgcn_perm(cm[j >> 2], im[j >> 2], sel);
            const uint32_t ad = ((j & 3) == 0 ? byte_mul<0>(wq[j >> 2], wx4) : (j & 3) == 1 ? byte_mul<1>(wq[j >> 2], wx4)
                                 : (j & 3) == 2 ? byte_mul<2>(wq[j >> 2], wx4) : byte_mul<3>(wq[j >> 2], wx4)) +
                                ((ROT && j >= CQ - (T::NROT - 1) && rot >= (uint32_t)(CQ - j)) ? hq_lane - (uint32_t)(CQ * 4) : hq_lane);
#ifdef FAQCS_LDS_DIAG_CHECK_QB_ADDR // (diagnostic build: the slot's validity rule -- every Q-B add, zero increments included, inside the quality matrix)
            {
                const uint32_t a_ = ad + 4u * (uint32_t)j;
                if (a_ < (uint32_t)(Cfg::O_HQ * 4) || a_ >= (uint32_t)((Cfg::O_HQ + Cfg::HQ) * 4)) atomicAdd(dbg_qb_bad, 1ull);
                else if (lane == 0 && j == 0) atomicAdd(dbg_qb_bad + 1, 1ull);
            }
#endif
#ifdef FAQCS_LDS_QB_NOCONFLICT // (diagnostic build, wrong results: every lane on a bank of its own -- what the conflicts of these adds cost)
            lds_add_u32((ad & 0x3u) + (uint32_t)(Cfg::O_HQ * 4) + (uint32_t)lane * 4u + 256u * (uint32_t)j, data);
#elif !defined(FAQCS_LDS_NO_QB_ATOMICS)
            lds_add_u32(ad + 4u * (uint32_t)j, data);
#else
            asm volatile("" :: "v"(ad), "v"(data));
#endif
        }
    };
    // ---- S: position x base in registers (6-bit count fields), AND the read's own base counts, from ONE 8-byte table entry per base:
    // .x = the pre-trim increment, .y = the post-trim increment.  A base outside the kept window is looked up at (byte | 0x80), whose
    // entry carries the pre increment only (input bytes >= 0x80 are no bases: a chunk that holds one is undone, patched and redone).
    // The lane's 19 increments are summed on the side (<= 19 per field), the 8 lanes of the read are added up by DPP (fields
    // widened to 12 bits after the first step: a read has up to 152 bases) and the lane whose turn it is (rl == t) keeps the
    // totals: what S-A used to compute with a second lookup per base in the lane-per-read layout.
    // MODE 0: count ; 1: take back the post increments of the reads flagged in i1 (vetoed after counting) ; 2: take back everything
    uint32_t tot_pe = 0, tot_po = 0, tot_ce = 0, tot_co = 0; // this lane's read: pre A | C << 12 | N << 24, pre T | G << 12, post ...
    uint32_t tot_pn = 0;                                      // (WIDE) pre N | post N << 16
    uint32_t seen7 = 0;                                       // OR of every counted base byte of the chunk (bit 7: abnormal input)
    bool pairhit = false;                                     // this lane's read: two adjacent upper-case N inside the kept window
    auto base_step = [&](const int t, const uint32_t i0, const uint32_t i1, auto mode_t) {
        constexpr int MODE = decltype(mode_t)::value;
        const int len = (int)(i0 >> 16), a = (int)(i1 & AM), n = (int)((i1 >> AB) & AM);
        const bool post = ((i1 >> FB) & 1u) != 0u, counted = ((i1 >> (FB + 1)) & 1u) != 0u, chk = ((i1 >> (FB + 2)) & 1u) != 0u;
        const int vb = counted ? med3i(len - pbase, 0, C + 1) : 0; // (a read that is not counted: every byte reads as "past the read")
        // the kept window as the table lookups see it: empty for a read that is not kept (every base outside: pre increments only)
        const int lo = med3i(a - pbase, 0, C + 1), hi = (post || chk) ? med3i(a + n - pbase, 0, C + 1) : lo;
        const uint32_t qa = (slot_b + (i0 & 0xffffu) + (uint32_t)pbase) & ~3u;
        const uint32_t sh = ((i0 & 0xffffu) + (uint32_t)pbase + slot_b) & 3u;
        uint32_t r[D + 1], w[D], inw[D];
#ifdef FAQCS_LDS_DIAG_LINEAR_LOADS
#pragma unroll
        for (int k = 0; k <= D; ++k) r[k] = lds_ld(slot_b + (uint32_t)lane * 4u + 256u * (uint32_t)k);
#else
#pragma unroll
        for (int k = 0; k <= D; ++k) r[k] = lds_ld(qa + 4u * (uint32_t)k);
#endif
#pragma unroll
#ifdef FAQCS_LDS_DIAG_UNIFORM_MASKROWS
        for (int k = 0; k < D; ++k) inw[k] = t_bm[BMW * (C + 1) + k] ^ t_bm[k];
#else
        for (int k = 0; k < D; ++k) inw[k] = t_bm[BMW * hi + k] ^ t_bm[BMW * lo + k]; // 0xff: inside the kept window
#endif
        // (the third row is NOT made conditional here as it is in load_b: with both conditional the allocator spills two registers per
        // chunk to scratch -- 8.4 -> 8.8 G reads/s without, same-box A/B profiles/r3g/ab_cond.txt, and 15 B/read of scratch traffic)
#pragma unroll
#ifdef FAQCS_LDS_DIAG_UNIFORM_MASKROWS
        for (int k = 0; k < D; ++k) w[k] = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh) & t_bm[BMW * (C + 1) + k];
#else
        for (int k = 0; k < D; ++k) w[k] = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh) & t_bm[BMW * vb + k]; // a byte past the read: 0, no class
#endif
        uint32_t tp = 0, tq = 0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int k = j >> 2;
            if ((j & 3) == 0) {
                if (MODE == 0) seen7 |= w[k];
                w[k] ^= ~((EXT && chk) ? 0u : inw[k]) & 0x88888888u; // (w[k] is the table index from here on; the byte is put back for the N tests)
            }
            const uint32_t ad = ((j & 3) == 0 ? byte_x8<0>(w[k], three) : (j & 3) == 1 ? byte_x8<1>(w[k], three)
                                 : (j & 3) == 2 ? byte_x8<2>(w[k], three) : byte_x8<3>(w[k], three)) + (uint32_t)(T::O_T3 * 4);
#ifdef FAQCS_LDS_DIAG_UNIFORM_SLOOKUP // (diagnostic build, wrong results: every lane looks up the same entry)
            const LdsPair2 e = *(lds_u2c_ptr)(size_t)((ad & 0u) + (uint32_t)(T::O_T3 * 4 + 8 * 'A'));
#else
            const LdsPair2 e = *(lds_u2c_ptr)(size_t)ad;
#endif
            if (MODE == 0) { bpre[j] += e.x; bpost[j] += e.y; tp += e.x; tq += e.y; }
            if (MODE == 1) bpost[j] -= e.y;
            if (MODE == 2) { bpre[j] -= e.x; bpost[j] -= e.y; }
        }
        if (MODE == 0) {
            // the read's totals: 8 lanes, <= 19 per 6-bit field each; one step in 6-bit fields (<= 38), the rest in 12-bit fields
            tp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tp, 0xB1, 0xf, 0xf, false);
            tq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tq, 0xB1, 0xf, 0xf, false);
            // (WIDE: a read can hold more than 255 N -- the two N counts get a register of their own, pre in the low and post in the high half)
            uint32_t pe = tp & (WIDE ? 0x0003f03fu : 0x3f03f03fu), po = (tp >> 6) & 0x0003f03fu, ce = tq & (WIDE ? 0x0003f03fu : 0x3f03f03fu), co = (tq >> 6) & 0x0003f03fu;
            uint32_t pn = WIDE ? ((tp >> 24) | ((tq >> 24) << 16)) : 0u;
            if (WIDE) {
                pn += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pn, 0x4E, 0xf, 0xf, false);
                pn += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pn, 0x141, 0xf, 0xf, false);
                pn += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pn, 0x140, 0xf, 0xf, false);
            }
            pe += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x4E, 0xf, 0xf, false);
            po += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)po, 0x4E, 0xf, 0xf, false);
            ce += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ce, 0x4E, 0xf, 0xf, false);
            co += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)co, 0x4E, 0xf, 0xf, false);
            if (LPR >= 8) { // (4 lanes per read: the two steps above are the whole quad)
                pe += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x141, 0xf, 0xf, false);
                po += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)po, 0x141, 0xf, 0xf, false);
                ce += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ce, 0x141, 0xf, 0xf, false);
                co += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)co, 0x141, 0xf, 0xf, false);
            }
            if (LPR == 16) { // the other half of the row (a read of <= 252 bases: 12-bit fields, an N count below 256)
                pe += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x140, 0xf, 0xf, false);
                po += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)po, 0x140, 0xf, 0xf, false);
                ce += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ce, 0x140, 0xf, 0xf, false);
                co += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)co, 0x140, 0xf, 0xf, false);
            }
            const bool turn = rl == t;
            tot_pe = turn ? pe : tot_pe; tot_po = turn ? po : tot_po; tot_ce = turn ? ce : tot_ce; tot_co = turn ? co : tot_co;
            if (WIDE) tot_pn = turn ? pn : tot_pn;
            // ---- upper-case N inside the kept window (count_poly_n, trim.cpp:578-597): looked at only when the read has enough N
            // (any case) in its window to matter, or when it is judged without being counted (chk) ----
            const uint32_t cN = WIDE ? pn >> 16 : ce >> 24;
            // -n 2 (the default): two adjacent N, tested here; any other -n: judged after the loop, from the staged bases
            if ((!EXT || P.max_poly_n == 2u) && __any(cN >= 2u || chk)) {
                uint32_t nb[D]; // bit 7 of a byte: upper-case 'N' inside the kept window
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const uint32_t orig = w[k] ^ (~((EXT && chk) ? 0u : inw[k]) & 0x88888888u); // (the byte itself again)
                    const uint32_t x = (orig & inw[k]) ^ 0x4e4e4e4eu;
                    const uint32_t sx = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
                    nb[k] = ~(sx | x) & 0x80808080u;
                }
                { // two adjacent N: positions (j, j + 1), j one of the lane's C positions
                    uint32_t hit = 0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        uint32_t here = nb[k];
                        if (4 * k + 4 > C) here &= low_bytes_(C - 4 * k); // j < C
                        const uint32_t next = k + 1 < D ? __builtin_amdgcn_alignbyte(nb[k + 1], nb[k], 1u) : (nb[k] >> 8);
                        hit |= here & next;
                    }
                    // (C a multiple of 4: the neighbour of the lane's last position is not among its D dwords -- the next lane's first byte)
                    if (C % 4 == 0) hit |= (nb[D - 1] >> 24) & RW::next(nb[0] & 0x80u);
                    const bool rowhit = RW::all_or(hit) != 0u;
                    pairhit = turn ? rowhit : pairhit;
                }
            } else if (MODE == 0) {
                pairhit = turn ? false : pairhit;
            }
        }
    };
    // (Measured and not kept: a 160-wide table grid with the four rows of a half wave started 0..3 positions apart, which makes
    // the ds_add of the position x quality pass conflict-free by construction -- SQ_LDS_BANK_CONFLICT 22.6 -> 13.7 and
    // SQ_LDS_IDX_ACTIVE 51 -> 44 clocks per read, but +3 VALU per read for the 20th position and the wrapped steps, and the
    // same 5.7 G reads/s: the waves wait on LDS round trips, not on LDS throughput.)
    // the loop over the 8 reads of a lane's row (a deeper software pipeline -- info two reads ahead, LDS bytes one read ahead --
    // was measured: +5 VALU per read for the register copies and no shorter waits; the LDS pipeline itself is the limit)
#define FAQCS_B_LOOP(I0, I1, CELLS)                                                                                   \
    {                                                                                                                 \
        _Pragma("unroll 1") for (int t = 0; t < TPR; ++t) {                                                          \
            if (base + (uint32_t)t >= n_reads) break; /* wave-uniform: no row has a read left */                      \
            const uint32_t a0_ = (uint32_t)__shfl((int)(I0), rowb + t), a1_ = (uint32_t)__shfl((int)(I1), rowb + t);   \
            RawB cur_;                                                                                                \
            load_b(a0_, a1_, cur_);                                                                                   \
            CELLS(cur_, a0_, a1_, t);                                                                                 \
        }                                                                                                             \
    }

#ifdef FAQCS_LDS_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_flush = 0, st_spill = 0;
#define FAQCS_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_prev; st_prev = now_; }
#else
#define FAQCS_STAMP(i)
#endif
    uint32_t p_off = 0, p_end = 0;
    auto fetch_offsets = [&](const uint32_t chunk_) { // (a chunk past the last one of the launch: a chunk of no reads)
        if (chunk_ != NO_CHUNK) {
            // (a lane that owns no read sits at the end of the chunk's last read, with length 0)
            const uint32_t lim_ = umin_(chunk_ * RPC + RPC, n_reads), my_ = owner ? chunk_ * RPC + ridx : lim_;
            p_off = off[my_ < lim_ ? my_ : lim_];
            p_end = off[my_ < lim_ ? my_ + 1 : lim_];
        }
    };
    // What a chunk needs from global memory before its first pass: the quality span (DMA into the wave's slot), the adapter pre-pass's
    // words, the first / last base of every read (mask_quality_terminal_N looks at them before the qualities are used).
    struct ChunkLoads { uint32_t v_off, v_end, v_sl, v_hit, pad_len; }; // pad_len (wave-uniform): 0, or the length all reads of the chunk have (padded rows)
    ChunkLoads ld = {0, 0, 0, 0, 0};
    // one arena's bytes of the chunk -> the wave's slot: the contiguous span, or one padded row per read (dma_rows)
    auto stage = [&](const uint8_t *arena, const uint32_t cs_, const uint32_t ce_, const uint32_t sh_, const uint32_t pad_len_) {
        if (pad_len_) dma_rows<NI>(arena + cs_ - sh_, pad_len_, (uint32_t)RPC, slot, lane);
        else dma_span<NI>(arena + cs_ - sh_, ce_ - cs_ + sh_, slot, lane);
    };
    bool pre_issued = false;
    auto issue_loads = [&](const uint32_t chunk_, const uint32_t o_, const uint32_t e_, ChunkLoads &L) {
        const uint32_t my_ = chunk_ * RPC + ridx;
        const bool mine_ = owner && my_ < n_reads;
        const uint32_t len_ = e_ - o_;
        const uint32_t cs_ = uniu(o_), ce_ = (uint32_t)__builtin_amdgcn_readlane((int)e_, 63);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (every read of the slot's previous content has returned)
        const uint32_t shq_ = (uint32_t)((size_t)(qual + cs_) & 15u);
        // every read of the chunk as long as the first one, a multiple of 32 bases: padded rows (a chunk with fewer than RPC reads: never)
        const uint32_t len0_ = uniu(len_);
        const bool rows_ = (len0_ & 31u) == 0u && len0_ != 0u && len0_ <= (uint32_t)T::PADLEN && __all(!owner || (mine_ && len_ == len0_));
        L.pad_len = rows_ ? len0_ : 0u;
        stage(qual, cs_, ce_, shq_, L.pad_len);
        L.v_off = o_; L.v_end = e_;
        L.v_sl = (WINDOWED && ad_sl && mine_) ? ad_sl[my_] : (len_ << 16);
        L.v_hit = (ad_hit && mine_) ? ad_hit[my_] : 0u;
    };
    uint32_t c_cur = (uint32_t)wave, n_flushed = 0, since_spill = 0;
    uint32_t chunk_cur = chunk_of(c_cur);
    fetch_offsets(chunk_cur);
#ifdef FAQCS_LDS_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime(); // shader clock = d(memtime) / d(memrealtime) x 100 MHz
#endif
#pragma unroll 1
    for (;;) {
#ifdef FAQCS_LDS_STAMPS
        st_prev = __builtin_amdgcn_s_memtime();
#endif
        {   // the flushes that come before chunk c_cur (every wave of the block passes each of them exactly once)
            const uint32_t n_local = lds_word(1); // (known by the time a wave holds a chunk number past it)
            const uint32_t due = chunk_cur != NO_CHUNK ? c_cur / FLUSH_CHUNKS : (n_local + FLUSH_CHUNKS - 1) / FLUSH_CHUNKS;
#pragma unroll 1
            while (n_flushed < due) {
                spill_base_regs();
                flush_block_partial<C, LPR, NW>(smem, P.partials + ((size_t)blockIdx.x * FAQCS_PARTIAL_FLUSHES + n_flushed) * FAQCS_PARTIAL_ROW, tid);
                ++n_flushed; since_spill = 0;
            }
        }
#ifdef FAQCS_LDS_STAMPS
        st_flush += __builtin_amdgcn_s_memtime() - st_prev;
        st_prev = __builtin_amdgcn_s_memtime();
#endif
        if (chunk_cur == NO_CHUNK) break;
        uint32_t c_next = 0;
        if (lane == 0) c_next = __hip_atomic_fetch_add((lds_u32_mut)(size_t)(uint32_t)(T::O_CTR * 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        c_next = uniu(c_next);
        // the first chunk of group number L: ask for the id of group number L + 1 (published below, behind this chunk's loads)
        // (a block has FAQCS_PARTIAL_FLUSHES rows to flush into: before it would need another one it stops asking for groups, and the
        // blocks that have room take what is left -- faqcs_launch_trim_lds makes sure they have)
        const bool at_group_start = c_next % NW == 0 && c_next < lds_word(1);
        const bool fetch_group = at_group_start && c_next / NW + 1u < MAX_GROUPS;
        uint32_t g_new = 0;
        if (fetch_group && lane == 0) g_new = atomicAdd(g_next, 1u);
        const uint32_t chunk_next = chunk_of(c_next);
        {
            const uint32_t chunk = chunk_cur;
            const uint32_t base = chunk * RPC;
            const uint32_t my = base + ridx;
            const bool mine = owner && my < n_reads;
            // ---- the span of the QUALITY arena -> LDS, and the per-read words that come from global memory.  Requested at the end
            // of the previous chunk (issue_loads below, once the slot is no longer read); here for a wave's first chunk only.
            if (!pre_issued) issue_loads(chunk_cur, p_off, p_end, ld);
            // a lane without a read sits at the end of the last read (length 0): the span ends where lane 63 ends
            const uint32_t v_off = ld.v_off, v_end = ld.v_end;
            const uint32_t v_len = v_end - v_off;
            const uint32_t v_sl = ld.v_sl, v_hit = ld.v_hit;
            const uint32_t cs = uniu(v_off), ce = (uint32_t)__builtin_amdgcn_readlane((int)v_end, 63);
            const int len = (int)v_len;
            const uint32_t shq = (uint32_t)((size_t)(qual + cs) & 15u);
            const uint32_t pad_len = uniu(ld.pad_len), pad_stride = pad_len + 16u;
            const uint32_t rowq = pad_len ? ridx * pad_stride + shq : v_off - cs + shq; // this lane's read inside the slot
            // bytes of the slot up to the end of the chunk's last read (padded rows: the 16 bytes behind the last row's read are whatever follows
            // the chunk in the arena -- they belong to the pad that is filled below, like the bytes behind a contiguous span)
            const uint32_t span_q = pad_len ? (uint32_t)(RPC - 1) * pad_stride + pad_len + shq : ce - cs + shq;
            // first / last base (mask_quality_terminal_N needs them before the qualities are looked at).  Requested HERE, not with the
            // early loads: a sector of the base arena touched a whole chunk ahead of the base DMA has left the L2 by then and comes
            // over the fabric twice (+100 B/read of fetches, measured)
            uint32_t bfirst = 0, blast = 0;
#ifndef FAQCS_LDS_NO_TERMINAL_LOADS // (diagnostic build: what these two scattered loads cost in fabric requests)
            if (tn_flags) { // the submitter's per-read flags (faqcs_batch::terminal_n): one coalesced byte per read instead
                const uint32_t f = mine ? (uint32_t)tn_flags[my] : 0u;
                bfirst = (f & 1u) ? (uint32_t)'N' : 0u; blast = (f & 2u) ? (uint32_t)'N' : 0u;
            } else if (len) { bfirst = (uint32_t)seq[(size_t)v_off]; blast = (uint32_t)seq[(size_t)v_off + len - 1]; }
#endif
            fetch_offsets(chunk_next);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (at_group_start) { // publish group number L + 1 of the block, or the block's chunk count when the launch has no group left
                const uint32_t L1 = c_next / NW + 1u, gid = fetch_group ? gridDim.x + uniu(g_new) : n_groups;
                if (lane == 0) {
                    if (gid < n_groups) *(volatile __attribute__((address_space(3))) uint32_t *)(size_t)(uint32_t)((T::O_CTR + 4 + (int)(L1 & 3u)) * 4) = ((L1 & 0xfffu) << 20) | gid;
                    else *(volatile __attribute__((address_space(3))) uint32_t *)(size_t)(uint32_t)((T::O_CTR + 1) * 4) = L1 * NW;
                }
            }
            // pad behind the span: the position-parallel passes read up to W + 5 bytes past a short last read, and what they
            // find there must be a valid quality byte (see quality_cells): the ADDRESS of a Q-B add comes from the byte even when its
            // increment is zero, and an add of zero is still a read-modify-write.  With a base letter there ('a' ... 't' are rows 64 ... 83 of
            // a 42-row table) it lands in ANOTHER wave's slot, and when that wave's LDS-DMA writes the same dword between the add's read and
            // its write, the add puts the old bytes back -- one stale dword, once in a thousand chunks (found by re-seeded runs in round 4:
            // profiles/r4c/restage_race.txt).  So the pad is written again whenever qualities are staged again (the take-back pass below).
            pad_behind_span<W, T::STG_BYTES>(slot_b, span_q, shq, lane, (uint32_t)in_off);
            FAQCS_STAMP(0)

            // ================= Q-A: one read per lane ===============================================================
            ReadOutcome oc;
            uint32_t v_patch = (uint32_t)len << 16; // lead | trail << 16
            // ---- mask_quality_terminal_N (trim.cpp:1191-1216): upper-case 'N' runs at either end get Q0, in place ----
            const bool tn = len > 0 && (bfirst == 'N' || blast == 'N');
            if (__any(tn)) {
                int lead = 0, trail = len;
                bool go = tn && bfirst == 'N';
#pragma unroll 1
                while (__any(go)) {
                    if (go) { ++lead; go = lead < len && seq[(size_t)v_off + lead] == 'N'; }
                }
                go = tn && blast == 'N' && lead < len;
                if (tn && lead >= len) trail = 0;
#pragma unroll 1
                while (__any(go)) {
                    if (go) { --trail; go = trail > 0 && seq[(size_t)v_off + trail - 1] == 'N'; }
                }
                if (tn) v_patch = (uint32_t)lead | ((uint32_t)trail << 16);
                const int ntail = tn ? len - trail : 0;
#pragma unroll 1
                for (int i = 0; __any(tn && (i < lead || i < ntail)); ++i) {
                    if (tn && i < lead) lds_st_u8(slot_b + rowq + (uint32_t)i, (uint32_t)in_off);
                    if (tn && i < ntail) lds_st_u8(slot_b + rowq + (uint32_t)(trail + i), (uint32_t)in_off);
                }
            }

            // ---- quality: sum and range check, four bytes per instruction --------------------------------
            // V = sum(raw - offset); a byte outside [offset, offset + 41] (or >= 128) sends the read to the exact pass
            uint32_t qsum = 0, qacc = 0;
            int n_dw = 0; // dwords every lane has summed (wave-uniform): a dword past a lane's read counts as four offset bytes
            const uint32_t c_hi = 0x56565656u; // x = raw - offset ; x + 0x56: bit 7 <=> x > 41
            // The range check first, over the span as it lies in the slot (16 aligned bytes per lane and instruction, no bank
            // conflicts, whatever read a byte belongs to).  No byte out of range -- every chunk of valid data -- and no option that
            // needs a read's sums before Q-B: the lane-per-read pass below is skipped and Q-B takes the sums (its lanes hold the bytes
            // anyway).  The few bytes either side of the span inside its first / last 16-byte piece are other reads' qualities (or the
            // arena's padding at its two ends: then the pass below runs, which is always right).
            bool sum_pass = !swar_ok || WINDOWED || (EXT && P.avgq_on);
            if (!sum_pass) {
                typedef uint32_t U4 __attribute__((ext_vector_type(4)));
                uint32_t acc = 0;
                const uint32_t n16 = (span_q + 15u) >> 4;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if ((uint32_t)(i * 64) >= n16) break; // wave-uniform
                    if ((uint32_t)lane < n16 - (uint32_t)(i * 64)) {
                        const U4 v = *(const __attribute__((address_space(3))) U4 *)(size_t)((slot_b + (uint32_t)(lane * 16)) + (uint32_t)(i * 1024));
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const uint32_t x = v[e] - offb; acc |= x | (x + c_hi) | v[e]; }
                    }
                }
                sum_pass = __any((acc & 0x80808080u) != 0u);
            }
            if (sum_pass) {
                const int nfull = len >> 2, rem = len & 3;
                const int kmax = uni((int)wave_max_u32((uint32_t)nfull));
                const uint32_t qa = slot_b + rowq, qa4 = qa & ~3u, qsh = qa & 3u;
                uint32_t prev = lds_ld(qa4);
#pragma unroll FAQCS_LDS_SUM_UNROLL
                for (int k = 0; k < kmax; ++k) {
                    const uint32_t nxt = lds_ld(qa4 + 4u * (uint32_t)k + 4u);
                    uint32_t w = __builtin_amdgcn_alignbyte(nxt, prev, qsh);
                    prev = nxt;
                    w = k < nfull ? w : offb;
                    qsum = __builtin_amdgcn_sad_u8(w, 0u, qsum);
                    const uint32_t x = w - offb; // a byte below the offset borrows: bit 7 of that byte (or of a lower one) is set
                    qacc |= x | (x + c_hi) | w;  // bit 7: raw < offset, raw > offset + 41 or raw >= 128, in some byte
                }
                n_dw = kmax;
                if (__any(rem != 0)) { // the bytes of the last, partial dword (past the read: quality == offset, adds 0)
                    const uint32_t m = low_bytes_(rem); // (rem == 0: no byte)
                    const uint32_t w = (lds_ld_any(qa + 4u * (uint32_t)nfull) & m) | (offb & ~m);
                    qsum = __builtin_amdgcn_sad_u8(w, 0u, qsum);
                    const uint32_t x = w - offb;
                    qacc |= x | (x + c_hi) | w;
                    n_dw = kmax + 1;
                }
            }
            const bool badq = !swar_ok || ((qacc & 0x80808080u) != 0u);
            int V_pre = (int)qsum - 4 * n_dw * in_off;
            bool read_err = false;
            if (__any(badq)) {
                // the trimmers clamp a negative score to 0 (fastq.h:17-36): clamp the staged bytes in place; the sums and the
                // > 41 verdict of such a read come from the exact pass over the arena further down
#pragma unroll 1
                for (int p = 0; __any(badq && p < len); ++p) {
                    if (badq && p < len) {
                        const uint32_t r = lds_ld_u8(slot_b + rowq + (uint32_t)p);
                        int v = (int)(int8_t)r - in_off;
                        v = v < 0 ? 0 : (v > 41 ? 41 : v);
                        lds_st_u8(slot_b + rowq + (uint32_t)p, (uint32_t)(v + in_off));
                    }
                }
            }

            FAQCS_STAMP(1)
            // ---- the window the reference trims: after the adapter pre-pass and --5end/--3end (trim.cpp:270-314) ----
            int wa = 0, wn = len;
            uint32_t flags = 0, filt = 0;
            if (WINDOWED && P.has_adapters) {
                const int first = (int)(v_sl & 0xffffu), second = (int)(v_sl >> 16);
                const bool mod = len != second;
                wa = mod ? first : 0; wn = mod ? second : len;
                flags = mod ? FAQCS_F_ADAPTER : 0u;
            }
            const bool do_trim = !(EXT && P.qc_only); // --qc_only: adapters are still cut, nothing else is (trim.cpp:279,299,325)
            if (WINDOWED && P.trim5 && do_trim) {
                const bool over = (int)P.trim5 > wn;
                wa = over ? wa : wa + (int)P.trim5;
                wn = over ? 0 : wn - (int)P.trim5;
            }
            if (WINDOWED && P.trim3 && do_trim) wn = (int)P.trim3 > wn ? 0 : wn - (int)P.trim3;

            // ---- BWA_plus (trim.cpp:714-793), walked as the reference walks it ---------------------------------
            // Step s of a walk visits window position wn - 1 - s (3') or s (5').  at_least_scan == 0 after step `bud`: the walk
            // covers min(5, n) positions and a reset at step s (area >= 0 before it, s < rlim: the position is still more than
            // n2 away from the far end) moves its end to s + 2 (n2 == 2 whenever a reset can fire).
            // key = area << 8 | code, code falling with time: the FIRST maximum wins (walk_step above).
            const int a5 = wn < 5 ? wn : 5, nn2 = wn < 2 ? wn : 2, qoff = Q + in_off;
            const int dc = uni(qoff * (1 << CB) - 1);
            int S3 = 0, fp3 = wn - 1, S5 = 0, fp5 = 0;
            const int mode = EXT ? P.mode : FAQCS_MODE_BWA_PLUS;
            bool sum_kept = false; // the kept window's quality sum needs its own pass (no walk areas to derive it from)
            if (EXT && do_trim && mode == FAQCS_MODE_BWA) {
                // trim.cpp:675-709: from the 3' end while position > 0 and area >= 0; the first maximum of the area cuts
                int area = 0, best = 0, p = wn - 1;
                const uint32_t q0 = slot_b + rowq + (uint32_t)wa;
#pragma unroll 1
                while (__any(p > 0 && area >= 0)) {
                    if (p > 0 && area >= 0) {
                        area += qoff - (int)lds_ld_u8(q0 + (uint32_t)p);
                        if (area > best) { best = area; fp3 = p - 1; }
                        --p;
                    }
                }
                S3 = best;
            } else if (EXT && do_trim && mode == FAQCS_MODE_HARD) {
                // trim.cpp:629-672: 3' = first position from the end (above 0) with q > Q, 5' = first position below it with q > Q
                const uint32_t q0 = slot_b + rowq + (uint32_t)wa;
                int p3 = wn - 1;
                bool go = p3 > 0;
#pragma unroll 1
                while (__any(go)) {
                    if (go) {
                        if (qoff < (int)lds_ld_u8(q0 + (uint32_t)p3)) { fp3 = p3; go = false; }
                        else { --p3; go = p3 > 0; }
                    }
                }
                if (!P.protect5) {
                    int p5 = 0;
                    go = p5 < p3;
#pragma unroll 1
                    while (__any(go)) {
                        if (go) {
                            if (qoff < (int)lds_ld_u8(q0 + (uint32_t)p5)) { fp5 = p5; go = false; }
                            else { ++p5; go = p5 < p3; }
                        }
                    }
                }
                sum_kept = true;
            } else if (do_trim) {
                int rlim = wn - 1 - nn2; // a reset at step s needs wn - 1 - s > n2
                const uint32_t endq = slot_b + rowq + (uint32_t)(wa + wn);
                const uint32_t esh = endq & 3u;
                uint32_t e4 = endq & ~3u;
                uint32_t hi = lds_ld(e4), cur = lds_ld(e4 - 4u);
                // Fast forward over a run of identical dwords b,b,b,b with Q - q(b) > 0 at the 3' end (the '#' tail of an Illumina
                // read): every position of such a run is visited, resets the scan and sets a new maximum, so m dwords of it leave
                // area = 4 m (Q - q), the maximum at the run's last position and two more positions to visit.
                int m = 0;
                const uint32_t w0 = __builtin_amdgcn_alignbyte(hi, cur, esh);
                const uint32_t pat = __builtin_amdgcn_perm(w0, w0, 0u);
                const int dq0 = qoff - (int)(w0 & 0xffu);
                {
                    const int mcap = rlim >> 2; // steps 4 i .. 4 i + 3 all reset: 4 i + 3 < rlim
                    bool run = dq0 > 0;
                    uint32_t h = hi, c = cur;
                    // four dwords per LDS round trip (the loop is a chain of round trips: a Q2 tail of 75 bases is 19 dwords)
#pragma unroll 1
                    for (int i = 0; ; i += 4) {
                        const uint32_t n0 = lds_ld(e4 - 8u - 4u * (uint32_t)i), n1 = lds_ld(e4 - 12u - 4u * (uint32_t)i);
                        const uint32_t n2 = lds_ld(e4 - 16u - 4u * (uint32_t)i), n3 = lds_ld(e4 - 20u - 4u * (uint32_t)i);
                        run = run && __builtin_amdgcn_alignbyte(h, c, esh) == pat && i < mcap;
                        m += run ? 1 : 0;
                        run = run && __builtin_amdgcn_alignbyte(c, n0, esh) == pat && i + 1 < mcap;
                        m += run ? 1 : 0;
                        run = run && __builtin_amdgcn_alignbyte(n0, n1, esh) == pat && i + 2 < mcap;
                        m += run ? 1 : 0;
                        run = run && __builtin_amdgcn_alignbyte(n1, n2, esh) == pat && i + 3 < mcap;
                        m += run ? 1 : 0;
                        h = n2; c = n3;
                        if (!__any(run)) break;
                    }
                }
                int rb = a5 - 1, area = 0, best = CMAX, K = CMAX; // rb: steps still to go after the current one
                if (__any(m > 0)) {
                    const int A0 = 4 * m * dq0;
                    area = A0;
                    best = K = (A0 << CB) + CMAX; // (m == 0: CMAX)
                    rb = m > 0 ? 1 : rb;
                    rlim -= 4 * m;
                    e4 -= 4u * (uint32_t)m;
                    hi = lds_ld(e4); cur = lds_ld(e4 - 4u);
                }
#pragma unroll 1
                for (int it = 0; __any(rb >= 0); it += 4) {
                    const uint32_t nx = lds_ld(e4 - 8u - (uint32_t)it);
                    const uint32_t w = __builtin_amdgcn_alignbyte(hi, cur, esh); // positions wend-4-it .. wend-1-it
                    hi = cur; cur = nx;
                    walk_step<true, CB>(w, it, (it + 1) * qoff * (1 << CB) + CMAX - 1 - it, dc, rb, rlim, area, K, best);
                }
                S3 = best >> CB;
                const int code = best & CMAX;
                const int sb = code == CMAX ? 4 * m - 1 : 4 * m + CMAX - 1 - code;
                fp3 = S3 > 0 ? (wn - 1 - sb) - 1 : wn - 1;
            }
            FAQCS_STAMP(2)
            if (do_trim && mode == FAQCS_MODE_BWA_PLUS && !(EXT && P.protect5)) { // --5trim_off (trim.cpp:752)
                int rb = a5 - 1, area = 0, best = CMAX, K = CMAX;
                const int rlim = fp3 - nn2; // a reset at step s needs s < final_pos_3 - n2
                const uint32_t begq = slot_b + rowq + (uint32_t)wa;
                const uint32_t b4 = begq & ~3u, bsh = begq & 3u;
                uint32_t lo = lds_ld(b4), cur = lds_ld(b4 + 4u);
#pragma unroll 1
                for (int it = 0; __any(rb >= 0); it += 4) {
                    const uint32_t nx = lds_ld(b4 + 8u + (uint32_t)it);
                    const uint32_t w = __builtin_amdgcn_alignbyte(cur, lo, bsh);
                    lo = cur; cur = nx;
                    walk_step<false, CB>(w, it, (it + 1) * qoff * (1 << CB) + CMAX - 1 - it, dc, rb, rlim, area, K, best);
                }
                S5 = best >> CB;
                fp5 = S5 > 0 ? (CMAX - 1 - (best & CMAX)) + 1 : 0;
            }

            // ---- length filters and the kept window (trim.cpp:317-360) -------------------------------------
            int a = wa, n = wn;
            bool ret = mine;
            if (ret && (n < (int)P.min_len || n == 0)) { ret = false; filt = FAQCS_FILT_LENGTH_PRE; }
            if (ret && do_trim) {
                // BWA_plus: final_pos_3 <= final_pos_5 empties the read (trim.cpp:781-790); BWA keeps [0, final_pos_3]; HARD [5', 3']
                const int kept = mode == FAQCS_MODE_BWA_PLUS ? (fp3 <= fp5 ? 0 : fp3 - fp5 + 1) : fp3 - fp5 + 1;
                if (kept != n) { flags |= FAQCS_F_QUAL_TRIMMED | ((uint32_t)(n - kept) << 20); } // (bits 20..: the bases the quality trim removed)
                a += fp5;
                n = kept;
                if (n < (int)P.min_len || n == 0) { ret = false; filt = FAQCS_FILT_LENGTH_POST; }
            }

            // ---- sum(raw - offset) over the kept window (trim.cpp:374, :553-576) ---------------------------
            int V_win = V_pre;
            if (WINDOWED) { // the window's own sum first
                uint32_t sm = 0;
                const int k0 = wa >> 2, k1 = (wa + wn + 3) >> 2; // dwords that hold a window byte
                const int kmax = uni((int)wave_max_u32((uint32_t)(k1 - k0)));
#pragma unroll 2
                for (int i = 0; i < kmax; ++i) {
                    const int k = k0 + i;
                    const uint32_t w = lds_ld_any(slot_b + rowq + 4u * (uint32_t)k); // (k >= k1: masked to nothing)
                    const uint32_t m = low_bytes_(med3i(wa + wn - 4 * k, 0, 4)) & ~low_bytes_(med3i(wa - 4 * k, 0, 4));
                    sm = __builtin_amdgcn_sad_u8(w & m, 0u, sm);
                }
                V_win = (int)sm - wn * in_off;
            }
            // all v == q here (no byte below the offset): what the two walks cut off is their maximal areas
            int V_post = n * Q - ((wn * Q - V_win) - (S3 > 0 ? S3 : 0) - (S5 > 0 ? S5 : 0));
            if (EXT && sum_kept) { // (wave-uniform: the trim mode)
                uint32_t sm = 0;
                const int k0 = a >> 2, k1 = (a + n + 3) >> 2;
                const int kmax = uni((int)wave_max_u32((uint32_t)(k1 > k0 ? k1 - k0 : 0)));
#pragma unroll 2
                for (int i = 0; i < kmax; ++i) {
                    const int k = k0 + i;
                    const uint32_t w = lds_ld_any(slot_b + rowq + 4u * (uint32_t)k);
                    const uint32_t m = low_bytes_(med3i(a + n - 4 * k, 0, 4)) & ~low_bytes_(med3i(a - 4 * k, 0, 4));
                    sm = __builtin_amdgcn_sad_u8(w & m, 0u, sm);
                }
                V_post = (int)sm - n * in_off;
            }
            // exact pass for a read with a raw quality outside [offset, offset + 41] (negative scores clamp to 0 in the
            // trimmers but not in the averages; > 41 aborts the run)
            if (__any(badq)) {
                const ExactQ xq = exact_quality(qual, v_off, len, v_patch, a, n, in_off, badq);
                if (badq) { V_pre = xq.sv; V_post = xq.svp; read_err = xq.mq > 41; }
            }
            // ---- average quality (trim.cpp:374-382): after poly-N in the reference; poly-N is judged in S-A, and a read it
            // rejects never reaches this test, so the order is restored there (avgq_fail is only applied to a read that passes) ----
            const bool avgq_fail = EXT && P.avgq_on && V_post < ((const int32_t *)(smem + Cfg::O_TAVGQ))[n];

            FAQCS_STAMP(3)
            // ================= Q-B: 8 lanes per read, position x quality ===========================================
            // (post cells are added for every read that is still valid; S-A's vetoes are taken back below)
            const uint32_t qi0 = rowq | ((uint32_t)len << 16);
            // retc: the read is still kept as far as the qualities can tell (an average-quality failure is final whatever poly-N says later)
            const bool retc = ret && !avgq_fail;
            const uint32_t qi1 = (uint32_t)a | ((uint32_t)n << AB) | (retc ? 1u << FB : 0u) | ((mine && !read_err) ? 1u << (FB + 1) : 0u);
#define FAQCS_QCELLS(X, A, B, T_) quality_cells(X, A, B, std::false_type{}, T_, !sum_pass)
            FAQCS_B_LOOP(qi0, qi1, FAQCS_QCELLS)
#undef FAQCS_QCELLS
            if (!sum_pass) { // the sums that the lane-per-read pass did not take (no byte out of range: all v == q)
                V_pre = (int)qb_sum - len * in_off;
                if (!(EXT && sum_kept)) V_post = n * Q - ((wn * Q - V_pre) - (S3 > 0 ? S3 : 0) - (S5 > 0 ? S5 : 0));
            }

            FAQCS_STAMP(4)
            // ---- the span of the BASE arena -> the same slot ---------------------------------------------------------
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every LDS read of the slot has returned
            const uint32_t shs = (uint32_t)((size_t)(seq + cs) & 15u);
            stage(seq, cs, ce, shs, pad_len);
            const uint32_t rows = pad_len ? ridx * pad_stride + shs : v_off - cs + shs;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FAQCS_STAMP(5)

            // ================= S: 8 lanes per read, the bases (base_step) ==============================================
            // i1: a | n << 8 | post << 16 | counted << 17 | chk << 18.  chk: the read fails the average quality but is still judged for
            // poly-N, which the reference tests first (trim.cpp:363-382).
            const uint32_t si0 = rows | ((uint32_t)len << 16);
            const uint32_t si1 = (uint32_t)a | ((uint32_t)n << AB) | (retc ? 1u << FB : 0u) | ((mine && !read_err) ? 1u << (FB + 1) : 0u) |
                                 ((ret && !retc) ? 1u << (FB + 2) : 0u);
            // (|sums| < 2^15 up to 252 bases: at most 127 per base; the 304-base variant keeps a register for each)
            const uint32_t vpk = WIDE ? (uint32_t)V_pre : (((uint32_t)V_pre & 0xffffu) | ((uint32_t)V_post << 16)), vpk2 = WIDE ? (uint32_t)V_post : 0u;
            const uint32_t fpk = flags | (filt << FAQCS_F_FILTER_SHIFT);
            const uint32_t v_hit_w = v_hit;
#define FAQCS_S_LOOP(I1, MODE)                                                                                        \
    {                                                                                                                 \
        _Pragma("unroll 1") for (int t = 0; t < TPR; ++t) {                                                          \
            if (base + (uint32_t)t >= n_reads) break; /* wave-uniform: no row has a read left */                      \
            const uint32_t a1_ = (uint32_t)__shfl((int)(I1), rowb + t);                                               \
            if (MODE == 1 && !__any(((a1_ >> FB) & 1u) != 0u)) continue;                                              \
            const uint32_t a0_ = (uint32_t)__shfl((int)si0, rowb + t);                                                \
            base_step(t, a0_, a1_, std::integral_constant<int, MODE>{});                                              \
        }                                                                                                             \
    }
            seen7 = 0;
            FAQCS_S_LOOP(si1, 0)
            if (__any((seen7 & 0x80808080u) != 0u)) {
                // a byte >= 0x80 among the bases: it was looked up as "outside the kept window".  Take the chunk back, blank such bytes
                // in the slot (no class: what the reference's switch does with them, trim.cpp:831-857) and count again.
                FAQCS_S_LOOP(si1, 2)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
                for (int p = 0; __any(p < len); ++p) {
                    if (p < len && (lds_ld_u8(slot_b + rows + (uint32_t)p) & 0x80u)) lds_st_u8(slot_b + rows + (uint32_t)p, 0u);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                FAQCS_S_LOOP(si1, 0)
            }

            FAQCS_STAMP(6)
            // ================= one read per lane again: the verdicts that depend on the bases ==========================
            // (The S loop is the kernel's register peak.  What it does not use is packed into vpk / fpk in front of it, and the read's
            // offsets and window are taken back out of the two words the loop was given: nothing else stays live across it.)
            {
            const int len = (int)(si0 >> 16), a = (int)(si1 & AM), n = (int)((si1 >> AB) & AM);
            const uint32_t rows = si0 & 0xffffu, rowq = rows - shs + shq, v_off = pad_len ? cs + ridx * pad_len : cs + rows - shs, v_len = (uint32_t)len;
            const uint32_t qi0 = rowq | ((uint32_t)len << 16);
            int V_pre = WIDE ? (int)vpk : (int)(int16_t)(uint16_t)(vpk & 0xffffu), V_post = WIDE ? (int)vpk2 : (int)vpk >> 16;
            uint32_t flags = fpk & ~(uint32_t)FAQCS_F_FILTER_MASK, filt = (fpk & (uint32_t)FAQCS_F_FILTER_MASK) >> FAQCS_F_FILTER_SHIFT;
            const uint32_t v_hit = WINDOWED ? v_hit_w : 0u;
            uint32_t pA = tot_pe & 0xfffu, pC = (tot_pe >> 12) & 0xfffu, pN = WIDE ? tot_pn & 0xffffu : tot_pe >> 24, pT = tot_po & 0xfffu, pG = tot_po >> 12;
            uint32_t cA = tot_ce & 0xfffu, cC = (tot_ce >> 12) & 0xfffu, cN = WIDE ? tot_pn >> 16 : tot_ce >> 24, cT = tot_co & 0xfffu, cG = tot_co >> 12;
            {
                // ---- poly-N (trim.cpp:363-371, :578-597): -n 2 = two adjacent upper-case N inside the kept window ----
                bool polyn;
                if (EXT && P.max_poly_n != 2u) { // -n 0: every read trips; -n k: a run of k upper-case N inside the kept window
                    uint32_t run[NWORD], hit = 0, nub[NWORD]; // nub: upper-case 'N' of the read, one bit per position
#pragma unroll
                    for (int wd = 0; wd < NWORD; ++wd) nub[wd] = 0;
                    // a run of k needs k N (any case) inside the window: only such reads (and the ones judged without having been
                    // counted) are scanned, one read per lane, from the staged bases
                    const bool scan = P.max_poly_n != 0u && ret && (cN >= P.max_poly_n || !retc);
                    if (__any(scan)) {
                        const int kmax = uni((int)wave_max_u32(scan ? (uint32_t)((len + 3) >> 2) : 0u));
#pragma unroll
                        for (int wd = 0; wd < NWORD; ++wd) {
                            if (8 * wd < kmax) { // (wave-uniform)
                                uint32_t nw = 0;
#pragma unroll 2
                                for (int k = 8 * wd; k < 8 * wd + 8 && k < ND; ++k) {
                                    uint32_t w = lds_ld_any(slot_b + rows + 4u * (uint32_t)k) & low_bytes_(med3i(len - 4 * k, 0, 4));
                                    w ^= 0x4e4e4e4eu;
                                    const uint32_t nz = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w; // bit 7 of a byte: the byte is not 'N'
                                    nw |= ((((~nz & 0x80808080u) >> 7) * 0x00204081u >> 21) & 0xfu) << (4 * (k & 7));
                                }
                                nub[wd] = scan ? nw : 0u;
                            }
                        }
                    }
                    if (P.max_poly_n != 0u) {
#pragma unroll
                        for (int w = 0; w < NWORD; ++w) run[w] = nub[w] & bit_range_(med3i(a - 32 * w, 0, 32), med3i(a + n - 32 * w, 0, 32));
                        // bit e of `run` after i rounds: N at e - i .. e, all inside the window
#pragma unroll 1
                        for (uint32_t i = 1; i < P.max_poly_n && i <= (uint32_t)NPOS; ++i) {
                            uint32_t carry = 0;
#pragma unroll
                            for (int w = 0; w < NWORD; ++w) {
                                const uint32_t shifted = (run[w] << 1) | carry;
                                carry = run[w] >> 31;
                                run[w] = shifted & nub[w] & bit_range_(med3i(a - 32 * w, 0, 32), med3i(a + n - 32 * w, 0, 32));
                            }
                        }
#pragma unroll
                        for (int w = 0; w < NWORD; ++w) hit |= run[w];
                    }
                    polyn = P.max_poly_n == 0u || (hit != 0u && P.max_poly_n <= (uint32_t)NPOS);
                } else {
                    polyn = pairhit;
                }
                if (ret && polyn) {
                    flags |= FAQCS_F_POLY_N_SEEN;
                    if (do_trim) { ret = false; filt = FAQCS_FILT_POLY_N; } // --qc_only counts it and keeps the read (trim.cpp:368-370)
                }
                // ---- average quality (judged in Q-A, applied here: after poly-N, trim.cpp:374-382) ----
                if (EXT && ret && avgq_fail) { ret = false; filt = FAQCS_FILT_AVG_Q; }

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
                // Dinucleotide candidates (two bases that each fill >= dthr of the kept window: AT- or GC-rich genomes).  At most two
                // classes can reach dthr when it exceeds a third of the window (the default --lc 0.85: 42.5 %), and then only the two
                // transitions between them can trip: count those two over the window, from the staged bases.  More candidate
                // classes (a small --lc) take the general pass below.
                bool dinuc_general = false;
                if (__any(dinuc)) {
                    const int ncand = (int)(cA >= dthr) + (int)(cT >= dthr) + (int)(cC >= dthr) + (int)(cG >= dthr);
                    dinuc_general = dinuc && ncand > 2;
                    const bool two = dinuc && ncand == 2;
                    // one-hot byte masks (the table's A, T, C, G fields) of the lane's two candidate classes
                    const uint32_t mA = cA >= dthr ? 0x000000ffu : 0u, mT = cT >= dthr ? 0x0000ff00u : 0u, mC = cC >= dthr ? 0x00ff0000u : 0u,
                                   mG = cG >= dthr ? 0xff000000u : 0u;
                    const uint32_t both = mA | mT | mC | mG;
                    const uint32_t mx = (both & 0xffu) ? 0xffu : (both & 0xff00u) ? 0xff00u : (both & 0xff0000u) ? 0xff0000u : (both ? 0xff000000u : 0u);
                    const uint32_t my_ = both & ~mx; // X = the lower field, Y = the other one
                    uint32_t dxy = 0, dyx = 0, prevx = 0, prevy = 0; // prev*: the previous window position held X / Y
                    const int k0 = a >> 2, k1 = (a + n + 3) >> 2;
                    const int kmax2 = uni((int)wave_max_u32((uint32_t)(two && k1 > k0 ? k1 - k0 : 0)));
#pragma unroll 1
                    for (int i = 0; i < kmax2; ++i) {
                        const int k = k0 + i;
                        const uint32_t w = lds_ld_any(slot_b + rows + 4u * (uint32_t)k);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int p = 4 * k + j;
                            const uint32_t ad = (j == 0 ? byte_x8<0>(w, three) : j == 1 ? byte_x8<1>(w, three)
                                                 : j == 2 ? byte_x8<2>(w, three) : byte_x8<3>(w, three)) + (uint32_t)(T::O_T2 * 4);
                            uint32_t ex = ((lds_u2c_ptr)(size_t)ad)->x;
                            ex = ((unsigned)(p - a) < (unsigned)n) ? ex : 0u; // outside the window: no class (resets `last`, trim.cpp:405-513)
                            const uint32_t isx = (ex & mx) ? 1u : 0u, isy = (ex & my_) ? 1u : 0u;
                            dxy += prevx & isy;
                            dyx += prevy & isx;
                            prevx = isx; prevy = isy;
                        }
                    }
                    if (two) lc_trip = lc_trip || dxy >= dthr || dyx >= dthr;
                }
                // exact per-position pass over the arena: every transition count for the general dinucleotide case (rare)
                if (__any(dinuc_general)) {
                    const ExactB xb = exact_bases(seq, v_off, len, a, n, dinuc_general, dinuc_general, dthr);
                    if (dinuc_general) lc_trip = lc_trip || xb.trip;
                }
                if (ret && lc_trip) { ret = false; filt = FAQCS_FILT_LOW_COMPLEXITY; }
            }

            FAQCS_STAMP(7)
            // ---- a read rejected here after its post-trim cells were counted: take them back, the bases first (still staged), then
            // the qualities (staged again) ----
            const bool veto = retc && !ret;
            if (__any(veto)) {
                const uint32_t vi1 = (uint32_t)a | ((uint32_t)n << AB) | (veto ? 1u << FB : 0u) | (veto ? 1u << (FB + 1) : 0u);
                FAQCS_S_LOOP(vi1, 1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stage(qual, cs, ce, shq, pad_len);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FAQCS_LDS_DIAG_VERIFY_RESTAGE // (diagnostic: is every byte of the re-staged span in the slot once the wait above has passed?)
                if (!pad_len) {
                    const uint32_t nb_ = ce - cs + shq;
                    uint32_t bad_ = 0, first_ = 0xffffffffu;
                    for (uint32_t u_ = (uint32_t)lane; u_ * 16u < nb_; u_ += 64u) {
                        const uint4 g_ = *reinterpret_cast<const uint4 *>(qual + cs - shq + (size_t)u_ * 16u);
                        const uint32_t l0_ = lds_ld(slot_b + u_ * 16u), l1_ = lds_ld(slot_b + u_ * 16u + 4u), l2_ = lds_ld(slot_b + u_ * 16u + 8u), l3_ = lds_ld(slot_b + u_ * 16u + 12u);
                        if (g_.x != l0_ || g_.y != l1_ || g_.z != l2_ || g_.w != l3_) { ++bad_; first_ = first_ == 0xffffffffu ? u_ : first_; }
                    }
                    if (bad_) {
                        atomicAdd(reinterpret_cast<unsigned long long *>(err + 16), (unsigned long long)bad_);
                        atomicMax(reinterpret_cast<unsigned long long *>(err + 16) + 1, ((unsigned long long)nb_ << 32) | first_);
                    }
                    atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + 2, 1ull);
                }
#endif
                // (the bases of the S pass lie behind the span now; the span's end is worked out again here rather than kept across the S pass)
                pad_behind_span<W, T::STG_BYTES>(slot_b, pad_len ? (uint32_t)(RPC - 1) * pad_stride + pad_len + shq : ce - cs + shq, shq, lane, (uint32_t)in_off);
                // the in-place edits of Q-A again: terminal-N runs and clamped bytes
                // (a read with an out-of-range byte is clamped again whether it is taken back or not: its neighbours' lanes read into it)
                if (__any((veto && tn) || badq)) {
#pragma unroll 1
                    for (int p = 0; __any(((veto && tn) || badq) && p < len); ++p) {
                        if (((veto && tn) || badq) && p < len) {
                            const uint32_t r = lds_ld_u8(slot_b + rowq + (uint32_t)p);
                            int v = (int)(int8_t)r - in_off;
                            v = v < 0 ? 0 : (v > 41 ? 41 : v);
                            if (p < (int)(v_patch & 0xffffu) || p >= (int)(v_patch >> 16)) v = 0;
                            lds_st_u8(slot_b + rowq + (uint32_t)p, (uint32_t)(v + in_off));
                        }
                    }
                }
                const uint32_t ui1 = (uint32_t)a | ((uint32_t)n << AB) | (veto ? 1u << FB : 0u);
#pragma unroll 1
                for (int t = 0; t < TPR; ++t) {
                    const uint32_t i1 = (uint32_t)__shfl((int)ui1, rowb + t);
                    if (!__any(((i1 >> FB) & 1u) != 0u)) continue;
                    const uint32_t i0 = (uint32_t)__shfl((int)qi0, rowb + t);
                    RawB x;
                    load_b(i0, i1, x);
                    quality_cells(x, i0, i1, std::true_type{}, t, false);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#undef FAQCS_S_LOOP

            // the slot is not read any more: request the next chunk's loads now, under the epilogue (p_off / p_end: its offsets,
            // fetched at the top of this chunk)
            ChunkLoads ld_next = ld;
            pre_issued = chunk_next != NO_CHUNK;
            if (pre_issued) issue_loads(chunk_next, p_off, p_end, ld_next);

            if (read_err) { any_err = 1; flags |= FAQCS_F_ERR_QUALITY; }
            oc.an = (uint32_t)a | ((uint32_t)n << 16);
            oc.fl = flags | (ret ? FAQCS_F_VALID : 0u) | (filt << FAQCS_F_FILTER_SHIFT);
            oc.pAT = pA | (pT << 16); oc.pCG = pC | (pG << 16);
            oc.cAT = cA | (cT << 16); oc.cCG = cC | (cG << 16);
            oc.N = pN | (cN << 16);
            oc.Vpre = V_pre; oc.Vpost = V_post;

            // ---- chunk epilogue: one read per lane ----------------------------------------------------------
            chunk_epilogue<LPR>(oc, mine, my, v_len, v_hit, lane, smem + Cfg::O_LEN, smem + Cfg::O_RQ, smem + Cfg::O_BQPRE,
                                smem + Cfg::O_BQPOST, smem + Cfg::O_FS, smem + Cfg::O_TMAGIC, out, rec_pre, rec_post, EXT && P.avgq_on != 0, 0u, &fs_acc,
                                WIDE && P.wide_records != 0u); // (a batch with a read past 256 bases: the two-word records composition_histogram then expects)
            ld = ld_next;
            }
        }

        FAQCS_STAMP(8)
        if (++since_spill == REG_FLUSH_EVERY) { spill_base_regs(); since_spill = 0; }
#ifdef FAQCS_LDS_STAMPS
        st_spill += __builtin_amdgcn_s_memtime() - st_prev; // (the register spill: outside the nine sections, like the block flush)
#endif
        c_cur = c_next; chunk_cur = chunk_next;
    }
    if (__any(any_err != 0) && lane == 0) atomicOr(err, 1u);
    if (tid == 0) P.partial_rows[blockIdx.x] = n_flushed; // rows of P.partials this block wrote (fold_partials)
    // ---- the block has run out of reads: its LDS (every wave has passed the last flush) folds composition records of the PREVIOUS launch
    // while the slowest blocks of this one finish (round 6; faqcs_trim_common.h: comp_fold_tail)
    if (lds_tail_folds_v<C, NW, LPR, RPC> && P.fold_n) {
        __syncthreads();
        comp_fold_tail<NW * 64>(smem, P, tid, blockIdx.x & 1u);
    }
#ifdef FAQCS_LDS_STAMPS
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + i, st_acc[i]);
        atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + 9, __builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + 11, st_flush);
#ifdef FAQCS_LDS_BLOCKLOG
        if (wave == 0) printf("block %u xcc %u chunks %u start %llu end %llu\n", blockIdx.x, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)), smem[T::O_CTR + 1],
                              (unsigned long long)rt0, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
        if (wave == 0) { // when the blocks finish: latest and earliest s_memrealtime (10 ns ticks) of a block's wave 0
            const unsigned long long te = __builtin_amdgcn_s_memrealtime();
            atomicMax(reinterpret_cast<unsigned long long *>(err + 16) + 13, te);
            atomicMax(reinterpret_cast<unsigned long long *>(err + 16) + 14, ~te);
            atomicMax(reinterpret_cast<unsigned long long *>(err + 16) + 15, ~rt0); // (the earliest loop start)
        }
        atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + 12, st_spill);
        (void)clk_entry;
        atomicAdd(reinterpret_cast<unsigned long long *>(err + 16) + 10, __builtin_amdgcn_s_memrealtime() - rt0);
    }
#endif
#undef FAQCS_STAMP
}

// faqcs_terminal_n_flags(): bit 0 = the read's first base is an upper-case 'N', bit 1 = its last base is
__global__ __launch_bounds__(256) void terminal_n_flags(const uint8_t *__restrict__ seq, const uint32_t *__restrict__ off, const uint32_t n, uint8_t *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = off[i], b = off[i + 1];
    flags[i] = b > a ? (uint8_t)((seq[a] == 'N' ? 1u : 0u) | (seq[(size_t)b - 1] == 'N' ? 2u : 0u)) : (uint8_t)0;
}
hipError_t faqcs_launch_terminal_n_flags(const uint8_t *seq, const uint32_t *off, uint32_t n_reads, uint8_t *flags, hipStream_t st)
{
    if (!n_reads) return hipSuccess;
    hipLaunchKernelGGL(terminal_n_flags, dim3((n_reads + 255) / 256), dim3(256), 0, st, seq, off, n_reads, flags);
    return hipGetLastError();
}

static thread_local bool g_trim_lds_tail_folded = false;
bool faqcs_trim_lds_tail_folded() { return g_trim_lds_tail_folded; }
template <int C, bool WINDOWED, bool EXT, int LPR = 8, int RPC = 64>
static hipError_t launch_trim_lds(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                  uint32_t n_reads, const uint32_t *ad_sl, const uint16_t *ad_hit, faqcs_read_result *out,
                                  unsigned long long *rec_pre, unsigned long long *rec_post, uint64_t *counters, uint32_t *err,
                                  int n_cu, hipStream_t st, const uint8_t *tn_flags)
{
    constexpr int NW = lds_waves(C, LPR, RPC);
    constexpr size_t lds = (size_t)LdsCfg<C, NW, LPR, RPC>::lds_dwords() * 4;
    static unsigned long long attr_done = 0;
    auto kern = trim_lds<C, NW, WINDOWED, EXT, LPR, RPC>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const uint32_t chunks = (n_reads + RPC - 1) / RPC;
    uint32_t grid = (chunks + NW - 1) / NW;
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu; // one block per CU: its LDS holds a slot per wave
    if (grid == 0) return hipSuccess;
    // every block can take FAQCS_PARTIAL_FLUSHES x 1020 chunks: a launch the blocks could not take between them goes to another kernel
#ifdef FAQCS_LDS_TEST_FLUSH_CHUNKS
    if ((uint64_t)chunks > (uint64_t)grid * FAQCS_PARTIAL_FLUSHES * ((uint32_t)(FAQCS_LDS_TEST_FLUSH_CHUNKS) / NW * NW)) return hipErrorNotSupported; // (up to the brim)
#else
    if ((uint64_t)chunks > (uint64_t)grid * FAQCS_PARTIAL_FLUSHES * (65535u / RPC / NW * NW) * 3 / 4) return hipErrorNotSupported;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, P, seq, qual, off, n_reads, ad_sl, ad_hit,
                       reinterpret_cast<uint2 *>(out), rec_pre, rec_post, counters, err, tn_flags);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    g_trim_lds_tail_folded = lds_tail_folds_v<C, NW, LPR, RPC> && P.fold_n != 0; // (the launch folds the composition records P.fold_* names, all of them)
    hipLaunchKernelGGL((fold_partials<C, LPR>), dim3((RowCfg<C, LPR, lds_wq(C, LPR)>::N_ZERO + 63) / 64), dim3(1024), 0, st, P.partials, P.partial_rows, grid, counters, P.lay, err + 8);
    return hipGetLastError();
}

// Returns hipErrorNotSupported when the configuration is not one trim_lds is compiled for (the caller then takes the
// other trim kernels).
hipError_t faqcs_launch_trim_lds(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                                 uint32_t n_reads, uint32_t max_len, const uint32_t *ad_sl, const uint16_t *ad_hit,
                                 faqcs_read_result *out, unsigned long long *rec_pre, unsigned long long *rec_post,
                                 uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st, const uint8_t *tn_flags)
{
    g_trim_lds_tail_folded = false;
    const bool windowed = P.has_adapters || ((P.trim5 || P.trim3) && !P.qc_only);
    const bool plain = P.mode == FAQCS_MODE_BWA_PLUS && !P.protect5 && !P.qc_only && P.replace_q == 0 && !P.avgq_on &&
                       P.max_poly_n == 2 && P.dbg == 0;
    // every option set except --replace_to_N_q (its G -> N edit needs base and quality of a position together) and the ablation bits
    const bool ext = !plain && P.replace_q == 0 && P.dbg == 0;
    if (!plain && !ext) return hipErrorNotSupported;
#define FAQCS_LDS_ARGS P, seq, qual, off, n_reads, ad_sl, ad_hit, out, rec_pre, rec_post, counters, err, n_cu, st, tn_flags
#define FAQCS_LDS_CASE(C)                                                                                                 \
    return ext ? (windowed ? launch_trim_lds<C, true, true>(FAQCS_LDS_ARGS) : launch_trim_lds<C, false, true>(FAQCS_LDS_ARGS)) \
               : (windowed ? launch_trim_lds<C, true, false>(FAQCS_LDS_ARGS) : launch_trim_lds<C, false, false>(FAQCS_LDS_ARGS))
    // measured on MI355X (kernel-only, G reads/s, trim_lds vs trim_tpr): 2x100 7.46 vs 7.10 (C = 13), 2x125 5.64 vs 5.06 on the
    // C = 19 grid (4.44 on C = 16, whose 128-dword rows put every read of a half wave on the same banks), 2x150 5.8 vs 5.2;
    // (rounds 2-3: 153..160 bases and every multiple of 32 stayed on trim_tpr -- 8 waves with 160-wide slots; 2x128 at 3.2 G reads/s
    // here against 5.0 there, every lane of a lane-per-read pass on one LDS bank.  Round 4: 16 lanes per read with smaller chunks
    // from 153 bases on, padded rows for equal-length chunks of a multiple of 32 bases: 2x128 8.3, 2x155 5.2)
    // 4 lanes per read: reads of up to 76 bases (2x50, 2x75) -- sixteen reads per step of the position-parallel passes; FAQCS_TRIM_LDS4=0 leaves
    // them to trim_tpr as in rounds 1-3
    static const bool lds4_on = [] { const char *e = getenv("FAQCS_TRIM_LDS4"); return !e || atoi(e) != 0; }();
    if (lds4_on && max_len > 0 && max_len <= 52)
        return ext ? (windowed ? launch_trim_lds<13, true, true, 4>(FAQCS_LDS_ARGS) : launch_trim_lds<13, false, true, 4>(FAQCS_LDS_ARGS))
                   : (windowed ? launch_trim_lds<13, true, false, 4>(FAQCS_LDS_ARGS) : launch_trim_lds<13, false, false, 4>(FAQCS_LDS_ARGS));
    if (lds4_on && max_len > 52 && max_len <= 76)
        return ext ? (windowed ? launch_trim_lds<19, true, true, 4>(FAQCS_LDS_ARGS) : launch_trim_lds<19, false, true, 4>(FAQCS_LDS_ARGS))
                   : (windowed ? launch_trim_lds<19, true, false, 4>(FAQCS_LDS_ARGS) : launch_trim_lds<19, false, false, 4>(FAQCS_LDS_ARGS));
    if (max_len > 76 && max_len <= 104) FAQCS_LDS_CASE(13);  // 2x100
    if (max_len > 104 && max_len <= 152) FAQCS_LDS_CASE(19); // 2x125, 2x150
#undef FAQCS_LDS_CASE
    // 16 lanes per read: 153 ... 252 bases (2x250, 2x251); FAQCS_TRIM_LDS16=0 switches it off (A/B against trim_filter_accumulate)
    static const bool lds16_on = [] { const char *e = getenv("FAQCS_TRIM_LDS16"); return !e || atoi(e) != 0; }();
    if (lds16_on && max_len > 152 && max_len <= (uint32_t)lds_maxlen(16, 16))
        return ext ? (windowed ? launch_trim_lds<16, true, true, 16, FAQCS_LDS16_RPC>(FAQCS_LDS_ARGS) : launch_trim_lds<16, false, true, 16, FAQCS_LDS16_RPC>(FAQCS_LDS_ARGS))
                   : (windowed ? launch_trim_lds<16, true, false, 16, FAQCS_LDS16_RPC>(FAQCS_LDS_ARGS) : launch_trim_lds<16, false, false, 16, FAQCS_LDS16_RPC>(FAQCS_LDS_ARGS));
    // 253 ... 304 bases (2x300, 2x301): 16 lanes x 19 positions, chunks of 20 reads (6 KB slots: 12 waves beside a [42][352] quality matrix)
    DevParams Pw = P;
    Pw.wide_records = max_len > 256 ? 1u : 0u;
#undef FAQCS_LDS_ARGS
#define FAQCS_LDS_ARGS Pw, seq, qual, off, n_reads, ad_sl, ad_hit, out, rec_pre, rec_post, counters, err, n_cu, st, tn_flags
    if (lds16_on && max_len > (uint32_t)lds_maxlen(16, 16) && max_len <= (uint32_t)lds_maxlen(19, 16))
        return ext ? (windowed ? launch_trim_lds<19, true, true, 16, 20>(FAQCS_LDS_ARGS) : launch_trim_lds<19, false, true, 16, 20>(FAQCS_LDS_ARGS))
                   : (windowed ? launch_trim_lds<19, true, false, 16, 20>(FAQCS_LDS_ARGS) : launch_trim_lds<19, false, false, 16, 20>(FAQCS_LDS_ARGS));
#undef FAQCS_LDS_ARGS
    return hipErrorNotSupported;
}
