// faqcs_dev.h -- shared between the host side (faqcs_capi.hip) and the gfx950 kernels.
//
// Data layout in HBM:
//   seq / qual      byte arenas, reads packed back to back, read i = [offset[i], offset[i+1])
//   offset          u32[n+1]
//   adapter_sl      u32[n]   (only with adapters) : first | second << 16  (Read::start_length, FaQCs.h:154)
//   adapter_hit     u16[n]   1 + credited adapter index
//   result          faqcs_read_result[n]  (8 B)
//   comp_pre/post   u64[n]   per-read composition record: valid<<63 | len | nA<<9 | nT<<18 | nC<<27 | nG<<36 | nN<<45
//                            (reads <= 256 bases; the long-read kernels write two words with 11-bit fields)
//   counters        u64[layout.total]     additive block (include/faqcs_mi.h faqcs_layout)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/faqcs_mi.h"

#define FAQCS_WAVE 64
#define FAQCS_FAST_READ_LENGTH 1024 /* longest read the chunked trim kernels and composition_histogram take; longer reads: trim_long */
#define FAQCS_TAB_LEN FAQCS_FAST_READ_LENGTH /* per-length lookup tables: every length the chunked kernels support */

// Everything the kernels need from faqcs_params + host-precomputed integer lookup tables, passed by value.
struct DevParams {
    int32_t mode, Q, in_off, out_off;
    uint32_t min_len, max_poly_n, trim5, trim3, replace_q;
    uint32_t protect5, qc_only, has_adapters, avgq_on;
    uint32_t R;                   // row capacity of the global matrices
    uint32_t n_adapters;
    uint32_t dbg;                 // FAQCS_DBG ablation bits (diagnostics only; 0 in production)
    uint32_t wide_records;        // (set per launch by faqcs_launch_trim_lds) the batch's longest read has more than 256 bases: two-word composition records
    float lc_ratio, avg_q;        // --lc / --avg_q as given (trim_long evaluates the reference's float expressions directly; the chunked
                                  // kernels use the per-length integer tables below)
    // per-length tables, index 0..FAQCS_TAB_LEN (float32 semantics folded into integers on the host)
    const uint32_t *lc_thr;       // lo16: min base count that trips `count*float(1.0/len) > lc` (trim.cpp:483-488)
                                  // hi16: min transition count that trips `dc*(norm*2) > lc`     (trim.cpp:499-503)
    const int32_t  *avgq_min_v;   // min V = sum(raw - offset) with NOT(ave_Q < --avg_q)          (trim.cpp:376)
    const uint32_t *div_magic;    // floor(2^32/len)+1: floor(V/len) == mulhi(V, magic), V < 2^16 (trim.cpp:254,539 int())
    const float    *comp_norm;    // float(10000)/len                                             (trim.cpp:860)
    const uint32_t *base_tab;     // [256] per input byte: 6-bit count fields A,T,C,G,N | isG<<30 | isN<<31
    uint32_t *partials;           // [n_cu][FAQCS_PARTIAL_FLUSHES][FAQCS_PARTIAL_ROW] + [n_cu] rows used: a row per block and flush (trim_lds: flush without
                                  // global atomics; a fold kernel behind the trim kernel adds the rows a block USED in this launch to the counter
                                  // block -- stale rows are never read, so nothing is zeroed)
    uint32_t *partial_rows;       // [n_cu] rows of `partials` each block of the last trim_lds launch wrote
    faqcs_layout lay;
    // (set per launch by the host; round 6) the composition records of the PREVIOUS launch: a trim_lds block that has run out of chunks folds
    // them in the LDS it no longer needs, while the slowest blocks finish -- the fold used to run in the seam between two launches because
    // its 120 KB table cannot share a CU with a 160 KB trim_lds block (8.5 % of the headline step).  fold_n == 0: nothing to fold.
    const unsigned long long *fold_pre, *fold_post;
    uint32_t fold_n;
    uint32_t *fold_claim;         // [2] chunks of fold_pre / fold_post handed out so far (zero when the launch starts)
    uint64_t *fold_dst_pre, *fold_dst_post;
};

enum { FS_SLOTS = 32 };
enum { FAQCS_PARTIAL_ROW = 16896 };  // >= N_ZERO of every trim_lds variant (RowCfg<19, 8, 160>: 7 962, <16, 16, 288>: 13 792, <19, 16, 352>: 16 768): dwords of one flushed copy of a block's LDS accumulators
enum { FAQCS_PARTIAL_FLUSHES = 8 };  // flushes (rows) a block has room for in one launch: it stops claiming chunks before it would need more

// base_tab fields (6 bits each so a lane can sum up to 63 reads before flushing)
#define BT_SHIFT(code) (6 * (code))
#define BT_FIELDS 0x3fffffffu
#define BT_IS_NU (1u << 31) /* upper-case 'N' exactly (count_poly_n / terminal-N masking are case sensitive); the sign
                               bit so that one v_alignbit per position gathers the flags into a bit mask */
#define BT_IS_GU (1u << 30) /* upper-case 'G' exactly (--replace_to_N_q) */

// composition record
#define CR_VALID (1ull << 63)

#ifdef __HIPCC__
// ---- DPP primitives ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// full-wave (64 lanes)
__device__ __forceinline__ int wave_incl_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_add(v), 63); }

// 16-lane rows: every lane of a row ends up with the row's result (butterfly: xor1, xor2, half-mirror, mirror)
#define FAQCS_ROW_ALL(OP)                                                              \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));                \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));                \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));               \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
__device__ __forceinline__ int op_add_(int a, int b) { return a + b; }
__device__ __forceinline__ int op_or_(int a, int b) { return a | b; }
__device__ __forceinline__ int op_umax_(int a, int b) { return (int)umax_((uint32_t)a, (uint32_t)b); }
__device__ __forceinline__ int op_imax_(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int row_all_sum(int v) { FAQCS_ROW_ALL(op_add_) return v; }
__device__ __forceinline__ uint32_t row_all_or(uint32_t x) { int v = (int)x; FAQCS_ROW_ALL(op_or_) return (uint32_t)v; }
__device__ __forceinline__ uint32_t row_all_umax(uint32_t x) { int v = (int)x; FAQCS_ROW_ALL(op_umax_) return (uint32_t)v; }
__device__ __forceinline__ int row_all_imax(int v) { FAQCS_ROW_ALL(op_imax_) return v; }
// inclusive prefix sum inside each row of 16 lanes
__device__ __forceinline__ int row_incl_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}
// value of the next / previous lane of the same row (0 at the row edge)
__device__ __forceinline__ uint32_t row_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t row_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t row_incl_scan_umax(uint32_t x)
{
    int v = (int)x;
    v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    return (uint32_t)v;
}

// The trim kernel is written against RowOps<LPR>: LPR = lanes that share one read.  16 = one DPP row per read
// (4 reads per wave, reads <= 256 bases), 64 = the whole wave on one read (long reads, <= 1024 bases).
template <int LPR> struct RowOps;
template <> struct RowOps<16> {
    static __device__ __forceinline__ int all_sum(int v) { return row_all_sum(v); }
    static __device__ __forceinline__ uint32_t all_or(uint32_t v) { return row_all_or(v); }
    static __device__ __forceinline__ uint32_t all_umax(uint32_t v) { return row_all_umax(v); }
    static __device__ __forceinline__ int incl_scan_add(int v) { return row_incl_scan_add(v); }
    static __device__ __forceinline__ uint32_t incl_scan_umax(uint32_t v) { return row_incl_scan_umax(v); }
    static __device__ __forceinline__ uint32_t next(uint32_t v) { return row_next(v); }
    static __device__ __forceinline__ uint32_t prev(uint32_t v) { return row_prev(v); }
};
// 8 lanes per read (8 reads per wave, reads <= 160 bases): the row-uniform scalar work is shared by twice as many reads.
// Reductions are three butterfly steps (xor 1, xor 2, mirror inside the half row); scans and neighbour shifts mask the
// lanes whose source would sit in the other read of the 16-lane DPP row.
template <> struct RowOps<8> {
#define FAQCS_HALF_ALL(OP)                                                             \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));                \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));                \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    static __device__ __forceinline__ int all_sum(int v) { FAQCS_HALF_ALL(op_add_) return v; }
    static __device__ __forceinline__ uint32_t all_or(uint32_t x) { int v = (int)x; FAQCS_HALF_ALL(op_or_) return (uint32_t)v; }
    static __device__ __forceinline__ uint32_t all_umax(uint32_t x) { int v = (int)x; FAQCS_HALF_ALL(op_umax_) return (uint32_t)v; }
#undef FAQCS_HALF_ALL
    // -1 when the lane's position inside its 8-lane group is >= k
    static __device__ __forceinline__ int ge_(int k) { return ((int)(threadIdx.x & 7u) >= k) ? -1 : 0; }
    static __device__ __forceinline__ int incl_scan_add(int v)
    {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false) & ge_(1);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false) & ge_(2);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xa, false); // row_shr:4 into banks 1 and 3 only (lanes 4-7, 12-15)
        return v;
    }
    static __device__ __forceinline__ uint32_t incl_scan_umax(uint32_t x)
    {
        int v = (int)x;
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false) & ge_(1));
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false) & ge_(2));
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false) & ge_(4));
        return (uint32_t)v;
    }
    static __device__ __forceinline__ uint32_t next(uint32_t v)
    {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, false) & (((threadIdx.x & 7u) == 7u) ? 0u : 0xffffffffu);
    }
    static __device__ __forceinline__ uint32_t prev(uint32_t v)
    {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false) & (uint32_t)ge_(1);
    }
};
// 32 lanes per read (two reads per wave, reads of 257..512 bases): row butterflies / scans plus ONE cross-row step --
// ds_swizzle (xor 16 inside each group of 32 lanes, no LDS memory touched) for reductions, row_bcast:15 into the odd rows
// for scans.
// 4 lanes per read (16 reads per wave, reads <= 76 bases): everything stays inside a DPP quad.
template <> struct RowOps<4> {
#define FAQCS_QUAD_ALL(OP)                                                             \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));                \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    static __device__ __forceinline__ int all_sum(int v) { FAQCS_QUAD_ALL(op_add_) return v; }
    static __device__ __forceinline__ uint32_t all_or(uint32_t x) { int v = (int)x; FAQCS_QUAD_ALL(op_or_) return (uint32_t)v; }
    static __device__ __forceinline__ uint32_t all_umax(uint32_t x) { int v = (int)x; FAQCS_QUAD_ALL(op_umax_) return (uint32_t)v; }
#undef FAQCS_QUAD_ALL
    static __device__ __forceinline__ int ge_(int k) { return ((int)(threadIdx.x & 3u) >= k) ? -1 : 0; }
    static __device__ __forceinline__ int incl_scan_add(int v)
    {
        v += __builtin_amdgcn_update_dpp(0, v, 0x90, 0xf, 0xf, false) & ge_(1); // quad_perm [0,0,1,2]: lane i <- i-1
        v += __builtin_amdgcn_update_dpp(0, v, 0x44, 0xf, 0xf, false) & ge_(2); // quad_perm [0,1,0,1]: lane i <- i-2
        return v;
    }
    static __device__ __forceinline__ uint32_t incl_scan_umax(uint32_t x)
    {
        int v = (int)x;
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x90, 0xf, 0xf, false) & ge_(1));
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x44, 0xf, 0xf, false) & ge_(2));
        return (uint32_t)v;
    }
    static __device__ __forceinline__ uint32_t next(uint32_t v)
    {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF9, 0xf, 0xf, false) & (((threadIdx.x & 3u) == 3u) ? 0u : 0xffffffffu); // [1,2,3,3]
    }
    static __device__ __forceinline__ uint32_t prev(uint32_t v)
    {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x90, 0xf, 0xf, false) & (uint32_t)ge_(1);
    }
};
template <> struct RowOps<32> {
    static __device__ __forceinline__ int swap_rows(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x401f); } // and 0x1f, or 0, xor 0x10
    static __device__ __forceinline__ int all_sum(int v) { v = row_all_sum(v); return v + swap_rows(v); }
    static __device__ __forceinline__ uint32_t all_or(uint32_t x) { const int v = (int)row_all_or(x); return (uint32_t)(v | swap_rows(v)); }
    static __device__ __forceinline__ uint32_t all_umax(uint32_t x) { const int v = (int)row_all_umax(x); return (uint32_t)op_umax_(v, swap_rows(v)); }
    static __device__ __forceinline__ int incl_scan_add(int v)
    {
        v = row_incl_scan_add(v);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
        return v;
    }
    static __device__ __forceinline__ uint32_t incl_scan_umax(uint32_t x)
    {
        int v = (int)row_incl_scan_umax(x);
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
        return (uint32_t)v;
    }
    static __device__ __forceinline__ uint32_t next(uint32_t v)
    {
        const int l = (int)(threadIdx.x & 63u);
        const uint32_t x = (uint32_t)__shfl((int)v, (l + 1) & 63);
        return (l & 31) == 31 ? 0u : x;
    }
    static __device__ __forceinline__ uint32_t prev(uint32_t v)
    {
        const int l = (int)(threadIdx.x & 63u);
        const uint32_t x = (uint32_t)__shfl((int)v, (l + 63) & 63);
        return (l & 31) == 0 ? 0u : x;
    }
};
template <> struct RowOps<64> {
    static __device__ __forceinline__ int all_sum(int v) { return wave_sum_i32(v); }
    static __device__ __forceinline__ uint32_t all_umax(uint32_t v) { return wave_max_u32(v); }
    static __device__ __forceinline__ uint32_t all_or(uint32_t x)
    {
        int v = (int)x;
        v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
        return (uint32_t)__builtin_amdgcn_readlane(v, 63);
    }
    static __device__ __forceinline__ int incl_scan_add(int v) { return wave_incl_scan_add(v); }
    static __device__ __forceinline__ uint32_t incl_scan_umax(uint32_t x)
    {
        int v = (int)row_incl_scan_umax(x);
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
        v = op_umax_(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
        return (uint32_t)v;
    }
    // value of lane +/- 1 across the whole wave, 0 at the wave edge (ds_bpermute: the long-read path is not DPP-tuned)
    static __device__ __forceinline__ uint32_t next(uint32_t v)
    {
        const int l = (int)(threadIdx.x & 63u);
        const uint32_t x = (uint32_t)__shfl((int)v, (l + 1) & 63);
        return l == 63 ? 0u : x;
    }
    static __device__ __forceinline__ uint32_t prev(uint32_t v)
    {
        const int l = (int)(threadIdx.x & 63u);
        const uint32_t x = (uint32_t)__shfl((int)v, (l + 63) & 63);
        return l == 0 ? 0u : x;
    }
};
#endif
