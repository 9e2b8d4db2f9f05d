// faqcs_dev.h -- shared between the host side (faqcs_capi.hip) and the gfx950 kernels.
//
// Data layout in HBM:
//   seq / qual      byte arenas, reads packed back to back, read i = [offset[i], offset[i+1])
//   offset          u32[n+1]
//   adapter_sl      u32[n]   (only with adapters) : first | second << 16  (Read::start_length, FaQCs.h:154)
//   adapter_hit     u16[n]   1 + credited adapter index
//   result          faqcs_read_result[n]  (8 B)
//   counters        u64[layout.total]     additive block (include/faqcs_mi.h faqcs_layout)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/faqcs_mi.h"

#define FAQCS_WAVE 64
#define FAQCS_TAB_LEN 4096 /* per-length lookup tables cover every supported read length */

// Everything the kernels need from faqcs_params + host-precomputed integer lookup tables, passed by value.
struct DevParams {
    int32_t mode, Q, in_off, out_off;
    uint32_t min_len, max_poly_n, trim5, trim3, replace_q;
    uint32_t protect5, qc_only, has_adapters, avgq_on;
    uint32_t R;                   // row capacity of the global matrices
    uint32_t n_adapters;
    // per-length tables, index 0..FAQCS_TAB_LEN (float32 semantics folded into integers on the host)
    const uint16_t *mono_thr;     // min base count that trips `count*float(1.0/len) > lc`        (trim.cpp:483-488)
    const uint16_t *di_thr;       // min transition count that trips `dc*(norm*2) > lc`             (trim.cpp:499-503)
    const uint32_t *avgq_min_sum; // min biased quality sum with NOT(ave_Q < --avg_q)               (trim.cpp:376)
    const float    *comp_norm;    // float(10000)/len                                               (trim.cpp:860)
    const uint64_t *div_magic;    // ceil(2^44/len): floor(S/len) == (S*magic)>>44 for S < 2^20.. (trim.cpp:572 int())
    faqcs_layout lay;
};

// per-wave filter-stat accumulators kept in LDS per block, flushed with one global atomic each
enum { FS_SLOTS = 32 };

// composition / small-histogram updates go through an LDS hash table: key = slot | bin << 5
enum {
    HS_PRE_COMP = 0,   // +kind (0..5)   bin = composition bin
    HS_POST_COMP = 6,  // +kind
    HS_PRE_LEN = 12, HS_POST_LEN = 13, HS_PRE_RQ = 14, HS_POST_RQ = 15, HS_PRE_BQ = 16, HS_POST_BQ = 17,
    HS_NSLOT = 18
};
#define HS_EMPTY 0xffffffffu

#ifdef __HIPCC__
// ---- wave64 primitives on DPP (no LDS traffic) ---------------------------------------------------------
#define FAQCS_DPP(op, v, ctrl, rm) op(v, __builtin_amdgcn_update_dpp(0, v, ctrl, rm, 0xf, false))
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
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
// max over the wave of an unsigned key (identity 0); result is wave-uniform
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
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#endif
