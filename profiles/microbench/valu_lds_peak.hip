// valu_lds_peak.hip -- what one MI355X CU sustains for the integer VALU and LDS instructions the trim / adapter kernels are
// made of, at 1 / 2 / 3 / 4 / 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_lds_peak valu_lds_peak.hip
// Output: cycles per wave-instruction per SIMD (shader clock from s_memtime, median over blocks) and chip-wide G wave-instr/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2000;
constexpr int PER_ITER = 32; // instructions per loop iteration (8 independent chains x 4)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(S) S S S S

// one instruction per chain; %[aK] are the 8 chain registers, %[b], %[c] constant VGPR operands
#define I_ADD(K) "v_add_u32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_PERM(K) "v_perm_b32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_ALIGNBIT(K) "v_alignbit_b32 %[a" #K "], %[a" #K "], %[b], 1\n\t"
#define I_ALIGNBYTE(K) "v_alignbyte_b32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_SDWA_SHL(K) "v_lshlrev_b32_sdwa %[a" #K "], %[b], %[a" #K "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define I_SDWA_MUL(K) "v_mul_u32_u24_sdwa %[a" #K "], %[b], %[a" #K "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
#define I_SDWA_SUB(K) "v_sub_u32_sdwa %[a" #K "], %[a" #K "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
#define I_SAD(K) "v_sad_u8 %[a" #K "], %[b], %[c], %[a" #K "]\n\t"
#define I_LSHL_ADD(K) "v_lshl_add_u32 %[a" #K "], %[a" #K "], 3, %[b]\n\t"
#define I_AND_OR(K) "v_and_or_b32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_BITOP3(K) "v_bitop3_b32 %[a" #K "], %[a" #K "], %[b], %[c] bitop3:0x96\n\t"
#define I_ADD3(K) "v_add3_u32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_BFE(K) "v_bfe_u32 %[a" #K "], %[a" #K "], 3, 9\n\t"
#define I_CNDMASK(K) "v_cndmask_b32 %[a" #K "], %[a" #K "], %[b], vcc\n\t"
#define I_CMP(K) "v_cmp_lt_u32 vcc, %[a" #K "], %[b]\n\t"
#define I_CMPX(K) "v_cmpx_le_i32 vcc, 0, %[b]\n\t"
#define I_MAX(K) "v_max_i32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_MAX3(K) "v_max3_u32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_DPP(K) "v_add_u32_dpp %[a" #K "], %[a" #K "], %[b] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MAD24(K) "v_mad_u32_u24 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_MUL_LO(K) "v_mul_lo_u32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_BCNT(K) "v_bcnt_u32_b32 %[a" #K "], %[b], %[a" #K "]\n\t"
#define I_PKADD(K) "v_pk_add_u16 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_DOT4(K) "v_dot4_u32_u8 %[a" #K "], %[b], %[c], %[a" #K "]\n\t"
#define I_MOV(K) "v_mov_b32 %[a" #K "], %[b]\n\t"
#define I_XOR(K) "v_xor_b32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_LSHR(K) "v_lshrrev_b32 %[a" #K "], 1, %[a" #K "]\n\t"
#define I_SALU(K) "s_add_u32 %[s0], %[s0], 1\n\t"
#define I_VALU_SALU(K) "v_add_u32 %[a" #K "], %[a" #K "], %[b]\n\ts_add_u32 %[s0], %[s0], 1\n\t"
#define I_READLANE(K) "v_readlane_b32 %[s0], %[a" #K "], 3\n\t"
#define I_FMA(K) "v_fma_f32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_AND(K) "v_and_b32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_OR(K) "v_or_b32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_SUB(K) "v_sub_u32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_SUBREV(K) "v_subrev_u32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_SHL(K) "v_lshlrev_b32 %[a" #K "], 3, %[a" #K "]\n\t"
#define I_SHLV(K) "v_lshlrev_b32 %[a" #K "], %[b], %[a" #K "]\n\t"
#define I_ASHR(K) "v_ashrrev_i32 %[a" #K "], 1, %[a" #K "]\n\t"
#define I_MIN(K) "v_min_u32 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_ADDLIT(K) "v_add_u32 %[a" #K "], 0x12345, %[a" #K "]\n\t"
#define I_ANDLIT(K) "v_and_b32 %[a" #K "], 0x7f7f7f7f, %[a" #K "]\n\t"
#define I_ADDS(K) "v_add_u32 %[a" #K "], %[s1], %[a" #K "]\n\t"
#define I_ADDCO(K) "v_add_co_u32 %[a" #K "], vcc, %[a" #K "], %[b]\n\t"
#define I_ADDC(K) "v_addc_co_u32 %[a" #K "], vcc, %[a" #K "], %[b], vcc\n\t"
#define I_MUL24(K) "v_mul_u32_u24 %[a" #K "], %[a" #K "], %[b]\n\t"
#define I_LSHL_OR(K) "v_lshl_or_b32 %[a" #K "], %[a" #K "], 3, %[b]\n\t"
#define I_XAD(K) "v_xad_u32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_OR3(K) "v_or3_b32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_BFI(K) "v_bfi_b32 %[a" #K "], %[c], %[a" #K "], %[b]\n\t"
#define I_NOT(K) "v_not_b32 %[a" #K "], %[a" #K "]\n\t"
#define I_CNDS(K) "v_cndmask_b32 %[a" #K "], %[a" #K "], %[b], %[sm]\n\t"
#define I_CMPS(K) "v_cmp_lt_u32 %[sm], %[a" #K "], %[b]\n\t"
#define I_CMP_CND(K) "v_cmp_lt_u32 vcc, %[a" #K "], %[b]\n\tv_cndmask_b32 %[a" #K "], %[a" #K "], %[c], vcc\n\t"
#define I_MBCNT(K) "v_mbcnt_lo_u32_b32 %[a" #K "], %[b], %[a" #K "]\n\t"
#define I_BITOP3_2(K) "v_bitop3_b32 %[a" #K "], %[a" #K "], %[b], %[c] bitop3:0xe8\n\t"
#define I_ADD_BITOP(K) "v_add_u32 %[a" #K "], %[a" #K "], %[b]\n\tv_bitop3_b32 %[a" #K "], %[a" #K "], %[b], %[c] bitop3:0x96\n\t"
#define I_ADD_PERM(K) "v_add_u32 %[a" #K "], %[a" #K "], %[b]\n\tv_perm_b32 %[a" #K "], %[a" #K "], %[b], %[c]\n\t"
#define I_LSHL64(K) "v_lshlrev_b64 %[p" #K "], 3, %[p" #K "]\n\t"
#define I_PKFMA(K) "v_add_u32 %[a" #K "], %[a" #K "], %[b]\n\t"

// LDS forms: address registers are the chains' own (fixed) addresses in d0..d7; the result lands in a0..a7
#define L_RD32(K) "ds_read_b32 %[a" #K "], %[d" #K "]\n\t"
#define L_RD64(K) "ds_read_b64 %[p" #K "], %[d" #K "]\n\t"
#define L_RD128(K) "ds_read_b128 %[q" #K "], %[d" #K "]\n\t"
#define L_RDU8(K) "ds_read_u8 %[a" #K "], %[d" #K "]\n\t"
#define L_ADD(K) "ds_add_u32 %[d" #K "], %[b]\n\t"
#define L_ADDRTN(K) "ds_add_rtn_u32 %[a" #K "], %[d" #K "], %[b]\n\t"
#define L_WR32(K) "ds_write_b32 %[d" #K "], %[b]\n\t"
#define L_BPERM(K) "ds_bpermute_b32 %[a" #K "], %[d" #K "], %[b]\n\t"

struct Res { unsigned long long clk; };

#define DEF_VALU(NAME, INS)                                                                                         \
    __global__ void k_##NAME(uint32_t *out, unsigned long long *clk, uint32_t bb, uint32_t cc)                      \
    {                                                                                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = bb + (threadIdx.x & 1), c = cc;                                                                 \
        uint32_t s0 = 0, s1 = bb >> 3;                                                                               \
        unsigned long long sm = cc;                                                                                  \
        uint64_t p0 = a0, p1 = a1, p2 = a2, p3 = a3, p4 = a4, p5 = a5, p6 = a6, p7 = a7;                             \
        asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");                                         \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                                            \
            asm volatile(BODY4(R8(INS))                                                                              \
                         : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), \
                           [a6] "+v"(a6), [a7] "+v"(a7), [s0] "+s"(s0), [sm] "+s"(sm), [p0] "+v"(p0), [p1] "+v"(p1),  \
                           [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6), [p7] "+v"(p7)  \
                         : [b] "v"(b), [c] "v"(c), [s1] "s"(s1)                                                      \
                         : "vcc");                                                                                   \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ (uint32_t)sm ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7);                     \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                                             \
    }

DEF_VALU(add, I_ADD) DEF_VALU(perm, I_PERM) DEF_VALU(alignbit, I_ALIGNBIT) DEF_VALU(alignbyte, I_ALIGNBYTE)
DEF_VALU(sdwa_shl, I_SDWA_SHL) DEF_VALU(sdwa_mul, I_SDWA_MUL) DEF_VALU(sdwa_sub, I_SDWA_SUB) DEF_VALU(sad, I_SAD)
DEF_VALU(lshl_add, I_LSHL_ADD) DEF_VALU(and_or, I_AND_OR) DEF_VALU(bitop3, I_BITOP3) DEF_VALU(add3, I_ADD3)
DEF_VALU(bfe, I_BFE) DEF_VALU(cndmask, I_CNDMASK) DEF_VALU(cmp, I_CMP) DEF_VALU(max, I_MAX) DEF_VALU(max3, I_MAX3)
DEF_VALU(dpp, I_DPP) DEF_VALU(mad24, I_MAD24) DEF_VALU(mul_lo, I_MUL_LO) DEF_VALU(bcnt, I_BCNT) DEF_VALU(pkadd, I_PKADD)
DEF_VALU(dot4, I_DOT4) DEF_VALU(mov, I_MOV) DEF_VALU(xor, I_XOR) DEF_VALU(lshr, I_LSHR) DEF_VALU(salu, I_SALU)
DEF_VALU(valu_salu, I_VALU_SALU) DEF_VALU(readlane, I_READLANE) DEF_VALU(fma, I_FMA)
DEF_VALU(and, I_AND) DEF_VALU(or, I_OR) DEF_VALU(sub, I_SUB) DEF_VALU(subrev, I_SUBREV) DEF_VALU(shl, I_SHL) DEF_VALU(shlv, I_SHLV) DEF_VALU(ashr, I_ASHR)
DEF_VALU(min, I_MIN) DEF_VALU(addlit, I_ADDLIT) DEF_VALU(andlit, I_ANDLIT) DEF_VALU(adds, I_ADDS) DEF_VALU(addco, I_ADDCO) DEF_VALU(addc, I_ADDC)
DEF_VALU(mul24, I_MUL24) DEF_VALU(lshl_or, I_LSHL_OR) DEF_VALU(xad, I_XAD) DEF_VALU(or3, I_OR3) DEF_VALU(bfi, I_BFI) DEF_VALU(not, I_NOT)
DEF_VALU(cnds, I_CNDS) DEF_VALU(cmps, I_CMPS) DEF_VALU(cmp_cnd, I_CMP_CND) DEF_VALU(mbcnt, I_MBCNT) DEF_VALU(bitop3_maj, I_BITOP3_2)
DEF_VALU(add_bitop, I_ADD_BITOP) DEF_VALU(add_perm, I_ADD_PERM) DEF_VALU(lshl64, I_LSHL64)

// LDS kernels.  MODE: address pattern of the 8 chains (lane stride in bytes, + chain offset)
#define DEF_LDS(NAME, INS, WAITEVERY)                                                                               \
    __global__ void k_##NAME(uint32_t *out, unsigned long long *clk, uint32_t stride, uint32_t cstride)            \
    {                                                                                                                \
        extern __shared__ uint32_t sm[];                                                                             \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = (uint32_t)i * 4u;                              \
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                                     \
        uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;                                     \
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));                                                     \
        u4 q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;                                           \
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;                                               \
        const uint32_t base = (lane * stride + wv * 64u) & 0xffffu;                                                  \
        uint32_t d0 = (base + 0 * cstride) & 0xfff0u, d1 = (base + 1 * cstride) & 0xfff0u, d2 = (base + 2 * cstride) & 0xfff0u, \
                 d3 = (base + 3 * cstride) & 0xfff0u, d4 = (base + 4 * cstride) & 0xfff0u, d5 = (base + 5 * cstride) & 0xfff0u, \
                 d6 = (base + 6 * cstride) & 0xfff0u, d7 = (base + 7 * cstride) & 0xfff0u;                           \
        if (stride & 3u) { d0 = base; d1 = base + cstride; d2 = base + 2 * cstride; d3 = base + 3 * cstride; d4 = base + 4 * cstride; d5 = base + 5 * cstride; d6 = base + 6 * cstride; d7 = base + 7 * cstride; } \
        uint32_t b = 1;                                                                                              \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                                            \
            asm volatile(BODY4(R8(INS)) WAITEVERY                                                                    \
                         : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), \
                           [a6] "+v"(a6), [a7] "+v"(a7), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), \
                           [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6), [p7] "+v"(p7), [q0] "+v"(q0), [q1] "+v"(q1), \
                           [q2] "+v"(q2), [q3] "+v"(q3), [q4] "+v"(q4), [q5] "+v"(q5), [q6] "+v"(q6), [q7] "+v"(q7)  \
                         : [b] "v"(b), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [d4] "v"(d4),        \
                           [d5] "v"(d5), [d6] "v"(d6), [d7] "v"(d7)                                                  \
                         : "memory");                                                                                \
        }                                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7) ^ q0.x ^ q1.y ^ q2.z ^ q3.w ^ q4.x ^ q5.x ^ q6.x ^ q7.x; \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                                             \
    }

DEF_LDS(rd32, L_RD32, "s_waitcnt lgkmcnt(0)\n\t") DEF_LDS(rd64, L_RD64, "s_waitcnt lgkmcnt(0)\n\t")
DEF_LDS(rd128, L_RD128, "s_waitcnt lgkmcnt(0)\n\t") DEF_LDS(rdu8, L_RDU8, "s_waitcnt lgkmcnt(0)\n\t")
DEF_LDS(ldsadd, L_ADD, "s_waitcnt lgkmcnt(0)\n\t") DEF_LDS(addrtn, L_ADDRTN, "s_waitcnt lgkmcnt(0)\n\t")
DEF_LDS(wr32, L_WR32, "s_waitcnt lgkmcnt(0)\n\t") DEF_LDS(bperm, L_BPERM, "s_waitcnt lgkmcnt(0)\n\t")

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t, uint32_t);
struct Entry { const char *name; kern_t k; bool lds; uint32_t p0, p1; int per_iter; };

int main(int argc, char **argv)
{
    int n_cu = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    n_cu = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;
    printf("# device %s, %d CUs, clockRate %.3f GHz\n", prop.name, n_cu, clock_ghz);
    uint32_t *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)n_cu * 2048 * 4 * 4));
    CHECK(hipMalloc(&clk, (size_t)n_cu * 8 * 8));
    std::vector<Entry> es = {
#define V(N) {#N, k_##N, false, 0x03020100u, 0x07060504u, PER_ITER}
        V(add), V(xor), V(lshr), V(mov), V(perm), V(alignbit), V(alignbyte), V(sdwa_shl), V(sdwa_mul), V(sdwa_sub), V(sad), V(lshl_add), V(and_or),
        V(bitop3), V(add3), V(bfe), V(cndmask), V(cmp), V(max), V(max3), V(dpp), V(mad24), V(mul_lo), V(bcnt), V(pkadd), V(dot4),
        V(salu), V(valu_salu), V(readlane), V(fma),
        V(and), V(or), V(sub), V(subrev), V(shl), V(shlv), V(ashr), V(min), V(addlit), V(andlit), V(adds), V(addco), V(addc), V(mul24), V(lshl_or), V(xad), V(or3),
        V(bfi), V(not), V(cnds), V(cmps), V(mbcnt), V(bitop3_maj), V(lshl64),
        {"cmp_cnd", k_cmp_cnd, false, 0x03020100u, 0x07060504u, 2 * PER_ITER}, {"add_bitop", k_add_bitop, false, 0x03020100u, 0x07060504u, 2 * PER_ITER},
        {"add_perm", k_add_perm, false, 0x03020100u, 0x07060504u, 2 * PER_ITER},
#undef V
        // LDS: (lane stride bytes, chain stride bytes)
        {"ds_read_b32 lane*4 (conflict-free)", k_rd32, true, 4, 256, PER_ITER},
        {"ds_read_b32 lane*152 (row per lane, L=152)", k_rd32, true, 152, 4, PER_ITER},
        {"ds_read_b32 lane*150&~3 (L=150)", k_rd32, true, 150, 4, PER_ITER},
        {"ds_read_b32 same address (broadcast)", k_rd32, true, 0, 4, PER_ITER},
        {"ds_read_b64 lane*8", k_rd64, true, 8, 512, PER_ITER},
        {"ds_read_b64 lane*152", k_rd64, true, 152, 8, PER_ITER},
        {"ds_read_b128 lane*16", k_rd128, true, 16, 1024, PER_ITER},
        {"ds_read_b128 lane*152 (8B aligned rows)", k_rd128, true, 152, 16, PER_ITER},
        {"ds_read_b128 lane*160", k_rd128, true, 160, 16, PER_ITER},
        {"ds_read_b128 lane*144", k_rd128, true, 144, 16, PER_ITER},
        {"ds_read_u8 lane*1", k_rdu8, true, 1, 64, PER_ITER},
        {"ds_add_u32 lane*4", k_ldsadd, true, 4, 256, PER_ITER},
        {"ds_add_u32 lane*8 (2-way)", k_ldsadd, true, 8, 256, PER_ITER},
        {"ds_add_u32 lane*128 (32-way)", k_ldsadd, true, 128, 4, PER_ITER},
        {"ds_add_u32 same address", k_ldsadd, true, 0, 4, PER_ITER},
        {"ds_add_rtn_u32 lane*4", k_addrtn, true, 4, 256, PER_ITER},
        {"ds_write_b32 lane*4", k_wr32, true, 4, 256, PER_ITER},
        {"ds_bpermute_b32", k_bperm, true, 4, 0, PER_ITER},
    };
    const char *only = argc > 1 ? argv[1] : nullptr;
    printf("%-46s %5s %12s %14s %14s\n", "instruction", "w/SIMD", "cyc/instr/SIMD", "cyc/instr/CU", "chip G instr/s");
    for (auto &e : es) {
        if (only && strcmp(e.name, only) != 0 && !(only[0] == '~' && strstr(e.name, only + 1))) continue;
        for (int wps : {1, 2, 3, 4, 6, 8}) {
            const int threads = 256, blocks_per_cu = wps; // 256 threads = one wave per SIMD
            const int grid = n_cu * blocks_per_cu;
            const size_t lds = e.lds ? 65536 / 1 : 0;
            if (e.lds && wps * 65536 > 160 * 1024 && wps > 2) {
                // more than two 64 KB blocks do not fit: use 16 KB of table per block beyond that (addresses are masked to 64 KB, so keep 64 KB and fewer waves)
            }
            int bpc = blocks_per_cu, thr = threads;
            size_t l = lds;
            if (e.lds) { bpc = 1; thr = 256 * wps; if (thr > 1024) { bpc = 2; thr /= 2; } l = 65536; } // one or two big blocks: all waves share one 64 KB table
            const int g = n_cu * bpc;
            CHECK(hipMemset(clk, 0, (size_t)n_cu * 8 * 8));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(e.k, dim3(g), dim3(thr), l, 0, out, clk, e.p0, e.p1); // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(g), dim3(thr), l, 0, out, clk, e.p0, e.p1);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(g);
            CHECK(hipMemcpy(h.data(), clk, g * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[g / 2];
            const double instr_per_wave = (double)ITERS * e.per_iter;
            const double cyc_per_instr_simd = med / (instr_per_wave * wps);      // each SIMD hosts wps waves
            const double total_instr = instr_per_wave * (double)g * (thr / 64);
            printf("%-46s %5d %14.2f %14.3f %14.1f   (%.3f ms, s_memtime-clock %.2f GHz)\n", e.name, wps, cyc_per_instr_simd, cyc_per_instr_simd / 4.0,
                   total_instr / (ms * 1e-3) * 1e-9, ms, med / (ms * 1e-3) * 1e-9 );
            fflush(stdout);
            (void)bpc;
        }
    }
    return 0;
}
