/*
 * faqcs_oracle.c -- TEST INFRASTRUCTURE ONLY (see faqcs_oracle.h).
 *
 * Plain-C restatement of the reference algorithm of the FaQCs per-read hot path.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference).  Written for clarity, not
 * speed: one read at a time, byte loops, no SIMD.  Float arithmetic is kept in binary32 exactly where
 * the reference computes in `float`; build with -ffp-contract=off.
 */
#include "faqcs_oracle.h"

#include <stdlib.h>
#include <string.h>

#define NQ FAQCS_NQ

/* ------------------------------------------------------------------------------------------------
 * counter layout -- duplicated on purpose from the product library (tests assert both agree)
 * ---------------------------------------------------------------------------------------------- */
static void oracle_layout(uint32_t R, uint32_t n_adapters, faqcs_layout *L)
{
    uint64_t o = 0;
    memset(L, 0, sizeof(*L));
    L->max_read_length = R;
    L->n_adapters = n_adapters;
    L->filter_stats = o;    o += 32;
    L->pre_read_qhist = o;  o += NQ;
    L->pre_base_qhist = o;  o += NQ;
    L->post_read_qhist = o; o += NQ;
    L->post_base_qhist = o; o += NQ;
    L->pre_len_hist = o;    o += (uint64_t)R + 1;
    L->post_len_hist = o;   o += (uint64_t)R + 1;
    L->pre_qual = o;        o += (uint64_t)R * NQ;
    L->post_qual = o;       o += (uint64_t)R * NQ;
    L->pre_base = o;        o += (uint64_t)R * FAQCS_NBASE;
    L->post_base = o;       o += (uint64_t)R * FAQCS_NBASE;
    L->pre_comp = o;        o += (uint64_t)FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND;
    L->post_comp = o;       o += (uint64_t)FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND;
    L->adapter_stats = o;   o += (uint64_t)n_adapters * 2;
    L->total = o;
}

/* ------------------------------------------------------------------------------------------------
 * k-mer table: stands in for std::unordered_map<size_t,size_t> (trim.cpp:82,928)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t *key; uint64_t *cnt; uint64_t cap, used; } kmer_map;

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void kmer_map_init(kmer_map *m, uint64_t cap)
{
    m->cap = cap; m->used = 0;
    m->key = (uint64_t *)calloc(cap, 8); m->cnt = (uint64_t *)calloc(cap, 8);
}
static void kmer_map_add(kmer_map *m, uint64_t k, uint64_t c);
static void kmer_map_grow(kmer_map *m)
{
    kmer_map n; kmer_map_init(&n, m->cap * 2);
    for (uint64_t i = 0; i < m->cap; ++i) if (m->key[i]) kmer_map_add(&n, m->key[i] - 1, m->cnt[i]);
    free(m->key); free(m->cnt); *m = n;
}
static void kmer_map_add(kmer_map *m, uint64_t k, uint64_t c)
{
    if ((m->used + 1) * 10 > m->cap * 7) kmer_map_grow(m);
    uint64_t h = mix64(k) & (m->cap - 1);
    while (m->key[h] && m->key[h] != k + 1) h = (h + 1) & (m->cap - 1);
    if (!m->key[h]) { m->key[h] = k + 1; m->used++; }
    m->cnt[h] += c;
}

struct faqcs_oracle {
    faqcs_params p;
    char **adapter;       /* owned copies */
    uint32_t *adapter_len;
    faqcs_layout lay;
    kmer_map kmers;
    int kmer_active;      /* Options::kmer_rarefaction, mutable (trim.cpp:183) */
    faqcs_rarefaction *points; uint32_t n_points, cap_points;
    uint64_t *hist_count, *hist_nkeys, n_hist;   /* PlotInfo::kmer_frequency_histogram (FaQCs.h:228) */
};

faqcs_oracle *faqcs_oracle_create(const faqcs_params *p)
{
    if (!p || p->max_read_length == 0 || p->max_read_length > FAQCS_MAX_READ_LENGTH) return NULL;
    faqcs_oracle *o = (faqcs_oracle *)calloc(1, sizeof(*o));
    o->p = *p;
    o->adapter = (char **)calloc(p->n_adapters + 1, sizeof(char *));
    o->adapter_len = (uint32_t *)calloc(p->n_adapters + 1, sizeof(uint32_t));
    for (uint32_t j = 0; j < p->n_adapters; ++j) {
        o->adapter_len[j] = (uint32_t)strlen(p->adapter_seq[j]);
        o->adapter[j] = (char *)malloc(o->adapter_len[j] + 1);
        memcpy(o->adapter[j], p->adapter_seq[j], o->adapter_len[j] + 1);
    }
    o->p.adapter_seq = NULL;
    oracle_layout(p->max_read_length, p->n_adapters, &o->lay);
    kmer_map_init(&o->kmers, 1u << 16);
    o->kmer_active = p->kmer_rarefaction != 0;
    return o;
}

void faqcs_oracle_destroy(faqcs_oracle *o)
{
    if (!o) return;
    for (uint32_t j = 0; j < o->p.n_adapters; ++j) free(o->adapter[j]);
    free(o->adapter); free(o->adapter_len); free(o->kmers.key); free(o->kmers.cnt); free(o->points);
    free(o->hist_count); free(o->hist_nkeys); free(o);
}

/* fastq.h:17-36 quality_score(): max(0, q - offset); > 41 throws.  Returns -1 for the throw. */
static int qscore(uint8_t q, int offset)
{
    int v = (int)(int8_t)q - offset; /* `char` is signed on x86-64 */
    if (v < 0) v = 0;
    return v > 41 ? -1 : v;
}

/* ------------------------------------------------------------------------------------------------
 * quality trimmers
 * ---------------------------------------------------------------------------------------------- */
/* trim.cpp:714-793 */
static uint32_t bwa_plus_trim(const uint8_t *qual, int len, int Q, int off, int protect5, uint32_t *kept)
{
    int at_least_scan = len < 5 ? len : 5;
    const int num_after_neg = len < 2 ? len : 2;
    int pos_3 = len - 1, final_pos_5 = 0, final_pos_3 = pos_3, area = 0, maxArea = 0;
    while (at_least_scan) {                                   /* :732-749 */
        --at_least_scan;
        if (pos_3 > num_after_neg && area >= 0) at_least_scan = num_after_neg;
        area += Q - qscore(qual[pos_3], off);
        if (area > maxArea) { maxArea = area; final_pos_3 = pos_3 - 1; }
        --pos_3;
    }
    if (!protect5) {                                          /* :752-779 */
        int pos_5 = 0;
        maxArea = 0; area = 0;
        at_least_scan = len < 5 ? len : 5;
        while (at_least_scan) {
            --at_least_scan;
            if (pos_5 < final_pos_3 - num_after_neg && area >= 0) at_least_scan = num_after_neg;
            area += Q - qscore(qual[pos_5], off);
            if (area > maxArea) { maxArea = area; final_pos_5 = pos_5 + 1; }
            ++pos_5;
        }
    }
    *kept = final_pos_3 <= final_pos_5 ? 0u : (uint32_t)(final_pos_3 - final_pos_5 + 1); /* :781-790 */
    return (uint32_t)final_pos_5;
}

/* trim.cpp:675-709 */
static uint32_t bwa_trim(const uint8_t *qual, int len, int Q, int off, uint32_t *kept)
{
    int pos_3 = len - 1, final_pos_3 = pos_3, area = 0, maxArea = 0;
    while (pos_3 > 0 && area >= 0) {
        area += Q - qscore(qual[pos_3], off);
        if (area > maxArea) { maxArea = area; final_pos_3 = pos_3 - 1; }
        --pos_3;
    }
    *kept = (uint32_t)(final_pos_3 + 1);
    return 0;
}

/* trim.cpp:629-672 */
static uint32_t hard_trim(const uint8_t *qual, int len, int Q, int off, int protect5, uint32_t *kept)
{
    int pos_3 = len - 1, final_pos_5 = 0, final_pos_3 = pos_3;
    while (pos_3 > 0) {
        if (Q < qscore(qual[pos_3], off)) { final_pos_3 = pos_3; break; }
        --pos_3;
    }
    if (!protect5) {
        int pos_5 = final_pos_5;
        while (pos_5 < pos_3) {
            if (Q < qscore(qual[pos_5], off)) { final_pos_5 = pos_5; break; }
            ++pos_5;
        }
    }
    *kept = (uint32_t)(final_pos_3 - final_pos_5 + 1);
    return (uint32_t)final_pos_5;
}

uint32_t faqcs_oracle_quality_trim(int mode, const uint8_t *qual, uint32_t len, int Q, int offset, int protect5,
                                   uint32_t *kept_len)
{
    if (len == 0) { *kept_len = 0; return 0; }
    switch (mode) {
    case FAQCS_MODE_HARD: return hard_trim(qual, (int)len, Q, offset, protect5, kept_len);
    case FAQCS_MODE_BWA:  return bwa_trim(qual, (int)len, Q, offset, kept_len);
    default:              return bwa_plus_trim(qual, (int)len, Q, offset, protect5, kept_len);
    }
}

/* ------------------------------------------------------------------------------------------------
 * adapter aligner: SO::SeqOverlap::align_smith_waterman with ALLOW_GAPS / INCLUDE_TARGET_RANGE /
 * TREAT_N_AS_MASK undefined (Makefile:11), one lane of the 8.
 * ---------------------------------------------------------------------------------------------- */
/* seq_overlap.cpp:372-411 ; 0 == throw "Unknown base" */
static uint8_t na_to_bits(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 1;  case 'C': case 'c': return 2;  case 'G': case 'g': return 4;
    case 'T': case 't': return 8;  case 'M': case 'm': return 3;  case 'R': case 'r': return 5;
    case 'S': case 's': return 6;  case 'V': case 'v': return 7;  case 'W': case 'w': return 9;
    case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12;
    case 'D': case 'd': return 13; case 'B': case 'b': return 14; case 'N': case 'n': return 15;
    case '-': return 16;
    }
    return 0;
}

int faqcs_oracle_align(const uint8_t *query, uint32_t qlen, const uint8_t *target, uint32_t tlen, int *score,
                       int *start_i, int *stop_i)
{
    /* two DP rows of {M, M_start_i}; seq_overlap.cpp:85-101 initial row: M = 0, start = 0 */
    int *lastM = (int *)calloc(tlen + 1, sizeof(int)), *lastS = (int *)calloc(tlen + 1, sizeof(int));
    int *curM = (int *)calloc(tlen + 1, sizeof(int)),  *curS = (int *)calloc(tlen + 1, sizeof(int));
    int best = 0, have = 0;                       /* max_elem.M reset to 0 per call, :104 */
    for (uint32_t i = 0; i < qlen; ++i) {
        curM[0] = 0; curS[0] = (int)i + 1;        /* :109-117 first column */
        const uint8_t q = na_to_bits(query[i]);
        for (uint32_t j = 0; j < tlen; ++j) {
            const int A = lastM[j], As = lastS[j];
            const int s = (q & na_to_bits(target[j])) ? 1 : -1;      /* :157-161 */
            const int M = (A > 0 ? A : 0) + s;                        /* :185-188 */
            const int S = (0 > A) ? (int)i : As;                      /* :255,272-275 */
            curM[j + 1] = M; curS[j + 1] = S;
            if (!(M < best)) { best = M; *start_i = S; *stop_i = (int)i; have = 1; } /* :338-354 */
        }
        int *t = lastM; lastM = curM; curM = t; t = lastS; lastS = curS; curS = t;    /* :368 */
    }
    free(lastM); free(lastS); free(curM); free(curS);
    *score = best;
    return have;
}

/* trim.cpp:1144-1189, literal (including the Q4 quirk: a non-record run is not reset at a masked base) */
void faqcs_oracle_find_mask_range(const uint8_t *mask, uint32_t len, uint32_t *start, uint32_t *length)
{
    uint32_t longest_run_start = 0, longest_run_length = 0, run_start = 0, run_length = 0;
    for (uint32_t i = 0; i < len; ++i) {
        if (!mask[i]) {
            if (run_length > longest_run_length) {
                longest_run_length = run_length; longest_run_start = run_start; run_length = 0;
            }
        } else {
            if (run_length == 0) run_start = i;
            ++run_length;
        }
    }
    if (run_length > longest_run_length) { longest_run_length = run_length; longest_run_start = run_start; }
    if (longest_run_length == 0) { *start = 0; *length = 0; return; }
    *start = longest_run_start; *length = longest_run_length;
}

/* trim.cpp:961-1142 with -t 1 semantics: groups of SO_LEN = 8 consecutive reads; a full
 * group's threshold uses the LAST read of the group (:1007-1008); the tail group drops the min (:1082).
 * Stale aligner state (H2) is modelled only where it is deterministic: within one read the range left
 * by adapter j-1 is what adapter j reports when none of its cells reaches M >= 0 (score 0).  Before the
 * first adapter of a read the state is "unknown" -> no hit. */
static int adapter_prepass(const faqcs_oracle *o, const uint8_t *seq, const uint32_t *offset, uint32_t n,
                           uint32_t *sl_first, uint32_t *sl_second, uint16_t *credited, uint64_t *adapter_stats)
{
    const faqcs_params *p = &o->p;
    const float rate = (float)(1.0 - (double)p->filterAdapterMismatchRate); /* :969 */
    uint8_t *mask = (uint8_t *)malloc(FAQCS_MAX_READ_LENGTH + 1);
    int rc = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += 8) {
        const uint32_t gsz = (n - g0 < 8) ? n - g0 : 8;
        const int tail = gsz < 8;
        const uint32_t last_len = offset[g0 + gsz] - offset[g0 + gsz - 1];
        for (uint32_t s = 0; s < gsz; ++s) {
            const uint32_t i = g0 + s, len = offset[i + 1] - offset[i];
            const uint8_t *r = seq + offset[i];
            sl_first[i] = 0; sl_second[i] = len; credited[i] = 0;           /* :988-989 */
            for (uint32_t k = 0; k < len; ++k) if (!na_to_bits(r[k])) rc = FAQCS_E_BASE; /* :399 -> :409 */
            if (rc) continue;
            memset(mask, 1, len);
            int best_score = 0; uint32_t best_j = 0;
            int have = 0, rs = 0, re = 0;
            for (uint32_t j = 0; j < p->n_adapters; ++j) {
                const uint32_t alen = o->adapter_len[j];
                const uint32_t m = tail ? alen : (last_len < alen ? last_len : alen);
                const int thr = (int)(rate * (float)m);                     /* :1007-1008 / :1082 */
                int score, a = rs, b = re;
                if (faqcs_oracle_align(r, len, (const uint8_t *)o->adapter[j], alen, &score, &a, &b)) {
                    have = 1; rs = a; re = b;
                } else if (!have) {
                    continue;              /* unknown stale state: treated as no hit (H2) */
                }
                const int match_length = re - rs + 1;
                const int num_match = (match_length + score) / 2;           /* :1024-1025 */
                if (num_match >= thr) {
                    for (int k = rs; k <= re; ++k) mask[k] = 0;             /* :1032-1034 */
                    if (score > best_score) { best_score = score; best_j = j; }
                }
            }
            if (best_score > 0) {                                           /* :1048-1065 */
                uint32_t st, ln;
                faqcs_oracle_find_mask_range(mask, len, &st, &ln);
                sl_first[i] = st; sl_second[i] = ln;
                credited[i] = (uint16_t)(best_j + 1);
                adapter_stats[2 * best_j] += 1;
                adapter_stats[2 * best_j + 1] += len - ln;
            }
        }
    }
    free(mask);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * accumulators
 * ---------------------------------------------------------------------------------------------- */
/* trim.cpp:795-808 ; returns -1 on the fastq.h:31 throw */
static int update_quality_matrix(uint64_t *M, uint32_t R, const uint8_t *qual, uint32_t len, uint32_t offset_5, int qoff)
{
    for (uint32_t i = 0; i < len; ++i) {
        const int q = qscore(qual[i], qoff);
        if (q < 0) return -1;
        if (i + offset_5 < R) M[(uint64_t)(i + offset_5) * NQ + q]++;
    }
    return 0;
}

/* trim.cpp:810-875 */
static void update_base_statistics(uint64_t *B, uint64_t *comp, uint32_t R, const uint8_t *seq, uint32_t len, uint32_t offset_5)
{
    unsigned num_A = 0, num_T = 0, num_C = 0, num_G = 0, num_N = 0;
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t row = i + offset_5;
        int col = -1;
        switch (seq[i]) {
        case 'A': case 'a': ++num_A; col = 0; break;
        case 'T': case 't': ++num_T; col = 1; break;
        case 'C': case 'c': ++num_C; col = 2; break;
        case 'G': case 'g': ++num_G; col = 3; break;
        case 'N': case 'n': ++num_N; col = 4; break;
        }
        if (col >= 0 && row < R) B[(uint64_t)row * FAQCS_NBASE + col]++;
    }
    const float norm = (len > 0) ? (float)(FAQCS_NCOMP_BIN - 1) / (float)len : 0.0f;   /* :860 */
    comp[(size_t)(norm * (float)num_A) * FAQCS_NCOMP_KIND + 0]++;
    comp[(size_t)(norm * (float)num_T) * FAQCS_NCOMP_KIND + 1]++;
    const unsigned index_C = (unsigned)(norm * (float)num_C);
    comp[(size_t)index_C * FAQCS_NCOMP_KIND + 2]++;
    const unsigned index_G = (unsigned)(norm * (float)num_G);
    comp[(size_t)index_G * FAQCS_NCOMP_KIND + 3]++;
    comp[(size_t)(norm * (float)num_N) * FAQCS_NCOMP_KIND + 4]++;
    comp[(size_t)(index_G + index_C) * FAQCS_NCOMP_KIND + 5]++;
}

/* trim.cpp:553-576 */
static float average_quality(const uint8_t *qual, uint32_t len, int qoff)
{
    int total = 0;
    for (uint32_t i = 0; i < len; ++i) total += (int)(int8_t)qual[i];
    if (len) {
        const float v = (float)total / (float)len - (float)qoff;
        return v > 0.0f ? v : 0.0f;
    }
    return 0.0f;
}

/* trim.cpp:578-597 */
static unsigned count_poly_n(const uint8_t *seq, uint32_t len)
{
    unsigned mx = 0, cur = 0;
    for (uint32_t i = 0; i < len; ++i) {
        if (seq[i] == 'N') { ++cur; if (cur > mx) mx = cur; } else cur = 0;
    }
    return mx;
}

/* trim.cpp:887-931 */
static void update_kmer(kmer_map *tab, const uint8_t *seq, uint32_t len, unsigned k)
{
    const uint64_t comp_shift = 2 * (k - 1);
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    uint64_t w = 0, comp = 0; unsigned word_len = 0;
    for (uint32_t i = 0; i < len; ++i) {
        ++word_len;
        switch (seq[i]) {
        case 'A': case 'a': w = (w << 2) | 0; comp = (comp >> 2) | (1ULL << comp_shift); break;
        case 'T': case 't': w = (w << 2) | 1; comp = (comp >> 2) | (0ULL << comp_shift); break;
        case 'G': case 'g': w = (w << 2) | 3; comp = (comp >> 2) | (2ULL << comp_shift); break;
        case 'C': case 'c': w = (w << 2) | 2; comp = (comp >> 2) | (3ULL << comp_shift); break;
        default: word_len = 0; break;
        }
        if (word_len >= k) {
            const uint64_t a = w & mask, b = comp & mask;
            kmer_map_add(tab, a < b ? a : b, 1);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * trim_read -- trim.cpp:225-551.  Works on a private copy of the read (seq/qual mutated in place as in
 * the reference) and reports (offset_5, len, flags).
 * ---------------------------------------------------------------------------------------------- */
static int trim_read(faqcs_oracle *o, uint8_t *seq, uint8_t *qual, uint32_t len0, uint32_t sl_first,
                     uint32_t sl_second, uint64_t *C, faqcs_read_result *res)
{
    const faqcs_params *p = &o->p;
    const faqcs_layout *L = &o->lay;
    const uint32_t R = p->max_read_length;
    uint64_t *fs = C + L->filter_stats;
    const int in_off = p->input_quality_offset, out_off = p->output_quality_offset;
    int ret = 1;
    uint32_t len = len0, offset_5 = 0, flags = 0, filt = 0;
    uint8_t *s = seq, *q = qual;   /* current substring */

    ++fs[FAQCS_TOTAL_COUNT]; ++fs[FAQCS_TOTAL_NUMBER]; fs[FAQCS_TOTAL_LENGTH] += len;   /* :238-240 */

    /* mask_quality_terminal_N, trim.cpp:1191-1216 */
    for (uint32_t i = 0; i < len && s[i] == 'N'; ++i) q[i] = (uint8_t)in_off;
    for (uint32_t i = len; i > 0 && s[i - 1] == 'N'; --i) q[i - 1] = (uint8_t)in_off;

    if (update_quality_matrix(C + L->pre_qual, R, q, len, 0, in_off)) return FAQCS_E_QUALITY;   /* :247 */
    update_base_statistics(C + L->pre_base, C + L->pre_comp, R, s, len, 0);                    /* :249 */
    C[L->pre_len_hist + len]++;                                                                /* :251 */
    int quality_bin = (int)average_quality(q, len, in_off);                                    /* :254 */
    C[L->pre_read_qhist + quality_bin]++; C[L->pre_base_qhist + quality_bin] += len;          /* :257-258 */

    if (p->qc_only && o->kmer_active) update_kmer(&o->kmers, s, len, p->kmer);                 /* :260-262 */

    if (p->n_adapters) {                                                                       /* :270-277, :934-954 */
        if (len != sl_second) {
            s += sl_first; q += sl_first;
            offset_5 += (sl_second == 0) ? len : sl_first;
            len = sl_second;
            flags |= FAQCS_F_ADAPTER;
        }
    }
    if (p->trim_5 && !p->qc_only) {                                                            /* :279-297 */
        if (p->trim_5 > len) { len = 0; /* offset_5 += 0: Q7 */ }
        else { s += p->trim_5; q += p->trim_5; len -= p->trim_5; offset_5 += p->trim_5; }
    }
    if (p->trim_3 && !p->qc_only) {                                                            /* :299-314 */
        if (p->trim_3 > len) len = 0; else len -= p->trim_3;
    }
    if (len < p->min_read_length || len == 0) {                                                /* :317-323 */
        fs[FAQCS_BASE_LENGTH] += len; ++fs[FAQCS_READ_LENGTH]; ret = 0; filt = FAQCS_FILT_LENGTH_PRE;
    }
    if (!p->qc_only && ret) {                                                                  /* :325-360 */
        const uint32_t init_len = len;
        uint32_t kept;
        const uint32_t cut5 = faqcs_oracle_quality_trim(p->mode, q, len, p->quality, in_off, (int)p->protect_5, &kept);
        offset_5 += cut5; s += cut5; q += cut5; len = kept;
        if (init_len != len) {
            fs[FAQCS_BASE_QUAL_TRIM] += init_len - len; ++fs[FAQCS_READ_QUAL_TRIM]; flags |= FAQCS_F_QUAL_TRIMMED;
        }
        if (len < p->min_read_length || len == 0) {
            fs[FAQCS_BASE_LENGTH] += len; ++fs[FAQCS_READ_LENGTH]; ret = 0; filt = FAQCS_FILT_LENGTH_POST;
        }
    }
    if (ret && count_poly_n(s, len) >= p->max_num_poly_N) {                                    /* :363-371 */
        fs[FAQCS_BASE_NN] += len; ++fs[FAQCS_READ_NN]; flags |= FAQCS_F_POLY_N_SEEN;
        if (!p->qc_only) { ret = 0; filt = FAQCS_FILT_POLY_N; }
    }
    const float ave_Q = average_quality(q, len, in_off);                                       /* :374 */
    if (ret && ave_Q < p->average_quality) {                                                   /* :376-382 */
        fs[FAQCS_BASE_AVG_Q] += len; ++fs[FAQCS_READ_AVG_Q]; ret = 0; filt = FAQCS_FILT_AVG_Q;
    }
    if (ret && len != 0) {                                                                     /* :388-513 */
        if (p->replace_to_N_q > 0)
            for (uint32_t i = 0; i < len; ++i)
                if (s[i] == 'G' && qscore(q[i], in_off) < (int)p->replace_to_N_q) s[i] = 'N';
        unsigned num_A = 0, num_T = 0, num_G = 0, num_C = 0, dc[16];
        memset(dc, 0, sizeof(dc));
        unsigned last = 4; /* INVALID_BASE */
        for (uint32_t i = 0; i < len; ++i) {
            unsigned cur;
            switch (s[i]) {
            case 'A': case 'a': ++num_A; cur = 0; break;
            case 'T': case 't': ++num_T; cur = 1; break;
            case 'G': case 'g': ++num_G; cur = 3; break;
            case 'C': case 'c': ++num_C; cur = 2; break;
            default: cur = 4; break;
            }
            if (cur != 4 && cur != last && last != 4) ++dc[(last << 2) | cur];
            last = cur;
        }
        float norm = (float)(1.0 / (double)len);                                               /* :483 */
        const float lc = p->low_complexity_cutoff_ratio;
        if ((float)num_A * norm > lc || (float)num_T * norm > lc || (float)num_G * norm > lc || (float)num_C * norm > lc) {
            fs[FAQCS_BASE_LOW_COMPLEXITY] += len; fs[FAQCS_READ_LOW_COMPLEXITY]++; ret = 0; filt = FAQCS_FILT_LOW_COMPLEXITY;
        } else {
            norm = (float)((double)norm * 2.0);                                                /* :499 */
            for (int k = 0; k < 16; ++k)
                if ((float)dc[k] * norm > lc) {
                    fs[FAQCS_BASE_LOW_COMPLEXITY] += len; fs[FAQCS_READ_LOW_COMPLEXITY]++; ret = 0;
                    filt = FAQCS_FILT_LOW_COMPLEXITY; break;
                }
        }
    }
    if (ret && in_off != out_off)                                                              /* :516-525 */
        for (uint32_t i = 0; i < len; ++i) q[i] = (uint8_t)(qscore(q[i], in_off) + out_off);
    if (ret) {                                                                                 /* :527-548 */
        fs[FAQCS_TOTAL_TRIMMED_LENGTH] += len; ++fs[FAQCS_TOTAL_TRIMMED_NUMBER];
        if (update_quality_matrix(C + L->post_qual, R, q, len, offset_5, out_off)) return FAQCS_E_QUALITY;
        update_base_statistics(C + L->post_base, C + L->post_comp, R, s, len, offset_5);
        C[L->post_len_hist + len]++;
        quality_bin = (int)ave_Q;
        C[L->post_read_qhist + quality_bin]++; C[L->post_base_qhist + quality_bin] += len;
        if (!p->qc_only && o->kmer_active) update_kmer(&o->kmers, s, len, p->kmer);
    }
    res->start = (uint16_t)(ret ? offset_5 : 0);
    res->len = (uint16_t)(ret ? len : 0);
    res->flags = (uint16_t)(flags | (ret ? FAQCS_F_VALID : 0) | (filt << FAQCS_F_FILTER_SHIFT));
    return 0;
}

static void push_point(faqcs_oracle *o, faqcs_rarefaction r);

/* trim.cpp:67-186 */
int faqcs_oracle_trim(faqcs_oracle *o, const uint8_t *seq, const uint8_t *qual, const uint32_t *offset,
                      uint32_t n, faqcs_read_result *results, uint64_t *counters)
{
    const faqcs_params *p = &o->p;
    uint32_t *slf = (uint32_t *)calloc(n + 1, 4), *sls = (uint32_t *)calloc(n + 1, 4);
    uint16_t *cred = (uint16_t *)calloc(n + 1, 2);
    uint8_t *s = (uint8_t *)malloc(FAQCS_MAX_READ_LENGTH + 1), *q = (uint8_t *)malloc(FAQCS_MAX_READ_LENGTH + 1);
    int rc = 0;
    for (uint32_t i = 0; i < n && !rc; ++i)
        if (offset[i + 1] - offset[i] > p->max_read_length) rc = FAQCS_E_INVAL;
    if (!rc && p->n_adapters)                                                                  /* :86-88 */
        rc = adapter_prepass(o, seq, offset, n, slf, sls, cred, counters + o->lay.adapter_stats);
    for (uint32_t i = 0; i < n && !rc; ++i) {                                                  /* :90-117 */
        const uint32_t len = offset[i + 1] - offset[i];
        memcpy(s, seq + offset[i], len); memcpy(q, qual + offset[i], len);
        rc = trim_read(o, s, q, len, slf[i], sls[i], counters, &results[i]);
        results[i].adapter = cred[i];
    }
    free(slf); free(sls); free(cred); free(s); free(q);
    if (rc) return rc;

    if (o->kmer_active) {                                                                      /* :157-185 */
        const uint64_t total_number = counters[o->lay.filter_stats + FAQCS_TOTAL_NUMBER];
        const uint64_t index = total_number / p->split_size;
        const uint64_t num_rarefaction = o->n_points;
        if (index > num_rarefaction && num_rarefaction < p->num_subsample) {
            faqcs_rarefaction r; r.num_seq = total_number;
            faqcs_oracle_kmer_totals(o, &r.distinct_kmer, &r.total_kmer);
            push_point(o, r);
        }
        if (num_rarefaction >= p->num_subsample) o->kmer_active = 0;
    }
    return 0;
}

int faqcs_oracle_kmer_active(const faqcs_oracle *o) { return o->kmer_active; }

uint32_t faqcs_oracle_kmer_points(const faqcs_oracle *o, faqcs_rarefaction *out, uint32_t cap)
{
    for (uint32_t i = 0; i < o->n_points && i < cap; ++i) out[i] = o->points[i];
    return o->n_points;
}

void faqcs_oracle_kmer_totals(const faqcs_oracle *o, uint64_t *distinct, uint64_t *total)
{
    uint64_t t = 0;
    for (uint64_t i = 0; i < o->kmers.cap; ++i) if (o->kmers.key[i]) t += o->kmers.cnt[i];
    *distinct = o->kmers.used; *total = t;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static void push_point(faqcs_oracle *o, faqcs_rarefaction r)
{
    if (o->n_points == o->cap_points) {
        o->cap_points = o->cap_points ? 2 * o->cap_points : 64;
        o->points = (faqcs_rarefaction *)realloc(o->points, o->cap_points * sizeof(faqcs_rarefaction));
    }
    o->points[o->n_points++] = r;
}

/* End of process_paired()/process_unpaired(): FaQCs.cpp:518-537 (and :737-756).  total_number is
 * filter_stats[TOTAL_NUMBER] at that moment. */
void faqcs_oracle_kmer_end_table(faqcs_oracle *o, uint64_t total_number)
{
    /* ++kmer_frequency_histogram[count] for every key (FaQCs.cpp:519-521) */
    uint64_t n = o->kmers.used, k = 0;
    uint64_t *c = (uint64_t *)malloc((n + 1) * 8);
    for (uint64_t i = 0; i < o->kmers.cap; ++i) if (o->kmers.key[i]) c[k++] = o->kmers.cnt[i];
    qsort(c, n, 8, cmp_u64);
    uint64_t *mc = (uint64_t *)malloc((n + o->n_hist + 1) * 8), *mk = (uint64_t *)malloc((n + o->n_hist + 1) * 8);
    uint64_t a = 0, i = 0, m = 0;
    while (a < o->n_hist || i < n) {
        uint64_t j = i;
        while (j < n && c[j] == c[i]) ++j;
        if (i >= n || (a < o->n_hist && o->hist_count[a] < c[i])) { mc[m] = o->hist_count[a]; mk[m++] = o->hist_nkeys[a++]; }
        else if (a < o->n_hist && o->hist_count[a] == c[i]) { mc[m] = c[i]; mk[m++] = o->hist_nkeys[a++] + (j - i); i = j; }
        else { mc[m] = c[i]; mk[m++] = j - i; i = j; }
    }
    free(o->hist_count); free(o->hist_nkeys); free(c);
    o->hist_count = mc; o->hist_nkeys = mk; o->n_hist = m;
    if (o->kmer_active && o->n_points == 0) {               /* FaQCs.cpp:523-537 */
        faqcs_rarefaction r; r.num_seq = total_number;
        faqcs_oracle_kmer_totals(o, &r.distinct_kmer, &r.total_kmer);
        push_point(o, r);
    }
    memset(o->kmers.key, 0, o->kmers.cap * 8); memset(o->kmers.cnt, 0, o->kmers.cap * 8); o->kmers.used = 0;
}

/* accumulated (count, nkeys) pairs, ascending count: what plot.cpp:683-714 prints */
uint64_t faqcs_oracle_kmer_histogram(const faqcs_oracle *o, uint64_t *count, uint64_t *nkeys, uint64_t cap)
{
    for (uint64_t i = 0; i < o->n_hist && i < cap; ++i) { count[i] = o->hist_count[i]; nkeys[i] = o->hist_nkeys[i]; }
    return o->n_hist;
}
