/*
 * faqcs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's per-read hot path (trim.cpp,
 * seq_overlap.cpp, fastq.h:17-36).  It is the *checker*: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (libfaqcs_mi.so) never links or calls it.
 * Parity status: PINNED -- tests/test_oracle_golden.py checks it against outputs of the real reference
 * (oracle/_ref/FaQCs_ref, built from /root/reference by oracle/Makefile) committed under tests/golden/.
 * Known unpinned corners: SURVEY.md H1 (defined as the reference's -t 1 behaviour) and H2.
 *
 * It shares the POD data model (params / batch / result / counter layout) with include/faqcs_mi.h so
 * the same harness can drive either side.
 */
#ifndef FAQCS_ORACLE_H
#define FAQCS_ORACLE_H

#include "../include/faqcs_mi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct faqcs_oracle faqcs_oracle;

faqcs_oracle *faqcs_oracle_create(const faqcs_params *p);
void faqcs_oracle_destroy(faqcs_oracle *o);

/* One reference trim() call (trim.cpp:67-186) over n reads of one segment.
 * counters (layout faqcs_counters_layout(p->max_read_length, p->n_adapters)) are += updated.
 * Returns 0, FAQCS_E_QUALITY or FAQCS_E_BASE. */
int faqcs_oracle_trim(faqcs_oracle *o, const uint8_t *seq, const uint8_t *qual, const uint32_t *offset,
                      uint32_t n, faqcs_read_result *results, uint64_t *counters);

int      faqcs_oracle_kmer_active(const faqcs_oracle *o);
uint32_t faqcs_oracle_kmer_points(const faqcs_oracle *o, faqcs_rarefaction *out, uint32_t cap);
void     faqcs_oracle_kmer_totals(const faqcs_oracle *o, uint64_t *distinct, uint64_t *total);
/* end of a process_paired/process_unpaired pass (FaQCs.cpp:518-537): fold table into the count histogram,
 * add the fallback rarefaction point, clear the table */
void     faqcs_oracle_kmer_end_table(faqcs_oracle *o, uint64_t total_number);
/* accumulated (count, nkeys) pairs ascending; returns number of pairs (may exceed cap: call again) */
uint64_t faqcs_oracle_kmer_histogram(const faqcs_oracle *o, uint64_t *count, uint64_t *nkeys, uint64_t cap);

/* stage-level entry points for unit goldens */
/* BWA_plus / BWA / HARD on a quality string; returns final_pos_5, writes kept length (0 = emptied) */
uint32_t faqcs_oracle_quality_trim(int mode, const uint8_t *qual, uint32_t len, int Q, int offset, int protect5,
                                   uint32_t *kept_len);
/* ungapped local alignment, seq_overlap.cpp:46-370 semantics for one (query, target) pair.
 * returns 1 if any cell reached M >= 0 (range valid) else 0 (reference leaves stale state, H2) */
int faqcs_oracle_align(const uint8_t *query, uint32_t qlen, const uint8_t *target, uint32_t tlen, int *score,
                       int *start_i, int *stop_i);
void faqcs_oracle_find_mask_range(const uint8_t *mask, uint32_t len, uint32_t *start, uint32_t *length);

#ifdef __cplusplus
}
#endif
#endif
