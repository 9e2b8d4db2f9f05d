/*
 * faqcs_mi.h -- C ABI of libfaqcs_mi.so: the MI355X (gfx950) implementation of the FaQCs per-read
 * trim / filter / accumulate hot path.
 *
 * Drop-in boundary.  The reference (LANL-Bioinformatics/FaQCs v2.10) has no FFI; the seam this
 * library replaces is the C++ function
 *
 *     void trim(std::vector<Read>&, std::vector<size_t>& filter_stats,
 *               MAP<std::string, std::pair<size_t,size_t>>& adapter_stats,
 *               MAP<Word,size_t>& kmer_table, PlotInfo&, Options&);        // FaQCs.h:245-248
 *
 * whose six call sites are FaQCs.cpp:287,290,424,427 (paired) and :628,:692 (unpaired).  One
 * reference trim() call == one *segment* of a faqcs_batch here.  integration/trim_shim.cpp is the
 * ~150-line C++ adapter a maintainer would compile in place of trim.o.
 *
 * Conventions: plain pointers and sizes only, no C++ / torch types; the caller owns every host
 * buffer; the library owns device memory and HIP streams; every entry point returns 0 or a negative
 * FAQCS_E_* code and faqcs_last_error() gives the text; no exceptions cross the boundary.
 */
#ifndef FAQCS_MI_H
#define FAQCS_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: faqcs_batch.terminal_n, faqcs_terminal_n_flags, the k-mer timers of faqcs_kernel_times.  Entry points added since then leave
 * every existing structure and call as it was, so the number stands: faqcs_kmer_forward, faqcs_comm_* (round 4).  Round 5 changes what the
 * 16-byte items of the k-mer exchange MEAN (faqcs_kmer_outbox / _insert_device / _forward: opaque to every caller in this repository, which
 * only moves them) -- a run of up to 17 consecutive k-mers each instead of one (key, epoch) pair -- and nothing about their size or the calls.
 * Round 6 adds faqcs_kmer_finish_pass, and faqcs_sync() no longer counts the k-mers that wait in the open group (see below). */
#define FAQCS_ABI_VERSION 2

/* FilterStat enum order, FaQCs.h:46-75 */
enum {
    FAQCS_TOTAL_COUNT = 0, FAQCS_TOTAL_NUMBER, FAQCS_TOTAL_LENGTH, FAQCS_TOTAL_TRIMMED_NUMBER,
    FAQCS_TOTAL_TRIMMED_LENGTH, FAQCS_PAIRED_READ_NUMBER, FAQCS_PAIRED_BASE_LENGTH, FAQCS_READ_LENGTH,
    FAQCS_BASE_LENGTH, FAQCS_READ_NN, FAQCS_BASE_NN, FAQCS_READ_PHIX, FAQCS_BASE_PHIX,
    FAQCS_READ_ADAPTER, FAQCS_BASE_ADAPTER, FAQCS_READ_AVG_Q, FAQCS_BASE_AVG_Q, FAQCS_READ_QUAL_TRIM,
    FAQCS_BASE_QUAL_TRIM, FAQCS_READ_LOW_COMPLEXITY, FAQCS_BASE_LOW_COMPLEXITY, FAQCS_N_TO_A,
    FAQCS_N_TO_T, FAQCS_N_TO_G, FAQCS_N_TO_C, FAQCS_NUM_STAT
};

#define FAQCS_NQ 42            /* MAX_QUALITY_SCORE + 1, fastq.h:15 */
#define FAQCS_NBASE 5          /* A,T,C,G,N -- enum order FaQCs.h:35-42 */
#define FAQCS_NCOMP_BIN 10001  /* NUM_COMPOSITION_BIN, FaQCs.h:20 */
#define FAQCS_NCOMP_KIND 6     /* NucleotideCount fields A,T,C,G,N,GC, FaQCs.h:167-174 */
#define FAQCS_SEGMENT_READS 32768 /* buffer_size, FaQCs.cpp:232,585 */
#define FAQCS_MAX_READ_LENGTH 32767 /* longest read the HIP kernels take: the reference's aligner works in int16 (seq_overlap.h:80), and
                                       faqcs_read_result holds window coordinates in 16 bits.  Reads of up to 1 024 bases run on the
                                       chunked kernels (64 reads per wave pass); a batch with a longer read runs on trim_long /
                                       adapter_overlap<1, 32768> (one wave per read) */
#define FAQCS_ARENA_PAD_BEFORE 16 /* readable bytes required in front of / behind a batch's arenas (faqcs_batch) */
#define FAQCS_ARENA_PAD_AFTER 64
#define FAQCS_MAX_ADAPTERS 64
#define FAQCS_MAX_ADAPTER_LENGTH 8192

enum { FAQCS_MODE_HARD = 0, FAQCS_MODE_BWA = 1, FAQCS_MODE_BWA_PLUS = 2 }; /* Options::Mode, FaQCs.h:90-95 */

/* error codes */
enum {
    FAQCS_OK = 0,
    FAQCS_E_INVAL = -1,      /* bad argument / unsupported size */
    FAQCS_E_NODEVICE = -2,   /* no HIP device or HIP runtime failure */
    FAQCS_E_QUALITY = -3,    /* a quality score > 41 after the offset (fastq.h:31-33 throws) */
    FAQCS_E_BASE = -4,       /* non-IUPAC base reached the aligner (seq_overlap.cpp:409 throws) */
    FAQCS_E_NOMEM = -5,
    FAQCS_E_KMER_FULL = -6   /* device k-mer table exhausted */
};

/* The subset of the reference's Options (FaQCs.h:77-144) the hot path reads, flattened to a POD.
 * The host resolves everything the reference resolves before/around trim(): quality-offset
 * auto-detect (trim.cpp:599-617), the NextSeq -q bump (trim.cpp:619-626, FaQCs.cpp:404-414) and the
 * adapter list (options.cpp:576-694). */
typedef struct faqcs_params {
    uint32_t abi_version;                 /* FAQCS_ABI_VERSION */
    int32_t  mode;                        /* FAQCS_MODE_* */
    int32_t  quality;                     /* -q ; Options::quality is a (signed) char */
    int32_t  input_quality_offset;        /* 33 / 64 (already auto-detected) */
    int32_t  output_quality_offset;
    uint32_t min_read_length;             /* --min_L */
    uint32_t max_num_poly_N;              /* -n */
    uint32_t trim_5;                      /* --5end */
    uint32_t trim_3;                      /* --3end */
    uint32_t replace_to_N_q;              /* --replace_to_N_q */
    float    average_quality;             /* --avg_q */
    float    low_complexity_cutoff_ratio; /* --lc  (float in the reference: FaQCs.h:115) */
    float    filterAdapterMismatchRate;   /* --rate */
    uint32_t protect_5;                   /* --5trim_off */
    uint32_t qc_only;                     /* --qc_only */
    uint32_t kmer_rarefaction;            /* --kmer_rarefaction */
    uint32_t kmer;                        /* -m, 2..31 */
    uint32_t split_size;                  /* --split_size */
    uint32_t num_subsample;               /* --subset (already doubled per options.cpp:506-523) */
    uint32_t max_read_length;             /* capacity R of the per-position matrices (<= FAQCS_MAX_READ_LENGTH); no read of a batch may be longer */
    uint32_t n_adapters;                  /* 0 == !(filter_adapter || filter_phiX) */
    const char *const *adapter_seq;       /* n_adapters NUL-terminated IUPAC strings (Options::adapter[j].second) */
    uint64_t kmer_table_slots;            /* device hash-table capacity (0 = library default 2^28; rounded up to a power of two in [2^22, 2^32]:
                                             the table is cut into 65 536 slices, one per key partition (a partition = the k-mers whose minimizer
                                             hashes to it); a key that finds the 128 slots behind its home slot taken lives in an overflow area of
                                             slots / 16 behind the table (round 5: a full slice is no longer an error), and FAQCS_E_KMER_FULL is
                                             raised when that area cannot take a key either -- size the table for <= 0.6 x slots distinct k-mers;
                                             16 bytes x 1.0625 x slots of device memory) */
} faqcs_params;

/* One submission: reads packed back to back in two byte arenas (structure of arrays).
 * Read i occupies seq[offset[i] .. offset[i+1]) and qual[offset[i] .. offset[i+1]) -- the reference
 * rejects |seq| != |qual| at parse time (fastq.cpp:117-121) so one offset array serves both.
 * PADDING CONTRACT: both arenas must be readable from 16 bytes before seq/qual + offset[0] up to 64 bytes past
 * seq/qual + offset[n] (FAQCS_ARENA_PAD_BEFORE / FAQCS_ARENA_PAD_AFTER): the kernels fetch a read with unaligned
 * multi-dword loads predicated on the read's length only (up to 19 bytes past its end) and whole 16-byte aligned
 * pieces of a 64-read span (up to 15 bytes either side); the padding bytes are never interpreted.  faqcs_submit()
 * and faqcs_submit_async() copy into padded device buffers themselves, so the contract binds faqcs_submit_device()
 * callers.  segment_start[] partitions the reads into reference
 * trim() calls (adapter groups of 8 restart at a segment start, trim.cpp:977-1071; k-mer rarefaction
 * points are taken at segment ends, trim.cpp:157-185). */
typedef struct faqcs_batch {
    const uint8_t  *seq;
    const uint8_t  *qual;
    const uint32_t *offset;         /* n_reads + 1 entries, non-decreasing */
    uint32_t        n_reads;
    uint32_t        n_segments;     /* >= 1 when n_reads > 0 */
    const uint32_t *segment_start;  /* n_segments + 1 entries; [0] = 0, [n_segments] = n_reads */
    uint32_t        max_read_len;   /* upper bound on the read lengths of this batch (selects the kernel
                                       variant); 0 = unknown: faqcs_submit() scans the offsets,
                                       faqcs_submit_device() falls back to the context capacity */
    const uint8_t  *terminal_n;     /* OPTIONAL (ABI 2): one byte per read, bit 0 = the read's first base is an upper-case 'N', bit 1 = its
                                       last base is (what mask_quality_terminal_N, trim.cpp:1191-1216, looks at first).  A host pointer for
                                       faqcs_submit() / faqcs_submit_async() (uploaded with the offsets), a device pointer for
                                       faqcs_submit_device() (e.g. from faqcs_terminal_n_flags()).  NULL = the kernels look themselves: two
                                       scattered byte loads per read into the base arena ahead of its streaming copy, +100 B/read of fabric
                                       requests.  A parser has both bytes in hand when it lays a read down. */
} faqcs_batch;

/* Per-read outcome (8 bytes).  For a valid read the reference's output record is
 *   seq  = in.seq [start, start+len)  with 'G' -> 'N' where Q < replace_to_N_q   (trim.cpp:390-403)
 *   qual = in.qual[start, start+len)  with the read's leading / trailing upper-case-'N' runs set to
 *          the input offset (trim.cpp:1191-1216) and then re-based input->output offset (trim.cpp:516-525)
 * faqcs_apply_edits() performs exactly these byte edits on the host. */
typedef struct faqcs_read_result {
    uint16_t start;   /* == offset_5 of trim_read() for a valid read */
    uint16_t len;
    uint16_t flags;   /* FAQCS_F_* */
    uint16_t adapter; /* 1 + index of the adapter credited for this read (trim.cpp:1036-1064), 0 = none */
} faqcs_read_result;

#define FAQCS_F_VALID        0x0001u
#define FAQCS_F_FILTER_MASK  0x000eu /* which filter fired first (trim_read short-circuit order) */
#define FAQCS_F_FILTER_SHIFT 1
enum { FAQCS_FILT_NONE = 0, FAQCS_FILT_LENGTH_PRE = 1, FAQCS_FILT_LENGTH_POST = 2, FAQCS_FILT_POLY_N = 3,
       FAQCS_FILT_AVG_Q = 4, FAQCS_FILT_LOW_COMPLEXITY = 5 };
#define FAQCS_F_QUAL_TRIMMED 0x0010u /* READ_QUAL_TRIM counted for this read */
#define FAQCS_F_ADAPTER      0x0020u /* start_length changed by the adapter pre-pass */
#define FAQCS_F_POLY_N_SEEN  0x0040u /* READ_NN counted (qc_only keeps the read valid, trim.cpp:368-370) */
/* The read is where the reference's trim() call throws; the submission as a whole also fails at faqcs_sync()/
 * faqcs_finish().  A driver that writes output per trim() call (the reference: per 32 768-read buffer) checks these to
 * stop before the buffer that holds such a read, as the reference does (FaQCs.cpp:287-361). */
#define FAQCS_F_ERR_QUALITY  0x0100u /* a quality above MAX_QUALITY_SCORE (fastq.h:31-33) */
#define FAQCS_F_ERR_BASE     0x0200u /* adapters active and a base na_to_bits() rejects (seq_overlap.cpp:409) */

/* Layout (in uint64 units) of the additive counter block.  Everything the reference accumulates in
 * filter_stats / PlotInfo / adapter_stats is a sum of per-read integers, so one block == one
 * all-reduce(sum).  Matrices are row-major [position][column] exactly like matrix<size_t>
 * (matrix.h:56-64); the reference's "rows grow on demand" is recovered on the host as
 * 1 + (last non-zero row) -- see faqcs_counter_rows(). */
typedef struct faqcs_layout {
    uint32_t max_read_length; /* R */
    uint32_t n_adapters;
    uint64_t filter_stats;    /* [FAQCS_NUM_STAT] */
    uint64_t pre_read_qhist, pre_base_qhist, post_read_qhist, post_base_qhist; /* [42] each */
    uint64_t pre_len_hist, post_len_hist;   /* [R+1] */
    uint64_t pre_qual, post_qual;           /* [R][42] */
    uint64_t pre_base, post_base;           /* [R][5]  */
    uint64_t pre_comp, post_comp;           /* [10001][6]  (A,T,C,G,N,GC) */
    uint64_t adapter_stats;                 /* [n_adapters][2] = (reads, bases) */
    uint64_t total;                         /* number of uint64 in the block */
} faqcs_layout;

typedef struct faqcs_rarefaction { uint64_t num_seq, distinct_kmer, total_kmer; } faqcs_rarefaction; /* FaQCs.h:194-199 */

typedef struct faqcs_ctx faqcs_ctx;

/* ---- layout / host helpers (no GPU needed) ---- */
int  faqcs_abi_version(void);
int  faqcs_counters_layout(uint32_t max_read_length, uint32_t n_adapters, faqcs_layout *out);
/* rows the reference's growing matrix<size_t> would have: 1 + last non-zero row (0 if all zero) */
uint32_t faqcs_counter_rows(const uint64_t *matrix, uint32_t max_rows, uint32_t n_cols);
/* applies the rule-based byte edits documented at faqcs_read_result; out_* need res->len bytes */
int  faqcs_apply_edits(const faqcs_params *p, const uint8_t *seq, const uint8_t *qual, uint32_t read_len,
                       const faqcs_read_result *res, uint8_t *out_seq, uint8_t *out_qual);
/* trim.cpp:599-617 -- returns 33, 64 or 0 (undecided: the reference throws) */
int  faqcs_auto_detect_quality_offset(const uint8_t *qual, const uint32_t *offset, uint32_t n_reads);
const char *faqcs_last_error(void);

/* ---- device path ---- */
/* device_id < 0 selects the current HIP device.  Copies the adapter strings. */
int  faqcs_create(const faqcs_params *params, int device_id, faqcs_ctx **out);
void faqcs_destroy(faqcs_ctx *ctx);

/* Process one batch whose arrays live in HOST memory: async H2D on the context's copy stream, kernels
 * on its compute stream, async D2H of the per-read results into `results` (n_reads entries).  Returns
 * when the work is enqueued; faqcs_sync() waits.  Counters accumulate on the device. */
int  faqcs_submit(faqcs_ctx *ctx, const faqcs_batch *batch, faqcs_read_result *results);

/* Same, but batch->seq/qual/offset and d_results are DEVICE pointers (inputs already resident in HBM:
 * the configuration bench.py times).  segment_start stays a host pointer. d_results may be NULL. */
int  faqcs_submit_device(faqcs_ctx *ctx, const faqcs_batch *batch, faqcs_read_result *d_results);

int  faqcs_sync(faqcs_ctx *ctx);

/* Pipelined form of faqcs_submit(): returns a ticket; faqcs_wait(ticket) blocks until THAT batch's results have
 * landed in `results` (later batches may still be in flight: two input staging slots let the H2D copy of batch
 * k+1 overlap the kernels of batch k).  Host arenas / result arrays obtained from faqcs_host_alloc() are pinned,
 * which makes both copies true asynchronous DMA. */
int  faqcs_submit_async(faqcs_ctx *ctx, const faqcs_batch *batch, faqcs_read_result *results, uint64_t *ticket);
int  faqcs_wait(faqcs_ctx *ctx, uint64_t ticket);
void *faqcs_host_alloc(size_t bytes);
void faqcs_host_free(void *p);

/* Options::quality is mutable during a run: the NextSeq check bumps -q to 20 (FaQCs.cpp:272-277,404-414).
 * Takes effect for batches submitted afterwards. */
int  faqcs_set_quality(faqcs_ctx *ctx, int quality);

/* Device address + length (uint64 units) of the additive counter block, so the host can run the one
 * collective this path needs -- all-reduce(sum, uint64) over RCCL -- in place before faqcs_finish(). */
int  faqcs_counters_device(faqcs_ctx *ctx, void **d_ptr, uint64_t *n_u64);

/* The same collective on a buffer the CALLER owns (what bench.py / faqcs_amd/parallel.py do: RCCL registers its own
 * allocations for peer access, so the all-reduce runs on a torch tensor): export copies the block (device to device,
 * d_dst/d_src are device pointers of >= n_u64 words) after the work submitted so far, import stores the reduced block. */
int  faqcs_counters_export(faqcs_ctx *ctx, void *d_dst, uint64_t n_u64);
int  faqcs_counters_import(faqcs_ctx *ctx, const void *d_src, uint64_t n_u64);

/* Copies the counter block to the host (layout: faqcs_counters_layout); syncs first.
 * Raises FAQCS_E_QUALITY / FAQCS_E_BASE if any read tripped the reference's throw sites. */
int  faqcs_finish(faqcs_ctx *ctx, uint64_t *counters, uint64_t n_u64);
int  faqcs_reset_counters(faqcs_ctx *ctx);

/* The merge of the reference (trim.cpp:120-154: every OpenMP thread adds its private counters to the caller's under `omp critical`) across
 * GPUs: all-reduce(sum) of the additive counter block IN PLACE, by RCCL over xGMI, enqueued on the context's compute stream behind its
 * kernels -- no staging copy, no host round trip.  librccl.so is loaded at the first call (the library does not
 * link it); every call fails with FAQCS_E_NODEVICE and a message when it is missing or reports an error, and the caller can fall back on
 * faqcs_counters_export / _import + its own collective.
 *   one process per GPU:  rank 0 calls faqcs_comm_id() and hands the FAQCS_COMM_ID_BYTES bytes to the other ranks (any channel); every rank
 *                         calls faqcs_comm_init(ctx, id, rank, world) (collective), then faqcs_comm_allreduce_counters(ctx) per pass.
 *   one process, n GPUs:  faqcs_comm_init_all(ctxs, n) once (the contexts must sit on n different devices), then
 *                         faqcs_comm_allreduce_counters_all(ctxs, n) -- one grouped call for all of them.
 * The communicator is released by faqcs_destroy(). */
#define FAQCS_COMM_ID_BYTES 128
int  faqcs_comm_id(void *id);
int  faqcs_comm_init(faqcs_ctx *ctx, const void *id, uint32_t rank, uint32_t world);
int  faqcs_comm_allreduce_counters(faqcs_ctx *ctx);
int  faqcs_comm_init_all(faqcs_ctx *const *ctxs, uint32_t n);
int  faqcs_comm_allreduce_counters_all(faqcs_ctx *const *ctxs, uint32_t n);

/* k-mer rarefaction (trim.cpp:157-185, FaQCs.cpp:518-537).  The k-mers of a submission are extracted when it is submitted, as 16-byte runs
 * into group buffers sized from the free HBM, and are COUNTED LATER (combine-before-insert): when the pass ends
 * (faqcs_kmer_end_table / faqcs_kmer_finish_pass) -- if the whole pass fits the buffers it is then counted in one piece and its keys never
 * reach the device table --, when the buffers are full, or when one of faqcs_kmer_points / _totals / _epoch_counts asks for the curve so far
 * (the open group then goes into the table, the slower path: a caller that only wants the finished curve calls faqcs_kmer_end_table FIRST and
 * reads the points afterwards -- they keep their values).  Points are appended at segment ends during submit; their (distinct, total)
 * are filled in by these calls:  */
int  faqcs_kmer_points(faqcs_ctx *ctx, faqcs_rarefaction *out, uint32_t cap, uint32_t *n_points);
/* (count, number of keys with that count) pairs, ascending count -- PlotInfo::kmer_frequency_histogram */
int  faqcs_kmer_histogram(faqcs_ctx *ctx, uint64_t *count, uint64_t *nkeys, uint64_t cap, uint64_t *n_pairs);
/* distinct keys / sum of counts of the pass in progress (the FaQCs.cpp:523-537 fallback point); when nothing has been counted since
 * faqcs_kmer_end_table(): of the pass that call finished */
int  faqcs_kmer_totals(faqcs_ctx *ctx, uint64_t *distinct, uint64_t *total);
/* Options::kmer_rarefaction is switched off by trim() once the curve is complete (trim.cpp:180-184) */
int  faqcs_kmer_active(faqcs_ctx *ctx);
/* End of a process_paired()/process_unpaired() pass (FaQCs.cpp:518-537, :737-756): folds the table into the
 * count histogram, appends the guaranteed single rarefaction point if none was taken, and starts a fresh
 * table (each process_* owns its own MAP<Word,size_t>, FaQCs.cpp:235,588). */
int  faqcs_kmer_end_table(faqcs_ctx *ctx);
/* The counting half of faqcs_kmer_end_table() (which calls it): the pass is complete, its open group is counted and every point and epoch
 * histogram is final; submissions with k-mers are refused until faqcs_kmer_end_table() has started the next pass.  For callers that read
 * faqcs_kmer_epoch_counts() (owner ranks) or the points before they end the table.  Replaces nothing of its own in the reference: the
 * end of process_paired() / process_unpaired(), FaQCs.cpp:518-537. */
int  faqcs_kmer_finish_pass(faqcs_ctx *ctx);

/* What a kmer_rarefaction context made from `params` will allocate on a device with free_bytes of free memory, before anything is
 * allocated (host only; bench.py --config kmer prints it per rank and refuses a run that cannot fit): out[0] the table with its overflow
 * area, out[1] / out[2] the level-1 / level-2 group buffers, out[3] the k-mer occurrences one group takes -- a pass below it is counted in
 * one piece, without the table --, out[4] the small arrays.  n_out >= 5. */
int  faqcs_kmer_memory_plan(const faqcs_params *params, uint64_t free_bytes, uint64_t *out, uint32_t n_out);

/* ---- k-mers across GPUs --------------------------------------------------------------
 * The reference keeps ONE MAP<Word,size_t> per process (trim.cpp:82,133-135) and samples (distinct, total) after
 * trim() calls (trim.cpp:157-185); distinct counts are not additive over shards.  In this mode every canonical
 * k-mer has one owner rank -- the owner of the partition its MINIMIZER hashes to (csrc/faqcs_skm.h), so that consecutive k-mers of a
 * read that share their minimizer travel together: a rank turns its shard into 16-byte ITEMS (a run of up to 17 consecutive 31-mers as
 * 2-bit bases + the epoch, about 8 occurrences per item: under 2 bytes per occurrence on the wire where a (key, epoch) pair per
 * occurrence was 16), grouped by owner; the caller moves them with an all-to-all (RCCL: faqcs_amd/parallel.py), and the owner expands,
 * combines and inserts them keeping the smallest epoch per key.  The items are opaque to the caller.
 * epoch = index of the first rarefaction point that includes the segment (a host function of the GLOBAL
 * read counts only, trim.cpp:157-185); FAQCS_EPOCH_NONE = the curve was already complete.  Then
 *   distinct(point i) = sum over ranks of #{keys with first epoch <= i},  total(point i) = sum of occurrences with
 *   epoch <= i -- both additive, i.e. one all-reduce of 2 x n_epochs integers. */
#define FAQCS_EPOCH_NONE 0xffffffffu
/* First call on a fresh kmer_rarefaction context.  n_epochs = number of epoch slots (num_subsample + 1).  Up to 1 000 (an item has 10 bits
 * for its epoch) the exchange moves runs of k-mers and the owner combines them; a job with more sampling points goes through (key, epoch)
 * pairs instead -- 16 bytes per occurrence, one atomic per pair on the owner: exact, any --subset, slow. */
int  faqcs_kmer_partition(faqcs_ctx *ctx, uint32_t rank, uint32_t world, uint32_t n_epochs);
/* Epoch of every segment of the NEXT submission (which then buckets instead of inserting). */
int  faqcs_kmer_set_epochs(faqcs_ctx *ctx, const uint32_t *segment_epoch, uint32_t n_segments);
/* After a submission: device array of 16-byte items grouped by destination rank 0..world-1 and the number of ITEMS per destination
 * (counts[world]).  Valid until the next submission.  A second call without a submission in between reports nothing to send (all
 * counts 0): a rank of a collective loop whose part of the input was empty does not send the previous outbox again. */
int  faqcs_kmer_outbox(faqcs_ctx *ctx, void **d_items, uint64_t *counts);
/* The canonical keys of EVERY OCCURRENCE of the last submission's outbox, expanded from its items on the host (all destinations; keys ==
 * NULL or cap too small: only the count is returned).  For a caller that owns the table itself, like the reference's trim() seam (MAP<Word,size_t>,
 * trim.cpp:133-135): the key values are an injective re-encoding of the canonical k-mers, so counts and distinct counts
 * are the reference's, the key values are not. */
int  faqcs_kmer_outbox_host(faqcs_ctx *ctx, uint64_t *keys, uint64_t cap, uint64_t *n_keys);
/* Owner side: takes n_items received items (device pointer; returns when the buffer may be reused). */
int  faqcs_kmer_insert_device(faqcs_ctx *ctx, const void *d_items, uint64_t n_items);
/* (Replaces, like the calls around it, the per-call merge of thread-local k-mer tables into ONE map, trim.cpp:133-135, for a map that is
 * partitioned over devices.)  The exchange inside ONE process that drives several devices (faqcs_mi --gpus N --kmer_rarefaction): moves the last submission's
 * outbox of `from` to the owner contexts (owners[r] = the context faqcs_kmer_partition() made rank r of `world`) and inserts
 * it there; a peer copy when the owner sits on another device.  Returns when the outbox may be overwritten. */
int  faqcs_kmer_forward(faqcs_ctx *from, faqcs_ctx *const *owners, uint32_t world);
/* Owner side: keys by first epoch and occurrences by epoch of THIS rank's table ([n_epochs] each, cap >= n_epochs). */
int  faqcs_kmer_epoch_counts(faqcs_ctx *ctx, uint64_t *distinct_by_first_epoch, uint64_t *total_by_epoch, uint32_t cap);

/* ---- measurement helpers (used by bench.py; not part of the reference seam) ---- */
/* Fills device arenas with the SURVEY section-8(d) synthetic reads (counter-based PRNG keyed by
 * (seed, first_read + i)); stride == L (packed).  d_offset gets n_reads+1 entries. */
int  faqcs_synth_fill(int device_id, uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads,
                      uint32_t L, uint64_t seed, uint64_t first_read, float adapter_frac);
/* terminal_n flags (faqcs_batch) of a device-resident batch, computed on the device: d_flags[i], i < n_reads */
int  faqcs_terminal_n_flags(int device_id, const uint8_t *d_seq, const uint32_t *d_offset, uint32_t n_reads, uint8_t *d_flags);
/* The k-mer configuration of SURVEY section 8(d): reads are windows of a fixed synthetic genome of genome_len bases
 * (either strand, 0.5 % substitutions, the same quality recipe), so distinct k-mers grow as on real data. */
int  faqcs_synth_fill_genome(int device_id, uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads,
                             uint32_t L, uint64_t seed, uint64_t first_read, uint64_t genome_len);
/* diagnostic builds only: section clocks accumulated by the trim kernel (16 words; read and cleared) */
int  faqcs_debug_words(faqcs_ctx *ctx, uint64_t *out, uint32_t n);
/* average duration (ms) of the dominant kernel over the launches since the last call, measured with
 * HIP events recorded on the compute stream around each launch */
int  faqcs_kernel_time_ms(faqcs_ctx *ctx, double *avg_ms, uint64_t *n_launches);
/* the same per kernel: the trim kernel (and which variant ran: "trim_lds", "trim_tpr", "trim_filter_accumulate") and the
 * adapter pre-pass adapter_overlap (0 without adapters); both measured with HIP events on the compute stream */
typedef struct faqcs_kernel_times {
    double trim_ms, adapter_ms; uint64_t n_launches; const char *trim_kernel;
    double kmer_ms;        /* k-mer kernels of a submission (kmer_count; kmer_extract in the owner-partitioned mode), per submission */
    double kmer_insert_ms; /* faqcs_kmer_insert_device (owner-partitioned mode), per submission */
} faqcs_kernel_times;
int  faqcs_kernel_report(faqcs_ctx *ctx, faqcs_kernel_times *out);

#ifdef __cplusplus
}
#endif
#endif /* FAQCS_MI_H */
