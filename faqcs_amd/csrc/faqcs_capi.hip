// faqcs_capi.hip -- host side of libfaqcs_mi.so: the C ABI declared in include/faqcs_mi.h.
//
// One faqcs_ctx == the (filter_stats, adapter_stats, PlotInfo, Options) quadruple the reference keeps in
// main() (FaQCs.cpp:67-69) plus the device state: a compute stream, a copy stream, device staging arenas,
// the additive u64 counter block, the adapter tables and the k-mer hash table.
// There is NO CPU implementation of the hot path in this library: without a HIP device faqcs_create() fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "faqcs_dev.h"
#include "faqcs_kmer.h"
#include "faqcs_skm.h"

// kernels (other translation units)
struct AdapterDev {
    const uint8_t *bits;
    const uint32_t *start;
    const uint32_t *planes;
    const uint32_t *wstart;
    uint32_t n_adapters;
    float match_rate;
    uint32_t longest, plane_dwords;
};
const char *faqcs_last_trim_kernel();
bool faqcs_last_trim_folded();
hipError_t faqcs_launch_trim(const DevParams &P, const uint8_t *seq, const uint8_t *qual, const uint32_t *off,
                             uint32_t n_reads, uint32_t max_len, const uint32_t *ad_sl, const uint16_t *ad_hit,
                             faqcs_read_result *out, unsigned long long *rec_pre, unsigned long long *rec_post,
                             uint64_t *counters, uint32_t *err, int n_cu, hipStream_t st, const uint8_t *tn_flags);
hipError_t faqcs_launch_terminal_n_flags(const uint8_t *seq, const uint32_t *off, uint32_t n_reads, uint8_t *flags, hipStream_t st);
hipError_t faqcs_launch_composition(const unsigned long long *rec_pre, const unsigned long long *rec_post, uint32_t n, bool wide,
                                    const float *comp_norm, uint64_t *dst_pre, uint64_t *dst_post, int n_cu, hipStream_t st);
hipError_t faqcs_launch_adapter(const AdapterDev &A, const uint8_t *seq, const uint32_t *off, uint32_t n_reads,
                                uint32_t max_len, const uint32_t *seg_start, uint32_t n_segments, uint32_t *ad_sl,
                                uint16_t *ad_hit, uint64_t *adapter_stats, uint32_t *err, uint32_t dbg, int n_cu, hipStream_t st);
hipError_t faqcs_launch_synth(uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads, uint32_t L,
                              uint64_t seed, uint64_t first_read, float adapter_frac, uint64_t genome_len, float at_frac, hipStream_t st);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                         \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess)                                                                             \
            return fail(FAQCS_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_));                \
    } while (0)

namespace {

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// p .. a: the adapter pre-pass (when there is one), a .. b: the trim kernel, k0 .. k1: the submission's k-mer kernels (kmer_count, or
// kmer_extract in the owner-partitioned mode)
struct Timing { hipEvent_t a, b, p, k0, k1; bool adapter, kmer; };

} // namespace

struct faqcs_ctx {
    faqcs_params prm;
    int device = 0, n_cu = 256;
    hipStream_t compute = nullptr, copy = nullptr;
    hipEvent_t copied = nullptr;
    DevParams dp;
    faqcs_layout lay;
    // device tables
    uint32_t *d_lcthr = nullptr, *d_magic = nullptr, *d_basetab = nullptr;
    int32_t *d_avgq = nullptr;
    float *d_norm = nullptr;
    uint64_t *d_counters = nullptr;
    uint32_t *d_err = nullptr;
    uint32_t *d_partials = nullptr;
    // adapters
    std::vector<std::string> adapters;
    uint8_t *d_abits = nullptr;
    uint32_t *d_astart = nullptr, *d_aplanes = nullptr, *d_awstart = nullptr;
    float match_rate = 0.f;
    uint32_t adapter_longest = 0, adapter_plane_dwords = 0;
    // staging for host submissions: two input slots so the H2D copy of batch k+1 overlaps the kernels of batch k
    struct Slot { DevBuf<uint8_t> seq, qual, tn; DevBuf<uint32_t> off; hipEvent_t done = nullptr; bool used = false; };
    Slot slot[2];
    uint64_t n_submits = 0;
    hipEvent_t ticket_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf<uint32_t> s_seg, s_sl;
    DevBuf<uint16_t> s_hit;
    DevBuf<faqcs_read_result> s_res;
    // per-read composition records (trim kernel -> composition_histogram).  Two sets: the histogram kernels of
    // submission k run on the aux stream next to the trim kernel of submission k+1 (LDS-bound next to VALU-bound).
    struct RecSet { DevBuf<unsigned long long> pre, post; hipEvent_t trimmed = nullptr, folded = nullptr; bool used = false; };
    RecSet rec[2];
    uint64_t n_enqueued = 0;
    // round 6: the records of a launch are folded one launch LATE -- by the next trim_lds launch's blocks as they run out of reads (DevParams::fold_*),
    // else by composition_histogram on the aux stream beside the next launch, or on the compute stream when somebody needs the counters
    int pending_fold = -1;         // the record set that still has to be folded (-1: none)
    uint32_t pending_n = 0;
    bool pending_wide = false;
    uint32_t *d_fold_claim = nullptr;
    hipStream_t aux = nullptr;
    // rarefaction state (trim.cpp:157-185): host-deterministic from read counts, values filled from the device
    uint64_t total_number = 0;
    int kmer_active = 0;
    std::vector<faqcs_rarefaction> points;
    struct PendingPoint { size_t point_index; size_t snap_index; };
    std::vector<PendingPoint> pending;
    KmerTable kt{nullptr, 0, nullptr, 0};
    // owner-partitioned multi-GPU k-mer mode (faqcs_kmer_partition)
    bool partitioned = false;
    uint32_t part_rank = 0, part_world = 1, n_epochs = 0;
    std::vector<uint32_t> seg_epoch;             // epochs of the NEXT submission's segments
    DevBuf<ulonglong2> ob_items;
    bool ob_fresh = false;                       // the outbox holds a submission that faqcs_kmer_outbox() has not handed out yet
    DevBuf<uint32_t> ob_wave_count;
    DevBuf<unsigned long long> ob_wave_offset;
    unsigned long long *d_ob = nullptr;          // [3 * world]: dest_count, dest_offset, dest_cursor
    unsigned long long *d_tot_by_epoch = nullptr, *d_first_hist = nullptr; // [n_epochs] each
    unsigned long long *d_snaps = nullptr; // [snap_cap][2]
    size_t snap_cap = 0, n_snaps = 0;
    std::map<uint64_t, uint64_t> kmer_hist; // PlotInfo::kmer_frequency_histogram
    // combine-before-insert k-mer counting (every context that is not owner-partitioned; faqcs_kmer_skm_kernel.hip): the
    // k-mers of a run of segments are appended to bucket buffers at submission time and reach the table group by group
    struct KmerGroup {
        bool ready = false;
        bool direct = false;          // FAQCS_KMER_DIRECT=1 (diagnostics): one atomic insert per occurrence (kmer_count), as in rounds 1-3
        bool owner = false;           // owner-partitioned context whose received pairs go through the group buffers too (n_epochs <= KG_EPOCH_SPAN)
        bool skm = false;             // 16-byte super-k-mer items (faqcs_kmer_skm_kernel.hip): every context that is not owner-partitioned
        uint32_t skm_w = 1;           // k-mers an item can hold (k - min(k, 15) + 1)
        DevBuf<uint32_t> defer;       // [0]: how many, [1 ..]: the reads skm_extract16 left to skm_extract (k = 31, reads of up to 256 bases)
        uint64_t cap_items = 0;       // item bound of a group
        uint64_t bound_items = 0;     // upper bound of the items the open group holds
        std::vector<uint32_t> run_epoch, upload[2]; // epochs (relative to epoch_base) of the open group's runs; host copies in flight
        unsigned n_flushes = 0;
        uint64_t n_launches = 0;      // (owner side) launches so far: rotates the sub-regions
        uint32_t epoch_base = 0;
        std::vector<uint64_t> sub_fill; // [256] upper bound of the items in the level-1 sub-regions written by block slot i
        KmerGroupDev dev{};           // (n_runs / epoch_base filled in at flush time)
        uint32_t ep_cap = 0;          // entries of dev.first_hist / dev.tot_by_epoch
        uint32_t ep_used = 0;         // 1 + largest epoch seen
        size_t points_final = 0;      // points whose (distinct, total) are final (resolved before the table restarted)
        std::vector<std::pair<hipEvent_t, hipEvent_t>> flush_ev; size_t flush_ev_used = 0;
        // round 6: a pass whose items fit the group buffers is counted in ONE piece when it ends (faqcs_kmer_finish_pass / faqcs_kmer_end_table)
        bool table_live = false;      // a group of this pass has been flushed into the table: its last group goes there too, and the table is swept
        bool pass_done = false;       // the pass has been counted (faqcs_kmer_finish_pass): nothing can join it; faqcs_kmer_end_table starts the next one
        bool pass_used = false;       // k-mers have joined the pass
        bool hist_in_table = false;   // the histogram of counts still has to be read off the table (table_live)
        bool hist_in_overflow = false; // ... off the overflow area behind it only (a pass counted in one piece whose slices' probe windows filled up)
        uint64_t last_distinct = 0, last_total = 0; // totals of the pass faqcs_kmer_end_table finished last
    } kg;
    // sender staging of the multi-GPU k-mer exchange (super-k-mer items of ONE submission, grouped by destination rank afterwards)
    struct KmerSend {
        KmerGroupDev dev{};
        DevBuf<ulonglong2> l1, spill;
        DevBuf<uint32_t> cur1, run_epoch, defer;
        DevBuf<unsigned long long> scratch;
        uint32_t *spill_n = nullptr;
    } ks;
    // kernel timing
    std::vector<Timing> timings;
    size_t timing_used = 0;
    double kernel_ms = 0.0, adapter_ms = 0.0, kmer_ms = 0.0, kmer_insert_ms = 0.0, kmer_flush_ms = 0.0;
    uint64_t kernel_launches = 0;
    hipEvent_t ins_a = nullptr, ins_b = nullptr;
    // faqcs_kmer_forward, owner side: two staging buffers for items that arrive by peer copy; [k]: the insert that read buffer k is done / the copy into it is
    DevBuf<ulonglong2> fwd_items[2];
    hipEvent_t fwd_free[2] = {nullptr, nullptr}, fwd_copied[2] = {nullptr, nullptr};
    unsigned fwd_n = 0;
    const char *trim_kernel = "";
    void *comm = nullptr;          // ncclComm_t (faqcs_comm_init / faqcs_comm_init_all)
    hipEvent_t comm_ev = nullptr;  // the aux stream's work (composition fold) before the collective
};

// ---------------------------------------------------------------------------------------------------------
// layout + host helpers
// ---------------------------------------------------------------------------------------------------------
extern "C" int faqcs_abi_version(void) { return FAQCS_ABI_VERSION; }

extern "C" int faqcs_counters_layout(uint32_t R, uint32_t n_adapters, faqcs_layout *L)
{
    if (!L || R == 0 || R > FAQCS_MAX_READ_LENGTH || n_adapters > FAQCS_MAX_ADAPTERS) return fail(FAQCS_E_INVAL, "faqcs_counters_layout: bad size");
    uint64_t o = 0;
    memset(L, 0, sizeof(*L));
    L->max_read_length = R;
    L->n_adapters = n_adapters;
    L->filter_stats = o;    o += 32;
    L->pre_read_qhist = o;  o += FAQCS_NQ;
    L->pre_base_qhist = o;  o += FAQCS_NQ;
    L->post_read_qhist = o; o += FAQCS_NQ;
    L->post_base_qhist = o; o += FAQCS_NQ;
    L->pre_len_hist = o;    o += (uint64_t)R + 1;
    L->post_len_hist = o;   o += (uint64_t)R + 1;
    L->pre_qual = o;        o += (uint64_t)R * FAQCS_NQ;
    L->post_qual = o;       o += (uint64_t)R * FAQCS_NQ;
    L->pre_base = o;        o += (uint64_t)R * FAQCS_NBASE;
    L->post_base = o;       o += (uint64_t)R * FAQCS_NBASE;
    L->pre_comp = o;        o += (uint64_t)FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND;
    L->post_comp = o;       o += (uint64_t)FAQCS_NCOMP_BIN * FAQCS_NCOMP_KIND;
    L->adapter_stats = o;   o += (uint64_t)n_adapters * 2;
    L->total = o;
    return 0;
}

extern "C" uint32_t faqcs_counter_rows(const uint64_t *m, uint32_t max_rows, uint32_t n_cols)
{
    for (uint32_t r = max_rows; r > 0; --r)
        for (uint32_t c = 0; c < n_cols; ++c)
            if (m[(uint64_t)(r - 1) * n_cols + c]) return r;
    return 0;
}

extern "C" int faqcs_apply_edits(const faqcs_params *p, const uint8_t *seq, const uint8_t *qual, uint32_t read_len,
                                 const faqcs_read_result *res, uint8_t *out_seq, uint8_t *out_qual)
{
    if (!p || !res || (uint32_t)res->start + res->len > read_len) return fail(FAQCS_E_INVAL, "faqcs_apply_edits: window outside the read");
    uint32_t lead = 0, trail = read_len; // [lead, trail) keeps its quality (trim.cpp:1191-1216)
    while (lead < read_len && seq[lead] == 'N') ++lead;
    while (trail > 0 && seq[trail - 1] == 'N') --trail;
    const int in = p->input_quality_offset, out = p->output_quality_offset;
    for (uint32_t k = 0; k < res->len; ++k) {
        const uint32_t i = res->start + k;
        const int raw = (i < lead || i >= trail) ? in : (int)(int8_t)qual[i];
        int qs = raw - in;
        if (qs < 0) qs = 0;
        uint8_t b = seq[i];
        if (p->replace_to_N_q > 0 && b == 'G' && qs < (int)p->replace_to_N_q) b = 'N'; // trim.cpp:390-403
        out_seq[k] = b;
        out_qual[k] = (in != out) ? (uint8_t)(qs + out) : (uint8_t)raw;                  // trim.cpp:516-525
    }
    return 0;
}

extern "C" int faqcs_auto_detect_quality_offset(const uint8_t *qual, const uint32_t *offset, uint32_t n_reads)
{
    if (!n_reads) return 0;
    for (uint32_t i = offset[0]; i < offset[n_reads]; ++i) { // trim.cpp:599-617
        const int c = (int)(int8_t)qual[i];
        if (c > 74) return 64;
        if (c < 59) return 33;
    }
    return 0;
}

extern "C" const char *faqcs_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------------------------------------
// host-side integer tables (the reference's float32 expressions, evaluated once per length)
// ---------------------------------------------------------------------------------------------------------
static uint8_t na_to_bits_host(char c)
{
    switch (c) { // seq_overlap.cpp:372-411
    case 'A': case 'a': return 1;  case 'C': case 'c': return 2;  case 'G': case 'g': return 4;
    case 'T': case 't': return 8;  case 'M': case 'm': return 3;  case 'R': case 'r': return 5;
    case 'S': case 's': return 6;  case 'V': case 'v': return 7;  case 'W': case 'w': return 9;
    case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12;
    case 'D': case 'd': return 13; case 'B': case 'b': return 14; case 'N': case 'n': return 15;
    case '-': return 16;
    }
    return 0;
}

static void build_tables(const faqcs_params &p, std::vector<uint32_t> &lcthr, std::vector<int32_t> &avgq,
                         std::vector<float> &norm, std::vector<uint32_t> &magic, std::vector<uint32_t> &base)
{
    const int N = FAQCS_TAB_LEN + 1;
    lcthr.assign(N, 0xffffffffu); avgq.assign(N, 0); norm.assign(N, 0.f); magic.assign(N, 0);
    const volatile float lc = p.low_complexity_cutoff_ratio;
    const volatile float avg = p.average_quality;
    for (int len = 1; len < N; ++len) {
        uint32_t mono = 0xffff, di = 0xffff;
        volatile float nrm = (float)(1.0 / (double)len);                 // trim.cpp:483
        for (int c = 0; c <= len; ++c) { volatile float v = (float)(unsigned)c * nrm; if (v > lc) { mono = (uint32_t)c; break; } }
        volatile float nrm2 = (float)((double)nrm * 2.0);                // trim.cpp:499
        for (int c = 0; c <= len; ++c) { volatile float v = (float)(unsigned)c * nrm2; if (v > lc) { di = (uint32_t)c; break; } }
        lcthr[len] = mono | (di << 16);
        norm[len] = (float)(FAQCS_NCOMP_BIN - 1) / (float)len;           // trim.cpp:860
        magic[len] = (uint32_t)((1ull << 32) / (uint64_t)len) + 1u;      // exact floor(V/len) for V < 2^16, len >= 2
        if (avg > 0.0f) {
            // smallest V = sum(raw - offset) >= 0 for which NOT(ave_Q < avg); ave_Q per trim.cpp:568-572.
            // (V < 0 gives ave_Q = 0 < avg, and V < table[len] is then true as well.)
            auto pass = [&](int V) {
                volatile float t = (float)(V + p.input_quality_offset * len) / (float)len;
                volatile float v = t - (float)p.input_quality_offset;
                const float a = v > 0.0f ? v : 0.0f;
                return !(a < avg);
            };
            int lo = 0, hi = 256 * len;
            if (!pass(hi)) avgq[len] = 0x7fffffff;
            else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (pass(mid)) hi = mid; else lo = mid + 1; } avgq[len] = lo; }
        }
    }
    // per input byte: 6-bit count fields in FaQCs enum order A,T,C,G,N (FaQCs.h:35-42), case-insensitive like
    // update_base_statistics (trim.cpp:831-857); flag bits for the case-sensitive tests (trim.cpp:396,586,1196)
    base.assign(256, 0u);
    const char *letters = "ATCGN";
    for (int k = 0; k < 5; ++k) {
        base[(unsigned char)letters[k]] = 1u << BT_SHIFT(k);
        base[(unsigned char)(letters[k] | 0x20)] = 1u << BT_SHIFT(k);
    }
    base['N'] |= BT_IS_NU;
    base['G'] |= BT_IS_GU;
}

template <class T> static hipError_t upload(T **dst, const std::vector<T> &v)
{
    hipError_t e = hipMalloc((void **)dst, v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

// ---------------------------------------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------------------------------------
extern "C" int faqcs_create(const faqcs_params *p, int device_id, faqcs_ctx **out)
{
    if (!p || !out) return fail(FAQCS_E_INVAL, "faqcs_create: null argument");
    if (p->abi_version != FAQCS_ABI_VERSION) return fail(FAQCS_E_INVAL, "faqcs_create: ABI version mismatch");
    if (p->max_read_length == 0 || p->max_read_length > FAQCS_MAX_READ_LENGTH) return fail(FAQCS_E_INVAL, "faqcs_create: max_read_length out of range");
    if (p->n_adapters > FAQCS_MAX_ADAPTERS) return fail(FAQCS_E_INVAL, "faqcs_create: too many adapters");
    if (p->mode < 0 || p->mode > 2) return fail(FAQCS_E_INVAL, "trim.cpp:trim_read: Undefined trimming mode!");
    // the argmax keys of the trim kernel hold |sum of (Q - q)| <= 1024 * (|Q| + 41) in 18 bits (16 for reads <= 256 bases)
    if (p->quality < -93 || p->quality > 93) return fail(FAQCS_E_INVAL, "faqcs_create: quality threshold outside [-93, 93]");
    if (p->kmer_rarefaction && (p->kmer < 2 || p->kmer > 31 || p->split_size == 0)) return fail(FAQCS_E_INVAL, "faqcs_create: kmer / split_size out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FAQCS_E_NODEVICE, "faqcs_create: no HIP device (the FaQCs MI355X hot path has no CPU fallback)");
    if (device_id < 0) HIPCHK(hipGetDevice(&device_id));
    HIPCHK(hipSetDevice(device_id));
    // owned by a guard until every allocation below has succeeded: an early HIPCHK return must not leak the context
    std::unique_ptr<faqcs_ctx, void (*)(faqcs_ctx *)> guard(new faqcs_ctx(), faqcs_destroy);
    faqcs_ctx *c = guard.get();
    c->prm = *p;
    c->prm.adapter_seq = nullptr;
    c->device = device_id;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    for (auto &rs : c->rec) { HIPCHK(hipEventCreateWithFlags(&rs.trimmed, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&rs.folded, hipEventDisableTiming)); }
    HIPCHK(hipEventCreateWithFlags(&c->copied, hipEventDisableTiming));
    for (auto &sl : c->slot) HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    for (auto &e : c->ticket_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    faqcs_counters_layout(p->max_read_length, p->n_adapters, &c->lay);

    std::vector<uint32_t> lcthr, magic, basetab; std::vector<int32_t> avgq; std::vector<float> norm;
    build_tables(*p, lcthr, avgq, norm, magic, basetab);
    HIPCHK(upload(&c->d_lcthr, lcthr)); HIPCHK(upload(&c->d_avgq, avgq)); HIPCHK(upload(&c->d_norm, norm));
    HIPCHK(upload(&c->d_magic, magic)); HIPCHK(upload(&c->d_basetab, basetab));
    HIPCHK(hipMalloc((void **)&c->d_counters, c->lay.total * sizeof(uint64_t)));
    HIPCHK(hipMemset(c->d_counters, 0, c->lay.total * sizeof(uint64_t)));
    HIPCHK(hipMalloc((void **)&c->d_err, 256)); // [0] error bits ; bytes 64.. : 16 diagnostic u64 words (FAQCS_LDS_STAMPS builds)
    HIPCHK(hipMemset(c->d_err, 0, 256));
    {   // trim_lds: one row per block and flush, then the rows each block used
        const size_t dwords = (size_t)c->n_cu * FAQCS_PARTIAL_FLUSHES * FAQCS_PARTIAL_ROW + (size_t)c->n_cu;
        HIPCHK(hipMalloc((void **)&c->d_partials, dwords * sizeof(uint32_t)));
        HIPCHK(hipMemset(c->d_partials, 0, dwords * sizeof(uint32_t)));
    }

    if (p->n_adapters) {
        std::vector<uint8_t> bits; std::vector<uint32_t> start(1, 0);
        for (uint32_t j = 0; j < p->n_adapters; ++j) {
            const char *s = p->adapter_seq[j];
            const size_t L = strlen(s);
            if (L == 0 || L > FAQCS_MAX_ADAPTER_LENGTH) return fail(FAQCS_E_INVAL, "faqcs_create: adapter length out of range");
            c->adapters.emplace_back(s);
            for (size_t k = 0; k < L; ++k) {
                const uint8_t b = na_to_bits_host(s[k]);
                if (!b) return fail(FAQCS_E_BASE, "seq_overlap.cpp:na_to_bits: Unknown base!");
                bits.push_back(b);
            }
            start.push_back((uint32_t)bits.size());
        }
        // bit-planes for the prefilter: per adapter word w, 4 dwords (A,C,G,T) over bases [32w, 32w+32)
        std::vector<uint32_t> planes, wstart(1, 0);
        for (uint32_t j = 0; j < p->n_adapters; ++j) {
            const uint32_t L = start[j + 1] - start[j], nw = (L + 31) / 32;
            for (uint32_t w = 0; w < nw; ++w) {
                uint32_t pl[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < 32 && 32 * w + k < L; ++k)
                    for (int b = 0; b < 4; ++b) pl[b] |= (uint32_t)((bits[start[j] + 32 * w + k] >> b) & 1u) << k;
                planes.insert(planes.end(), pl, pl + 4);
            }
            wstart.push_back(wstart.back() + nw);
        }
        HIPCHK(upload(&c->d_abits, bits)); HIPCHK(upload(&c->d_astart, start));
        HIPCHK(upload(&c->d_aplanes, planes)); HIPCHK(upload(&c->d_awstart, wstart));
        c->adapter_plane_dwords = (uint32_t)planes.size();
        for (uint32_t j = 0; j < p->n_adapters; ++j) c->adapter_longest = std::max(c->adapter_longest, start[j + 1] - start[j]);
        c->match_rate = (float)(1.0 - (double)p->filterAdapterMismatchRate); // trim.cpp:969
    }

    DevParams &d = c->dp;
    memset(&d, 0, sizeof(d));
    d.mode = p->mode; d.Q = p->quality; d.in_off = p->input_quality_offset; d.out_off = p->output_quality_offset;
    // (anything above the longest read behaves like the longest read + 1; the kernels compare these as ints)
    auto cap16 = [](uint32_t v) { return v > 65535u ? 65535u : v; };
    d.min_len = cap16(p->min_read_length); d.max_poly_n = cap16(p->max_num_poly_N); d.trim5 = cap16(p->trim_5); d.trim3 = cap16(p->trim_3);
    d.replace_q = p->replace_to_N_q; d.protect5 = p->protect_5; d.qc_only = p->qc_only;
    d.has_adapters = p->n_adapters ? 1 : 0; d.avgq_on = p->average_quality > 0.0f ? 1 : 0;
    d.R = p->max_read_length; d.n_adapters = p->n_adapters;
    d.lc_ratio = p->low_complexity_cutoff_ratio; d.avg_q = p->average_quality;
    if (const char *e = getenv("FAQCS_DBG")) d.dbg = (uint32_t)strtoul(e, nullptr, 0);
    d.lc_thr = c->d_lcthr; d.avgq_min_v = c->d_avgq; d.comp_norm = c->d_norm; d.div_magic = c->d_magic; d.base_tab = c->d_basetab;
    d.partials = c->d_partials;
    d.partial_rows = c->d_partials + (size_t)c->n_cu * FAQCS_PARTIAL_FLUSHES * FAQCS_PARTIAL_ROW;
    d.lay = c->lay;

    c->kmer_active = p->kmer_rarefaction ? 1 : 0;
    if (p->kmer_rarefaction) {
        uint64_t slots = p->kmer_table_slots ? p->kmer_table_slots : (1ull << 28);
        uint64_t pow2 = 1; while (pow2 < slots) pow2 <<= 1;
        // every partition of the combine-before-insert path owns a slice of the table (faqcs_kmer_skm_kernel.hip)
        if (pow2 < (uint64_t)KG_SLICE_MIN << 16) pow2 = (uint64_t)KG_SLICE_MIN << 16;
        if (pow2 > (uint64_t)KG_SLICE_MAX << 16) return fail(FAQCS_E_INVAL, "faqcs_create: kmer_table_slots above 2^32");
        c->kt.mask = pow2 - 1;
        { uint32_t lg = 0; while ((1ull << lg) < pow2) ++lg; if (lg > 46) return fail(FAQCS_E_INVAL, "faqcs_create: kmer_table_slots too large"); c->kt.shift = 62 - lg; }
        if (const char *e = getenv("FAQCS_KMER_DIRECT")) c->kg.direct = atoi(e) != 0;
        // the overflow area (1/16 of the table, at least 2^16 slots): keys whose probe window in their partition's slice is full
        c->kt.ovf_mask = std::max<uint64_t>(pow2 >> 4, 1ull << 16) - 1;
        // fine partitions (faqcs_kmer.h): a table that is full (0.6 keys per slot) holds about 1 200 keys per fine partition up to 2^30 slots --
        // a third of what one round of the counting kernel's LDS table takes; 2^31 slots: 2 450, 2^32: 4 900 (several rounds then)
        { const uint32_t lg = 62 - c->kt.shift;
          int f = (int)lg - 27; f = f < 0 ? 0 : (f > 3 ? 3 : f);
          if (const char *e = getenv("FAQCS_KMER_FINE_BITS")) { const int v = atoi(e); if (v >= 0 && v <= 3) f = v; } // (tests: every F on a small table)
          while (f > 0 && (pow2 >> (16 + f)) < 16) --f;                    // a slice has at least 16 slots
          while (f < 3 && (pow2 >> (16 + f)) > (uint64_t)KG_SLICE_MAX / 8) ++f; // ... and at most what the counting kernel's claim bitmap covers
          c->kt.fine = (uint32_t)f; }
        HIPCHK(hipMalloc((void **)&c->kt.dirty, (size_t)(1u << (16 + c->kt.fine)) / 8));
        HIPCHK(hipMemset(c->kt.dirty, 0, (size_t)(1u << (16 + c->kt.fine)) / 8));
        HIPCHK(hipMalloc((void **)&c->kt.slots, kmer_table_total(c->kt) * sizeof(KmerSlot)));
        HIPCHK(hipMalloc((void **)&c->kt.stats, 64));
        HIPCHK(faqcs_launch_kmer_table_init(c->kt, c->n_cu, c->compute)); // empty key, count - 1 = 0, no epoch
        HIPCHK(hipMemset(c->kt.stats, 0, 64));
        c->snap_cap = 4096;
        HIPCHK(hipMalloc((void **)&c->d_snaps, c->snap_cap * 16));
    }
    *out = guard.release();
    return 0;
}

static void comm_release(void *comm); // (ncclCommDestroy: defined with the rest of the RCCL glue further down)

extern "C" void faqcs_destroy(faqcs_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->compute) (void)hipStreamSynchronize(c->compute);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    for (auto &t : c->timings) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); (void)hipEventDestroy(t.p); (void)hipEventDestroy(t.k0); (void)hipEventDestroy(t.k1); }
    if (c->comm) comm_release(c->comm);
    if (c->comm_ev) (void)hipEventDestroy(c->comm_ev);
    if (c->ins_a) (void)hipEventDestroy(c->ins_a);
    if (c->ins_b) (void)hipEventDestroy(c->ins_b);
    for (int k = 0; k < 2; ++k) { if (c->fwd_free[k]) (void)hipEventDestroy(c->fwd_free[k]); if (c->fwd_copied[k]) (void)hipEventDestroy(c->fwd_copied[k]); c->fwd_items[k].release(); }
    void *ptrs[] = {c->d_lcthr, c->d_basetab, c->d_avgq, c->d_norm, c->d_magic, c->d_counters, c->d_err, c->d_partials, c->d_abits, c->d_astart, c->d_aplanes, c->d_awstart,
                    c->kt.slots, c->kt.stats, c->kt.dirty, c->d_fold_claim, c->d_snaps, c->d_ob, c->d_tot_by_epoch, c->d_first_hist};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    for (auto &sl : c->slot) { sl.seq.release(); sl.qual.release(); sl.tn.release(); sl.off.release(); if (sl.done) (void)hipEventDestroy(sl.done); }
    for (auto &e : c->ticket_ev) if (e) (void)hipEventDestroy(e);
    c->s_seg.release(); c->s_sl.release(); c->s_hit.release(); c->s_res.release();
    for (auto &rs : c->rec) { rs.pre.release(); rs.post.release(); if (rs.trimmed) (void)hipEventDestroy(rs.trimmed); if (rs.folded) (void)hipEventDestroy(rs.folded); }
    if (c->aux) (void)hipStreamDestroy(c->aux);
    c->ob_items.release(); c->ob_wave_count.release(); c->ob_wave_offset.release();
    { void *kg_ptrs[] = {c->kg.dev.l1, c->kg.dev.l2, c->kg.dev.cur1, c->kg.dev.cur2, c->kg.dev.run_epoch, c->kg.dev.first_hist, c->kg.dev.tot_by_epoch, c->kg.dev.dense, c->kg.dev.big, c->kg.dev.n_big, c->kg.dev.redo};
      for (void *q : kg_ptrs) if (q) (void)hipFree(q);
      c->kg.defer.release();
      c->ks.l1.release(); c->ks.spill.release(); c->ks.cur1.release(); c->ks.run_epoch.release(); c->ks.defer.release(); c->ks.scratch.release();
      for (auto &ev : c->kg.flush_ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); } }
    if (c->copied) (void)hipEventDestroy(c->copied);
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    delete c;
}

extern "C" int faqcs_set_quality(faqcs_ctx *c, int quality)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (quality < -93 || quality > 93) return fail(FAQCS_E_INVAL, "faqcs_set_quality: quality threshold outside [-93, 93]");
    c->prm.quality = quality;
    c->dp.Q = quality;
    return 0;
}


// ---------------------------------------------------------------------------------------------------------
// combine-before-insert k-mer counting: host side (kernels: faqcs_kmer_skm_kernel.hip)
// ---------------------------------------------------------------------------------------------------------
// Sizes of the group buffers for a table of `slots` slots, k-mers of k bases and `free_b` bytes of free device memory (kg_init, and
// faqcs_kmer_memory_plan for a caller that wants to know before anything is allocated).
// Either scatter level writes 65 536 sub-regions: (bucket, writing block) at level 1, partitions at level 2.  An item is 16 bytes and
// holds a run of up to w k-mers, (w + 1) / 2 on average; the buffers are sized for three items per w + 1 occurrences (the bench's reads
// make one per 8), a level-1 sub-region for 1.25 x its even share, a partition -- minimizer bins vary more than hash bins do -- for
// 1.5 x.  What overflows is counted occurrence by occurrence (exact, slow).
// The item bound of a group.  Round 6: a pass that fits ONE group never reaches the table (faqcs_kmer.h), so a group is as large as the
// table's own sizing rule makes a pass -- six occurrences per slot: 0.6 distinct keys per slot at a coverage of 10 -- and the HBM
// allows: at most 60 % of what is free (the caller's batches are resident already in every use of this library).  2^31 slots:
// groups of 12.9 G occurrences, 44 + 53 GB of buffers next to the 36.5 GB table.
struct KgPlan { uint64_t G; uint32_t cap1, cap2; size_t l1_bytes, l2_bytes; };
static KgPlan kg_plan(uint64_t slots, uint32_t kmer, size_t free_b)
{
    const uint32_t w = kmer > 15 ? kmer - 14 : 1;
    KgPlan pl{};
    auto caps = [&](uint64_t G) {
        const double G_items = w == 1 ? (double)G : (double)G * 3.0 / (w + 1);
        const double mean = G_items / (KG_FAN * KG_FAN);
        pl.cap1 = (uint32_t)(mean * 1.25 + 8.0 * std::sqrt(mean) + 64.0);
        pl.cap2 = (uint32_t)(mean * 1.5 + 8.0 * std::sqrt(mean) + 64.0);
        if (pl.cap1 < (uint32_t)KG_MIN_CAP) pl.cap1 = KG_MIN_CAP;
        if (pl.cap2 < (uint32_t)KG_MIN_CAP) pl.cap2 = KG_MIN_CAP;
        pl.cap2 = (pl.cap2 + 7u) & ~7u; // (cut 2^F ways for the fine partitions of a pass counted at its end)
        pl.l1_bytes = (size_t)pl.cap1 * KG_FAN * KG_FAN * 16; pl.l2_bytes = (size_t)pl.cap2 * KG_FAN * KG_FAN * 16;
        return pl.l1_bytes + pl.l2_bytes;
    };
    uint64_t G = slots * 6;
    if (G < (1ull << 18)) G = 1ull << 18;
    if (G > (1ull << 36)) G = 1ull << 36;
    bool fixed = false;
    if (const char *e = getenv("FAQCS_KMER_GROUP_ITEMS")) { const uint64_t v = strtoull(e, nullptr, 0); if (v >= (1ull << 14) && v <= (1ull << 36)) { G = v; fixed = true; } }
    while (!fixed && G > (1ull << 18) && caps(G) > free_b / 10 * 6) G -= G / 4;
    (void)caps(G);
    pl.G = G;
    return pl;
}

// What a kmer_rarefaction context will hold on its device: out[0] the table (with its overflow area), out[1] / out[2] the level-1 / level-2
// group buffers, out[3] the occurrences a group takes (a pass below it is counted in one piece), out[4] the small arrays.  Host only.
extern "C" int faqcs_kmer_memory_plan(const faqcs_params *p, uint64_t free_bytes, uint64_t *out, uint32_t n_out)
{
    if (!p || !out || n_out < 5) return fail(FAQCS_E_INVAL, "faqcs_kmer_memory_plan: null argument / fewer than 5 words");
    for (uint32_t i = 0; i < n_out; ++i) out[i] = 0;
    if (!p->kmer_rarefaction) return 0;
    uint64_t slots = p->kmer_table_slots ? p->kmer_table_slots : (1ull << 28);
    uint64_t pow2 = 1; while (pow2 < slots) pow2 <<= 1;
    if (pow2 < (uint64_t)KG_SLICE_MIN << 16) pow2 = (uint64_t)KG_SLICE_MIN << 16;
    const uint64_t table = (pow2 + std::max<uint64_t>(pow2 >> 4, 1ull << 16)) * sizeof(KmerSlot);
    const KgPlan pl = kg_plan(pow2, p->kmer, free_bytes > table ? (size_t)(free_bytes - table) : 0);
    out[0] = table; out[1] = pl.l1_bytes; out[2] = pl.l2_bytes; out[3] = pl.G;
    out[4] = (uint64_t)KG_FAN * KG_FAN * 4 * 18 + (1u << 16) * 8 + (1u << 20) * 8 + (1ull << 19) / 8; // cursors, redo list, histogram of counts, dirty bits
    return 0;
}

static int kg_init(faqcs_ctx *c)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (g.ready) return 0;
    const uint64_t slots = c->kt.mask + 1;
    KmerGroupDev &d = g.dev;
    g.skm = !g.direct; // (FAQCS_KMER_DIRECT=1 on an owner-partitioned context: round 3's pairs and per-pair atomics)
    g.skm_w = c->prm.kmer > 15 ? c->prm.kmer - 14 : 1;
    const size_t item_bytes = 16;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const KgPlan pl = kg_plan(slots, c->prm.kmer, free_b);
    const uint64_t G = pl.G;
    d.cap1 = pl.cap1; d.cap2 = pl.cap2;
    g.cap_items = G;
    d.split = 1u; // (a partition's items in one piece: skm_combine fetches them by index)
    d.cap2f = d.cap2 >> c->kt.fine;
    HIPCHK(hipMalloc((void **)&d.l1, (size_t)KG_FAN * KG_FAN * d.cap1 * item_bytes));
    HIPCHK(hipMalloc((void **)&d.l2, (size_t)KG_FAN * KG_FAN * d.cap2 * item_bytes));
    HIPCHK(hipMalloc((void **)&d.cur1, (size_t)KG_FAN * KG_FAN * 4));
    HIPCHK(hipMalloc((void **)&d.cur2, (size_t)KG_FAN * KG_FAN * 8 * 4)); // (up to 2^19 fine partitions)
    HIPCHK(hipMalloc((void **)&d.run_epoch, (size_t)KG_MAX_RUNS * 4));
    // histogram of counts of a pass counted at its end (the kernels add to it; faqcs_kmer_end_table reads it and starts it again)
    d.dense_n = 1u << 16; d.big_cap = 1u << 20;
    HIPCHK(hipMalloc((void **)&d.dense, (size_t)d.dense_n * 8)); HIPCHK(hipMalloc((void **)&d.big, (size_t)d.big_cap * 8)); HIPCHK(hipMalloc((void **)&d.n_big, 8));
    HIPCHK(hipMemsetAsync(d.dense, 0, (size_t)d.dense_n * 8, c->compute)); HIPCHK(hipMemsetAsync(d.n_big, 0, 8, c->compute));
    HIPCHK(hipMalloc((void **)&d.redo, ((size_t)KG_FAN * KG_FAN * 8 + 1) * 4));
    d.n_redo = d.redo + (size_t)KG_FAN * KG_FAN * 8;
    if (g.owner && c->part_world > 1) { // the partitions this rank owns, mapped onto [0, 2^19) (faqcs_kmer.h: part_mul)
        const uint32_t lo_b = (c->part_rank * 256u + c->part_world - 1u) / c->part_world, hi_b = ((c->part_rank + 1u) * 256u + c->part_world - 1u) / c->part_world;
        d.part_lo = lo_b << 11;
        d.part_mul = hi_b > lo_b ? (1ull << 51) / ((uint64_t)(hi_b - lo_b) << 11) : 0ull;
    }
    HIPCHK(faqcs_launch_skm_reset(d, c->compute));
    g.sub_fill.assign(KG_FAN, 0);
    g.ready = true;
    return 0;
}

// first_hist / tot_by_epoch hold at least `need` epochs (grown with their contents)
static int kg_ensure_epochs(faqcs_ctx *c, uint32_t need)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (need > g.ep_used) g.ep_used = need;
    if (need <= g.ep_cap) return 0;
    const uint32_t cap = need + need / 2 + 4096;
    unsigned long long *f = nullptr, *t = nullptr;
    HIPCHK(hipMalloc((void **)&f, (size_t)cap * 8)); HIPCHK(hipMalloc((void **)&t, (size_t)cap * 8));
    HIPCHK(hipMemsetAsync(f, 0, (size_t)cap * 8, c->compute)); HIPCHK(hipMemsetAsync(t, 0, (size_t)cap * 8, c->compute));
    if (g.ep_cap) {
        HIPCHK(hipMemcpyAsync(f, g.dev.first_hist, (size_t)g.ep_cap * 8, hipMemcpyDeviceToDevice, c->compute));
        HIPCHK(hipMemcpyAsync(t, g.dev.tot_by_epoch, (size_t)g.ep_cap * 8, hipMemcpyDeviceToDevice, c->compute));
        HIPCHK(hipStreamSynchronize(c->compute)); // (kernels in flight hold the old pointers)
        (void)hipFree(g.dev.first_hist); (void)hipFree(g.dev.tot_by_epoch);
    }
    g.dev.first_hist = f; g.dev.tot_by_epoch = t; g.dev.n_epochs = cap; g.ep_cap = cap;
    return 0;
}

// FAQCS_KMER_DEBUG=1 (diagnostics): a flush step by step with the buffers checked on the host in between -- every level-1 item sits
// in the bucket of its partition's top 8 bits, every level-2 item in its partition, no item holds more than w k-mers; prints the
// item / occurrence / distinct-key counts and the fullest partition (the host expands the items with the kernels' own faqcs_skm.h)
static int kg_debug_flush(faqcs_ctx *c, bool final)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    const KmerGroupDev &d = g.dev;
    const SkmGeom geo = skm_geom(c->prm.kmer);
    HIPCHK(hipStreamSynchronize(c->compute));
    // (final: the pass is counted in one piece -- 2^(16 + F) fine partitions of cap2f items instead of 65 536 of cap2)
    const uint32_t n_parts = final ? 1u << (16 + c->kt.fine) : (uint32_t)KG_FAN * KG_FAN, pcap = final ? d.cap2f : d.cap2;
    auto part_of = [&](unsigned long long w1) { return final ? skm_item_part(w1) >> (3 - c->kt.fine) : skm_item_p16(w1); };
    auto flush_stage = [&](uint32_t st) { return final ? faqcs_launch_skm_finish(d, c->kt, c->prm.kmer, c->n_cu, c->compute, st) : faqcs_launch_skm_flush(d, c->kt, c->prm.kmer, c->compute, st); };
    std::vector<uint32_t> cur1((size_t)KG_FAN * KG_FAN), cur2((size_t)n_parts);
    HIPCHK(hipMemcpy(cur1.data(), d.cur1, cur1.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long st[3];
    HIPCHK(hipMemcpy(st, c->kt.stats, 24, hipMemcpyDeviceToHost));
    uint64_t n1 = 0, occ1 = 0, bad1 = 0, long1 = 0;
    std::vector<ulonglong2> buf(d.cap1);
    for (uint32_t b = 0; b < (uint32_t)KG_FAN; ++b)
        for (uint32_t s = 0; s < (uint32_t)KG_FAN; ++s) {
            const uint32_t n = cur1[(size_t)s * KG_FAN + b];
            if (!n) continue;
            if (n > d.cap1) { fprintf(stderr, "[kmer debug] level-1 cursor %u > cap %u (bucket %u, sub %u)\n", n, d.cap1, b, s); continue; }
            HIPCHK(hipMemcpy(buf.data(), reinterpret_cast<const ulonglong2 *>(d.l1) + ((size_t)b * KG_FAN + s) * d.cap1, (size_t)n * 16, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i) {
                ++n1; occ1 += skm_item_kmers(buf[i].y);
                if (skm_item_bucket(buf[i].y) != b) ++bad1;
                if (skm_item_kmers(buf[i].y) > geo.w || skm_item_run(buf[i].y) >= g.run_epoch.size()) ++long1;
            }
        }
    { uint32_t nb_used = 0, first_b = 0xffffffffu, last_b = 0, ns_used = 0;
      for (uint32_t b = 0; b < (uint32_t)KG_FAN; ++b) { uint64_t t = 0; for (uint32_t s2 = 0; s2 < (uint32_t)KG_FAN; ++s2) t += cur1[(size_t)s2 * KG_FAN + b]; if (t) { ++nb_used; first_b = std::min(first_b, b); last_b = b; } }
      for (uint32_t s2 = 0; s2 < (uint32_t)KG_FAN; ++s2) { uint64_t t = 0; for (uint32_t b = 0; b < (uint32_t)KG_FAN; ++b) t += cur1[(size_t)s2 * KG_FAN + b]; if (t) ++ns_used; }
      fprintf(stderr, "[kmer debug] cap1 %u cap2 %u split %u runs %zu; buckets in use %u (%u .. %u), sub-regions in use %u\n", d.cap1, d.cap2, d.split, g.run_epoch.size(), nb_used, first_b, last_b, ns_used); }
    if (g.defer.p) { uint32_t nd = 0; HIPCHK(hipMemcpy(&nd, g.defer.p, 4, hipMemcpyDeviceToHost)); std::vector<uint32_t> dl(std::min<uint32_t>(nd, 8)); if (!dl.empty()) HIPCHK(hipMemcpy(dl.data(), g.defer.p + 1, dl.size() * 4, hipMemcpyDeviceToHost));
                     fprintf(stderr, "[kmer debug] the last 16-positions launch left %u reads to the general kernel:", nd); for (uint32_t v : dl) fprintf(stderr, " %u", v); fprintf(stderr, "\n"); }
    fprintf(stderr, "[kmer debug] before the flush: %llu level-1 items, %llu occurrences, %llu in a wrong bucket, %llu with a bad length / run; overflow flag %llu, total %llu\n",
            (unsigned long long)n1, (unsigned long long)occ1, (unsigned long long)bad1, (unsigned long long)long1, st[2], st[1]);
    HIPCHK(flush_stage(1u));
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(cur2.data(), d.cur2, cur2.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st, c->kt.stats, 24, hipMemcpyDeviceToHost));
    uint64_t n2 = 0, occ2 = 0, bad2 = 0, max_keys = 0, max_items = 0, distinct = 0;
    buf.resize((size_t)pcap);
    std::vector<unsigned long long> keys;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const uint32_t np = cur2[p];
        if (!np) continue;
        HIPCHK(hipMemcpy(buf.data(), reinterpret_cast<const ulonglong2 *>(d.l2) + (size_t)p * pcap, buf.size() * 16, hipMemcpyDeviceToHost));
        keys.clear();
        for (uint32_t j = 0; j < 1; ++j)
            for (uint32_t i = 0; i < np && i < pcap; ++i) {
                const ulonglong2 it = buf[i];
                ++n2; occ2 += skm_item_kmers(it.y);
                if (part_of(it.y) != p) ++bad2;
                SkmRoll r = skm_roll_begin(it.x, it.y, geo);
                for (uint32_t t = 0; t < skm_item_kmers(it.y) && t < 32; ++t) { keys.push_back(skm_mix62(skm_roll_key(r))); skm_roll_next(r, geo); }
            }
        std::sort(keys.begin(), keys.end());
        const uint64_t nd = (uint64_t)(std::unique(keys.begin(), keys.end()) - keys.begin());
        distinct += nd;
        max_keys = std::max<uint64_t>(max_keys, nd); max_items = std::max<uint64_t>(max_items, np);
    }
    fprintf(stderr, "[kmer debug] after the split: %llu level-2 items, %llu occurrences, %llu in a wrong partition; %llu distinct keys in this group, fullest partition %llu keys / %llu items (slice: %llu slots); overflow flag %llu\n",
            (unsigned long long)n2, (unsigned long long)occ2, (unsigned long long)bad2, (unsigned long long)distinct, (unsigned long long)max_keys, (unsigned long long)max_items,
            (unsigned long long)((c->kt.mask + 1) >> (final ? 16 + c->kt.fine : 16)), st[2]);
    HIPCHK(flush_stage(2u));
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(st, c->kt.stats, 24, hipMemcpyDeviceToHost));
    fprintf(stderr, "[kmer debug] after the %s: overflow flag %llu\n", final ? "count (the pass in one piece)" : "combine", st[2]);
    HIPCHK(flush_stage(4u));
    return 0;
}

// the open group's items are counted: level-2 scatter, combine, cursors back to zero (all on the compute stream).
// final == false: into the table, while the pass goes on (the group buffers are full, or a caller asks for the curve so far);
// final == true: the pass ends with this group and no group of it has gone into the table -- counted in one piece, the table untouched.
// (timed: a flush outside a submission's k0 .. k1 events brings its own pair)
static int kg_flush(faqcs_ctx *c, bool timed = false, bool final = false)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (!g.ready || g.run_epoch.empty()) return 0;
    std::vector<uint32_t> &up = g.upload[g.n_flushes++ & 1]; // (kept alive past the asynchronous copy)
    up = g.run_epoch;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (timed) {
        if (g.flush_ev_used == g.flush_ev.size()) {
            hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            g.flush_ev.emplace_back(a, b);
        }
        ev = g.flush_ev[g.flush_ev_used++];
        HIPCHK(hipEventRecord(ev.first, c->compute));
    }
    HIPCHK(hipMemcpyAsync(g.dev.run_epoch, up.data(), up.size() * 4, hipMemcpyHostToDevice, c->compute));
    g.dev.n_runs = (uint32_t)up.size(); g.dev.epoch_base = g.epoch_base;
    const char *dbg = getenv("FAQCS_KMER_DEBUG"); // (read at every flush: a test turns it on for one engine)
    const bool debug = dbg && atoi(dbg) != 0;
    const char *e_final = getenv("FAQCS_KMER_FINAL"); // (0: round 5's path for every group -- A/B runs, and the tests of that path; read at every flush)
    final = final && !g.table_live && !(e_final && atoi(e_final) == 0);
    if (g.skm && debug) { if (int rc = kg_debug_flush(c, final)) return rc; }
    else if (final) HIPCHK(faqcs_launch_skm_finish(g.dev, c->kt, c->prm.kmer, c->n_cu, c->compute));
    else HIPCHK(faqcs_launch_skm_flush(g.dev, c->kt, c->prm.kmer, c->compute));
    if (!final) g.table_live = true;
    if (timed) HIPCHK(hipEventRecord(ev.second, c->compute));
    g.run_epoch.clear(); g.bound_items = 0;
    std::fill(g.sub_fill.begin(), g.sub_fill.end(), 0);
    return 0;
}

// the k-mers of reads [r0, r1) -- one epoch -- join the open group.  host_off (may be null): the host copy of the offsets, for a
// tight item bound (an occurrence starts at a distinct base); otherwise per_read items bound every read.
static int kg_add_run(faqcs_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, const uint32_t *d_off, uint32_t r0, uint32_t r1,
                      const faqcs_read_result *d_res, uint64_t per_read, const uint32_t *host_off, uint32_t epoch, uint32_t max_len)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (g.pass_done) return fail(FAQCS_E_INVAL, "faqcs_submit: the k-mer pass has been counted (faqcs_kmer_finish_pass); faqcs_kmer_end_table starts the next one");
    if (int rc = kg_init(c)) return rc;
    if (int rc = kg_ensure_epochs(c, epoch + 1)) return rc;
    g.pass_used = true;
    if (per_read == 0) per_read = 1;
    while (r0 < r1) {
        // (round 6: an item carries its epoch RELATIVE TO THE GROUP'S FIRST directly -- the run -> epoch table of the group is the identity --, so a
        // group takes any number of extraction launches: faqcs_mi submits a launch per 32 768-read buffer, and a pass of more than 1 000 buffers
        // was cut into groups by the 10-bit run field alone.  What is left is the span of epochs an LDS histogram of the counting kernels takes.)
        if (!g.run_epoch.empty() && epoch - g.epoch_base >= (uint32_t)KG_EPOCH_SPAN) {
            if (int rc = kg_flush(c)) return rc;
        }
        // A launch of `take` reads runs on grid blocks; block i appends to sub-region (i + rot) % 256 of every bucket: a share
        // 1 / (256 grid) of the launch's items each (hashing spreads a block's items over the buckets; the blocks take equal
        // numbers of reads).  sub_fill bounds every sub-region from above, 1/8 + 64 items of variance included; the group is
        // flushed before a sub-region could overflow (an overflow would be exact too, but slow: kmer_insert_atomic).
        const uint32_t run = g.run_epoch.empty() ? 0u : epoch - g.epoch_base, rot = (uint32_t)((g.n_launches * 37u) % KG_FAN);
        // items a launch can be expected to write at most (super-k-mers: three per w + 1 occurrences and two per read)
        auto items_of = [&](uint32_t take, uint64_t bound) { return !g.skm || g.skm_w == 1 ? bound : bound * 3 / (g.skm_w + 1) + 2ull * take; };
        static const bool no16g = [] { const char *e = getenv("FAQCS_KMER_EXTRACT16"); return e && atoi(e) == 0; }();
        const bool x16 = g.skm && c->prm.kmer == 31 && max_len <= 256 && !no16g;
        auto grid_of = [&](uint32_t take) { return x16 ? faqcs_skm_grid16(take, c->n_cu) : faqcs_skm_grid(take, c->n_cu); };
        // sub_fill[i] = the EXPECTED number of items in the sub-regions block slot i has written (its even share of every launch so far); what a
        // sub-region may hold beyond that -- skew of the blocks' reads and of the buckets, 1/8, and the scatter of a sum of independent shares,
        // eight standard deviations + a granule -- is added ONCE, to the sum (round 5 added 64 items per launch: a pass of 32 768-read
        // submissions, 14 items per sub-region each, was flushed after 540 of them with its sub-regions a fifth full)
        auto over = [&](uint64_t mean) { return mean + mean / 8 + (uint64_t)(8.0 * std::sqrt((double)mean)) + 64 > g.dev.cap1; };
        auto fits = [&](uint32_t take, uint64_t bound) {
            const uint32_t grid = grid_of(take);
            const uint64_t ib = items_of(take, bound);
            const uint64_t share = ib / ((uint64_t)grid * KG_FAN) + 1;
            for (uint32_t i = 0; i < grid; ++i) if (over(g.sub_fill[(i + rot) % KG_FAN] + share)) return false;
            return g.bound_items + bound <= g.cap_items;
        };
        auto bound_of = [&](uint32_t take) { return host_off ? (uint64_t)(host_off[r0 + take] - host_off[r0]) : (uint64_t)take * per_read; };
        uint32_t take = r1 - r0;
        if (!fits(take, bound_of(take))) { // the largest prefix that fits (the bound grows with take; the per-block share may not)
            uint32_t lo = 0, hi = take;
            while (lo < hi) { const uint32_t mid = lo + (hi - lo + 1) / 2; if (fits(mid, bound_of(mid))) lo = mid; else hi = mid - 1; }
            take = lo;
        }
        if (take == 0) {
            if (!g.run_epoch.empty()) { if (int rc = kg_flush(c)) return rc; continue; }
            take = 1; // (a read that no empty group has room for: whatever overflows is counted by the per-occurrence path)
        }
        const uint64_t bound = bound_of(take), ib = items_of(take, bound);
        const uint32_t grid = grid_of(take);
        if (g.run_epoch.empty()) g.epoch_base = epoch;
        static const bool no16 = [] { const char *e = getenv("FAQCS_KMER_EXTRACT16"); return e && atoi(e) == 0; }(); // (A/B switch)
        if (g.skm && c->prm.kmer == 31 && max_len <= 256 && !no16) { // four reads per wave and round; what it cannot take goes to the general kernel behind it
            if ((size_t)take + 1 > g.defer.cap) HIPCHK(g.defer.reserve((size_t)(r1 - r0) + 1));
            HIPCHK(hipMemsetAsync(g.defer.p, 0, 4, c->compute));
            HIPCHK(faqcs_launch_skm_extract16(c->dp, g.dev, c->kt, run, rot, epoch, d_seq, d_qual, d_off, r0, r0 + take, d_res,
                                              g.defer.p + 1, g.defer.p, c->n_cu, c->compute));
            HIPCHK(faqcs_launch_skm_extract(c->dp, c->prm.kmer, g.dev, c->kt, run, rot, epoch, d_seq, d_qual, d_off, r0, r0 + take, d_res,
                                            c->n_cu, c->compute, g.defer.p + 1, g.defer.p, grid));
        } else HIPCHK(faqcs_launch_skm_extract(c->dp, c->prm.kmer, g.dev, c->kt, run, rot, epoch, d_seq, d_qual, d_off,
                                              r0, r0 + take, d_res, c->n_cu, c->compute));
        while (g.run_epoch.size() <= (size_t)run) g.run_epoch.push_back((uint32_t)g.run_epoch.size()); // (the identity, as long as the group's epochs span)
        ++g.n_launches;
        g.bound_items += bound;
        for (uint32_t i = 0; i < grid; ++i) g.sub_fill[(i + rot) % KG_FAN] += ib / ((uint64_t)grid * KG_FAN) + 1;
        r0 += take;
    }
    return 0;
}

// owner side of the multi-GPU exchange: n received items (device memory; their run fields hold absolute epochs) join the open group,
// whose run -> epoch table is the identity (at most KG_EPOCH_SPAN epochs: faqcs_kmer_partition sends a job with more through the
// (key, epoch) pairs of FAQCS_KMER_DIRECT instead)
static int kg_add_items(faqcs_ctx *c, const void *d_items, uint64_t n)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (g.pass_done) return fail(FAQCS_E_INVAL, "faqcs_kmer_insert_device: the k-mer pass has been counted (faqcs_kmer_finish_pass); faqcs_kmer_end_table starts the next one");
    if (int rc = kg_init(c)) return rc;
    if (int rc = kg_ensure_epochs(c, c->n_epochs)) return rc;
    g.pass_used = true;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(d_items);
    while (n) {
        const uint32_t rot = (uint32_t)((g.n_launches * 37u) % KG_FAN);
        // (an item holds up to skm_w occurrences: the group's bound counts occurrences, the sub-regions items)
        auto over = [&](uint64_t mean) { return mean + mean / 8 + (uint64_t)(8.0 * std::sqrt((double)mean)) + 64 > g.dev.cap1; }; // (as in kg_add_run)
        auto fits = [&](uint64_t take) {
            const uint32_t grid = faqcs_skm_items_grid(take, c->n_cu);
            const uint64_t share = take / ((uint64_t)grid * KG_FAN) + 1;
            for (uint32_t i = 0; i < grid; ++i) if (over(g.sub_fill[(i + rot) % KG_FAN] + share)) return false;
            return g.bound_items + take * g.skm_w <= g.cap_items;
        };
        uint64_t take = n;
        if (!fits(take)) {
            uint64_t lo = 0, hi = take;
            while (lo < hi) { const uint64_t mid = lo + (hi - lo + 1) / 2; if (fits(mid)) lo = mid; else hi = mid - 1; }
            take = lo;
        }
        if (take == 0) {
            if (g.bound_items) { if (int rc = kg_flush(c)) return rc; continue; }
            take = 1;
        }
        const uint32_t grid = faqcs_skm_items_grid(take, c->n_cu);
        g.epoch_base = 0;
        HIPCHK(faqcs_launch_skm_items(g.dev, c->kt, c->prm.kmer, rot, p, take, c->n_cu, c->compute));
        ++g.n_launches;
        if (g.run_epoch.empty()) { g.run_epoch.resize(c->n_epochs); for (uint32_t j = 0; j < c->n_epochs; ++j) g.run_epoch[j] = j; }
        g.bound_items += take * g.skm_w;
        for (uint32_t i = 0; i < grid; ++i) g.sub_fill[(i + rot) % KG_FAN] += take / ((uint64_t)grid * KG_FAN) + 1;
        p += take * 16; n -= take;
    }
    return 0;
}

// (distinct, total) of the points that are not final yet, from the epoch histograms: distinct(point i) = keys whose first epoch
// is <= i, total(point i) = occurrences with epoch <= i (both arrays restart with the table, faqcs_kmer_end_table)
static int kg_resolve_points(faqcs_ctx *c)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    if (!g.ready || c->points.size() <= g.points_final) return 0;
    const size_t n = std::min<size_t>(c->points.size(), g.ep_used);
    std::vector<unsigned long long> f(n), t(n);
    if (n) {
        HIPCHK(hipMemcpy(f.data(), g.dev.first_hist, n * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(t.data(), g.dev.tot_by_epoch, n * 8, hipMemcpyDeviceToHost));
    }
    unsigned long long df = 0, dt = 0;
    for (size_t i = 0; i < c->points.size(); ++i) {
        if (i < n) { df += f[i]; dt += t[i]; }
        if (i >= g.points_final) { c->points[i].distinct_kmer = df; c->points[i].total_kmer = dt; }
    }
    return 0;
}

static int kg_totals(faqcs_ctx *c, unsigned long long *distinct, unsigned long long *total)
{
    faqcs_ctx::KmerGroup &g = c->kg;
    *distinct = *total = 0;
    if (!g.ready || !g.ep_used) return 0;
    std::vector<unsigned long long> f(g.ep_used), t(g.ep_used);
    HIPCHK(hipMemcpy(f.data(), g.dev.first_hist, (size_t)g.ep_used * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t.data(), g.dev.tot_by_epoch, (size_t)g.ep_used * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < g.ep_used; ++i) { *distinct += f[i]; *total += t[i]; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// submission
// ---------------------------------------------------------------------------------------------------------
// seq/qual/off are device pointers valid for indices off[0]..off[n]; host_off is the host copy of the offsets
// (needed for the max read length and k-mer bookkeeping), d_res a device result array.
static int enqueue(faqcs_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, const uint32_t *d_off, uint32_t n,
                   uint32_t max_len, const uint32_t *seg, uint32_t n_seg, faqcs_read_result *d_res, const uint8_t *d_tn = nullptr,
                   const uint32_t *host_off = nullptr)
{
    const faqcs_params &p = c->prm;
    uint32_t *d_sl = nullptr; uint16_t *d_hit = nullptr;
    Timing *tm = nullptr;
    if (n) {
        if (c->timing_used == c->timings.size()) {
            Timing t; HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b)); HIPCHK(hipEventCreate(&t.p)); HIPCHK(hipEventCreate(&t.k0)); HIPCHK(hipEventCreate(&t.k1));
            t.adapter = false; t.kmer = false; c->timings.push_back(t);
        }
        tm = &c->timings[c->timing_used++];
        tm->adapter = p.n_adapters != 0;
        tm->kmer = false;
    }
    if (n && p.n_adapters) {
        HIPCHK(hipEventRecord(tm->p, c->compute));
        HIPCHK(c->s_sl.reserve(n)); HIPCHK(c->s_hit.reserve(n)); HIPCHK(c->s_seg.reserve(n_seg + 1));
        d_sl = c->s_sl.p; d_hit = c->s_hit.p;
        HIPCHK(hipMemcpyAsync(c->s_seg.p, seg, (n_seg + 1) * 4, hipMemcpyHostToDevice, c->compute));
        AdapterDev A{c->d_abits, c->d_astart, c->d_aplanes, c->d_awstart, p.n_adapters, c->match_rate, c->adapter_longest, c->adapter_plane_dwords};
        HIPCHK(faqcs_launch_adapter(A, d_seq, d_off, n, max_len, c->s_seg.p, n_seg, d_sl, d_hit,
                                    c->d_counters + c->lay.adapter_stats, c->d_err, c->dp.dbg, c->n_cu, c->compute));
    }
    if (n) {
        Timing &t = *tm;
        HIPCHK(hipEventRecord(t.a, c->compute));
        const bool wide = max_len > 256; // the long-read kernels write two-word composition records
        const int set = (int)(c->n_enqueued++ & 1);
        faqcs_ctx::RecSet &rs = c->rec[set];
        if (rs.used) { HIPCHK(hipStreamWaitEvent(c->compute, rs.folded, 0)); rs.used = false; } // (a fold of the set's previous records on the aux stream)
        const char *e_long = getenv("FAQCS_TRIM_LONG");
        const bool force_long = e_long && atoi(e_long) != 0;
        const bool long_reads = max_len > FAQCS_FAST_READ_LENGTH || force_long; // trim_long: composition bins are added by the kernel itself, no records
        const size_t need = long_reads ? (size_t)n / 2 + 1 : (size_t)n * (wide ? 2 : 1); // (trim_long: rec_pre is its scratch, one u32 per read)
        if (need > rs.pre.cap) { HIPCHK(hipStreamSynchronize(c->aux)); HIPCHK(hipStreamSynchronize(c->compute)); } // (the set's old records may still be read: by the fold kernel, or by the launch before this one)
        HIPCHK(rs.pre.reserve(need)); HIPCHK(rs.post.reserve(need));
        // the records of the launch BEFORE this one: this launch's blocks fold them when they run out of reads, if its kernel can (trim_lds with
        // a block of at least 122 KB of LDS: the 2x100 ... 2x150 variants); FAQCS_TAIL_FOLD=0: never (A/B)
        DevParams dp = c->dp;
        static const bool tail_on = [] { const char *e = getenv("FAQCS_TAIL_FOLD"); return !e || atoi(e) != 0; }();
        if (c->pending_fold >= 0 && !c->pending_wide && tail_on) {
            const faqcs_ctx::RecSet &ps = c->rec[c->pending_fold];
            if (!c->d_fold_claim) HIPCHK(hipMalloc((void **)&c->d_fold_claim, 8));
            HIPCHK(hipMemsetAsync(c->d_fold_claim, 0, 8, c->compute));
            dp.fold_pre = ps.pre.p; dp.fold_post = ps.post.p; dp.fold_n = c->pending_n; dp.fold_claim = c->d_fold_claim;
            dp.fold_dst_pre = c->d_counters + c->lay.pre_comp; dp.fold_dst_post = c->d_counters + c->lay.post_comp;
        }
        HIPCHK(faqcs_launch_trim(dp, d_seq, d_qual, d_off, n, max_len, d_sl, d_hit, d_res, rs.pre.p, rs.post.p,
                                 c->d_counters, c->d_err, c->n_cu, c->compute, d_tn));
        HIPCHK(hipEventRecord(t.b, c->compute));
        c->trim_kernel = faqcs_last_trim_kernel();
        if (c->pending_fold >= 0) { // ... else composition_histogram folds them on the aux stream, beside this launch
            if (!(dp.fold_n && faqcs_last_trim_folded())) {
                faqcs_ctx::RecSet &ps = c->rec[c->pending_fold];
                HIPCHK(hipStreamWaitEvent(c->aux, ps.trimmed, 0));
                HIPCHK(faqcs_launch_composition(ps.pre.p, ps.post.p, c->pending_n, c->pending_wide, c->d_norm, c->d_counters + c->lay.pre_comp,
                                                c->d_counters + c->lay.post_comp, c->n_cu, c->aux));
                HIPCHK(hipEventRecord(ps.folded, c->aux));
                ps.used = true;
            }
            c->pending_fold = -1;
        }
        static const bool no_comp = [] { const char *e = getenv("FAQCS_DIAG_NO_COMPOSITION"); return e && atoi(e) != 0; }(); // (diagnostic, wrong composition tables: what the fold costs a step)
        if (!(c->dp.dbg & 1u) && !long_reads && !no_comp) {
            HIPCHK(hipEventRecord(rs.trimmed, c->compute));
            c->pending_fold = set; c->pending_n = n; c->pending_wide = wide;
        }
    }
    // ---- owner-partitioned k-mer mode: this shard's runs of k-mers (super-k-mer items) grouped by owner rank; the caller exchanges them
    if (c->partitioned && !c->kg.direct) {
        if (!c->kmer_active || n == 0) { c->seg_epoch.clear(); return 0; }
        if (c->seg_epoch.size() != n_seg) return fail(FAQCS_E_INVAL, "faqcs_submit: faqcs_kmer_set_epochs() must give one epoch per segment of the submission");
        faqcs_ctx::KmerSend &ks = c->ks;
        const uint32_t w = p.kmer > 15 ? p.kmer - 14 : 1;
        const uint64_t occ = host_off ? (uint64_t)(host_off[n] - host_off[0]) : (uint64_t)n * (max_len >= p.kmer ? max_len - p.kmer + 1 : 1);
        const uint64_t ib = (w == 1 ? occ : occ * 3 / (w + 1)) + 2ull * n + 64; // items this submission can be expected to make at most
        // staging: a sub-region per (bucket, writing block) of twice its even share; whatever does not fit spills (room for all of it)
        const uint32_t cap1 = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(KG_MIN_CAP, 2 * ib / ((uint64_t)KG_FAN * KG_FAN) + 64), 0x7fffffffu);
        const uint32_t spill_cap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(2 * ib, occ + 64), 0x7fffffffu); // (the true bound is one run per occurrence)
        if ((size_t)KG_FAN * KG_FAN * cap1 > ks.l1.cap || spill_cap > ks.spill.cap) HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(ks.l1.reserve((size_t)KG_FAN * KG_FAN * cap1));
        HIPCHK(ks.spill.reserve(spill_cap));
        HIPCHK(ks.cur1.reserve((size_t)KG_FAN * KG_FAN + 64));
        HIPCHK(ks.run_epoch.reserve((size_t)KG_MAX_RUNS));
        HIPCHK(ks.scratch.reserve((size_t)KG_FAN * KG_FAN + 128));
        HIPCHK(ks.defer.reserve((size_t)n + 1));
        ks.spill_n = ks.cur1.p + (size_t)KG_FAN * KG_FAN;
        KmerGroupDev &d = ks.dev;
        d = KmerGroupDev{};
        d.l1 = reinterpret_cast<unsigned long long *>(ks.l1.p); d.cur1 = ks.cur1.p; d.run_epoch = ks.run_epoch.p; d.cap1 = cap1; d.split = 1;
        d.spill = reinterpret_cast<unsigned long long *>(ks.spill.p); d.spill_n = ks.spill_n; d.spill_cap = spill_cap;
        HIPCHK(hipMemsetAsync(ks.cur1.p, 0, ((size_t)KG_FAN * KG_FAN + 64) * 4, c->compute));
        if (tm) { HIPCHK(hipEventRecord(tm->k0, c->compute)); tm->kmer = true; }
        // one launch per run of segments with the same epoch; an item's run field (13 bits) indexes the submission's run -> epoch table
        std::vector<uint32_t> &run_epoch = c->kg.upload[c->kg.n_flushes++ & 1];
        run_epoch.clear();
        static const bool no16 = [] { const char *e = getenv("FAQCS_KMER_EXTRACT16"); return e && atoi(e) == 0; }();
        for (uint32_t s = 0; s < n_seg;) {
            uint32_t e = s + 1;
            while (e < n_seg && c->seg_epoch[e] == c->seg_epoch[s]) ++e;
            if (c->seg_epoch[s] != 0xffffffffu && seg[e] > seg[s]) {
                if (run_epoch.size() >= (size_t)KG_MAX_RUNS) return fail(FAQCS_E_INVAL, "faqcs_submit: more than 1000 runs of segments with one epoch in a submission");
                const uint32_t run = (uint32_t)run_epoch.size(), rot = (run * 37u) % KG_FAN;
                if (p.kmer == 31 && max_len <= 256 && !no16) {
                    HIPCHK(hipMemsetAsync(ks.defer.p, 0, 4, c->compute));
                    HIPCHK(faqcs_launch_skm_extract16(c->dp, d, c->kt, run, rot, c->seg_epoch[s], d_seq, d_qual, d_off, seg[s], seg[e], d_res,
                                                      ks.defer.p + 1, ks.defer.p, c->n_cu, c->compute));
                    HIPCHK(faqcs_launch_skm_extract(c->dp, p.kmer, d, c->kt, run, rot, c->seg_epoch[s], d_seq, d_qual, d_off, seg[s], seg[e], d_res,
                                                    c->n_cu, c->compute, ks.defer.p + 1, ks.defer.p, faqcs_skm_grid16(seg[e] - seg[s], c->n_cu)));
                } else
                    HIPCHK(faqcs_launch_skm_extract(c->dp, p.kmer, d, c->kt, run, rot, c->seg_epoch[s], d_seq, d_qual, d_off, seg[s], seg[e], d_res,
                                                    c->n_cu, c->compute));
                run_epoch.push_back(c->seg_epoch[s]);
            }
            s = e;
        }
        if (!run_epoch.empty()) HIPCHK(hipMemcpyAsync(ks.run_epoch.p, run_epoch.data(), run_epoch.size() * 4, hipMemcpyHostToDevice, c->compute));
        d.n_runs = (uint32_t)run_epoch.size(); d.epoch_base = 0;
        if ((size_t)std::min<uint64_t>((uint64_t)KG_FAN * KG_FAN * cap1 + spill_cap, occ + 1) > c->ob_items.cap) HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(c->ob_items.reserve((size_t)std::min<uint64_t>((uint64_t)KG_FAN * KG_FAN * cap1 + spill_cap, occ + 1)));
        HIPCHK(faqcs_launch_skm_outbox(d, c->part_world, c->d_ob, ks.scratch.p, c->ob_items.p, c->compute));
        if (tm) HIPCHK(hipEventRecord(tm->k1, c->compute));
        c->seg_epoch.clear();
        c->ob_fresh = true;
        return 0;
    }
    // (FAQCS_KMER_DIRECT=1: round 3's form -- (key, epoch) pairs bucketed by owner in two passes, one atomic insert per pair on the owner)
    if (c->partitioned) {
        if (!c->kmer_active || n == 0) { c->seg_epoch.clear(); return 0; }
        if (c->seg_epoch.size() != n_seg) return fail(FAQCS_E_INVAL, "faqcs_submit: faqcs_kmer_set_epochs() must give one epoch per segment of the submission");
        // an occurrence starts at a distinct base, so the arena size bounds the item count (host copy of the last offset
        // is not available for device-resident batches: use the per-read bound n * max_len)
        const size_t cap = (size_t)n * (size_t)(max_len > p.kmer ? max_len - p.kmer + 1 : 0) + 1;
        HIPCHK(c->ob_items.reserve(cap));
        // one launch per run of segments with the same epoch; the waves of all launches get consecutive rows of the
        // per-wave count / offset tables
        struct RunSpan { uint32_t s, e, wave_base; };
        std::vector<RunSpan> runs;
        uint32_t total_waves = 0;
        for (uint32_t s = 0; s < n_seg;) {
            uint32_t e = s + 1;
            while (e < n_seg && c->seg_epoch[e] == c->seg_epoch[s]) ++e;
            if (c->seg_epoch[s] != 0xffffffffu && seg[e] > seg[s]) {
                runs.push_back({s, e, total_waves});
                total_waves += faqcs_kmer_extract_waves(seg[e] - seg[s], c->n_cu);
            }
            s = e;
        }
        HIPCHK(c->ob_wave_count.reserve((size_t)total_waves * c->part_world + 1));
        HIPCHK(c->ob_wave_offset.reserve((size_t)total_waves * c->part_world + 1));
        KmerOutbox O{c->ob_items.p, c->d_ob, c->d_ob + c->part_world, c->d_ob + 2 * c->part_world, c->part_world,
                     c->ob_wave_count.p, c->ob_wave_offset.p, total_waves};
        HIPCHK(hipMemsetAsync(c->d_ob, 0, 3 * c->part_world * 8, c->compute));
        if (tm) { HIPCHK(hipEventRecord(tm->k0, c->compute)); tm->kmer = true; }
        for (int fill = 0; fill < 2; ++fill) {
            for (const RunSpan &r : runs)
                HIPCHK(faqcs_launch_kmer_extract(c->dp, p.kmer, O, fill != 0, d_seq, d_qual, d_off, seg[r.s], seg[r.e], d_res,
                                                 c->seg_epoch[r.s], r.wave_base, c->n_cu, c->compute));
            if (!fill) HIPCHK(faqcs_launch_kmer_outbox_offsets(O, c->compute));
        }
        if (tm) HIPCHK(hipEventRecord(tm->k1, c->compute));
        c->seg_epoch.clear();
        c->ob_fresh = true;
        return 0;
    }
    // ---- rarefaction bookkeeping per reference trim() call (trim.cpp:157-185) -------------------------------
    // The k-mers of consecutive segments go to the device in ONE launch per run of segments that ends at a sampling point
    // (or at the end of the batch / of the curve): only there does the order of insertion become observable.
    uint32_t run_begin = seg[0];
    if (tm && c->kmer_active) { HIPCHK(hipEventRecord(tm->k0, c->compute)); tm->kmer = true; }
    const bool direct = c->kg.direct;
    const uint64_t per_read = max_len >= p.kmer ? (uint64_t)(max_len - p.kmer + 1) : 1;
    auto flush_run = [&](uint32_t run_end) -> int {
        if (run_end > run_begin) {
            if (direct) HIPCHK(faqcs_launch_kmer(c->dp, p.kmer, c->kt, d_seq, d_qual, d_off, run_begin, run_end, d_res, c->n_cu, c->compute));
            // the run's epoch: the index of the next sampling point (the first one that will include these occurrences)
            else if (int rc = kg_add_run(c, d_seq, d_qual, d_off, run_begin, run_end, d_res, per_read, host_off, (uint32_t)c->points.size(), max_len)) return rc;
        }
        run_begin = run_end;
        return 0;
    };
    for (uint32_t s = 0; s < n_seg; ++s) {
        const uint32_t r0 = seg[s], r1 = seg[s + 1];
        if (!c->kmer_active) run_begin = r1; // (segments after the curve completed are not counted, trim.cpp:180-184)
        c->total_number += (r1 - r0);
        if (c->kmer_active) {
            const uint64_t index = c->total_number / p.split_size;
            const size_t num_rarefaction = c->points.size();
            if (s + 1 == n_seg || num_rarefaction >= p.num_subsample ||
                (index > num_rarefaction && num_rarefaction < p.num_subsample))
                if (int rc = flush_run(r1)) return rc;
            if (index > num_rarefaction && num_rarefaction < p.num_subsample) {
                faqcs_rarefaction pt{c->total_number, 0, 0};
                c->points.push_back(pt);
                if (direct) { // the table's running (distinct, total) in stream order
                    if (c->n_snaps == c->snap_cap) { // drain the snapshots taken so far
                        HIPCHK(hipStreamSynchronize(c->compute));
                        std::vector<unsigned long long> h(c->n_snaps * 2);
                        HIPCHK(hipMemcpy(h.data(), c->d_snaps, c->n_snaps * 16, hipMemcpyDeviceToHost));
                        for (auto &pp : c->pending) { c->points[pp.point_index].distinct_kmer = h[2 * pp.snap_index]; c->points[pp.point_index].total_kmer = h[2 * pp.snap_index + 1]; }
                        c->pending.clear(); c->n_snaps = 0;
                    }
                    HIPCHK(hipMemcpyAsync(c->d_snaps + 2 * c->n_snaps, c->kt.stats, 16, hipMemcpyDeviceToDevice, c->compute));
                    c->pending.push_back({c->points.size() - 1, c->n_snaps});
                    ++c->n_snaps;
                }
            }
            if (num_rarefaction >= p.num_subsample) c->kmer_active = 0; // trim.cpp:180-184
        }
    }
    if (tm && tm->kmer) HIPCHK(hipEventRecord(tm->k1, c->compute));
    return 0;
}

static int scan_offsets(const uint32_t *off, uint32_t n, uint32_t cap, uint32_t *max_len)
{
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) return fail(FAQCS_E_INVAL, "faqcs_submit: offsets must be non-decreasing");
        const uint32_t l = off[i + 1] - off[i];
        m = l > m ? l : m;
    }
    if (m > cap) return fail(FAQCS_E_INVAL, "faqcs_submit: read longer than max_read_length");
    if (m > FAQCS_MAX_READ_LENGTH) return fail(FAQCS_E_INVAL, "faqcs_submit: reads longer than FAQCS_MAX_READ_LENGTH bases are not supported by the HIP kernels");
    *max_len = m;
    return 0;
}

static int check_segments(const faqcs_batch *b)
{
    if (!b || !b->offset) return fail(FAQCS_E_INVAL, "faqcs_submit: null batch");
    if (!b->segment_start || b->segment_start[0] != 0 || b->segment_start[b->n_segments] != b->n_reads)
        return fail(FAQCS_E_INVAL, "faqcs_submit: segment_start must span [0, n_reads]");
    for (uint32_t s = 0; s < b->n_segments; ++s)
        if (b->segment_start[s + 1] < b->segment_start[s]) return fail(FAQCS_E_INVAL, "faqcs_submit: segment_start must be non-decreasing");
    return 0;
}

extern "C" int faqcs_submit_async(faqcs_ctx *c, const faqcs_batch *b, faqcs_read_result *results, uint64_t *ticket)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (int rc = check_segments(b)) return rc;
    HIPCHK(hipSetDevice(c->device));
    const uint32_t n = b->n_reads;
    uint32_t max_len = 0;
    if (int rc = scan_offsets(b->offset, n, c->prm.max_read_length, &max_len)) return rc;
    const uint32_t o0 = b->offset[0], o1 = b->offset[n];
    const size_t bytes = (size_t)(o1 - o0);
    faqcs_ctx::Slot &sl = c->slot[c->n_submits & 1];
    hipEvent_t tk = c->ticket_ev[c->n_submits & 7];
    if (c->n_submits >= 8) HIPCHK(hipEventSynchronize(tk)); // the ticket ring is 8 deep
    // the slot may still feed the kernels of submission k-2: growing it (hipFree) needs them finished, reusing it
    // only needs the copy stream to wait for them
    if (sl.used) {
        if (bytes + FAQCS_ARENA_PAD_BEFORE + FAQCS_ARENA_PAD_AFTER > sl.seq.cap || (size_t)n + 1 > sl.off.cap || (b->terminal_n && (size_t)n + 64 > sl.tn.cap)) HIPCHK(hipEventSynchronize(sl.done));
        else HIPCHK(hipStreamWaitEvent(c->copy, sl.done, 0));
    }
    HIPCHK(sl.seq.reserve(bytes + FAQCS_ARENA_PAD_BEFORE + FAQCS_ARENA_PAD_AFTER)); HIPCHK(sl.qual.reserve(bytes + FAQCS_ARENA_PAD_BEFORE + FAQCS_ARENA_PAD_AFTER)); HIPCHK(sl.off.reserve((size_t)n + 1));
    if (b->terminal_n) HIPCHK(sl.tn.reserve((size_t)n + 64));
    if ((size_t)n + 1 > c->s_res.cap) HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(c->s_res.reserve((size_t)n + 1));
    // arena bytes land 16 bytes into the staging buffer; the kernels index with the ORIGINAL offsets
    if (bytes) {
        HIPCHK(hipMemcpyAsync(sl.seq.p + 16, b->seq + o0, bytes, hipMemcpyHostToDevice, c->copy));
        HIPCHK(hipMemcpyAsync(sl.qual.p + 16, b->qual + o0, bytes, hipMemcpyHostToDevice, c->copy));
    }
    HIPCHK(hipMemcpyAsync(sl.off.p, b->offset, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->copy));
    const uint8_t *d_tn = nullptr;
    if (b->terminal_n && n) { // the caller's parser has looked at the ends of each read: n more bytes instead of two scattered loads per read
        HIPCHK(hipMemcpyAsync(sl.tn.p, b->terminal_n, n, hipMemcpyHostToDevice, c->copy));
        d_tn = sl.tn.p;
    }
    HIPCHK(hipEventRecord(c->copied, c->copy));
    HIPCHK(hipStreamWaitEvent(c->compute, c->copied, 0));
    const uint8_t *d_seq = sl.seq.p + 16 - o0, *d_qual = sl.qual.p + 16 - o0;
    if (int rc = enqueue(c, d_seq, d_qual, sl.off.p, n, max_len, b->segment_start, b->n_segments, c->s_res.p, d_tn, b->offset)) return rc;
    if (n && results) HIPCHK(hipMemcpyAsync(results, c->s_res.p, (size_t)n * sizeof(faqcs_read_result), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipEventRecord(sl.done, c->compute));
    HIPCHK(hipEventRecord(tk, c->compute));
    sl.used = true;
    if (ticket) *ticket = c->n_submits;
    ++c->n_submits;
    return 0;
}

extern "C" int faqcs_submit(faqcs_ctx *c, const faqcs_batch *b, faqcs_read_result *results)
{
    return faqcs_submit_async(c, b, results, nullptr);
}

extern "C" int faqcs_wait(faqcs_ctx *c, uint64_t ticket)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (ticket >= c->n_submits) return fail(FAQCS_E_INVAL, "faqcs_wait: unknown ticket");
    if (ticket + 8 < c->n_submits) return 0; // its event slot has been recycled: the submission finished long ago
    HIPCHK(hipEventSynchronize(c->ticket_ev[ticket & 7]));
    return 0;
}

extern "C" void *faqcs_host_alloc(size_t bytes)
{
    void *p = nullptr;
    // portable: a buffer is handed to whichever device context takes the next 32 768-read block (faqcs_mi --gpus N)
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}

extern "C" void faqcs_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int faqcs_submit_device(faqcs_ctx *c, const faqcs_batch *b, faqcs_read_result *d_results)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (!b || !b->segment_start) return fail(FAQCS_E_INVAL, "faqcs_submit_device: null batch");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t n = b->n_reads;
    // offsets live on the device: the caller vouches for max_read_length; pick the kernel from the ctx capacity
    uint32_t max_len = b->max_read_len ? b->max_read_len : c->prm.max_read_length;
    if (max_len > c->prm.max_read_length) return fail(FAQCS_E_INVAL, "faqcs_submit_device: max_read_len exceeds the context capacity");
    if (max_len > FAQCS_MAX_READ_LENGTH) return fail(FAQCS_E_INVAL, "faqcs_submit_device: reads longer than FAQCS_MAX_READ_LENGTH bases are not supported by the HIP kernels");
    if (!d_results) { HIPCHK(c->s_res.reserve((size_t)n + 1)); d_results = c->s_res.p; }
    return enqueue(c, b->seq, b->qual, b->offset, n, max_len, b->segment_start, b->n_segments, d_results, b->terminal_n);
}

extern "C" int faqcs_terminal_n_flags(int device_id, const uint8_t *d_seq, const uint32_t *d_offset, uint32_t n_reads, uint8_t *d_flags)
{
    if (!d_seq || !d_offset || !d_flags) return fail(FAQCS_E_INVAL, "null argument");
    if (device_id >= 0) HIPCHK(hipSetDevice(device_id));
    HIPCHK(faqcs_launch_terminal_n_flags(d_seq, d_offset, n_reads, d_flags, nullptr));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

static int resolve_points(faqcs_ctx *c)
{
    if (c->pending.empty()) return 0;
    std::vector<unsigned long long> h(c->n_snaps * 2);
    HIPCHK(hipMemcpy(h.data(), c->d_snaps, c->n_snaps * 16, hipMemcpyDeviceToHost));
    for (auto &pp : c->pending) {
        c->points[pp.point_index].distinct_kmer = h[2 * pp.snap_index];
        c->points[pp.point_index].total_kmer = h[2 * pp.snap_index + 1];
    }
    c->pending.clear();
    c->n_snaps = 0;
    return 0;
}

// the records of the last launch are folded now, on the compute stream (somebody is about to read, move or reset the counter block)
static int fold_pending_now(faqcs_ctx *c)
{
    if (c->pending_fold < 0) return 0;
    faqcs_ctx::RecSet &ps = c->rec[c->pending_fold];
    HIPCHK(faqcs_launch_composition(ps.pre.p, ps.post.p, c->pending_n, c->pending_wide, c->d_norm, c->d_counters + c->lay.pre_comp,
                                    c->d_counters + c->lay.post_comp, c->n_cu, c->compute));
    c->pending_fold = -1;
    return 0;
}

extern "C" int faqcs_sync(faqcs_ctx *c)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->copy));
    if (int rc = fold_pending_now(c)) return rc;
    // (round 6: the k-mers waiting in the open group stay there -- they are counted when the pass ends, faqcs_kmer_end_table /
    // faqcs_kmer_finish_pass, or when a caller asks for the curve so far, faqcs_kmer_points / _totals / _epoch_counts: kmer_catch_up)
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipStreamSynchronize(c->aux));
    for (size_t i = 0; i < c->kg.flush_ev_used; ++i) { // group flushes (they run behind a later submission or here): part of the k-mer time
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->kg.flush_ev[i].first, c->kg.flush_ev[i].second) == hipSuccess) c->kmer_flush_ms += ms;
    }
    c->kg.flush_ev_used = 0;
    for (size_t i = 0; i < c->timing_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->timings[i].a, c->timings[i].b) == hipSuccess) { c->kernel_ms += ms; ++c->kernel_launches; }
        if (c->timings[i].kmer && hipEventElapsedTime(&ms, c->timings[i].k0, c->timings[i].k1) == hipSuccess) c->kmer_ms += ms;
        if (c->timings[i].adapter && hipEventElapsedTime(&ms, c->timings[i].p, c->timings[i].a) == hipSuccess) c->adapter_ms += ms;
    }
    c->timing_used = 0;
    if (int rc = resolve_points(c)) return rc;
    uint32_t e = 0;
    HIPCHK(hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
    if (e & 1u) return fail(FAQCS_E_QUALITY, "fastq.h:quality_score: Found a quality score value that is greater than the maximum allowed quality score");
    if (e & 2u) return fail(FAQCS_E_BASE, "seq_overlap.cpp:na_to_bits: Unknown base!");
    if (e & 4u) return fail(FAQCS_E_INVAL, "faqcs: internal error, the trim kernel's LDS block does not start at address 0");
    if (c->kt.stats) {
        unsigned long long st[3];
        HIPCHK(hipMemcpy(st, c->kt.stats, 24, hipMemcpyDeviceToHost));
        // (bit 1: the sender staging of the multi-GPU exchange ran out of room for runs that did not fit their sub-regions -- an input whose minimizer changes
        // at almost every position makes more runs than the staging is sized for; not a full table: ADVICE r5)
        if (st[2] & 2ull) return fail(FAQCS_E_NOMEM, "faqcs: the k-mer exchange staging overflowed (more runs of k-mers per read than it is sized for): submit smaller batches");
        if (st[2]) return fail(FAQCS_E_KMER_FULL, "faqcs: device k-mer table is full (raise faqcs_params.kmer_table_slots)");
    }
    return 0;
}

extern "C" int faqcs_counters_device(faqcs_ctx *c, void **d_ptr, uint64_t *n_u64)
{
    if (!c || !d_ptr || !n_u64) return fail(FAQCS_E_INVAL, "null argument");
    *d_ptr = c->d_counters;
    *n_u64 = c->lay.total;
    return 0;
}

// The collective runs on a buffer the CALLER owns (e.g. a torch tensor that RCCL registers for IPC): export the block into it,
// all-reduce it, import the sum.  Both copies wait for the work submitted so far and return when the bytes have moved.
extern "C" int faqcs_counters_export(faqcs_ctx *c, void *d_dst, uint64_t n_u64)
{
    if (!c || !d_dst || n_u64 < c->lay.total) return fail(FAQCS_E_INVAL, "faqcs_counters_export: buffer too small");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = fold_pending_now(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->aux));
    HIPCHK(hipMemcpyAsync(d_dst, c->d_counters, c->lay.total * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

extern "C" int faqcs_counters_import(faqcs_ctx *c, const void *d_src, uint64_t n_u64)
{
    if (!c || !d_src || n_u64 < c->lay.total) return fail(FAQCS_E_INVAL, "faqcs_counters_import: buffer too small");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = fold_pending_now(c)) return rc; // (what is still to be folded belongs to the block that is being replaced... by its own sum: export came first)
    HIPCHK(hipStreamSynchronize(c->aux));
    HIPCHK(hipMemcpyAsync(c->d_counters, d_src, c->lay.total * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// native collective: the counter block all-reduced in place by RCCL (loaded at run time; see include/faqcs_mi.h)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct RcclId { char internal[FAQCS_COMM_ID_BYTES]; }; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES == 128), passed by value
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclUint64 = 5, kNcclSum = 0; // ncclDataType_t / ncclRedOp_t of rccl.h (ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)
Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char *names[] = {getenv("FAQCS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { if (n && *n && (x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
        if (!x.lib) return x;
        auto sym = [&](const char *n) { return dlsym(x.lib, n); };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommInitAll && x.CommDestroy && x.AllReduce && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}
int rccl_fail(const char *what, int rc)
{
    Rccl &r = rccl();
    std::string m = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
    return fail(FAQCS_E_NODEVICE, m.c_str());
}
#define RCCLCHK(call, what) do { const int rc_ = (call); if (rc_ != 0) return rccl_fail(what, rc_); } while (0)
// the collective of one context, enqueued on its compute stream behind everything the context has submitted (both streams)
int comm_enqueue(faqcs_ctx *c)
{
    Rccl &r = rccl();
    HIPCHK(hipSetDevice(c->device));
    if (int rc = fold_pending_now(c)) return rc;
    if (!c->comm_ev) HIPCHK(hipEventCreateWithFlags(&c->comm_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->comm_ev, c->aux));             // (the composition fold adds to the block on the aux stream)
    HIPCHK(hipStreamWaitEvent(c->compute, c->comm_ev, 0));
    RCCLCHK(r.AllReduce(c->d_counters, c->d_counters, (size_t)c->lay.total, kNcclUint64, kNcclSum, c->comm, c->compute), "ncclAllReduce");
    return 0;
}
} // namespace

static void comm_release(void *comm) { if (rccl().CommDestroy) (void)rccl().CommDestroy(comm); }

extern "C" int faqcs_comm_id(void *id)
{
    if (!id) return fail(FAQCS_E_INVAL, "faqcs_comm_id: null id");
    Rccl &r = rccl();
    if (!r.ok) return fail(FAQCS_E_NODEVICE, "faqcs_comm_id: librccl.so could not be loaded (FAQCS_RCCL_LIB names another file)");
    RcclId u;
    RCCLCHK(r.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, FAQCS_COMM_ID_BYTES);
    return 0;
}

extern "C" int faqcs_comm_init(faqcs_ctx *c, const void *id, uint32_t rank, uint32_t world)
{
    if (!c || !id || world == 0 || rank >= world) return fail(FAQCS_E_INVAL, "faqcs_comm_init: bad ctx / id / rank / world");
    if (c->comm) return fail(FAQCS_E_INVAL, "faqcs_comm_init: the context already has a communicator");
    Rccl &r = rccl();
    if (!r.ok) return fail(FAQCS_E_NODEVICE, "faqcs_comm_init: librccl.so could not be loaded (FAQCS_RCCL_LIB names another file)");
    HIPCHK(hipSetDevice(c->device));
    RcclId u;
    memcpy(u.internal, id, FAQCS_COMM_ID_BYTES);
    RCCLCHK(r.CommInitRank(&c->comm, (int)world, u, (int)rank), "ncclCommInitRank");
    return 0;
}

extern "C" int faqcs_comm_allreduce_counters(faqcs_ctx *c)
{
    if (!c || !c->comm) return fail(FAQCS_E_INVAL, "faqcs_comm_allreduce_counters: faqcs_comm_init() first");
    return comm_enqueue(c);
}

extern "C" int faqcs_comm_init_all(faqcs_ctx *const *ctxs, uint32_t n)
{
    if (!ctxs || n == 0 || n > 64) return fail(FAQCS_E_INVAL, "faqcs_comm_init_all: bad context list");
    Rccl &r = rccl();
    if (!r.ok) return fail(FAQCS_E_NODEVICE, "faqcs_comm_init_all: librccl.so could not be loaded (FAQCS_RCCL_LIB names another file)");
    std::vector<int> dev(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->comm) return fail(FAQCS_E_INVAL, "faqcs_comm_init_all: null context or one that has a communicator");
        if (ctxs[i]->lay.total != ctxs[0]->lay.total) return fail(FAQCS_E_INVAL, "faqcs_comm_init_all: the contexts' counter blocks differ in size");
        dev[i] = ctxs[i]->device;
        for (uint32_t j = 0; j < i; ++j) if (dev[j] == dev[i]) return fail(FAQCS_E_INVAL, "faqcs_comm_init_all: two contexts on one device (RCCL wants one rank per device)");
    }
    std::vector<void *> comms(n, nullptr);
    RCCLCHK(r.CommInitAll(comms.data(), (int)n, dev.data()), "ncclCommInitAll");
    for (uint32_t i = 0; i < n; ++i) ctxs[i]->comm = comms[i];
    return 0;
}

extern "C" int faqcs_comm_allreduce_counters_all(faqcs_ctx *const *ctxs, uint32_t n)
{
    if (!ctxs || n == 0) return fail(FAQCS_E_INVAL, "faqcs_comm_allreduce_counters_all: bad context list");
    for (uint32_t i = 0; i < n; ++i) if (!ctxs[i] || !ctxs[i]->comm) return fail(FAQCS_E_INVAL, "faqcs_comm_allreduce_counters_all: faqcs_comm_init_all() first");
    Rccl &r = rccl();
    RCCLCHK(r.GroupStart(), "ncclGroupStart");
    int rc = 0;
    for (uint32_t i = 0; i < n && rc == 0; ++i) rc = comm_enqueue(ctxs[i]);
    const int ge = r.GroupEnd();
    if (rc) return rc;
    RCCLCHK(ge, "ncclGroupEnd");
    return 0;
}

extern "C" int faqcs_finish(faqcs_ctx *c, uint64_t *counters, uint64_t n_u64)
{
    if (!c || !counters || n_u64 < c->lay.total) return fail(FAQCS_E_INVAL, "faqcs_finish: counter buffer too small");
    if (int rc = faqcs_sync(c)) return rc;
    HIPCHK(hipMemcpy(counters, c->d_counters, c->lay.total * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int faqcs_reset_counters(faqcs_ctx *c)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    // (records that are still to be folded belong to the job that ends here: they are dropped with its block -- a caller reads the block first)
    c->pending_fold = -1;
    HIPCHK(hipStreamSynchronize(c->aux));
    HIPCHK(hipMemsetAsync(c->d_counters, 0, c->lay.total * sizeof(uint64_t), c->compute));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// k-mer rarefaction results
// ---------------------------------------------------------------------------------------------------------
extern "C" int faqcs_kmer_active(faqcs_ctx *c) { return c ? c->kmer_active : 0; }

// The curve so far: the open group is counted INTO THE TABLE (the pass goes on behind this call, so its keys have to live somewhere) and
// the points taken so far get their values.  Callers that only want the finished curve call faqcs_kmer_end_table() first -- the pass
// is then counted in one piece without the table -- and read the points afterwards.
static int kmer_catch_up(faqcs_ctx *c)
{
    if (int rc = faqcs_sync(c)) return rc;
    if (c->kg.ready && !c->kg.run_epoch.empty()) {
        if (int rc = kg_flush(c, true)) return rc;
        if (int rc = faqcs_sync(c)) return rc; // (the flush's time joins the k-mer time; the table-full flag)
    }
    return kg_resolve_points(c);
}

extern "C" int faqcs_kmer_points(faqcs_ctx *c, faqcs_rarefaction *out, uint32_t cap, uint32_t *n_points)
{
    if (!c || !n_points) return fail(FAQCS_E_INVAL, "null argument");
    if (int rc = kmer_catch_up(c)) return rc;
    *n_points = (uint32_t)c->points.size();
    for (uint32_t i = 0; out && i < cap && i < c->points.size(); ++i) out[i] = c->points[i];
    return 0;
}

// (distinct, total) of the pass in progress -- or, when nothing has been counted since faqcs_kmer_end_table(), of the pass that call finished
extern "C" int faqcs_kmer_totals(faqcs_ctx *c, uint64_t *distinct, uint64_t *total)
{
    if (!c || !distinct || !total) return fail(FAQCS_E_INVAL, "null argument");
    *distinct = *total = 0;
    if (!c->kt.stats) return 0;
    if (int rc = kmer_catch_up(c)) return rc;
    unsigned long long st[2];
    if ((c->partitioned && !c->kg.owner) || c->kg.direct) HIPCHK(hipMemcpy(st, c->kt.stats, 16, hipMemcpyDeviceToHost));
    else if (!c->kg.pass_used) { st[0] = c->kg.last_distinct; st[1] = c->kg.last_total; }
    else if (int rc = kg_totals(c, &st[0], &st[1])) return rc;
    *distinct = st[0]; *total = st[1];
    return 0;
}

// The pass ends: its open group is counted -- in one piece and without the table when no group of the pass has been flushed before
//  -- and the points get their final values.  Nothing can join the pass afterwards; faqcs_kmer_end_table()
// (which calls this) starts the next one.  Callers that read results other than the points before faqcs_kmer_end_table() -- the epoch
// histograms of an owner rank, faqcs_kmer_epoch_counts -- call it themselves.
extern "C" int faqcs_kmer_finish_pass(faqcs_ctx *c)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (!c->kt.stats) return 0;
    if (int rc = faqcs_sync(c)) return rc;
    faqcs_ctx::KmerGroup &g = c->kg;
    if (g.ready && !g.pass_done) {
        if (int rc = kg_flush(c, true, true)) return rc;
        g.pass_done = true;
        if (int rc = faqcs_sync(c)) return rc;
        // keys in the overflow area (a probe window of a slice was full) are not swept by the counting kernel
        unsigned long long st[4];
        HIPCHK(hipMemcpy(st, c->kt.stats, 32, hipMemcpyDeviceToHost));
        g.hist_in_table = g.table_live;
        g.hist_in_overflow = !g.table_live && st[3] != 0;
        const char *e_stats = getenv("FAQCS_KMER_STATS"); // (read at every pass: a test turns it on for one engine)
        if (e_stats && atoi(e_stats) != 0) { // (diagnostics: how the pass was counted)
            uint32_t n_redo = 0;
            if (g.dev.n_redo && !g.table_live) HIPCHK(hipMemcpy(&n_redo, g.dev.n_redo, 4, hipMemcpyDeviceToHost));
            fprintf(stderr, "[kmer stats] pass counted %s; %u of %u fine partitions through their table slices; %llu inserts into the overflow area; group bound %llu occurrences, cap1 %u cap2f %u\n",
                    g.table_live ? "through the table (a group was flushed before the pass ended)" : "in one piece", n_redo, 1u << (16 + c->kt.fine), st[3],
                    (unsigned long long)g.cap_items, g.dev.cap1, g.dev.cap2f);
        }
    }
    return kg_resolve_points(c);
}

extern "C" int faqcs_kmer_end_table(faqcs_ctx *c)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (!c->kt.stats) return 0;
    if (int rc = faqcs_kmer_finish_pass(c)) return rc;
    faqcs_ctx::KmerGroup &g = c->kg;
    const bool table_only = (c->partitioned && !g.owner) || g.direct; // (round 3's per-occurrence paths: everything is in the table)
    unsigned long long st[2] = {0, 0};
    if (table_only) HIPCHK(hipMemcpy(st, c->kt.stats, 16, hipMemcpyDeviceToHost));
    else if (int rc = kg_totals(c, &st[0], &st[1])) return rc;
    // ++kmer_frequency_histogram[count] for every key, FaQCs.cpp:518-521: what the counting kernel of a pass in one piece has added up
    // already (KmerGroupDev::dense / big), plus -- when keys of the pass live in the table -- a read-only pass over it
    const bool sweep = st[0] != 0 && (table_only || g.hist_in_table || g.hist_in_overflow);
    const bool ovf_only = !table_only && !g.hist_in_table; // (the slices are empty: only the area behind the table holds keys)
    if (st[0]) {
        const uint32_t DENSE = 1u << 16, BIGCAP = 1u << 20;
        unsigned long long *d_dense = g.dev.dense, *d_big = g.dev.big, *d_nbig = g.dev.n_big;
        const bool own = !d_dense; // (a context that never made a group: the per-occurrence paths)
        if (own) {
            HIPCHK(hipMalloc((void **)&d_dense, DENSE * 8)); HIPCHK(hipMalloc((void **)&d_big, (size_t)BIGCAP * 8)); HIPCHK(hipMalloc((void **)&d_nbig, 8));
            HIPCHK(hipMemsetAsync(d_dense, 0, DENSE * 8, c->compute)); HIPCHK(hipMemsetAsync(d_nbig, 0, 8, c->compute));
        }
        auto release = [&]() { if (own) { (void)hipFree(d_dense); (void)hipFree(d_big); (void)hipFree(d_nbig); } };
        // (a read-only pass, then kmer_table_init below: 13.5 ms on the bench's 2^31-slot table; one pass that also cleared the live
        // sectors -- scattered 64-byte stores between the reads -- took 17.5)
        if (sweep) HIPCHK(faqcs_launch_kmer_histogram(c->kt, d_dense, DENSE, d_big, d_nbig, BIGCAP, c->n_cu, c->compute, ovf_only));
        HIPCHK(hipStreamSynchronize(c->compute));
        std::vector<unsigned long long> dense(DENSE);
        unsigned long long nbig = 0;
        HIPCHK(hipMemcpy(dense.data(), d_dense, DENSE * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&nbig, d_nbig, 8, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < DENSE; ++i) if (dense[i]) c->kmer_hist[i] += dense[i];
        if (nbig > BIGCAP) { release(); return fail(FAQCS_E_NOMEM, "faqcs_kmer_end_table: too many k-mers with count >= 65536"); }
        if (nbig) {
            std::vector<unsigned long long> big(nbig);
            HIPCHK(hipMemcpy(big.data(), d_big, nbig * 8, hipMemcpyDeviceToHost));
            for (auto v : big) c->kmer_hist[v] += 1;
        }
        if (!own) { HIPCHK(hipMemsetAsync(d_dense, 0, DENSE * 8, c->compute)); HIPCHK(hipMemsetAsync(d_nbig, 0, 8, c->compute)); }
        release();
    }
    if (c->partitioned) { // the points belong to the driver (faqcs_kmer_epoch_counts); only the table restarts here
        HIPCHK(hipMemsetAsync(c->d_tot_by_epoch, 0, (size_t)c->n_epochs * 8, c->compute));
    } else if (c->kmer_active && c->points.empty()) { // FaQCs.cpp:523-537
        faqcs_rarefaction pt{c->total_number, st[0], st[1]};
        c->points.push_back(pt);
    }
    // the next pass starts on an empty table: a stream of stores over all of it only when this pass has put keys there
    if (sweep || (st[0] == 0 && (table_only || g.hist_in_table || g.hist_in_overflow))) HIPCHK(faqcs_launch_kmer_table_init(c->kt, c->n_cu, c->compute, ovf_only));
    HIPCHK(hipMemsetAsync(c->kt.stats, 0, 64, c->compute));
    HIPCHK(hipMemsetAsync(c->kt.dirty, 0, (size_t)(1u << (16 + c->kt.fine)) / 8, c->compute));
    g.last_distinct = st[0]; g.last_total = st[1];
    g.table_live = false; g.pass_done = false; g.pass_used = false; g.hist_in_table = false; g.hist_in_overflow = false;
    if (g.ready && g.ep_cap) { // the epoch histograms restart with the table; the points taken so far keep their values
        g.points_final = c->points.size();
        HIPCHK(hipMemsetAsync(g.dev.first_hist, 0, (size_t)g.ep_cap * 8, c->compute));
        HIPCHK(hipMemsetAsync(g.dev.tot_by_epoch, 0, (size_t)g.ep_cap * 8, c->compute));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// owner-partitioned k-mers across GPUs
// ---------------------------------------------------------------------------------------------------------
extern "C" int faqcs_kmer_partition(faqcs_ctx *c, uint32_t rank, uint32_t world, uint32_t n_epochs)
{
    if (!c) return fail(FAQCS_E_INVAL, "null ctx");
    if (!c->kt.slots) return fail(FAQCS_E_INVAL, "faqcs_kmer_partition: the context was created without kmer_rarefaction");
    if (world == 0 || world > 64 || rank >= world || n_epochs == 0) return fail(FAQCS_E_INVAL, "faqcs_kmer_partition: bad rank / world / n_epochs");
    if (c->n_submits || c->total_number || c->partitioned) return fail(FAQCS_E_INVAL, "faqcs_kmer_partition: must be the first call on a fresh context");
    HIPCHK(hipSetDevice(c->device));
    c->kt.partitioned = 1;
    HIPCHK(hipMalloc((void **)&c->d_ob, 3 * (size_t)world * 8));
    HIPCHK(hipMalloc((void **)&c->d_tot_by_epoch, (size_t)n_epochs * 8));
    HIPCHK(hipMalloc((void **)&c->d_first_hist, (size_t)n_epochs * 8));
    HIPCHK(hipMemset(c->d_tot_by_epoch, 0, (size_t)n_epochs * 8));
    c->partitioned = true; c->part_rank = rank; c->part_world = world; c->n_epochs = n_epochs;
    // the pairs this rank receives are combined before they reach its table like a single GPU's own occurrences (a group's items carry
    // their epoch in 10 bits: up to KG_EPOCH_SPAN epochs; more, or FAQCS_KMER_DIRECT=1: one atomic insert per pair, kmer_insert_items)
    // what this rank receives -- super-k-mer items whose run field holds the absolute epoch, 13 bits -- joins the group buffers like a single
    // GPU's own runs (up to KG_EPOCH_SPAN epochs), or is counted occurrence by occurrence (more).  FAQCS_KMER_DIRECT=1: round 3's pairs.
    // More sampling epochs than an item's 10-bit epoch field and a group's LDS histogram take (KG_EPOCH_SPAN): the job goes through the
    // (key, epoch) pairs of FAQCS_KMER_DIRECT -- 16 bytes per occurrence on the wire, one atomic per pair on the owner: exact, any --subset.
    if (n_epochs > (uint32_t)KG_EPOCH_SPAN) c->kg.direct = true;
    c->kg.owner = !c->kg.direct;
    return 0;
}

extern "C" int faqcs_kmer_set_epochs(faqcs_ctx *c, const uint32_t *segment_epoch, uint32_t n_segments)
{
    if (!c || (!segment_epoch && n_segments)) return fail(FAQCS_E_INVAL, "null argument");
    if (!c->partitioned) return fail(FAQCS_E_INVAL, "faqcs_kmer_set_epochs: call faqcs_kmer_partition first");
    for (uint32_t s = 0; s < n_segments; ++s)
        if (segment_epoch[s] != 0xffffffffu && segment_epoch[s] >= c->n_epochs) return fail(FAQCS_E_INVAL, "faqcs_kmer_set_epochs: epoch out of range");
    c->seg_epoch.assign(segment_epoch, segment_epoch + n_segments);
    return 0;
}

extern "C" int faqcs_kmer_outbox(faqcs_ctx *c, void **d_items, uint64_t *counts)
{
    if (!c || !d_items || !counts) return fail(FAQCS_E_INVAL, "null argument");
    if (!c->partitioned) return fail(FAQCS_E_INVAL, "faqcs_kmer_outbox: call faqcs_kmer_partition first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->compute));
    std::vector<unsigned long long> h(c->part_world, 0ull);
    // (a second call without a submission in between -- a rank of a collective loop whose part of the input was empty -- has nothing to send:
    // the outbox of the submission before it has been handed out already)
    if (c->ob_fresh) HIPCHK(hipMemcpy(h.data(), c->d_ob, (size_t)c->part_world * 8, hipMemcpyDeviceToHost));
    c->ob_fresh = false;
    for (uint32_t d = 0; d < c->part_world; ++d) counts[d] = h[d];
    *d_items = c->ob_items.p;
    return 0;
}

// Host copy of the keys of the last submission's outbox (all destinations, in bucket order): what a single-process caller
// that keeps its own MAP<Word, size_t> (the reference's trim() seam, integration/trim_shim.cpp) merges per call.
extern "C" int faqcs_kmer_outbox_host(faqcs_ctx *c, uint64_t *keys, uint64_t cap, uint64_t *n_keys)
{
    if (!c || !n_keys) return fail(FAQCS_E_INVAL, "null argument");
    if (!c->partitioned) return fail(FAQCS_E_INVAL, "faqcs_kmer_outbox_host: call faqcs_kmer_partition first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->compute));
    std::vector<unsigned long long> h(c->part_world);
    HIPCHK(hipMemcpy(h.data(), c->d_ob, (size_t)c->part_world * 8, hipMemcpyDeviceToHost));
    uint64_t n_items = 0;
    for (uint32_t d = 0; d < c->part_world; ++d) n_items += h[d];
    // the outbox holds runs of k-mers (faqcs_skm.h): the number of occurrences is the sum of their lengths, counted on the host
    uint64_t total = n_items;
    if (!c->kg.direct && n_items) {
        std::vector<unsigned long long> w1((size_t)n_items * 2);
        HIPCHK(hipMemcpy(w1.data(), c->ob_items.p, (size_t)n_items * 16, hipMemcpyDeviceToHost));
        total = 0;
        for (uint64_t i = 0; i < n_items; ++i) total += skm_item_kmers(w1[2 * i + 1]);
    }
    *n_keys = total;
    if (!keys || cap < total || total == 0) return 0;
    std::vector<unsigned long long> items((size_t)n_items * 2);
    HIPCHK(hipMemcpy(items.data(), c->ob_items.p, (size_t)n_items * 16, hipMemcpyDeviceToHost));
    if (c->kg.direct) { for (uint64_t i = 0; i < n_items; ++i) keys[i] = items[2 * i]; return 0; }
    const SkmGeom geo = skm_geom(c->prm.kmer);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_items; ++i) {
        SkmRoll r = skm_roll_begin(items[2 * i], items[2 * i + 1], geo);
        for (uint32_t j = 0, nk = skm_item_kmers(items[2 * i + 1]); j < nk; ++j) { keys[at++] = skm_roll_key(r); skm_roll_next(r, geo); }
    }
    return 0;
}

// the received items join the owner's group buffers (or its table), enqueued on its compute stream
static int kmer_insert_enqueue(faqcs_ctx *c, const void *d_items, uint64_t n_items)
{
    if (c->kg.owner) return kg_add_items(c, d_items, n_items);
    HIPCHK(faqcs_launch_kmer_insert_items(c->kt, d_items, n_items, c->d_tot_by_epoch, c->n_epochs, c->n_cu, c->compute));
    return 0;
}

extern "C" int faqcs_kmer_insert_device(faqcs_ctx *c, const void *d_items, uint64_t n_items)
{
    if (!c || (!d_items && n_items)) return fail(FAQCS_E_INVAL, "null argument");
    if (!c->partitioned) return fail(FAQCS_E_INVAL, "faqcs_kmer_insert_device: call faqcs_kmer_partition first");
    HIPCHK(hipSetDevice(c->device));
    if (!c->ins_a) { HIPCHK(hipEventCreate(&c->ins_a)); HIPCHK(hipEventCreate(&c->ins_b)); }
    HIPCHK(hipEventRecord(c->ins_a, c->compute));
    if (int rc = kmer_insert_enqueue(c, d_items, n_items)) return rc;
    HIPCHK(hipEventRecord(c->ins_b, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute)); // the caller may recycle d_items as soon as this returns
    { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ins_a, c->ins_b) == hipSuccess) c->kmer_insert_ms += ms; }
    return 0;
}

// In-process form of the exchange (one process driving several devices: faqcs_mi --gpus N --kmer_rarefaction): the last submission's
// outbox of `from` goes to the owner contexts -- owners[r] = the context of rank r -- and is inserted there.  Same device: the
// owner reads the outbox in place; another device: a peer copy into the owner's staging buffer first.
extern "C" int faqcs_kmer_forward(faqcs_ctx *from, faqcs_ctx *const *owners, uint32_t world)
{
    if (!from || !owners) return fail(FAQCS_E_INVAL, "null argument");
    if (!from->partitioned || world != from->part_world) return fail(FAQCS_E_INVAL, "faqcs_kmer_forward: the context is not partitioned over `world` ranks");
    for (uint32_t r = 0; r < world; ++r)
        if (!owners[r] || !owners[r]->partitioned || owners[r]->part_rank != r || owners[r]->part_world != world)
            return fail(FAQCS_E_INVAL, "faqcs_kmer_forward: owners[r] must be the context of rank r");
    HIPCHK(hipSetDevice(from->device));
    HIPCHK(hipStreamSynchronize(from->compute));
    std::vector<unsigned long long> cnt(world);
    HIPCHK(hipMemcpy(cnt.data(), from->d_ob, (size_t)world * 8, hipMemcpyDeviceToHost));
    // (FAQCS_KMER_FORCE_PEER_COPY=1: an owner on the SAME device is treated like one on another device -- its items go through hipMemcpyPeer into
    // its staging buffer -- so that the branch a multi-GPU node takes is exercised on a one-GPU box: tests/test_gpu_parity.py)
    const char *e_peer = getenv("FAQCS_KMER_FORCE_PEER_COPY");
    const bool force_peer = e_peer && atoi(e_peer) != 0;
    size_t at = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const unsigned long long n = cnt[r];
        if (!n) continue;
        const ulonglong2 *src = from->ob_items.p + at;
        at += (size_t)n;
        faqcs_ctx *o = owners[r];
        if (o->device == from->device && !force_peer) { if (int rc = faqcs_kmer_insert_device(o, src, n)) return rc; continue; }
        HIPCHK(hipSetDevice(o->device));
        // Two staging buffers per owner.  The copy runs on the owner's COPY stream and waits (on the device) only for the insert that last read
        // this staging buffer; the owner's compute stream waits (on the device) for the copy before the insert; the host waits for the copy
        // alone -- the sender's outbox may be overwritten then --, not for what the owner's compute stream has queued (the forward
        // blocked on the owner's whole stream, one buffer late).
        const unsigned k = o->fwd_n++ & 1u;
        if (!o->fwd_free[k]) { HIPCHK(hipEventCreateWithFlags(&o->fwd_free[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&o->fwd_copied[k], hipEventDisableTiming)); }
        else HIPCHK(hipStreamWaitEvent(o->copy, o->fwd_free[k], 0));
        if ((size_t)n > o->fwd_items[k].cap) HIPCHK(hipEventSynchronize(o->fwd_free[k])); // (about to be reallocated: nothing may still read it)
        HIPCHK(o->fwd_items[k].reserve((size_t)n));
        HIPCHK(hipMemcpyPeerAsync(o->fwd_items[k].p, o->device, src, from->device, (size_t)n * 16, o->copy));
        HIPCHK(hipEventRecord(o->fwd_copied[k], o->copy));
        HIPCHK(hipStreamWaitEvent(o->compute, o->fwd_copied[k], 0));
        if (int rc = kmer_insert_enqueue(o, o->fwd_items[k].p, n)) return rc;
        HIPCHK(hipEventRecord(o->fwd_free[k], o->compute));
        HIPCHK(hipEventSynchronize(o->fwd_copied[k]));
        HIPCHK(hipSetDevice(from->device));
    }
    return 0;
}

extern "C" int faqcs_kmer_epoch_counts(faqcs_ctx *c, uint64_t *distinct_by_first_epoch, uint64_t *total_by_epoch, uint32_t cap)
{
    if (!c || !distinct_by_first_epoch || !total_by_epoch) return fail(FAQCS_E_INVAL, "null argument");
    if (!c->partitioned || cap < c->n_epochs) return fail(FAQCS_E_INVAL, "faqcs_kmer_epoch_counts: not partitioned / buffers too small");
    if (int rc = kmer_catch_up(c)) return rc; // (after faqcs_kmer_finish_pass: nothing is open; before it: the open group goes into the table)
    if (c->kg.owner) { // kept up to date by the combine kernel: no pass over the table
        for (uint32_t i = 0; i < c->n_epochs; ++i) { distinct_by_first_epoch[i] = 0; total_by_epoch[i] = 0; }
        if (c->kg.ready && c->kg.ep_cap) {
            HIPCHK(hipMemcpy(distinct_by_first_epoch, c->kg.dev.first_hist, (size_t)c->n_epochs * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(total_by_epoch, c->kg.dev.tot_by_epoch, (size_t)c->n_epochs * 8, hipMemcpyDeviceToHost));
        }
        return 0;
    }
    HIPCHK(hipMemsetAsync(c->d_first_hist, 0, (size_t)c->n_epochs * 8, c->compute));
    HIPCHK(faqcs_launch_kmer_first_epoch_histogram(c->kt, c->d_first_hist, c->n_epochs, c->n_cu, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(distinct_by_first_epoch, c->d_first_hist, (size_t)c->n_epochs * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(total_by_epoch, c->d_tot_by_epoch, (size_t)c->n_epochs * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int faqcs_kmer_histogram(faqcs_ctx *c, uint64_t *count, uint64_t *nkeys, uint64_t cap, uint64_t *n_pairs)
{
    if (!c || !n_pairs) return fail(FAQCS_E_INVAL, "null argument");
    *n_pairs = c->kmer_hist.size();
    uint64_t i = 0;
    for (auto &kv : c->kmer_hist) {
        if (i >= cap || !count || !nkeys) break;
        count[i] = kv.first; nkeys[i] = kv.second; ++i;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// measurement helpers
// ---------------------------------------------------------------------------------------------------------
static int synth_fill_impl(int device_id, uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads, uint32_t L,
                           uint64_t seed, uint64_t first_read, float adapter_frac, uint64_t genome_len)
{
    if (device_id >= 0) HIPCHK(hipSetDevice(device_id));
    if ((uint64_t)n_reads * L + L > 0xffffffffull) return fail(FAQCS_E_INVAL, "faqcs_synth_fill: arena exceeds 32-bit offsets");
    const char *at = getenv("FAQCS_SYNTH_AT"); // (A+T fraction of the synthetic bases; unset: uniform ACGT)
    HIPCHK(faqcs_launch_synth(d_seq, d_qual, d_offset, n_reads, L, seed, first_read, adapter_frac, genome_len, at ? (float)atof(at) : -1.0f, nullptr));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

extern "C" int faqcs_synth_fill(int device_id, uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads,
                                uint32_t L, uint64_t seed, uint64_t first_read, float adapter_frac)
{
    return synth_fill_impl(device_id, d_seq, d_qual, d_offset, n_reads, L, seed, first_read, adapter_frac, 0);
}

extern "C" int faqcs_synth_fill_genome(int device_id, uint8_t *d_seq, uint8_t *d_qual, uint32_t *d_offset, uint32_t n_reads,
                                       uint32_t L, uint64_t seed, uint64_t first_read, uint64_t genome_len)
{
    if (genome_len <= L) return fail(FAQCS_E_INVAL, "faqcs_synth_fill_genome: the genome must be longer than a read");
    return synth_fill_impl(device_id, d_seq, d_qual, d_offset, n_reads, L, seed, first_read, 0.f, genome_len);
}

// Diagnostic builds only (-DFAQCS_LDS_STAMPS): the 16 u64 words the trim kernel accumulates its section clocks in; reads and clears them.
extern "C" int faqcs_debug_words(faqcs_ctx *c, uint64_t *out, uint32_t n)
{
    if (!c || !out || n > 16) return fail(FAQCS_E_INVAL, "faqcs_debug_words: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(out, reinterpret_cast<const uint8_t *>(c->d_err) + 64, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(reinterpret_cast<uint8_t *>(c->d_err) + 64, 0, 128));
    return 0;
}

extern "C" int faqcs_kernel_time_ms(faqcs_ctx *c, double *avg_ms, uint64_t *n_launches)
{
    if (!c || !avg_ms || !n_launches) return fail(FAQCS_E_INVAL, "null argument");
    if (int rc = faqcs_sync(c)) return rc;
    *n_launches = c->kernel_launches;
    *avg_ms = c->kernel_launches ? c->kernel_ms / (double)c->kernel_launches : 0.0;
    c->kernel_ms = 0.0; c->adapter_ms = 0.0; c->kmer_ms = 0.0; c->kmer_insert_ms = 0.0; c->kmer_flush_ms = 0.0; c->kernel_launches = 0;
    return 0;
}

// Per-kernel averages since the last call of this or of faqcs_kernel_time_ms(): the trim kernel (with the name of the variant
// the last submission ran) and the adapter pre-pass (0 without adapters).  Resets the sums like faqcs_kernel_time_ms().
extern "C" int faqcs_kernel_report(faqcs_ctx *c, faqcs_kernel_times *out)
{
    if (!c || !out) return fail(FAQCS_E_INVAL, "null argument");
    if (int rc = faqcs_sync(c)) return rc;
    out->n_launches = c->kernel_launches;
    out->trim_ms = c->kernel_launches ? c->kernel_ms / (double)c->kernel_launches : 0.0;
    out->adapter_ms = c->kernel_launches ? c->adapter_ms / (double)c->kernel_launches : 0.0;
    out->trim_kernel = c->trim_kernel;
    out->kmer_ms = c->kernel_launches ? (c->kmer_ms + c->kmer_flush_ms) / (double)c->kernel_launches : 0.0;
    out->kmer_insert_ms = c->kernel_launches ? c->kmer_insert_ms / (double)c->kernel_launches : 0.0;
    c->kernel_ms = 0.0; c->adapter_ms = 0.0; c->kmer_ms = 0.0; c->kmer_insert_ms = 0.0; c->kmer_flush_ms = 0.0; c->kernel_launches = 0;
    return 0;
}
