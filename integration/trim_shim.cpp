// trim_shim.cpp -- the reference-side binding of libfaqcs_mi.so.
//
// Compile THIS file instead of the reference's trim.cpp (and drop seq_overlap.cpp) inside a FaQCs v2.10
// source tree, link with -lfaqcs_mi, and the unmodified reference driver (FaQCs.cpp, options.cpp,
// fastq.cpp, plot.cpp, file_util.cpp) runs its per-read hot path on an MI355X:
//
//     g++ -O3 -fopenmp -std=c++0x -I. -I<repo>/include -c <repo>/integration/trim_shim.cpp -o trim.o
//     g++ -o FaQCs fastq.o options.o file_util.o trim.o plot.o FaQCs.o -L<repo>/faqcs_amd -lfaqcs_mi -lm -lz -fopenmp
//
// It defines exactly the four symbols FaQCs.h:245-252 declares "In trim.cpp":
//     trim(), auto_detect_next_seq(), auto_detect_quality_offset(), parse_id()
// `#include "FaQCs.h"` below is the REFERENCE's header, found on the include path of the tree this file is
// compiled in (oracle/Makefile target `ref_hip` points it at /root/reference; nothing is copied).
//
// What trim() does here: packs the vector<Read> into the structure-of-arrays batch of include/faqcs_mi.h
// (one segment == this one call), submits it, applies the per-read results to the reads exactly the way
// the reference mutates them (substring + rule-based byte edits, or seq = qual = "" when filtered), and adds
// the DELTA of the device counter block since the previous call into the caller's filter_stats,
// adapter_stats and PlotInfo, so write_stats() and plot() downstream see what they always saw.
//
// --kmer_rarefaction: the reference merges per-call thread-local tables into the caller's MAP<Word,size_t> (trim.cpp:133-135)
// and samples the curve inside trim() (:157-185).  Here the context counts into a one-rank owner partition, the call's k-mer
// occurrences are downloaded (faqcs_kmer_outbox_host) and added to that map, and the sampling rule is restated below, so
// FaQCs.cpp:518-537 consumes the map as it always did.
#include <algorithm>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "FaQCs.h"   // the reference's own header (Read, Options, PlotInfo, FilterStat, MAP)
#include "faqcs_mi.h"

using namespace std;

namespace {

struct Shim {
    faqcs_ctx *ctx = NULL;
    faqcs_layout lay;
    vector<uint64_t> prev, cur;
    vector<string> adapter_names;
    int quality = 0;
    uint32_t R = 1024;           // row capacity of the per-position matrices: grown (the context is rebuilt) when a call brings a longer read
    // reusable host arenas
    vector<uint8_t> seq, qual;
    vector<uint32_t> off;
    vector<faqcs_read_result> res;
    faqcs_params prm;
    vector<const char *> adapter_ptr;
    bool kmers = false;          // the context extracts the k-mers of every call for the caller's MAP<Word, size_t>
    vector<uint64_t> keys;

    ~Shim() { if (ctx) faqcs_destroy(ctx); }
};
Shim g;

void fail(int rc)
{
    static string msg; // the reference throws `const char*` and main() prints "Caught the error <msg>"
    msg = faqcs_last_error();
    if (rc == FAQCS_E_QUALITY) throw "fastq.h:quality_score: Found a quality score value that is greater than the maximum allowed quality score";
    if (rc == FAQCS_E_BASE) throw "seq_overlap.cpp:na_to_bits: Unknown base!";
    throw msg.c_str();
}

void ensure_ctx(const Options &o)
{
    if (g.ctx) {
        if (g.quality != (int)o.quality) { // NextSeq bump between calls, FaQCs.cpp:272-277,404-414
            faqcs_set_quality(g.ctx, (int)o.quality);
            g.quality = (int)o.quality;
        }
        return;
    }
    faqcs_params &p = g.prm;
    memset(&p, 0, sizeof(p));
    p.abi_version = FAQCS_ABI_VERSION;
    p.mode = o.mode == Options::HARD ? FAQCS_MODE_HARD : (o.mode == Options::BWA ? FAQCS_MODE_BWA : FAQCS_MODE_BWA_PLUS);
    p.quality = (int)o.quality;
    p.input_quality_offset = (int)o.input_quality_offset;
    p.output_quality_offset = (int)o.output_quality_offset;
    p.min_read_length = o.min_read_length;
    p.max_num_poly_N = o.max_num_poly_N;
    p.trim_5 = o.trim_5;
    p.trim_3 = o.trim_3;
    p.replace_to_N_q = o.replace_to_N_q;
    p.average_quality = o.average_quality;
    p.low_complexity_cutoff_ratio = o.low_complexity_cutoff_ratio;
    p.filterAdapterMismatchRate = o.filterAdapterMismatchRate;
    p.protect_5 = o.protect_5;
    p.qc_only = o.qc_only;
    p.kmer = o.kmer;
    // k-mers: the caller owns the table (MAP<Word, size_t>&): the device only EXTRACTS the canonical k-mers of every call
    // (owner-partitioned mode with one owner) and the shim merges them into the caller's map (trim.cpp:133-135)
    p.kmer_rarefaction = o.kmer_rarefaction ? 1u : 0u;
    p.kmer_table_slots = 1u << 16;
    p.split_size = o.split_size;
    p.num_subsample = o.num_subsample;
    p.max_read_length = g.R;
    if (o.filter_adapter || o.filter_phiX) { // trim.cpp:86
        for (size_t j = 0; j < o.adapter.size(); ++j) {
            g.adapter_names.push_back(o.adapter[j].first);
            g.adapter_ptr.push_back(o.adapter[j].second.c_str());
        }
        p.n_adapters = (uint32_t)o.adapter.size();
        p.adapter_seq = g.adapter_ptr.empty() ? NULL : &g.adapter_ptr[0];
    }
    int rc = faqcs_create(&p, -1, &g.ctx);
    if (rc) fail(rc);
    if (p.kmer_rarefaction) { rc = faqcs_kmer_partition(g.ctx, 0, 1, 1); if (rc) fail(rc); g.kmers = true; }
    faqcs_counters_layout(g.R, p.n_adapters, &g.lay);
    g.prev.assign(g.lay.total, 0);
    g.cur.assign(g.lay.total, 0);
    g.quality = (int)o.quality;
}

void add_vector(vector<size_t> &dst, const uint64_t *cur, const uint64_t *prev, size_t n)
{
    size_t used = 0;
    for (size_t i = 0; i < n; ++i) if (cur[i]) used = i + 1; // vectors grow on demand in the reference (trim.cpp:880)
    if (dst.size() < used) dst.resize(used);
    for (size_t i = 0; i < used; ++i) dst[i] += (size_t)(cur[i] - prev[i]);
}

void add_matrix(matrix<size_t> &dst, const uint64_t *cur, const uint64_t *prev, uint32_t rows_src, uint32_t cols)
{
    if (rows_src == 0) return;
    if (dst.get_num_row() < rows_src) dst.resize(rows_src, cols); // matrix.h:23-29 preserves the old rows
    for (uint32_t r = 0; r < rows_src; ++r)
        for (uint32_t c = 0; c < cols; ++c) dst(r, c) += (size_t)(cur[(size_t)r * cols + c] - prev[(size_t)r * cols + c]);
}

void add_comp(vector<NucleotideCount> &dst, const uint64_t *cur, const uint64_t *prev)
{
    for (size_t b = 0; b < FAQCS_NCOMP_BIN; ++b) {
        const uint64_t *c = cur + b * FAQCS_NCOMP_KIND, *p = prev + b * FAQCS_NCOMP_KIND;
        dst[b].num_A += (size_t)(c[0] - p[0]); dst[b].num_T += (size_t)(c[1] - p[1]); dst[b].num_C += (size_t)(c[2] - p[2]);
        dst[b].num_G += (size_t)(c[3] - p[3]); dst[b].num_N += (size_t)(c[4] - p[4]); dst[b].num_GC += (size_t)(c[5] - p[5]);
    }
}

} // namespace

// FaQCs.h:245-248
void trim(vector<Read> &m_buffer, vector<size_t> &m_filter_stats,
          MAP<string, pair<size_t, size_t> > &m_adapter_stats, MAP<Word, size_t> &m_kmer_table, PlotInfo &m_info,
          Options &m_opt)
{
    const uint32_t n = (uint32_t)m_buffer.size();
    {   // the reference's matrices grow with the reads (trim.cpp:797-805,880); the context has a fixed row capacity: every accumulator
        // of a call is handed to the caller at its end, so a longer read only needs a new context with more rows
        size_t longest = 0;
        for (uint32_t i = 0; i < n; ++i) longest = m_buffer[i].seq.size() > longest ? m_buffer[i].seq.size() : longest;
        if (longest > FAQCS_MAX_READ_LENGTH) throw "trim_shim: reads of more than 32767 bases are not supported (seq_overlap.h:80 aligns in int16)";
        if (longest > g.R) {
            uint32_t r = g.R;
            while (r < longest) r *= 2;
            g.R = r > FAQCS_MAX_READ_LENGTH ? FAQCS_MAX_READ_LENGTH : r;
            if (g.ctx) { faqcs_destroy(g.ctx); g.ctx = NULL; g.adapter_names.clear(); g.adapter_ptr.clear(); }
        }
    }
    ensure_ctx(m_opt);
    // ---- vector<Read> (array of 3 std::string) -> structure of arrays ---------------------------------------
    size_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += m_buffer[i].seq.size();
    g.seq.assign(total + 64, 0);
    g.qual.assign(total + 64, 0);
    g.off.resize(n + 1);
    g.res.resize(n + 1);
    size_t o = 32; // slack in front: the library stages [offset[0], offset[n]) only
    for (uint32_t i = 0; i < n; ++i) {
        const Read &r = m_buffer[i];
        if (r.seq.size() != r.qual.size()) throw "trim_shim: |Sequence| != |Quality|"; // fastq.cpp:117-121 already rejects this
        g.off[i] = (uint32_t)o;
        memcpy(&g.seq[o], r.seq.data(), r.seq.size());
        memcpy(&g.qual[o], r.qual.data(), r.qual.size());
        o += r.seq.size();
    }
    g.off[n] = (uint32_t)o;
    const uint32_t seg[2] = {0, n};
    faqcs_batch b;
    memset(&b, 0, sizeof(b));
    b.seq = &g.seq[0]; b.qual = &g.qual[0]; b.offset = &g.off[0]; b.n_reads = n; b.n_segments = 1; b.segment_start = seg;
    int rc = 0;
    if (g.kmers) { // this call's k-mers are wanted while the curve is open (trim.cpp:82, :180-184)
        const uint32_t epoch = m_opt.kmer_rarefaction ? 0u : FAQCS_EPOCH_NONE;
        rc = faqcs_kmer_set_epochs(g.ctx, &epoch, 1);
        if (rc) fail(rc);
    }
    rc = faqcs_submit(g.ctx, &b, &g.res[0]);
    if (rc) fail(rc);
    rc = faqcs_finish(g.ctx, &g.cur[0], g.cur.size()); // syncs; raises the reference's throw sites
    if (rc) fail(rc);

    // ---- mutate the reads the way trim_read() does (trim.cpp:103-105, :291-292, :390-403, :516-525, :1191-1216) ----
    string s, q;
    for (uint32_t i = 0; i < n; ++i) {
        Read &r = m_buffer[i];
        const faqcs_read_result &x = g.res[i];
        if (!(x.flags & FAQCS_F_VALID)) { r.seq = r.qual = ""; continue; }
        s.resize(x.len); q.resize(x.len);
        faqcs_apply_edits(&g.prm, (const uint8_t *)r.seq.data(), (const uint8_t *)r.qual.data(), (uint32_t)r.seq.size(), &x,
                          (uint8_t *)&s[0], (uint8_t *)&q[0]);
        r.seq = s; r.qual = q;
    }

    // ---- add this call's share of every accumulator (trim.cpp:120-154) ----------------------------------------
    const faqcs_layout &L = g.lay;
    const uint64_t *c = &g.cur[0], *p = &g.prev[0];
    if (m_filter_stats.size() < FilterStat::NUM_STAT) m_filter_stats.resize(FilterStat::NUM_STAT);
    for (int k = 0; k < FilterStat::NUM_STAT; ++k) m_filter_stats[k] += (size_t)(c[L.filter_stats + k] - p[L.filter_stats + k]);
    for (uint32_t j = 0; j < L.n_adapters; ++j) {
        const uint64_t dr = c[L.adapter_stats + 2 * j] - p[L.adapter_stats + 2 * j];
        const uint64_t db = c[L.adapter_stats + 2 * j + 1] - p[L.adapter_stats + 2 * j + 1];
        if (dr) { pair<size_t, size_t> &st = m_adapter_stats[g.adapter_names[j]]; st.first += (size_t)dr; st.second += (size_t)db; }
    }
    const uint32_t rq_pre = faqcs_counter_rows(c + L.pre_qual, g.R, FAQCS_NQ), rq_post = faqcs_counter_rows(c + L.post_qual, g.R, FAQCS_NQ);
    add_matrix(m_info.pre_quality_matrix, c + L.pre_qual, p + L.pre_qual, rq_pre, FAQCS_NQ);
    add_matrix(m_info.post_quality_matrix, c + L.post_qual, p + L.post_qual, rq_post, FAQCS_NQ);
    add_matrix(m_info.pre_base_matrix, c + L.pre_base, p + L.pre_base, rq_pre, FAQCS_NBASE);
    add_matrix(m_info.post_base_matrix, c + L.post_base, p + L.post_base, rq_post, FAQCS_NBASE);
    add_vector(m_info.pre_read_quality_histogram, c + L.pre_read_qhist, p + L.pre_read_qhist, FAQCS_NQ);
    add_vector(m_info.pre_base_quality_histogram, c + L.pre_base_qhist, p + L.pre_base_qhist, FAQCS_NQ);
    add_vector(m_info.post_read_quality_histogram, c + L.post_read_qhist, p + L.post_read_qhist, FAQCS_NQ);
    add_vector(m_info.post_base_quality_histogram, c + L.post_base_qhist, p + L.post_base_qhist, FAQCS_NQ);
    add_vector(m_info.pre_length_histogram, c + L.pre_len_hist, p + L.pre_len_hist, g.R + 1);
    add_vector(m_info.post_length_histogram, c + L.post_len_hist, p + L.post_len_hist, g.R + 1);
    add_comp(m_info.pre_nuc_composition, c + L.pre_comp, p + L.pre_comp);
    add_comp(m_info.post_nuc_composition, c + L.post_comp, p + L.post_comp);
    g.prev.swap(g.cur);

    // ---- k-mers of this call into the caller's table, then the sampling rule of trim.cpp:157-185 ---------------------
    if (g.kmers && m_opt.kmer_rarefaction) {
        uint64_t nk = 0;
        rc = faqcs_kmer_outbox_host(g.ctx, NULL, 0, &nk);
        if (rc) fail(rc);
        g.keys.resize(nk + 1);
        rc = faqcs_kmer_outbox_host(g.ctx, &g.keys[0], nk, &nk);
        if (rc) fail(rc);
        for (uint64_t i = 0; i < nk; ++i) ++m_kmer_table[(Word)g.keys[i]];
        const size_t reads_so_far = m_filter_stats[FilterStat::TOTAL_NUMBER];
        const size_t index = reads_so_far / m_opt.split_size, have = m_info.kmer_rarefaction.size();
        if (index > have && have < m_opt.num_subsample) {
            Rarefaction point;
            point.num_seq = reads_so_far;
            point.distinct_kmer = m_kmer_table.size();
            point.total_kmer = 0;
            for (MAP<Word, size_t>::const_iterator it = m_kmer_table.begin(); it != m_kmer_table.end(); ++it) point.total_kmer += it->second;
            m_info.kmer_rarefaction.push_back(point);
        }
        if (have >= m_opt.num_subsample) m_opt.kmer_rarefaction = false; // the curve is complete
    }
}

// trim.cpp:619-626
bool auto_detect_next_seq(const vector<Read> &m_buffer)
{
    return !m_buffer.empty() && m_buffer[0].def.compare(0, 3, "@NS") == 0;
}

// trim.cpp:599-617
char auto_detect_quality_offset(const vector<Read> &m_buffer)
{
    for (size_t i = 0; i < m_buffer.size(); ++i) {
        const string &q = m_buffer[i].qual;
        for (size_t k = 0; k < q.size(); ++k) {
            if (q[k] > 74) return 64;
            if (q[k] < 59) return 33;
        }
    }
    throw "trim_shim:auto_detect_quality_offset: Unknown quality format!";
}

// trim.cpp:188-222
string parse_id(const string &m_def)
{
    string::size_type loc = m_def.find(' ');
    if (loc == string::npos) loc = m_def.size();
    if (loc > 1 && isdigit((unsigned char)m_def[loc - 1]) && (m_def[loc - 2] == '.' || m_def[loc - 2] == '/')) loc -= 2;
    return m_def.substr(0, loc);
}
