// trim_ab.cpp -- A/B harness for builds of libfaqcs_mi.so without Python: the headline workload (synthetic 2x<L> reads
// generated on the device, BWA_plus -q 5 --min_L 50) through the C ABI of each library given on the command line, in one
// process, round-robin; prints the trim kernel's average time per launch and a hash of the counter block and of the result
// array (equal hashes across libraries = same results on this input).
//   g++ -O2 -std=c++17 -o trim_ab trim_ab.cpp -I../../include -ldl -L/opt/rocm/lib -lamdhip64 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//   ./trim_ab <n_reads> <L> <rounds> libA.so [libB.so ...]     (env TRIM_AB_ARGS="adapter" adds the built-in adapters + polyA: not yet)
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "faqcs_mi.h"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Lib {
    std::string path;
    void *h;
    decltype(&faqcs_create) create;
    decltype(&faqcs_destroy) destroy;
    decltype(&faqcs_submit_device) submit_device;
    decltype(&faqcs_sync) sync;
    decltype(&faqcs_finish) finish;
    decltype(&faqcs_reset_counters) reset;
    decltype(&faqcs_counters_layout) layout;
    decltype(&faqcs_synth_fill) synth;
    decltype(&faqcs_kernel_report) report;
    decltype(&faqcs_last_error) last_error;
    decltype(&faqcs_debug_words) debug_words;
    decltype(&faqcs_abi_version) abi;
    decltype(&faqcs_terminal_n_flags) tn_flags; // (ABI 2; absent from older builds)
    faqcs_ctx *ctx = nullptr;
    double ms_sum = 0;
    int n = 0;
    bool uses_flags = false;
};

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: trim_ab <n_reads> <L> <rounds> lib.so [lib.so ...]\n"); return 2; }
    const uint32_t n = (uint32_t)atof(argv[1]);
    const uint32_t L = (uint32_t)atoi(argv[2]);
    const int rounds = atoi(argv[3]);
    const int reps = getenv("TRIM_AB_REPS") ? atoi(getenv("TRIM_AB_REPS")) : 3;
    std::vector<Lib> libs;
    for (int i = 4; i < argc; ++i) {
        Lib l;
        l.path = argv[i];
        l.h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
#define SYM(f, name) l.f = (decltype(l.f))dlsym(l.h, name); if (!l.f) { fprintf(stderr, "%s: no %s\n", argv[i], name); return 1; }
        SYM(create, "faqcs_create") SYM(destroy, "faqcs_destroy") SYM(submit_device, "faqcs_submit_device") SYM(sync, "faqcs_sync")
        SYM(finish, "faqcs_finish") SYM(reset, "faqcs_reset_counters") SYM(layout, "faqcs_counters_layout") SYM(synth, "faqcs_synth_fill")
        SYM(report, "faqcs_kernel_report") SYM(last_error, "faqcs_last_error") SYM(debug_words, "faqcs_debug_words") SYM(abi, "faqcs_abi_version")
        l.tn_flags = (decltype(l.tn_flags))dlsym(l.h, "faqcs_terminal_n_flags");
        libs.push_back(l);
    }
    HC(hipSetDevice(0));
    uint8_t *d_seq, *d_qual;
    uint32_t *d_off;
    faqcs_read_result *d_res;
    const size_t bytes = (size_t)n * L + 64;
    HC(hipMalloc((void **)&d_seq, bytes + 64));
    HC(hipMalloc((void **)&d_qual, bytes + 64));
    HC(hipMalloc((void **)&d_off, ((size_t)n + 1) * 4));
    HC(hipMalloc((void **)&d_res, (size_t)n * 8));
    d_seq += 64; d_qual += 64; // FAQCS_ARENA_PAD_BEFORE
    const float adapter_frac = 0.0f;
    if (libs[0].synth(0, d_seq, d_qual, d_off, n, L, 20260101ull, 0, adapter_frac) != 0) { fprintf(stderr, "synth: %s\n", libs[0].last_error()); return 1; }
    HC(hipDeviceSynchronize());

    faqcs_params p;
    memset(&p, 0, sizeof p);
    p.abi_version = FAQCS_ABI_VERSION;
    p.mode = FAQCS_MODE_BWA_PLUS;
    p.quality = 5;
    p.input_quality_offset = 33;
    p.output_quality_offset = 33;
    p.min_read_length = 50;
    p.max_num_poly_N = 2;
    p.low_complexity_cutoff_ratio = 0.85f;
    p.filterAdapterMismatchRate = 0.2f;
    p.kmer = 31;
    p.split_size = 1000000;
    p.num_subsample = 10;
    p.max_read_length = L <= 256 ? 256 : FAQCS_MAX_READ_LENGTH;
    if (getenv("TRIM_AB_POLYN")) p.max_num_poly_N = (uint32_t)atoi(getenv("TRIM_AB_POLYN")); // (!= 2: the EXT variants of trim_lds)
    if (getenv("TRIM_AB_AVGQ")) p.average_quality = (float)atof(getenv("TRIM_AB_AVGQ"));
    if (getenv("TRIM_AB_TRIM5")) p.trim_5 = (uint32_t)atoi(getenv("TRIM_AB_TRIM5"));             // (the WINDOWED variants)
    std::vector<uint32_t> seg;
    for (uint32_t s = 0; s < n; s += 32768) seg.push_back(s);
    seg.push_back(n);
    faqcs_batch b;
    b.seq = d_seq; b.qual = d_qual; b.offset = d_off; b.n_reads = n; b.n_segments = (uint32_t)seg.size() - 1; b.segment_start = seg.data();
    b.max_read_len = L;
    b.terminal_n = nullptr;
    faqcs_layout lay;
    libs[0].layout(p.max_read_length, 0, &lay);
    std::vector<uint64_t> counters(lay.total);
    std::vector<faqcs_read_result> res(n);
    uint8_t *d_tn = nullptr;
    HC(hipMalloc((void **)&d_tn, (size_t)n + 64));
    for (auto &l : libs) {
        p.abi_version = (uint32_t)l.abi(); // (an ABI 1 build reads the batch up to max_read_len only)
        b.terminal_n = nullptr;
        if (l.tn_flags && !getenv("TRIM_AB_NO_FLAGS")) { l.tn_flags(0, d_seq, d_off, n, d_tn); b.terminal_n = d_tn; }
        l.uses_flags = b.terminal_n != nullptr;
        if (l.create(&p, 0, &l.ctx) != 0) { fprintf(stderr, "%s: create: %s\n", l.path.c_str(), l.last_error()); return 1; }
        // one checked pass: hashes
        if (l.submit_device(l.ctx, &b, d_res) != 0 || l.sync(l.ctx) != 0) { fprintf(stderr, "%s: submit: %s\n", l.path.c_str(), l.last_error()); return 1; }
        l.finish(l.ctx, counters.data(), lay.total);
        HC(hipMemcpy(res.data(), d_res, (size_t)n * 8, hipMemcpyDeviceToHost));
        faqcs_kernel_times kt;
        l.report(l.ctx, &kt);
        printf("%-44s kernel %-26s counters %016llx results %016llx  (reads %llu, trimmed %llu)\n", l.path.c_str(), kt.trim_kernel ? kt.trim_kernel : "?",
               (unsigned long long)fnv(counters.data(), lay.total * 8), (unsigned long long)fnv(res.data(), (size_t)n * 8),
               (unsigned long long)counters[lay.filter_stats + 1], (unsigned long long)counters[lay.filter_stats + 3]);
        l.reset(l.ctx);
    }
    for (int r = 0; r < rounds; ++r) {
        for (auto &l : libs) {
            faqcs_kernel_times kt;
            l.report(l.ctx, &kt); // resets the timers
            b.terminal_n = l.uses_flags ? d_tn : nullptr;
            for (int k = 0; k < reps; ++k) {
                if (l.submit_device(l.ctx, &b, d_res) != 0) { fprintf(stderr, "%s: submit: %s\n", l.path.c_str(), l.last_error()); return 1; }
                if (getenv("TRIM_AB_SYNC_EACH")) l.sync(l.ctx); // (the composition fold of a launch then never runs beside the next launch)
            }
            l.sync(l.ctx);
            l.report(l.ctx, &kt);
            printf("round %d %-44s trim %.4f ms/launch -> %.1f M reads/s\n", r, l.path.c_str(), kt.trim_ms, n / kt.trim_ms / 1e3);
            l.ms_sum += kt.trim_ms; l.n++;
            l.reset(l.ctx);
        }
    }
    if (getenv("TRIM_AB_STAMPS")) { // a -DFAQCS_LDS_STAMPS build: section clocks since the context was created (err words 16..)
        static const char *names[9] = {"load Q (offsets + DMA)", "terminal-N + sum pass", "3' walk", "5' walk + filters", "Q-B", "load S (DMA)",
                                       "S (fused pass)", "verdicts / dinucleotide", "undo / epilogue / flush"};
        for (auto &l : libs) {
            uint64_t w[16] = {0};
            l.debug_words(l.ctx, w, 16);
            double tot = 0;
            for (int i = 0; i < 9; ++i) tot += (double)w[i];
            if (tot <= 0) continue;
            const double chunks = (double)(1 + rounds * reps) * n / 64.0;
            printf("%s: section clocks per 64-read chunk per wave (total %.0f)\n", l.path.c_str(), tot / chunks);
            for (int i = 0; i < 9; ++i) printf("  %-26s %8.1f = %5.1f %%\n", names[i], w[i] / chunks, 100.0 * w[i] / tot);
            printf("  %-26s %8.1f   (outside the sections)\n  %-26s %8.1f   (per chunk; once per wave)\n  %-26s %8.1f\n", "block flushes (+ barrier)", w[11] / chunks,
                   "register spills", w[12] / chunks, "loop, entry to exit", w[9] / chunks);
            {   // one more launch on its own: when do the blocks finish?
                uint64_t z[16];
                l.debug_words(l.ctx, z, 16);
                b.terminal_n = l.uses_flags ? d_tn : nullptr;
                l.submit_device(l.ctx, &b, d_res); l.sync(l.ctx);
                l.debug_words(l.ctx, z, 16);
                const double t_last = (double)z[13], t_first = (double)~z[14], t_start = (double)~z[15];
                printf("  blocks finish between %.1f and %.1f us after the first wave entered its loop (wave 0 of each block)\n", (t_first - t_start) * 0.01, (t_last - t_start) * 0.01);
            }
            if (w[10]) printf("  shader clock while the kernel runs: %.3f GHz (s_memtime / s_memrealtime x 100 MHz, summed over the waves)\n", (double)w[9] / (double)w[10] * 0.1);
        }
    }
    for (auto &l : libs) printf("mean %-44s trim %.4f ms/launch -> %.1f M reads/s (frac of 8 TB/s at %u B/read: %.4f)\n", l.path.c_str(), l.ms_sum / l.n,
                                n / (l.ms_sum / l.n) / 1e3, 2 * L + 12, n * (2.0 * L + 12) / (l.ms_sum / l.n * 1e-3) / 8e12);
    fflush(stdout);
    if (libs.size() > 1) _Exit(0); // (skip the teardown of several HIP-linked libraries in one process)
    for (auto &l : libs) l.destroy(l.ctx);
    return 0;
}
