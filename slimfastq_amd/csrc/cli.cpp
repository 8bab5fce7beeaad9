// cli.cpp -- `slimfastq-amd`: the reference's command line (config.cpp:161-277) over the C ABI.
//
//   -u fastq  -f file.sfq  -d  -O  -l N | -1..-4  -q  -s  -v  -h        (same meaning as the reference)
//   -B reads  : records per independent block (default 1024; 0 = one block = a format-6 file the
//               reference itself can decode)
//   -A        : adaptive tables (every block runs the reference's per-symbol table updates) instead of the default
//               frozen tables (rows built by counting passes, one chain per GPU lane)
//   -g dev    : HIP device
// All model / coder work happens in libslimfastq_amd.so on the GPU; this file parses arguments, reads and
// writes files and fills the info page.
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "container.h"

static const int   kInternalVersion = 6;      // config.cpp:41
static const int   kBlockVersion = 10;         // 7: block format, the reference's quirks kept; 8: lossless ("gen.lc", 14 stream sizes per index entry);
                                                // 9: "chn.idx" may carry flag bits 2-5 (difference-coded lists, segments, Rice-coded base exceptions, the bases' match
                                                //    model): a reader of version 8 did not look at flags it did not know, so what sets them says 9 and is refused there
                                                // 10: flag bit 7 (bases without a model as two bits each, no coder): a reader of version 9 refuses the flag, so what sets it says 10
static const int   kBlockVersionMin = 7;
static const char* kUserVersion = "2.04-amd";

static bool g_encode = true;
static bool g_batch = false;                  // -b: jobs from stdin, one context for all of them; errors end the job, not the process
static std::string g_usr;

struct JobError { std::string msg; };

// SFQ_TIMING=1: where a run spends its wall time (stderr)
#include <chrono>
#include <thread>
#include <atomic>
static bool g_timing = false;                 // -z: phase timings on stderr
static void tick(const char* what) {
    static auto t0 = std::chrono::steady_clock::now(), last = t0;
    if (!g_timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] %-28s +%7.3f s  (%7.3f s)\n", what, std::chrono::duration<double>(now - last).count(), std::chrono::duration<double>(now - t0).count());
    last = now;
}

static std::string g_partial;                // an archive being written as the slabs come back: a failed job removes it
[[noreturn]] static void croak(const char* fmt, ...) {                 // config.cpp:54-68
    char msg[1024];
    va_list ap; va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    if (!g_partial.empty()) { unlink(g_partial.c_str()); g_partial.clear(); }
    // (thrown in one-shot mode too: the stack unwinds through the guards below, which join the reader / writer threads
    //  before main() prints the message and exits -- exit() from here would race a thread still inside pwrite / fwrite)
    throw JobError{msg};
}
// a thread that is joined wherever its scope ends, croak()'s unwinding included (a joinable std::thread that goes out of
// scope calls std::terminate)
struct Joiner {
    std::thread t;
    ~Joiner() { join(); }
    void join() { if (t.joinable()) t.join(); }
    template <typename F> void start(F&& f) { join(); t = std::thread(std::forward<F>(f)); }
};

static void usage() {
    printf("Usage: \n"
           "-u  usr-filename : (default: stdin)\n"
           "-f comp-filename : required - compressed\n"
           "-d               : decode (instead of encoding) \n"
           "-O               : silently overwrite existing files\n"
           "-l level         : compression level 1 to 4 (default is 3 ) \n"
           "-1, -2, -3, -4   : alias for -l 1, -l 2, etc \n"
           "-B reads         : records per independent GPU block (default: about 376 KiB of text, i.e. 1024 reads of 150 bp;\n"
           "                   0 = single block, reference-compatible file)\n"
           "-A               : adaptive tables: every block updates its rows per symbol, as the reference does (slower)\n"
           "-F               : frozen tables, built by counting passes, one coding chain per GPU lane\n"
           "                   default: -F from 64 MiB of text on, -A in blocks of 65536 reads below that (the priors\n"
           "                   frozen tables transmit weigh too much on a small file)\n"
           "-C reads         : frozen tables: records per chain (default: automatic)\n"
           "-S mbytes        : input is compressed in slabs of this many MiB, one archive segment each (default 512 for a\n"
           "                   regular file, read ahead while the GPU codes the slab before; 2048 for a pipe)\n"
           "-t threads       : threads reading a slab (default 6)\n"
           "-z               : print the time of every phase on stderr\n"
           "-g device        : HIP device index (default 0)\n"
           "-T percent       : share of the device memory this process may use for model tables (several processes on one GPU)\n"
           "-b               : batch: read '<fastq>\\t<sfq>' jobs (with -d: '<sfq>\\t<fastq>') from stdin, answer 'ok|fail\\t...' per job on stdout\n"
           "-v               : version : internal version \n"
           "-h               : help : this message \n"
           "-s               : stat : information about a compressed file \n"
           "-q               : suppress extra stats info that could have been seen by -s \n"
           "\nslimfastq-amd A B : compress A (a fastq file) to B, or decompress A (a slimfastq file) to B / stdout\n");
    exit(0);
}

static int clamp_level(int l) { return l > 4 ? 4 : l < 1 ? 1 : l; }   // config.cpp:232-237
static int level_gen_bits(int level) { switch (level) { case 1: return 18; case 2: return 22; case 3: return 24; default: return 26; } }

struct Opts {
    int level = 3, device = 0;
    long block_reads = -1;                                             // -1 = automatic (about 376 KiB of text per block)
    bool overwrite = false, quiet = false;
    uint64_t slab_bytes = 0;                      // -S; 0 = 512 MiB for a regular file (read ahead into pinned buffers), 2 GiB otherwise
    int io_threads = 6;                           // -t: threads that read a slab
    int table_pct = 0;                                                 // -T: share of the device memory for model tables (0 = the library's default)
    bool adaptive = false;                                             // -A
    bool force_frozen = false;                                         // -F
    long chain_reads = 0;                                              // -C
};

// "seg.idx": an archive is a sequence of SEGMENTS, each the result of one library call (one slab of a large
// input, or one rank of a multi-GPU job): its blocks, its share of every stream, its own quality prior.
struct Segment { uint64_t nblocks, prior_bytes, raw_bytes, chain_bytes, recpri_bytes; };
static void put_v(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
static bool get_v(const std::vector<uint8_t>& b, size_t& p, uint64_t& v) {
    v = 0;
    for (int sh = 0; sh < 64; sh += 7) { if (p >= b.size()) return false; const uint8_t c = b[p++]; v |= (uint64_t)(c & 0x7f) << sh; if (!(c & 0x80)) return true; }
    return false;
}

// Growable byte buffer that never zero-fills (a std::vector would touch gigabytes just to size them).
struct Bytes {
    uint8_t* p = nullptr; size_t n = 0, cap = 0, touched = 0;
    ~Bytes() { free(p); }
    // Fault the first `upto` bytes in now: a device-to-host copy into pages the process has never touched goes
    // through the driver page by page and is ~10x slower than touching them here first (measured).
    void touch(size_t upto) {
        if (upto > cap) upto = cap;
        for (size_t i = touched & ~(size_t)4095; i < upto; i += 4096) p[i] = 0;
        if (upto > touched) touched = upto;
    }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        uint8_t* q = (uint8_t*)realloc(p, want);
        if (!q) return false;
        p = q; cap = want; touched = std::min(touched, n);               // realloc may have moved to fresh pages
        return true;
    }
};
// Fill `buf` (which may hold a carried-over tail) up to `want` bytes; returns false on a read error.
static bool fill(FILE* f, Bytes& buf, size_t want, bool& eof) {
    if (!buf.reserve(want)) return false;
    while (buf.n < want) {
        const size_t got = fread(buf.p + buf.n, 1, want - buf.n, f);
        if (got == 0) { eof = true; break; }
        buf.n += got;
    }
    return !ferror(f);
}
// Bytes of [p, p+n) that hold whole records (a record is four lines), given that p starts at a record.
static size_t whole_records(const uint8_t* p, size_t n) {
    size_t nl = 0;
    for (size_t i = 0; i < n; i++) nl += p[i] == '\n';
    size_t drop = nl & 3, end = n;
    while (end > 0 && p[end - 1] != '\n') end--;                    // the partial last line
    while (drop && end > 0) { end--; while (end > 0 && p[end - 1] != '\n') end--; drop--; }
    return end;
}

static void encode_file(sfq_ctx* ctx, const Opts& o, const std::string& usr, const std::string& fil) {
    if (!o.overwrite && access(fil.c_str(), F_OK) == 0) {
        if (g_batch) croak("Can't write file '%s': File exists", fil.c_str());
        fprintf(stderr, "Can't write file '%s': File exists\n", fil.c_str()); exit(1);
    }
    FILE* in = usr.empty() ? stdin : fopen(usr.c_str(), "rb");
    if (!in) {
        if (g_batch) croak("Can't read file '%s'", usr.c_str());
        fprintf(stderr, "Can't read file '%s'\n", usr.c_str()); exit(1);
    }
    const bool legacy = o.block_reads == 0;
    sfq_params p; memset(&p, 0, sizeof p);
    p.level = o.level; p.block_reads = o.block_reads < 0 ? SFQ_BLOCK_AUTO : (uint32_t)o.block_reads;
    p.prior_step = legacy ? 0 : SFQ_PRIOR_AUTO;                        // warm start needs the block format
    // tables: by the size of the text unless asked for (one decision per file: the first slab's size stands for a pipe's)
    bool frozen = !legacy && !o.adaptive;
    bool tables_decided = legacy || o.adaptive || o.force_frozen;
    p.tables = frozen ? SFQ_TABLES_FROZEN : SFQ_TABLES_ADAPTIVE;
    p.chain_reads = (uint32_t)std::max(0l, o.chain_reads);

    std::vector<uint8_t> streams[SFQ_NSTREAMS], first_all, prior_all, chain_all, recpri_all;
    std::vector<sfq_block_info> blocks_all;
    std::vector<Segment> segs;
    uint64_t total_in = 0, total_records = 0;
    // (Mapping the file and handing the mapping to the library was measured: the copy engine pins page-cache pages one
    //  by one, 4-10x slower than reading into an ordinary buffer first.)
    size_t file_left = SIZE_MAX;                                       // bytes not yet read, when the input is a regular file
    if (in != stdin) { struct stat st; if (fstat(fileno(in), &st) == 0 && S_ISREG(st.st_mode)) file_left = (size_t)st.st_size; }
    auto decide_tables = [&](uint64_t text_bytes) {
        if (tables_decided) return;
        tables_decided = true;
        frozen = text_bytes >= (64ull << 20);
        p.tables = frozen ? SFQ_TABLES_FROZEN : SFQ_TABLES_ADAPTIVE;
        if (!frozen) { p.prior_step = 0; if (o.block_reads < 0) p.block_reads = 65536; }      // include/slimfastq_amd.h SFQ_TABLES_AUTO
    };
    if (file_left != SIZE_MAX) decide_tables(file_left);
    // A one-shot process pays for every byte of model tables it allocates (device allocations of tens of GB take
    // seconds), and its time is file I/O anyway: size the tables for a fraction of the block slots the text could use.
    if (!g_batch) {
        const uint64_t slab0 = o.slab_bytes ? o.slab_bytes : (!legacy && in != stdin && file_left != SIZE_MAX) ? (512ull << 20) : (2048ull << 20);
        const uint64_t text = std::min<uint64_t>(file_left == SIZE_MAX ? slab0 : file_left, legacy ? UINT64_MAX : slab0);
        uint64_t budget = std::max<uint64_t>(2ull << 30, 8 * text);
        if (o.table_pct) budget = std::min<uint64_t>(budget, sfq_ctx_device_memory(ctx) / 100 * (uint64_t)o.table_pct);
        sfq_ctx_set_table_budget(ctx, budget);
    }
    // A regular file in the block format: slabs are read AHEAD -- several threads pread the next slab into one of two
    // page-locked buffers while the GPU codes the current one -- and the streams come back into a page-locked buffer too.
    // (File to file the time is the host's: one thread fread()s at ~5 GB/s, pageable H2D/D2H copies crawl.)
    bool ahead = !legacy && in != stdin && file_left != SIZE_MAX;
    size_t slab = (size_t)(o.slab_bytes ? o.slab_bytes : ahead ? (512ull << 20) : (2048ull << 20));
    // (a small file: buffers of its size, not of the slab's -- page-locking 1.5 GiB for a 1 MB file costs hundreds of
    //  milliseconds and may not fit a small memlock limit)
    if (ahead) slab = std::min(slab, std::max<size_t>(file_left, (size_t)1 << 20));
    // kept for the life of the process: a -b worker reuses them, a one-shot run exits without the 0.2 s of unpinning
    static uint8_t* pinned[4] = { nullptr, nullptr, nullptr, nullptr }; static size_t pinned_cap = 0;
    if (ahead) {
        const size_t cap = slab + (64u << 20), out_pin = cap / 3 + (16u << 20);
        if (pinned_cap < cap) {
            for (auto& q : pinned) { if (q) sfq_host_free(ctx, q); q = nullptr; }
            pinned[0] = (uint8_t*)sfq_host_alloc(ctx, cap); pinned[1] = (uint8_t*)sfq_host_alloc(ctx, cap);
            pinned[2] = (uint8_t*)sfq_host_alloc(ctx, out_pin); pinned[3] = (uint8_t*)sfq_host_alloc(ctx, out_pin);
            pinned_cap = (pinned[0] && pinned[1] && pinned[2] && pinned[3]) ? cap : 0;
            if (!pinned_cap) {                                             // no page-locked memory to be had: the plain path below
                for (auto& q : pinned) { if (q) sfq_host_free(ctx, q); q = nullptr; }
                ahead = false;
                slab = (size_t)(o.slab_bytes ? o.slab_bytes : (2048ull << 20));
            }
        }
    }
    Bytes fq, out;
    bool eof = false;
    sfqc::PagedWriter pw; bool streamed = false;                       // the archive written as the slabs come back
    if (ahead) {
        const int fd = fileno(in);
        const size_t fsize = file_left;
        const size_t cap = slab + (64u << 20);                          // room for the carried-over partial record
        const size_t out_cap = (size_t)sfq_encode_bound(cap);
        // the streams of a slab are rarely a third of its text: that much is page-locked, twice (one buffer is written to
        // the archive while the next slab's streams arrive in the other); a slab that needs more goes through a plain buffer
        const size_t out_pin = cap / 3 + (16u << 20);
        uint8_t* buf[2] = { pinned[0], pinned[1] };
        Bytes big_out;                                                  // only for a slab whose streams outgrow the pinned buffer
        std::string werr;
        if (!pw.open(fil, werr)) croak("%s", werr.c_str());
        g_partial = fil;
        int sid[SFQ_NSTREAMS];
        for (int s2 = 0; s2 < SFQ_NSTREAMS; s2++) sid[s2] = -1;           // a stream gets its directory entry with its first bytes
        Joiner writer;                                                  // appends the previous slab's streams to the archive
        int ob = 0;
        tick("pinned buffers");
        // read file bytes [off, off + len) to dst with o.io_threads threads; newlines counted on the way
        struct Read {
            std::vector<std::thread> th; std::atomic<uint64_t> nl{0}; std::atomic<int> bad{0};
            ~Read() { for (auto& t : th) if (t.joinable()) t.join(); }
        };
        auto start_read = [&](Read& r, uint8_t* dst, size_t off, size_t len) {
            r.nl = 0; r.bad = 0;
            const int nt = std::max(1, o.io_threads);
            const size_t per = ((len + nt - 1) / nt + 4095) & ~(size_t)4095;
            for (int t = 0; t < nt; t++) {
                const size_t a0 = std::min(len, per * t), a1 = std::min(len, per * (t + 1));
                if (a0 >= a1) break;
                r.th.emplace_back([&r, fd, dst, off, a0, a1]() {
                    size_t at = a0; uint64_t nl = 0;
                    while (at < a1) {
                        const ssize_t got = pread(fd, dst + at, std::min<size_t>(a1 - at, 8u << 20), (off_t)(off + at));
                        if (got <= 0) { r.bad = 1; return; }
                        for (const uint8_t* q = dst + at, *e = q + got; q < e; q++) nl += *q == '\n';
                        at += (size_t)got;
                    }
                    r.nl += nl;
                });
            }
        };
        auto join_read = [&](Read& r) { for (auto& t : r.th) t.join(); r.th.clear(); if (r.bad) croak("read error"); };
        size_t off = 0;                                                 // file bytes handed to a reader so far
        size_t have[2] = {0, 0}; uint64_t nls[2] = {0, 0};
        int cur = 0;
        {
            Read r; const size_t len = std::min(slab, fsize);
            start_read(r, buf[0], 0, len); join_read(r);
            have[0] = len; nls[0] = r.nl; off = len;
        }
        tick("read slab");
        while (have[cur]) {
            const bool last = off >= fsize;
            uint8_t* text = buf[cur];
            size_t use = have[cur];
            if (!last) {                                                // whole records only: drop the partial last line and nl % 4 lines more
                size_t drop = (size_t)(nls[cur] & 3), end = use;
                while (end > 0 && text[end - 1] != '\n') end--;
                while (drop && end > 0) { end--; while (end > 0 && text[end - 1] != '\n') end--; drop--; }
                use = end;
                if (use == 0) croak("a record longer than the slab (%zu MiB): raise -S", slab >> 20);
            }
            // the next slab: the carried-over tail, then file bytes read by the threads while this one is coded
            const int nxt = cur ^ 1;
            const size_t carry = have[cur] - use;
            Read r;
            size_t len = 0;
            if (carry > (64u << 20)) croak("a record longer than 64 MiB next to a slab boundary: raise -S");
            memcpy(buf[nxt], text + use, carry);
            uint64_t carry_nl = 0; for (size_t i = 0; i < carry; i++) carry_nl += buf[nxt][i] == '\n';
            if (!last) { len = std::min(slab, fsize - off); start_read(r, buf[nxt] + carry, off, len); }
            sfq_result res;
            uint8_t* outp = pinned[2 + ob];
            int rc = sfq_encode_blocks_host(ctx, text, use, &p, outp, out_pin, &res);
            if (rc == SFQ_E_OVERFLOW) {
                if (!big_out.reserve(out_cap)) croak("out of memory");
                writer.join();
                outp = big_out.p;
                rc = sfq_encode_blocks_host(ctx, text, use, &p, outp, out_cap, &res);
            }
            if (rc) croak("%s", sfq_last_error(ctx));
            tick("sfq_encode_blocks_host");
            const size_t b0 = blocks_all.size();
            blocks_all.resize(b0 + res.n_blocks);
            sfq_get_block_index(ctx, blocks_all.data() + b0, res.n_blocks);
            for (size_t b = b0; b < blocks_all.size(); b++) { blocks_all[b].first_record += total_records; blocks_all[b].first_hdr_off += first_all.size(); }
            const size_t f0 = first_all.size();
            first_all.resize(f0 + (size_t)res.first_hdr_bytes + 1);
            sfq_get_first_headers(ctx, first_all.data() + f0, res.first_hdr_bytes);
            first_all.resize(f0 + (size_t)res.first_hdr_bytes);
            writer.join();
            {
                const sfq_result rr = res; sfqc::PagedWriter* w = &pw; int* ids = sid; const uint8_t* src = outp;
                writer.start([rr, w, ids, src]() {
                    for (int s2 = 0; s2 < SFQ_NSTREAMS; s2++) if (rr.stream_bytes[s2]) {
                        if (ids[s2] < 0) ids[s2] = w->stream(sfq_stream_name(s2));
                        w->append(ids[s2], src + rr.stream_offset[s2], (size_t)rr.stream_bytes[s2]);
                    }
                });
                if (outp == big_out.p) writer.join(); else ob ^= 1;
            }
            Segment sg{res.n_blocks, 0, use, 0, 0};
            const int64_t pn = sfq_get_qlt_prior(ctx, nullptr, 0);
            if (pn > 0) { const size_t q0 = prior_all.size(); prior_all.resize(q0 + (size_t)pn); sfq_get_qlt_prior(ctx, prior_all.data() + q0, (uint64_t)pn); sg.prior_bytes = (uint64_t)pn; }
            const int64_t cn = sfq_get_chain_index(ctx, nullptr, 0);
            if (cn > 0) { const size_t q0 = chain_all.size(); chain_all.resize(q0 + (size_t)cn); sfq_get_chain_index(ctx, chain_all.data() + q0, (uint64_t)cn); sg.chain_bytes = (uint64_t)cn; }
            const int64_t rn = sfq_get_rec_prior(ctx, nullptr, 0);
            if (rn > 0) { const size_t q0 = recpri_all.size(); recpri_all.resize(q0 + (size_t)rn); sfq_get_rec_prior(ctx, recpri_all.data() + q0, (uint64_t)rn); sg.recpri_bytes = (uint64_t)rn; }
            segs.push_back(sg);
            total_in += use; total_records += res.n_records;
            tick("collect slab");
            join_read(r);
            tick("wait for the next slab");
            have[nxt] = carry + len; nls[nxt] = carry_nl + r.nl; off += len;
            cur = nxt;
        }
        writer.join();
        streamed = true;
        eof = true;                                                     // nothing left for the loop below
    }
    for (;;) {
        if (eof && fq.n == 0) break;
        // the reference's single adaptive state (format 6) cannot be cut: one slab holds the whole file
        size_t want = legacy ? std::max<size_t>(fq.n * 2, 64u << 20) : slab;
        if (want <= fq.n) want = fq.n * 2;                              // a record longer than the slab: grow
        if (file_left != SIZE_MAX) want = legacy ? fq.n + file_left + 1 : std::min(want, fq.n + file_left + 1);   // +1: see the end of the file
        const size_t before = fq.n;
        if (!eof && !fill(in, fq, want, eof)) croak("read error");
        if (file_left != SIZE_MAX) file_left -= std::min(file_left, fq.n - before);
        tick("read slab");
        if (legacy && !eof) continue;
        size_t use = fq.n;
        if (!eof) { use = whole_records(fq.p, fq.n); if (use == 0) continue; }
        const uint8_t* text = fq.p;
        if (use == 0) break;
        decide_tables(eof ? use : (64ull << 20));                          // (a pipe: more than one slab of text is a big file)
        const size_t bound = (size_t)sfq_encode_bound(use);
        if (!out.reserve(bound)) croak("out of memory");
        out.touch(use / 3 + (1 << 20));                                    // the streams rarely exceed a third of the text
        sfq_result res;
        const int rc = sfq_encode_blocks_host(ctx, text, use, &p, out.p, bound, &res);
        if (rc) croak("%s", sfq_last_error(ctx));
        tick("sfq_encode_blocks_host");
        const size_t b0 = blocks_all.size();
        blocks_all.resize(b0 + res.n_blocks);
        sfq_get_block_index(ctx, blocks_all.data() + b0, res.n_blocks);
        for (size_t b = b0; b < blocks_all.size(); b++) { blocks_all[b].first_record += total_records; blocks_all[b].first_hdr_off += first_all.size(); }
        const size_t f0 = first_all.size();
        first_all.resize(f0 + (size_t)res.first_hdr_bytes + 1);
        sfq_get_first_headers(ctx, first_all.data() + f0, res.first_hdr_bytes);
        first_all.resize(f0 + (size_t)res.first_hdr_bytes);
        for (int s = 0; s < SFQ_NSTREAMS; s++)
            streams[s].insert(streams[s].end(), out.p + res.stream_offset[s], out.p + res.stream_offset[s] + res.stream_bytes[s]);
        Segment sg{res.n_blocks, 0, use, 0, 0};
        if (!legacy) {
            const int64_t pn = sfq_get_qlt_prior(ctx, nullptr, 0);
            if (pn > 0) { const size_t q0 = prior_all.size(); prior_all.resize(q0 + (size_t)pn); sfq_get_qlt_prior(ctx, prior_all.data() + q0, (uint64_t)pn); sg.prior_bytes = (uint64_t)pn; }
            const int64_t cn = sfq_get_chain_index(ctx, nullptr, 0);
            if (cn > 0) { const size_t q0 = chain_all.size(); chain_all.resize(q0 + (size_t)cn); sfq_get_chain_index(ctx, chain_all.data() + q0, (uint64_t)cn); sg.chain_bytes = (uint64_t)cn; }
            const int64_t rn = sfq_get_rec_prior(ctx, nullptr, 0);
            if (rn > 0) { const size_t q0 = recpri_all.size(); recpri_all.resize(q0 + (size_t)rn); sfq_get_rec_prior(ctx, recpri_all.data() + q0, (uint64_t)rn); sg.recpri_bytes = (uint64_t)rn; }
        }
        segs.push_back(sg);
        total_in += use; total_records += res.n_records;
        memmove(fq.p, fq.p + use, fq.n - use); fq.n -= use;               // keep the partial record for the next slab
    }
    if (in != stdin) fclose(in);
    if (segs.empty()) croak("fastq file: empty input");
    tick("collect streams");

    sfqc::Archive a;                                                   // info keys in the reference's order (config.cpp:334-347, usrs.cpp:262-266, recs.cpp:71, gens.cpp:104, usrs.cpp:405)
    a.set("whoami", "slimfastq");
    a.set("version", legacy ? kInternalVersion : kBlockVersion);
    a.set("config.level", o.level);
    a.set("orig.filename", usr.empty() ? "<< stdin >>" : usr);
    if (!usr.empty() || !legacy) a.set("orig.size", (long long)total_in);
    if (legacy) {
        const sfq_block_info& b = blocks_all[0];
        if (first_all.size() >= 400) croak("first header too long for the reference's info page (recs.cpp:30)");
        if (b.solid) a.set("usr.solid", 1);
        a.set("llen", b.llen);
        a.set("usr.2id", b.two_id);
        a.set("rec.first", std::string(first_all.begin(), first_all.end()));
        if (b.n_byte && b.n_byte != 'N') a.set("gen.N_byte", b.n_byte);
        a.set("num_records", (long long)total_records);
        if (!o.quiet && b.extra_hi) a.set("qlt.extra.hi", b.extra_hi);
    } else {
        a.set("blk.reads", (long long)blocks_all[0].n_records);            // of the first segment (each segment carries its own in the index)
        a.set("blk.count", (long long)blocks_all.size());
        a.set("num_records", (long long)total_records);
        if (segs.size() > 1) a.set("seg.count", (long long)segs.size());
        if (frozen) a.set("blk.tables", 1);                                // frozen tables: chn.idx / rec.pri per segment
    }
    if (streamed) {
        auto add = [&](const char* name, const std::vector<uint8_t>& v) { const int id = pw.stream(name); pw.append(id, v.data(), v.size()); };
        add("blk.idx", sfqc::pack_block_index(blocks_all));
        add("blk.hdr", first_all);
        if (!prior_all.empty()) add("qlt.pri", prior_all);
        if (!chain_all.empty()) add("chn.idx", chain_all);
        if (!recpri_all.empty()) add("rec.pri", recpri_all);
        if (segs.size() > 1) {
            std::vector<uint8_t> si;
            put_v(si, segs.size());
            for (auto& g : segs) {
                put_v(si, g.nblocks); put_v(si, g.prior_bytes); put_v(si, g.raw_bytes);
                if (frozen) { put_v(si, g.chain_bytes); put_v(si, g.recpri_bytes); }
            }
            add("seg.idx", si);
        }
        std::string werr;
        if (!pw.finish(a.info, werr)) croak("%s", werr.c_str());
        g_partial.clear();
        tick("finish archive");
        return;
    }
    for (int s = 0; s < SFQ_NSTREAMS; s++) if (!streams[s].empty()) a.add(sfq_stream_name(s), std::move(streams[s]));
    if (!legacy) {
        a.add("blk.idx", sfqc::pack_block_index(blocks_all));
        a.add("blk.hdr", first_all);
        if (!prior_all.empty()) a.add("qlt.pri", prior_all);
        if (!chain_all.empty()) a.add("chn.idx", chain_all);
        if (!recpri_all.empty()) a.add("rec.pri", recpri_all);
        if (segs.size() > 1) {
            std::vector<uint8_t> si;
            put_v(si, segs.size());
            for (auto& g : segs) {
                put_v(si, g.nblocks); put_v(si, g.prior_bytes); put_v(si, g.raw_bytes);
                if (frozen) { put_v(si, g.chain_bytes); put_v(si, g.recpri_bytes); }
            }
            a.add("seg.idx", si);
        }
    }
    std::string err;
    tick("build archive");
    if (!sfqc::write_file(fil, a, err)) croak("%s", err.c_str());
    tick("write file");
}

static void decode_file(sfq_ctx* ctx, const Opts& o, const std::string& usr, const std::string& fil) {
    std::string err;
    sfqc::Archive a;
    tick("start");
    if (!sfqc::read_file(fil, a, err)) croak("%s", err.c_str());
    tick("read archive");
    const int version = (int)a.get_long("version", 0);
    if (version > kBlockVersion) croak("%s was compressed with slimfastq version %d. My version is %d. Please upgrade me before decoing", fil.c_str(), version, kBlockVersion);
    const int level = clamp_level((int)a.get_long("config.level", 2));       // config.cpp:359
    std::vector<sfq_block_info> blocks;
    std::vector<uint8_t> first;
    std::vector<Segment> segs;
    const std::vector<uint8_t>* pri = a.find("qlt.pri");
    const std::vector<uint8_t>* chn = a.find("chn.idx");
    const std::vector<uint8_t>* rpr = a.find("rec.pri");
    const long long blk_tables = a.get_long("blk.tables", 0);
    if (blk_tables != 0 && blk_tables != 1) croak("%s: blk.tables = %lld is not a table mode this version knows. Please upgrade me before decoing", fil.c_str(), blk_tables);
    const bool frozen = blk_tables == 1;
    if (!frozen && chn) croak("archive holds a chain index (chn.idx) but does not say frozen tables (blk.tables)");
    const bool shared_prior = a.get_long("seg.shared_prior", 0) == 1;
    if (frozen && !chn) croak("archive says frozen tables but holds no chain index (chn.idx)");
    if (version >= kBlockVersionMin) {
        const std::vector<uint8_t>* idx = a.find("blk.idx");
        if (!idx || !sfqc::unpack_block_index(*idx, blocks, version >= 8 ? SFQ_NSTREAMS : 10)) croak("bad block index");
        if (const std::vector<uint8_t>* h = a.find("blk.hdr")) first = *h;
        if (const std::vector<uint8_t>* si = a.find("seg.idx")) {
            size_t q = 0; uint64_t n = 0;
            if (!get_v(*si, q, n) || n == 0 || n > blocks.size()) croak("bad segment index");
            segs.resize((size_t)n);
            for (auto& g : segs) {
                g.chain_bytes = g.recpri_bytes = 0;
                if (!get_v(*si, q, g.nblocks) || !get_v(*si, q, g.prior_bytes) || !get_v(*si, q, g.raw_bytes)) croak("bad segment index");
                if (frozen && (!get_v(*si, q, g.chain_bytes) || !get_v(*si, q, g.recpri_bytes))) croak("bad segment index");
            }
        } else segs.push_back(Segment{blocks.size(), pri ? pri->size() : 0, (uint64_t)a.get_long("orig.size", 0), chn ? chn->size() : 0, rpr ? rpr->size() : 0});
        if (blocks.empty()) croak("bad block index (no blocks)");
    } else {
        sfq_block_info b; memset(&b, 0, sizeof b);
        b.n_records = (uint32_t)a.get_long("num_records");
        if (!b.n_records) croak("Zero records, what's going on?");
        b.llen = (uint32_t)a.get_long("llen");
        b.solid = a.get_long("usr.solid") != 0; b.two_id = a.get_long("usr.2id") != 0;
        b.n_byte = (uint8_t)a.get_long("gen.N_byte", 0);
        b.gen_bits = (uint8_t)level_gen_bits(level);
        std::string f = a.get("rec.first");
        first.assign(f.begin(), f.end());
        b.first_hdr_len = (uint32_t)first.size();
        if (a.get_long("num_records") > 0xFFFFFFFFll) croak("archive too large for the single-block GPU path (more than 2^32 - 1 records)");
        for (int s = 0; s < SFQ_NSTREAMS; s++) if (auto* v = a.find(sfq_stream_name(s))) {
            if (v->size() > 0xFFFFFFFFull) croak("archive too large for the single-block GPU path (stream %s has %zu bytes)", sfq_stream_name(s), v->size());
            b.size[s] = v->size();
        }
        blocks.push_back(b);
        segs.push_back(Segment{1, 0, (uint64_t)a.get_long("orig.size", 0), 0, 0});
    }
    FILE* of = stdout;
    if (!usr.empty()) {
        if (!o.overwrite && access(usr.c_str(), F_OK) == 0) {
            if (g_batch) croak("Can't write file '%s': File exists", usr.c_str());
            fprintf(stderr, "Can't write file '%s': File exists\n", usr.c_str()); exit(1);
        }
        of = fopen(usr.c_str(), "wb");
        if (!of) {
            if (g_batch) croak("Can't write file '%s'", usr.c_str());
            fprintf(stderr, "Can't write file '%s'\n", usr.c_str()); exit(1);
        }
    }
    sfq_params p; memset(&p, 0, sizeof p);
    p.level = level; p.version = version >= kBlockVersionMin ? kInternalVersion : (uint32_t)version;
    // walk the segments: each one's blocks, its slice of every stream (streams are segment-major), its prior
    size_t b0 = 0, pri_off = 0, chn_off = 0, rpr_off = 0;
    uint64_t spos[SFQ_NSTREAMS] = {0};
    std::vector<uint8_t> data;
    Bytes out;
    // several segments of known size: the text of one is written (a thread) while the next is decoded, out of two
    // page-locked buffers (kept for the life of the process, like the encoder's)
    static uint8_t* opin[2] = { nullptr, nullptr }; static size_t opin_cap = 0;
    size_t max_raw = 0; bool sized = segs.size() > 1;
    for (const Segment& g : segs) { max_raw = std::max<size_t>(max_raw, (size_t)g.raw_bytes); if (!g.raw_bytes) sized = false; }
    if (sized && opin_cap < max_raw + 64) {
        for (auto& q : opin) { if (q) sfq_host_free(ctx, q); q = nullptr; }
        opin[0] = (uint8_t*)sfq_host_alloc(ctx, max_raw + 64); opin[1] = (uint8_t*)sfq_host_alloc(ctx, max_raw + 64);
        opin_cap = (opin[0] && opin[1]) ? max_raw + 64 : 0;
        if (!opin_cap) sized = false;
        tick("pinned buffers");
    }
    std::atomic<int> wbad{0}; int ob = 0;
    Joiner writer;                                                      // (declared after what its thread uses: joined first when the scope unwinds)
    for (const Segment& g : segs) {
        if (g.nblocks == 0 || g.nblocks > blocks.size() - b0) croak("bad segment index");
        if (pri ? g.prior_bytes > pri->size() - pri_off : g.prior_bytes != 0) croak("bad segment index");
        if (chn ? g.chain_bytes > chn->size() - chn_off : g.chain_bytes != 0) croak("bad segment index");
        if (rpr ? g.recpri_bytes > rpr->size() - rpr_off : g.recpri_bytes != 0) croak("bad segment index");
        std::vector<sfq_block_info> sb(blocks.begin() + (ptrdiff_t)b0, blocks.begin() + (ptrdiff_t)(b0 + g.nblocks));
        const uint64_t rec0 = sb[0].first_record, h0 = sb[0].first_hdr_off;
        uint64_t need[SFQ_NSTREAMS] = {0}, hbytes = 0;
        for (auto& b : sb) { b.first_record -= rec0; b.first_hdr_off -= h0; hbytes += b.first_hdr_len; for (int s = 0; s < SFQ_NSTREAMS; s++) need[s] += b.size[s]; }
        if (h0 + hbytes > first.size()) croak("bad block index (first headers)");
        data.clear();
        uint64_t soff[SFQ_NSTREAMS];
        for (int s = 0; s < SFQ_NSTREAMS; s++) {
            soff[s] = data.size();
            if (!need[s]) continue;
            const std::vector<uint8_t>* v = a.find(sfq_stream_name(s));
            if (!v || spos[s] + need[s] > v->size()) croak("stream %s is shorter than its block index says", sfq_stream_name(s));
            data.insert(data.end(), v->begin() + (ptrdiff_t)spos[s], v->begin() + (ptrdiff_t)(spos[s] + need[s]));
            spos[s] += need[s];
        }
        // (seg.shared_prior: a segment without priors of its own codes from the previous segment's, e.g. the ranks of a multi-GPU job)
        if (g.prior_bytes) { if (sfq_set_qlt_prior(ctx, pri->data() + pri_off, g.prior_bytes)) croak("%s", sfq_last_error(ctx)); }
        else if (!shared_prior) sfq_set_qlt_prior(ctx, nullptr, 0);
        if (sfq_set_chain_index(ctx, g.chain_bytes ? chn->data() + chn_off : nullptr, g.chain_bytes)) croak("%s", sfq_last_error(ctx));
        if (g.recpri_bytes) { if (sfq_set_rec_prior(ctx, rpr->data() + rpr_off, g.recpri_bytes)) croak("%s", sfq_last_error(ctx)); }
        else if (!shared_prior) sfq_set_rec_prior(ctx, nullptr, 0);
        uint64_t cap = g.raw_bytes, got = 0;
        if (!cap) cap = data.size() * 8 + (1 << 20);
        sfq_result res;
        int rc = 0;
        uint8_t* dst = nullptr;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (sized && cap + 16 <= opin_cap) dst = opin[ob];
            else {
                if (!out.reserve((size_t)cap + 16)) croak("out of memory");
                out.touch((size_t)cap);
                dst = out.p;
            }
            rc = sfq_decode_blocks_host(ctx, &p, sb.data(), (uint32_t)sb.size(), first.data() + h0, hbytes,
                                        data.data(), data.size(), soff, dst, cap, &got, &res);
            if (rc != SFQ_E_OVERFLOW || got <= cap) break;
            cap = got;                                                 // the call reports the size it needs
        }
        if (rc) croak("%s", sfq_last_error(ctx));
        tick("sfq_decode_blocks_host");
        writer.join();
        if (wbad) croak("USR: Error writing output");
        if (dst == out.p) { if (fwrite(dst, 1, (size_t)got, of) != got) croak("USR: Error writing output"); }
        else {
            writer.start([dst, got, of, &wbad]() { if (fwrite(dst, 1, (size_t)got, of) != got) wbad = 1; });
            ob ^= 1;
        }
        tick("write output");
        b0 += g.nblocks; pri_off += g.prior_bytes; chn_off += g.chain_bytes; rpr_off += g.recpri_bytes;
    }
    writer.join();
    if (wbad) croak("USR: Error writing output");
    if (of != stdout) fclose(of); else fflush(stdout);
}

int main(int argc, char** argv) {
    std::string fil;
    Opts o;
    bool statistics = false;
    if (argc == 1) usage();
    for (int opt; (opt = getopt(argc, argv, "qPsvhdObzAF1234u:f:l:B:g:S:T:C:t:")) != -1;) {
        switch (opt) {
        case 'u': g_usr = optarg; break;
        case 'f': fil = optarg; break;
        case 'l': o.level = (int)strtoll(optarg, 0, 0); break;
        case '1': case '2': case '3': case '4': o.level = opt - '0'; break;
        case 'd': g_encode = false; break;
        case 'O': o.overwrite = true; break;
        case 'P': break;
        case 'q': o.quiet = true; break;
        case 'B': o.block_reads = strtol(optarg, 0, 0); break;
        case 'S': o.slab_bytes = (uint64_t)std::max<long long>(1, strtoll(optarg, 0, 0)) << 20; break;
        case 'g': o.device = atoi(optarg); break;
        case 'T': o.table_pct = std::min(90, std::max(1, atoi(optarg))); break;
        case 'b': g_batch = true; break;
        case 'z': g_timing = true; break;
        case 'A': o.adaptive = true; break;
        case 'F': o.force_frozen = true; break;
        case 'C': o.chain_reads = strtol(optarg, 0, 0); break;
        case 't': o.io_threads = std::min(64, std::max(1, atoi(optarg))); break;
        case 'v': printf("Version %s\nInternal format version=%u (block format %u)\n", kUserVersion, kInternalVersion, kBlockVersion); exit(0);
        case 'h': usage();
        case 's': statistics = true; g_encode = false; break;
        default: fprintf(stderr, "slimfastq: Ilagal args: use -h for help\n"); exit(1);
        }
    }
    o.level = clamp_level(o.level);                                    // clamp at parse time (the reference records the clamped value only)

    if (g_batch) {
        sfq_ctx* ctx = nullptr;
        const int rc = sfq_ctx_create(&ctx, o.device);
        if (rc) { fprintf(stderr, "slimfastq: no usable HIP device (error %d): this build has no CPU path\n", rc); return 1; }
        if (o.table_pct) sfq_ctx_set_table_budget(ctx, sfq_ctx_device_memory(ctx) / 100 * (uint64_t)o.table_pct);
        char* line = nullptr; size_t cap = 0; ssize_t n;
        int failed = 0;
        while ((n = getline(&line, &cap, stdin)) > 0) {
            while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
            if (!n) continue;
            char* tab = strchr(line, '\t');
            if (!tab) { printf("fail\t%s\texpected '<source>\\t<target>'\n", line); fflush(stdout); failed++; continue; }
            *tab = 0;
            const std::string src = line, dst = tab + 1;
            try {
                if (g_encode) encode_file(ctx, o, src, dst); else decode_file(ctx, o, dst, src);
                printf("ok\t%s\t%s\n", src.c_str(), dst.c_str());
            } catch (const JobError& e) {
                printf("fail\t%s\t%s\n", src.c_str(), e.msg.c_str());
                failed++;
            } catch (const std::exception& e) {                        // e.g. bad_alloc on a hostile archive: the job fails, the worker lives
                if (!g_partial.empty()) { unlink(g_partial.c_str()); g_partial.clear(); }
                printf("fail\t%s\t%s\n", src.c_str(), e.what());
                failed++;
            }
            fflush(stdout);
        }
        free(line);
        sfq_ctx_destroy(ctx);
        return failed ? 2 : 0;
    }

    while (optind < argc) {                                            // DWIM, config.cpp:279-325 (simplified)
        const char* file = argv[optind++];
        FILE* fh = fopen(file, "rb");
        char head[20] = {0};
        size_t cnt = fh ? fread(head, 1, 19, fh) : 0;
        if (fh) fclose(fh);
        if (cnt && fil.empty() && !strncmp(head, "whoami=slimfastq", 16)) { fil = file; g_encode = !g_usr.empty(); }
        else if (cnt && g_usr.empty() && head[0] == '@') g_usr = file;
        else if (!g_encode && g_usr.empty()) g_usr = file;
        else if (g_encode && fil.empty()) fil = file;
        else { fprintf(stderr, "What am I suppose to do with '%s'?\n (please specify explicitly with -f/-u prefix)\n", file); exit(1); }
    }
    if (fil.empty()) { fprintf(stderr, "Missing essential argument: -f\n"); exit(1); }

    try {
    std::string err;
    if (statistics) {                                                  // config.cpp:76-85
        sfqc::Archive a;
        if (!sfqc::read_file(fil, a, err)) croak("%s", err.c_str());
        fprintf(stderr, ":::: Info ::::\n");
        for (auto& kv : a.info) fprintf(stderr, "%-16s = %s\n", kv.first.c_str(), kv.second.c_str());
        fprintf(stderr, "\n:::: Files stream ::::\n i: name      : bytes\n");
        int i = 1;
        for (auto& s : a.streams) fprintf(stderr, "%2d: %-10s: %zu\n", i++, s.first.c_str(), s.second.size());
        return 0;
    }

    sfq_ctx* ctx = nullptr;
    tick("process start");
    const int rc = sfq_ctx_create(&ctx, o.device);
    if (rc) croak("no usable HIP device (error %d): this build has no CPU path", rc);
    tick("sfq_ctx_create");
    if (o.table_pct) sfq_ctx_set_table_budget(ctx, sfq_ctx_device_memory(ctx) / 100 * (uint64_t)o.table_pct);
    if (g_encode) encode_file(ctx, o, g_usr, fil); else decode_file(ctx, o, g_usr, fil);
    tick("done");
    } catch (const JobError& e) {                                      // croak(): config.cpp:54-68
        fprintf(stderr, "slimfastq: %s %s: %s\n", g_encode ? "encoding" : "decoding", g_usr.empty() ? "<< stdin >>" : g_usr.c_str(), e.msg.c_str());
        fflush(nullptr);
        _exit(1);
    }
    // a one-shot process: the driver reclaims the context's 10s of GB faster than freeing them one by one
    fflush(nullptr);
    _exit(0);
}
