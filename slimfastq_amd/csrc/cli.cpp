// cli.cpp -- `slimfastq-amd`: the reference's command line (config.cpp:161-277) over the C ABI.
//
//   -u fastq  -f file.sfq  -d  -O  -l N | -1..-4  -q  -s  -v  -h        (same meaning as the reference)
//   -B reads  : records per independent block (default 1024; 0 = one block = a format-6 file the
//               reference itself can decode)
//   -g dev    : HIP device
// All model / coder work happens in libslimfastq_amd.so on the GPU; this file parses arguments, reads and
// writes files and fills the info page.
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "container.h"

static const int   kInternalVersion = 6;      // config.cpp:41
static const int   kBlockVersion = 7;
static const char* kUserVersion = "2.04-amd";

static bool g_encode = true;
static std::string g_usr;

[[noreturn]] static void croak(const char* fmt, ...) {                 // config.cpp:54-68
    va_list ap; va_start(ap, fmt);
    fprintf(stderr, "slimfastq: %s %s: ", g_encode ? "encoding" : "decoding", g_usr.empty() ? "<< stdin >>" : g_usr.c_str());
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    exit(1);
}

static void usage() {
    printf("Usage: \n"
           "-u  usr-filename : (default: stdin)\n"
           "-f comp-filename : required - compressed\n"
           "-d               : decode (instead of encoding) \n"
           "-O               : silently overwrite existing files\n"
           "-l level         : compression level 1 to 4 (default is 3 ) \n"
           "-1, -2, -3, -4   : alias for -l 1, -l 2, etc \n"
           "-B reads         : records per independent GPU block (default 1024; 0 = single block, reference-compatible file)\n"
           "-g device        : HIP device index (default 0)\n"
           "-v               : version : internal version \n"
           "-h               : help : this message \n"
           "-s               : stat : information about a compressed file \n"
           "-q               : suppress extra stats info that could have been seen by -s \n"
           "\nslimfastq-amd A B : compress A (a fastq file) to B, or decompress A (a slimfastq file) to B / stdout\n");
    exit(0);
}

static bool read_all(FILE* f, std::vector<uint8_t>& out) {
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    return !ferror(f);
}

static int clamp_level(int l) { return l > 4 ? 4 : l < 1 ? 1 : l; }   // config.cpp:232-237
static int level_gen_bits(int level) { switch (level) { case 1: return 18; case 2: return 22; case 3: return 24; default: return 26; } }

int main(int argc, char** argv) {
    std::string fil;
    bool overwrite = false, statistics = false, quiet = false;
    int level = 3, device = 0;
    long block_reads = 1024;
    if (argc == 1) usage();
    for (int opt; (opt = getopt(argc, argv, "qPsvhdO1234u:f:l:B:g:")) != -1;) {
        switch (opt) {
        case 'u': g_usr = optarg; break;
        case 'f': fil = optarg; break;
        case 'l': level = (int)strtoll(optarg, 0, 0); break;
        case '1': case '2': case '3': case '4': level = opt - '0'; break;
        case 'd': g_encode = false; break;
        case 'O': overwrite = true; break;
        case 'P': break;
        case 'q': quiet = true; break;
        case 'B': block_reads = strtol(optarg, 0, 0); break;
        case 'g': device = atoi(optarg); break;
        case 'v': printf("Version %s\nInternal format version=%u (block format %u)\n", kUserVersion, kInternalVersion, kBlockVersion); exit(0);
        case 'h': usage();
        case 's': statistics = true; g_encode = false; break;
        default: croak("Ilagal args: use -h for help");
        }
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
    level = clamp_level(level);                                        // clamp at parse time (the reference records the clamped value only)
    if (block_reads < 0) block_reads = 0;

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
    int rc = sfq_ctx_create(&ctx, device);
    if (rc) croak("no usable HIP device (error %d): this build has no CPU path", rc);

    if (g_encode) {
        if (!overwrite && access(fil.c_str(), F_OK) == 0) { fprintf(stderr, "Can't write file '%s': File exists\n", fil.c_str()); exit(1); }
        std::vector<uint8_t> fq;
        FILE* in = g_usr.empty() ? stdin : fopen(g_usr.c_str(), "rb");
        if (!in) { fprintf(stderr, "Can't read file '%s'\n", g_usr.c_str()); exit(1); }
        if (!read_all(in, fq)) croak("read error");
        if (in != stdin) fclose(in);
        sfq_params p; memset(&p, 0, sizeof p);
        p.level = level; p.block_reads = (uint32_t)block_reads;
        p.prior_step = block_reads ? SFQ_PRIOR_AUTO : 0;               // warm start needs the block format
        std::vector<uint8_t> out((size_t)sfq_encode_bound(fq.size()));
        sfq_result res;
        rc = sfq_encode_blocks_host(ctx, fq.data(), fq.size(), &p, out.data(), out.size(), &res);
        if (rc) croak("%s", sfq_last_error(ctx));
        std::vector<sfq_block_info> blocks(res.n_blocks);
        sfq_get_block_index(ctx, blocks.data(), res.n_blocks);
        std::vector<uint8_t> first((size_t)res.first_hdr_bytes + 1);
        sfq_get_first_headers(ctx, first.data(), res.first_hdr_bytes);
        first.resize((size_t)res.first_hdr_bytes);

        sfqc::Archive a;                                               // info keys in the reference's order (config.cpp:334-347, usrs.cpp:262-266, recs.cpp:71, gens.cpp:104, usrs.cpp:405)
        const bool legacy = block_reads == 0;
        a.set("whoami", "slimfastq");
        a.set("version", legacy ? kInternalVersion : kBlockVersion);
        a.set("config.level", level);
        a.set("orig.filename", g_usr.empty() ? "<< stdin >>" : g_usr);
        if (!g_usr.empty() || !legacy) a.set("orig.size", (long long)fq.size());
        if (legacy) {
            const sfq_block_info& b = blocks[0];
            if (first.size() >= 400) croak("first header too long for the reference's info page (recs.cpp:30)");
            if (b.solid) a.set("usr.solid", 1);
            a.set("llen", b.llen);
            a.set("usr.2id", b.two_id);
            a.set("rec.first", std::string(first.begin(), first.end()));
            if (b.n_byte && b.n_byte != 'N') a.set("gen.N_byte", b.n_byte);
            a.set("num_records", (long long)res.n_records);
            if (!quiet && b.extra_hi) a.set("qlt.extra.hi", b.extra_hi);
        } else {
            a.set("blk.reads", block_reads);
            a.set("blk.count", (long long)res.n_blocks);
            a.set("num_records", (long long)res.n_records);
        }
        for (int s = 0; s < SFQ_NSTREAMS; s++) {
            if (!res.stream_bytes[s]) continue;
            const uint8_t* p0 = out.data() + res.stream_offset[s];
            a.add(sfq_stream_name(s), std::vector<uint8_t>(p0, p0 + res.stream_bytes[s]));
        }
        if (!legacy) {
            a.add("blk.idx", sfqc::pack_block_index(blocks));
            a.add("blk.hdr", first);
            const int64_t pn = sfq_get_qlt_prior(ctx, nullptr, 0);
            if (pn > 0) { std::vector<uint8_t> pri((size_t)pn); sfq_get_qlt_prior(ctx, pri.data(), pri.size()); a.add("qlt.pri", pri); }
        }
        if (!sfqc::write_file(fil, a, err)) croak("%s", err.c_str());
    } else {
        sfqc::Archive a;
        if (!sfqc::read_file(fil, a, err)) croak("%s", err.c_str());
        const int version = (int)a.get_long("version", 0);
        if (version > kBlockVersion) croak("%s was compressed with slimfastq version %d. My version is %d. Please upgrade me before decoing", fil.c_str(), version, kBlockVersion);
        if (a.find("usr.lrec")) croak("archive holds oversize records (usr.lrec): not supported by the GPU decoder yet");
        level = clamp_level((int)a.get_long("config.level", 2));       // config.cpp:359
        std::vector<sfq_block_info> blocks;
        std::vector<uint8_t> first;
        if (version >= kBlockVersion) {
            const std::vector<uint8_t>* idx = a.find("blk.idx");
            if (!idx || !sfqc::unpack_block_index(*idx, blocks)) croak("bad block index");
            if (const std::vector<uint8_t>* h = a.find("blk.hdr")) first = *h;
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
            for (int s = 0; s < SFQ_NSTREAMS; s++) if (auto* v = a.find(sfq_stream_name(s))) b.size[s] = (uint32_t)v->size();
            blocks.push_back(b);
        }
        std::vector<uint8_t> data; uint64_t soff[SFQ_NSTREAMS];
        for (int s = 0; s < SFQ_NSTREAMS; s++) {
            soff[s] = data.size();
            if (auto* v = a.find(sfq_stream_name(s))) data.insert(data.end(), v->begin(), v->end());
        }
        if (const std::vector<uint8_t>* pri = a.find("qlt.pri")) sfq_set_qlt_prior(ctx, pri->data(), pri->size());
        sfq_params p; memset(&p, 0, sizeof p);
        p.level = level; p.version = version >= kBlockVersion ? kInternalVersion : (uint32_t)version;
        uint64_t cap = (uint64_t)a.get_long("orig.size", 0), got = 0;
        if (!cap) cap = data.size() * 8 + (1 << 20);
        std::vector<uint8_t> out;
        sfq_result res;
        for (int attempt = 0; attempt < 2; attempt++) {
            out.resize((size_t)cap + 16);
            rc = sfq_decode_blocks_host(ctx, &p, blocks.data(), (uint32_t)blocks.size(), first.data(), first.size(),
                                        data.data(), data.size(), soff, out.data(), cap, &got, &res);
            if (rc != SFQ_E_OVERFLOW || got <= cap) break;
            cap = got;                                                 // the call reports the size it needs
        }
        if (rc) croak("%s", sfq_last_error(ctx));
        FILE* o = stdout;
        if (!g_usr.empty()) {
            if (!overwrite && access(g_usr.c_str(), F_OK) == 0) { fprintf(stderr, "Can't write file '%s': File exists\n", g_usr.c_str()); exit(1); }
            o = fopen(g_usr.c_str(), "wb");
            if (!o) { fprintf(stderr, "Can't write file '%s'\n", g_usr.c_str()); exit(1); }
        }
        if (fwrite(out.data(), 1, (size_t)got, o) != got) croak("USR: Error writing output");
        if (o != stdout) fclose(o);
    }
    sfq_ctx_destroy(ctx);
    return 0;
}
