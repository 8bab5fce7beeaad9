// synth.cpp -- deterministic synthetic FASTQ, host only.
// Every read is generated from (seed, read index) alone, so any range of reads can be produced
// independently (per rank, per thread) and the concatenation is reproducible.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/slimfastq_amd.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {                       // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

void put_u(std::string& o, uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); o.append(b, (size_t)n); }

// kind 3: the bases come from a seeded random genome of GENOME_BP bases (never stored: base p is a hash of p),
// read = a uniformly placed window, either strand, 0.5 % substitutions -- coverage = reads x length / GENOME_BP
// (30x at 2 M reads of 150 bp).  This is the honest stress of the base model: the reference's file-wide
// context table learns the genome as coverage accumulates, an independent block cannot.
const uint64_t GENOME_BP = 10000000ull;
inline uint32_t genome_base(uint64_t seed, uint64_t p) {
    uint64_t z = (seed * 0x9E3779B97F4A7C15ull) ^ (p * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull);
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    return (uint32_t)(z >> 61) & 3u;
}

// 150 bp-style Illumina read: '@SIM.<i> M7:<run>:<flowcell>:<lane>:<tile>:<x>:<y> 1:N:0:<idx>'
void illumina_read(uint64_t idx, uint32_t len, uint64_t seed, std::string& o, bool binned = false, bool genome = false) {
    Rng r(seed ^ (idx * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    o += "@SIM."; put_u(o, idx + 1);
    o += " M7:42:000000000-A7XYZ:"; put_u(o, 1 + (idx / 2500000) % 8);
    o += ':'; put_u(o, 1101 + (idx / 5000) % 96);
    o += ':'; put_u(o, 1000 + r.below(29000));
    o += ':'; put_u(o, 1000 + r.below(24000));
    o += " 1:N:0:ATCACG\n";
    const size_t b0 = o.size();
    o.resize(b0 + len);
    std::string q(len, '#');
    // qualities: first-order Markov chain over Q2..Q41 with a position-dependent downward drift and a
    // terminal "B-tail" of Q2 once the read has collapsed
    int cur = 34 + (int)r.below(7);
    const uint32_t knee = len / 3 + r.below(len);          // where the decline starts
    const uint64_t gspan = GENOME_BP > len ? GENOME_BP - len : 1;
    const uint64_t gpos = genome ? r.next() % gspan : 0;
    const bool rev = genome && (r.next() >> 63);
    bool tail = false;
    for (uint32_t i = 0; i < len; i++) {
        if (!tail) {
            uint32_t u = r.below(1000);
            if (u >= 400) {
                int step;
                if (i < knee) step = (u < 650) ? 1 : (u < 880) ? -1 : (u < 950) ? -2 : (u < 985) ? -4 : 2;
                else          step = (u < 560) ? 1 : (u < 800) ? -1 : (u < 920) ? -3 : (u < 975) ? -6 : -10;
                cur += step;
                if (cur > 41) cur = 41;
                if (cur < 2) cur = 2;
            }
            if (cur <= 4 && i > knee) tail = true;
        }
        const int qv = tail ? 2 : cur;
        // kind 2: NovaSeq-style 4-level binning (Q2, Q12, Q23, Q37)
        q[i] = (char)('!' + (binned ? (qv < 3 ? 2 : qv < 15 ? 12 : qv < 30 ? 23 : 37) : qv));
        const uint32_t v = (uint32_t)(r.next() >> 40);
        char base = "ACGT"[v & 3];
        if (genome) {
            uint32_t g = rev ? 3u - genome_base(seed, gpos + (len - 1 - i)) : genome_base(seed, gpos + i);
            if (((v >> 12) % 200) == 0) g = (g + 1 + ((v >> 2) & 3) % 3) & 3;       // 0.5 % substitutions
            base = "ACGT"[g];
        }
        if ((v >> 2) % 1000 == 0) { base = 'N'; q[i] = '#'; }   // P(N) = 1e-3, N gets quality '#'
        o[b0 + i] = base;
    }
    o += "\n+\n";
    o += q;
    o += '\n';
}

// Nanopore-style long read: UUID header, length log-uniform in [10000, 50000], Q in [1, 60]
void long_read(uint64_t idx, uint64_t seed, std::string& o) {
    Rng r(seed ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
    char b[64];
    o += '@';
    const uint64_t a = r.next(), c = r.next();
    snprintf(b, sizeof b, "%08x-%04x-%04x-%04x-%012llx", (unsigned)(a >> 32), (unsigned)(a >> 16) & 0xffff, (unsigned)a & 0xffff,
             (unsigned)(c >> 48), (unsigned long long)(c & 0xffffffffffffull));
    o += b;
    o += " runid=e28af1fe5fa6e55295e7c381a135f5eea5a0cb19 read="; put_u(o, idx * 7 % 100000);
    o += " ch="; put_u(o, 1 + r.below(512));
    o += " start_time=2019-08-29T16:50:10Z flow_cell_id=FAL33580 protocol_group_id=PAN-0230-5 sample_id=pooled\n";
    // log-uniform length
    double u = r.unit();
    double lnlen = 9.210340371976184 /* ln 1e4 */ + u * 1.6094379124341003 /* ln 5 */;
    uint32_t len = 10000;
    { double e = 1.0, x = lnlen - 9.210340371976184, term = 1.0; for (int k = 1; k < 30; k++) { term *= x / k; e += term; } len = (uint32_t)(10000.0 * e); }
    if (len > 50000) len = 50000;
    const size_t b0 = o.size();
    o.resize(b0 + len);
    std::string q(len, '"');
    int cur = 12 + (int)r.below(20);
    for (uint32_t i = 0; i < len; i++) {
        uint32_t v = (uint32_t)(r.next() >> 36);
        o[b0 + i] = "ACGT"[v & 3];
        int step = (int)((v >> 2) % 9) - 4;
        if (((v >> 8) & 3) == 0) cur += step;
        if (cur < 1) cur = 1;
        if (cur > 60) cur = 60;
        q[i] = (char)('!' + cur);
    }
    o += "\n+\n";
    o += q;
    o += '\n';
}

}  // namespace

extern "C" int64_t sfq_synth_fastq(uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed, int kind,
                                   uint8_t* h_out, uint64_t cap) {
    if (kind < 0 || kind > 3) return SFQ_E_ARG;
    if (kind != 1 && (read_len == 0 || read_len > 65000)) return SFQ_E_ARG;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (n_reads < 4096) nt = 1;
    std::vector<std::string> parts(nt);
    std::vector<std::thread> th;
    const uint64_t per = (n_reads + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        th.emplace_back([&, t]() {
            const uint64_t a = std::min<uint64_t>(n_reads, (uint64_t)t * per), b = std::min<uint64_t>(n_reads, a + per);
            std::string& o = parts[t];
            if (kind != 1) o.reserve((size_t)(b - a) * (2 * read_len + 64));
            for (uint64_t i = a; i < b; i++) {
                if (kind == 1) long_read(first_read + i, seed, o);
                else illumina_read(first_read + i, read_len, seed, o, kind == 2, kind == 3);
            }
        });
    }
    for (auto& x : th) x.join();
    uint64_t total = 0;
    for (auto& s : parts) total += s.size();
    if (!h_out) return (int64_t)total;
    if (total > cap) return SFQ_E_OVERFLOW;
    uint64_t o = 0;
    for (auto& s : parts) { memcpy(h_out + o, s.data(), s.size()); o += s.size(); }
    return (int64_t)total;
}
