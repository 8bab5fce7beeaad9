// container.cpp -- see container.h.  Host-side file plumbing only; no entropy coding here.
#include "container.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sfqc {

static const size_t PAGE = 0x2000;            // filer.hpp:34
static const size_t NODE_IDS = PAGE / 4 - 1;  // filer.hpp:39 : 2047 ids, then the next-node id
static const size_t MAX_FILES = PAGE / 24;    // filer.cpp:49 : 341

#pragma pack(push, 1)
struct DirEnt { char name[8]; uint64_t size; uint32_t first; uint32_t node; };
#pragma pack(pop)
static_assert(sizeof(DirEnt) == 24, "directory entry is 24 bytes (filer.cpp:42-47)");

const char* Archive::get(const std::string& key) const {
    for (auto& kv : info) if (kv.first == key) return kv.second.c_str();
    return "";
}
long long Archive::get_long(const std::string& key, long long dflt) const {
    const char* s = get(key);
    return *s ? atoll(s) : dflt;                                       // config.cpp:127-130
}
void Archive::set(const std::string& key, const std::string& val) { info.emplace_back(key, val); }
void Archive::set(const std::string& key, long long val) { info.emplace_back(key, std::to_string(val)); }
const std::vector<uint8_t>* Archive::find(const std::string& name) const {
    for (auto& s : streams) if (s.first == name) return &s.second;
    return nullptr;
}
void Archive::add(const std::string& name, std::vector<uint8_t> bytes) { streams.emplace_back(name, std::move(bytes)); }
uint64_t Archive::payload_bytes() const {
    uint64_t n = 0;
    for (auto& kv : info) n += kv.first.size() + kv.second.size() + 2;
    for (auto& s : streams) n += s.second.size();
    return n;
}

// ---- reading ------------------------------------------------------------------------------------------
static bool read_chain(const uint8_t* img, size_t npages, const DirEnt& e, bool is_info, std::vector<uint8_t>& out, std::string& err) {
    out.clear();
    if (e.size > (uint64_t)npages * PAGE) { err = "corrupt directory: a stream larger than the file"; return false; }   // untrusted 64-bit size
    out.reserve((size_t)e.size);
    uint64_t left = e.size;
    auto page = [&](uint32_t id) -> const uint8_t* { return id < npages ? img + (size_t)id * PAGE : nullptr; };
    const uint8_t* pg = page(is_info ? 0 : e.first);
    const uint32_t* node = e.node ? (const uint32_t*)page(e.node) : nullptr;
    size_t ni = 0;
    while (left) {
        if (!pg) { err = "container: page chain leaves the file"; return false; }
        size_t take = left < PAGE ? (size_t)left : PAGE;
        out.insert(out.end(), pg, pg + take);
        left -= take;
        if (!left) break;
        if (!node) { err = "container: stream longer than its page chain"; return false; }
        if (ni == NODE_IDS) { node = (const uint32_t*)page(node[NODE_IDS]); ni = 0; if (!node) { err = "container: bad node chain"; return false; } }
        pg = page(node[ni++]);
    }
    return true;
}

bool parse_image(const uint8_t* img, size_t n, Archive& a, std::string& err) {
    a.info.clear(); a.streams.clear();
    if (n < 2 * PAGE) { err = "container: file too small"; return false; }
    const size_t npages = n / PAGE;
    const DirEnt* dir = (const DirEnt*)(img + PAGE);
    uint32_t count = dir[0].first;                                     // filer.cpp:95-96
    if (count == 0 || count > MAX_FILES) { err = "container: bad directory"; return false; }
    std::vector<uint8_t> text;
    if (!read_chain(img, npages, dir[0], true, text, err)) return false;
    size_t p = 0;                                                      // config.cpp:87-107
    while (p < text.size()) {
        size_t e = p;
        while (e < text.size() && text[e] != '\n' && text[e] != 0) e++;
        std::string line((const char*)&text[p], e - p);
        size_t eq = line.find('=');
        if (eq != std::string::npos) {
            std::string k = line.substr(0, eq);
            bool dup = false;
            for (auto& kv : a.info) if (kv.first == k) dup = true;     // std::map::insert keeps the first
            if (!dup) a.info.emplace_back(k, line.substr(eq + 1));
        }
        p = e + 1;
    }
    for (uint32_t i = 1; i < count; i++) {
        char nm[9]; memcpy(nm, dir[i].name, 8); nm[8] = 0;
        std::vector<uint8_t> bytes;
        if (!read_chain(img, npages, dir[i], false, bytes, err)) return false;
        a.streams.emplace_back(nm, std::move(bytes));
    }
    return true;
}

bool read_file(const std::string& path, Archive& a, std::string& err) {
    // a regular file is mapped and its page chains copied out of the mapping (reading it into a zero-filled buffer first
    // cost as much again as the copy); anything else is read
    unsigned long long n = 0;
    {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "Can't read file '" + path + "'"; return false; }
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
                n = (unsigned long long)sb.st_size;
                const bool ok = parse_image((const uint8_t*)m, (size_t)n, a, err);
                munmap(m, (size_t)sb.st_size);
                close(fd);
                if (!ok) return false;
                goto check;
            }
        }
        close(fd);
    }
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) { err = "Can't read file '" + path + "'"; return false; }
        std::vector<uint8_t> img;
        uint8_t buf[1 << 16];
        for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0; ) img.insert(img.end(), buf, buf + got);
        fclose(f);
        n = img.size();
        if (!parse_image(img.data(), img.size(), a, err)) return false;
    }
check:
    long long cs = a.get_long("comp.size", 0);                         // config.cpp:366-371
    if (cs > 0 && (unsigned long long)cs != n) { err = "expected compressed file size to be " + std::to_string(cs); return false; }
    return true;
}

// ---- writing ------------------------------------------------------------------------------------------
std::vector<uint8_t> build_image(const Archive& a) {
    // page budget: 2 + per stream (data pages + node pages)
    struct Lay { size_t pages, nodes; };
    std::vector<Lay> lay;
    size_t total = 2;
    for (auto& s : a.streams) {
        size_t pages = (s.second.size() + PAGE - 1) / PAGE;
        if (pages == 0) pages = 1;
        size_t nodes = pages > 1 ? (pages - 1 + NODE_IDS - 1) / NODE_IDS : 0;
        lay.push_back({pages, nodes});
        total += pages + nodes;
    }
    std::string text;
    bool has_size = false;
    for (auto& kv : a.info) { if (kv.first == "comp.size") has_size = true; }
    for (auto& kv : a.info) text += kv.first + "=" + kv.second + "\n";
    if (!has_size) text += "comp.size=" + std::to_string((unsigned long long)total * PAGE) + "\n";   // config.cpp:381-389
    std::vector<uint8_t> img(total * PAGE, 0);
    if (text.size() > PAGE) text.resize(PAGE);                        // (info beyond one page is not produced by this writer)
    memcpy(img.data(), text.data(), text.size());
    DirEnt* dir = (DirEnt*)(img.data() + PAGE);
    memset(dir, 0, PAGE);
    dir[0].size = text.size();
    dir[0].first = (uint32_t)(a.streams.size() + 1);
    size_t next = 2;
    for (size_t i = 0; i < a.streams.size() && i + 1 < MAX_FILES; i++) {
        const auto& s = a.streams[i];
        DirEnt& e = dir[i + 1];
        memset(e.name, 0, 8);
        memcpy(e.name, s.first.data(), s.first.size() < 8 ? s.first.size() : 8);
        e.size = s.second.size();
        const size_t pages = lay[i].pages, nodes = lay[i].nodes;
        const size_t data0 = next, node0 = next + pages;
        e.first = (uint32_t)data0;
        e.node = nodes ? (uint32_t)node0 : 0;
        for (size_t p = 0; p < pages; p++) {
            size_t off = p * PAGE, take = s.second.size() > off ? s.second.size() - off : 0;
            if (take > PAGE) take = PAGE;
            if (take) memcpy(img.data() + (data0 + p) * PAGE, s.second.data() + off, take);
        }
        for (size_t q = 0; q < nodes; q++) {
            uint32_t* node = (uint32_t*)(img.data() + (node0 + q) * PAGE);
            for (size_t k = 0; k < NODE_IDS; k++) {
                size_t pi = 1 + q * NODE_IDS + k;                      // page index within the stream
                node[k] = pi < pages ? (uint32_t)(data0 + pi) : 0;
            }
            node[NODE_IDS] = q + 1 < nodes ? (uint32_t)(node0 + q + 1) : 0;
        }
        next += pages + nodes;
    }
    return img;
}

bool write_file(const std::string& path, const Archive& a, std::string& err) {
    if (a.streams.size() + 1 > MAX_FILES) { err = "too many streams"; return false; }
    std::vector<uint8_t> img = build_image(a);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "Can't write file '" + path + "'"; return false; }
    bool ok = fwrite(img.data(), 1, img.size(), f) == img.size();
    ok = fclose(f) == 0 && ok;
    if (!ok) err = "write error on '" + path + "'";
    return ok;
}

// ---- writing as the streams grow ---------------------------------------------------------------------------
PagedWriter::~PagedWriter() { if (fd_ >= 0) close(fd_); }
bool PagedWriter::open(const std::string& path, std::string& err) {
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd_ < 0) { err = "Can't write file '" + path + "'"; return false; }
    wbuf_.reserve(17u << 20);
    return true;
}
int PagedWriter::stream(const std::string& name) { st_.emplace_back(); st_.back().name = name; st_.back().cur.reserve(PAGE); return (int)st_.size() - 1; }
bool PagedWriter::flush() {
    size_t at = 0;
    while (at < wbuf_.size() && !bad_) {
        const ssize_t w = pwrite(fd_, wbuf_.data() + at, wbuf_.size() - at, (off_t)(wbuf_page_ * PAGE + at));
        if (w <= 0) { bad_ = true; break; }
        at += (size_t)w;
    }
    wbuf_page_ += wbuf_.size() / PAGE;
    wbuf_.clear();
    return !bad_;
}
void PagedWriter::emit(const uint8_t* page) {                       // the page with id next_ - 1 (ids are handed out in emission order)
    wbuf_.insert(wbuf_.end(), page, page + PAGE);
    if (wbuf_.size() >= (16u << 20)) flush();
}
void PagedWriter::append(int id, const uint8_t* p, size_t n) {
    St& s = st_[(size_t)id];
    s.size += n;
    while (n) {
        if (s.cur.empty() && n >= PAGE) { s.pages.push_back(next_++); emit(p); p += PAGE; n -= PAGE; continue; }   // whole pages straight through
        const size_t take = std::min(n, PAGE - s.cur.size());
        s.cur.insert(s.cur.end(), p, p + take); p += take; n -= take;
        if (s.cur.size() == PAGE) { s.pages.push_back(next_++); emit(s.cur.data()); s.cur.clear(); }
    }
}
bool PagedWriter::finish(const std::vector<std::pair<std::string, std::string>>& info, std::string& err) {
    if (st_.size() + 1 > MAX_FILES) { err = "too many streams"; return false; }
    std::vector<uint8_t> page(PAGE);
    for (St& s : st_) {                                             // the partial last page (an empty stream still owns one page, as build_image lays it out)
        if (!s.cur.empty() || s.pages.empty()) {
            std::fill(page.begin(), page.end(), 0);
            if (!s.cur.empty()) memcpy(page.data(), s.cur.data(), s.cur.size());
            s.pages.push_back(next_++); emit(page.data()); s.cur.clear();
        }
    }
    std::vector<DirEnt> dir(MAX_FILES);
    memset(dir.data(), 0, dir.size() * sizeof(DirEnt));
    for (size_t i = 0; i < st_.size(); i++) {                       // node pages: the ids of a stream's pages after the first
        St& s = st_[i];
        DirEnt& e = dir[i + 1];
        memcpy(e.name, s.name.data(), std::min<size_t>(8, s.name.size()));
        e.size = s.size; e.first = s.pages[0]; e.node = 0;
        const size_t rest = s.pages.size() - 1;
        const size_t nodes = (rest + NODE_IDS - 1) / NODE_IDS;
        const uint32_t node0 = next_;
        if (nodes) e.node = node0;
        for (size_t q = 0; q < nodes; q++) {
            uint32_t* node = (uint32_t*)page.data();
            for (size_t k = 0; k < NODE_IDS; k++) { const size_t pi = 1 + q * NODE_IDS + k; node[k] = pi < s.pages.size() ? s.pages[pi] : 0; }
            node[NODE_IDS] = q + 1 < nodes ? (uint32_t)(node0 + q + 1) : 0;
            next_++; emit(page.data());
        }
    }
    if (!flush()) { err = "write error"; return false; }
    std::string text;
    bool has_size = false;
    for (auto& kv : info) if (kv.first == "comp.size") has_size = true;
    for (auto& kv : info) text += kv.first + "=" + kv.second + "\n";
    if (!has_size) text += "comp.size=" + std::to_string((unsigned long long)next_ * PAGE) + "\n";      // config.cpp:381-389
    if (text.size() > PAGE) text.resize(PAGE);
    std::vector<uint8_t> head(2 * PAGE, 0);
    memcpy(head.data(), text.data(), text.size());
    dir[0].size = text.size(); dir[0].first = (uint32_t)(st_.size() + 1);
    memcpy(head.data() + PAGE, dir.data(), MAX_FILES * sizeof(DirEnt));
    if (pwrite(fd_, head.data(), head.size(), 0) != (ssize_t)head.size()) { err = "write error"; return false; }
    const bool ok = close(fd_) == 0; fd_ = -1;
    if (!ok) err = "write error";
    return ok;
}

// ---- block index ----------------------------------------------------------------------------------------
static void put_v(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
static bool get_v(const std::vector<uint8_t>& b, size_t& p, uint64_t& v) {
    v = 0;
    for (int sh = 0; sh < 64; sh += 7) {
        if (p >= b.size()) return false;
        uint8_t c = b[p++];
        v |= (uint64_t)(c & 0x7f) << sh;
        if (!(c & 0x80)) return true;
    }
    return false;
}
std::vector<uint8_t> pack_block_index(const std::vector<sfq_block_info>& blocks) {
    std::vector<uint8_t> o;
    put_v(o, blocks.size());
    for (auto& b : blocks) {
        put_v(o, b.n_records); put_v(o, b.llen);
        put_v(o, (uint64_t)b.solid | ((uint64_t)b.two_id << 1)); put_v(o, b.n_byte); put_v(o, b.gen_bits);
        put_v(o, b.extra_hi); put_v(o, b.first_hdr_len); put_v(o, b.hdr_bytes);
        for (int s = 0; s < SFQ_NSTREAMS; s++) put_v(o, b.size[s]);
    }
    return o;
}
// nstreams: stream sizes per entry -- 10 in archives of block format 7, SFQ_NSTREAMS (14) since format 8
bool unpack_block_index(const std::vector<uint8_t>& bytes, std::vector<sfq_block_info>& blocks, int nstreams) {
    if (nstreams < 1 || nstreams > SFQ_NSTREAMS) return false;
    size_t p = 0; uint64_t n, v;
    if (!get_v(bytes, p, n) || n > (1u << 24)) return false;
    blocks.assign((size_t)n, sfq_block_info());
    uint64_t rec = 0, hoff = 0;
    for (auto& b : blocks) {
        memset(&b, 0, sizeof b);
        b.first_record = rec;
        if (!get_v(bytes, p, v)) return false; b.n_records = (uint32_t)v; rec += v;
        if (!get_v(bytes, p, v)) return false; b.llen = (uint32_t)v;
        if (!get_v(bytes, p, v)) return false; b.solid = v & 1; b.two_id = (v >> 1) & 1;
        if (!get_v(bytes, p, v)) return false; b.n_byte = (uint8_t)v;
        if (!get_v(bytes, p, v)) return false; b.gen_bits = (uint8_t)v;
        if (!get_v(bytes, p, v)) return false; b.extra_hi = (uint32_t)v;
        if (!get_v(bytes, p, v)) return false; b.first_hdr_len = (uint32_t)v; b.first_hdr_off = hoff; hoff += v;
        if (!get_v(bytes, p, v)) return false; b.hdr_bytes = (uint32_t)v;
        for (int s = 0; s < nstreams; s++) { if (!get_v(bytes, p, v)) return false; b.size[s] = v; }
    }
    return true;
}

}  // namespace sfqc
