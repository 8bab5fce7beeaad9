// container.h -- the paged ".sfq" single-file container (host plumbing around the hot path).
//
// Reads the reference's format-6 files and writes files the reference can read.  Layout facts restated
// from the reference (not its code): 8 KiB pages; page 0 = info text "key=value\n"; page 1 = directory
// of up to 341 entries {char name[8]; u64 size; u32 first; u32 node} (entry 0 = info stream, its `first`
// field holds the entry count on disk); a stream's first data page is `first`, further page ids come
// from `node` pages of 2047 ids + 1 next-node id.  (filer.hpp:34-42, filer.cpp:41-53, 88-97, 121-128,
// 217-242, 273-303.)
//
// Formats 7 and 8 (this project's block format; 8 = lossless, with the "gen.lc" stream and 14 stream sizes per index
// entry) keep the same container and stream names; every stream is the
// concatenation of its per-block parts, and two extra streams describe the blocks:
//   "blk.idx" : varint-coded sfq_block_info fields, one entry per block
//   "blk.hdr" : the blocks' first headers (the reference keeps one in info key "rec.first")
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/slimfastq_amd.h"

namespace sfqc {

struct Archive {
    std::vector<std::pair<std::string, std::string>> info;          // insertion order (first wins on lookup)
    std::vector<std::pair<std::string, std::vector<uint8_t>>> streams;   // directory order, without the info stream

    const char* get(const std::string& key) const;
    long long   get_long(const std::string& key, long long dflt = 0) const;
    void        set(const std::string& key, const std::string& val);
    void        set(const std::string& key, long long val);
    const std::vector<uint8_t>* find(const std::string& name) const;
    void        add(const std::string& name, std::vector<uint8_t> bytes);
    uint64_t    payload_bytes() const;                               // sum of stream sizes + info text
};

// I/O.  On failure return false and set err.
bool read_file(const std::string& path, Archive& a, std::string& err);
bool parse_image(const uint8_t* img, size_t n, Archive& a, std::string& err);
bool write_file(const std::string& path, const Archive& a, std::string& err);   // sets comp.size
std::vector<uint8_t> build_image(const Archive& a);

// The same file written as the streams grow (the reference's FilerSave does the same: pages are handed out in the order
// they fill, so the streams' pages interleave; filer.cpp:217-242): append() as data arrives, finish() writes the partial
// pages, the node pages, the directory and the info page.  Nothing of a stream is kept in memory but its page list.
class PagedWriter {
public:
    ~PagedWriter();
    bool open(const std::string& path, std::string& err);
    int  stream(const std::string& name);                                  // a new directory entry; returns its id
    void append(int id, const uint8_t* p, size_t n);
    bool finish(const std::vector<std::pair<std::string, std::string>>& info, std::string& err);   // adds comp.size
private:
    struct St { std::string name; uint64_t size = 0; std::vector<uint32_t> pages; std::vector<uint8_t> cur; };
    void emit(const uint8_t* page);
    bool flush();
    int fd_ = -1; bool bad_ = false;
    uint32_t next_ = 2;                  // page ids 0 and 1 are the info page and the directory
    uint64_t wbuf_page_ = 2;             // page id of wbuf_'s first page
    std::vector<uint8_t> wbuf_;
    std::vector<St> st_;
};

// block index <-> "blk.idx"
std::vector<uint8_t> pack_block_index(const std::vector<sfq_block_info>& blocks);
bool unpack_block_index(const std::vector<uint8_t>& bytes, std::vector<sfq_block_info>& blocks, int nstreams = SFQ_NSTREAMS);

}  // namespace sfqc
