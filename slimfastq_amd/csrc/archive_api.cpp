// archive_api.cpp -- the ".sfq" container behind the C ABI, for hosts that assemble an archive themselves
// (the writer rank of a multi-GPU job, a binding in another language).  Host only.
#include <cstring>
#include <string>
#include <vector>

#include "container.h"

extern "C" {

int sfq_archive_write(const char* path, const char* info_text, uint32_t n_streams,
                      const char* const* names, const uint8_t* const* data, const uint64_t* sizes) {
    if (!path || !info_text || (n_streams && (!names || !data || !sizes))) return SFQ_E_ARG;
    sfqc::Archive a;
    for (const char* p = info_text; *p;) {                         // "key=value\n" lines (config.cpp:334-347)
        const char* nl = strchr(p, '\n');
        const size_t len = nl ? (size_t)(nl - p) : strlen(p);
        const char* eq = (const char*)memchr(p, '=', len);
        if (eq) a.set(std::string(p, eq), std::string(eq + 1, p + len));
        p += len + (nl ? 1 : 0);
    }
    for (uint32_t s = 0; s < n_streams; s++) {
        if (!names[s] || strlen(names[s]) > 8 || (sizes[s] && !data[s])) return SFQ_E_ARG;     // directory names are 8 bytes (filer.cpp:42-47)
        a.add(names[s], std::vector<uint8_t>(data[s], data[s] + sizes[s]));
    }
    std::string err;
    return sfqc::write_file(path, a, err) ? SFQ_OK : SFQ_E_ARG;
}

int64_t sfq_pack_block_index(const sfq_block_info* blocks, uint32_t n, uint8_t* out, uint64_t cap) {
    if (n && !blocks) return SFQ_E_ARG;
    const std::vector<uint8_t> v = sfqc::pack_block_index(std::vector<sfq_block_info>(blocks, blocks + n));
    if (!out) return (int64_t)v.size();
    if (v.size() > cap) return SFQ_E_OVERFLOW;
    memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

}  // extern "C"
