// dev_common.h -- types shared by the host side of the C ABI and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "../../include/slimfastq_amd.h"

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t  i32;
typedef int64_t  i64;

// Limits of the reference's model path (usrs.hpp:34-36): longer lines go through its raw "oversize"
// side streams, which the block kernels do not implement (SFQ_E_UNSUPPORTED).
#define SFQ_MAX_ID_LLEN 0x2000
#define SFQ_MAX_GN_LLEN 0x10000

// ---- adaptive-table rows in HBM ---------------------------------------------------------------
// A ranger row is kept as 64 (Log64Ranger) or 256 (PowerRanger) dword slots, slot = freq | sym<<16,
// i.e. the reference's parallel arrays freq[]/syms[] (log64_ranger.hpp:45-49, power_ranger.hpp:43-48)
// interleaved so one coalesced dword-per-lane load fetches a whole Log64 row.  The scalar members
// (total, iend, count) live in a 16-byte header next to an epoch tag: a row whose epoch differs from
// the running block's epoch is "fresh" (the all-zero state of a new reference process), so tables are
// reused from block to block without clearing.
struct RowHdr {
    u32 total;
    u16 iend;
    u8  count;
    u8  pad;
    u32 epoch;
    u32 pad2;
};

// Epoch tags: plain row kernels and the wave quality kernel lay rows out differently inside the same
// table memory, so their tags live in disjoint ranges, and neither range can collide with slot data
// (freq | sym << 16 < 0x00400000).  Raw epochs stay below 2^30 (api.cpp advance_epoch).
#define EPOCH_L(e) (0x40000000u | (e))
#define EPOCH_W(e) (0x80000000u | (e))

#define L64_NSYM 64
#define PW_NSYM  256

// PowerRanger rows of one block slot.  Header model: ranger_t ranger[66] = {type, str, num[14]}
// (recs.hpp:42-48) -> row = field*16 + {0 type, 1 str, 2+k num.p[k]}.  Each XFile: 14 PowerRangerU
// rows + 1 string row (xfile.hpp:44-46) -> 16 rows.  Plus the quality escape row (qlts.hpp:45).
#define PR_REC_ROWS   (66 * 16)
#define PR_XF_BASE    PR_REC_ROWS
#define PR_XF_ROWS    16
#define PR_XF_COUNT   12         // gen.Ns gen.Nn rec.x usr.x usr.x.q usr.pfg usr.pfq gen.lc usr.lrec usr.lgen usr.lqlt (+1 spare)
#define PR_EXQ_ROW    (PR_XF_BASE + PR_XF_ROWS * PR_XF_COUNT)
#define PR_ROWS       (PR_EXQ_ROW + 8)   // padded

enum { XF_GEN_NS = 0, XF_GEN_NN = 1, XF_REC_X = 2, XF_USR_X = 3, XF_USR_XQ = 4, XF_USR_PFG = 5, XF_USR_PFQ = 6,
       XF_GEN_LC = 7, XF_USR_LREC = 8, XF_USR_LGEN = 9, XF_USR_LQLT = 10 };

// ---- per-block descriptor (device) ---------------------------------------------------------------
struct BlockDesc {
    u64 rec0;           // first record (global index)
    u32 nrec;
    u32 llen;           // usrs.cpp:265 (after the solid adjustment)
    u8  solid, two_id, gen_bits, pad;
    u32 status;         // 0 or SFQ_E_* (positive)
    u64 first_hdr_off;  // into the FASTQ buffer (text after '@')
    u32 first_hdr_len;
    u32 n_byte;         // gens.cpp:100-105
    u32 extra_hi;       // qlts.cpp:57-61
    u32 size[SFQ_NSTREAMS];
    u64 out_off[SFQ_NSTREAMS];  // where the block's stream lives in the scratch arena
    u32 out_cap[SFQ_NSTREAMS];
    u32 hdr_bytes;      // sum of header lengths in the block
};

// ---- the block format is lossless --------------------------------------------------------------------
// Where the reference would give back something else than it was given, the block format (sfq_params.block_reads != 0;
// ModelArgs::lossless) departs from the reference's bytes:
//  * a header field that numberwang (recs.cpp:192-262) types as a number although RecLoad::load (recs.cpp:430-456) would not
//    print the same bytes back -- an empty field (prints "0"), a decimal of more than 18 digits ("%lld" wraps or signs it) --
//    is coded as a string (ST_STR) instead; a header with a NUL inside (map_space stops there, recs.cpp:148) goes whole;
//  * lowercase bases (accepted gens.cpp:73-77, restored uppercase gens.cpp:171-178) are listed in the side stream "gen.lc";
//  * a '+' line that is neither empty nor the record's own header (usrs.cpp:236-239 keeps one flag per file) is refused.
// Format 6 (one block) keeps the reference's behaviour byte for byte.
// rec_number_prints_back: does a field of `len` bytes (first byte `c0`) typed as the number type `type` print back?
__device__ __forceinline__ bool rec_number_prints_back(u32 type, u32 len, u32 c0) {
    if (len == 0) return false;                                  // recs.cpp:209-210 types it decimal 0; recs.cpp:453-454 prints "0"
    const bool deci = type < 2u /* ST_STR */ || type >= 11u /* ST_DGT_Z */;
    const u32 digits = len - (c0 == '0' ? 1u : 0u);
    return !(deci && digits > 18u);                              // 10^18 < 2^63: no wrap in numberwang's loop, no sign from "%lld"
}
// the '+' line of record r: empty (two_id == 0) or the record's header again (two_id != 0); anything else the reference
// would replace by one of the two (usrs.cpp:236-239, 518-523)
__device__ __forceinline__ bool plus_line_is_regular(const u8* fq, const u64* line_off, u64 r, u32 two_id) {
    const u64 h0 = line_off[4 * r] + 1, h1 = line_off[4 * r + 1] - 1;
    const u64 p0 = line_off[4 * r + 2] + 1, p1 = line_off[4 * r + 3] - 1;
    if (!two_id) return p1 == p0;
    if (p1 - p0 != h1 - h0) return false;
    for (u64 i = 0; i < h1 - h0; i++) if (fq[h0 + i] != fq[p0 + i]) return false;
    return true;
}
__device__ __forceinline__ bool is_lower_base(u32 c) { return c == 'a' || c == 'c' || c == 'g' || c == 't' || c == 'n'; }

// character of a base line -> code (gens.cpp:72-77): 0..3, N-like -> 4, illegal -> 0x10
__device__ __forceinline__ u32 gen_code_of(u32 c) {        // gens.cpp:72-77: 0..3, N-like -> 4, illegal -> 0x10
    const u32 l = c | 0x20u;
    u32 n = 0x10u;
    n = (l == 'a' || c == '0') ? 0u : n;
    n = (l == 'c' || c == '1') ? 1u : n;
    n = (l == 'g' || c == '2') ? 2u : n;
    n = (l == 't' || c == '3') ? 3u : n;
    n = (l == 'n' || c == '.') ? 4u : n;
    return n;
}
