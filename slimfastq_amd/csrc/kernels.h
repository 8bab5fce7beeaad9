// kernels.h -- launch wrappers exported by the .hip translation units to the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.h"

// Arguments shared by every model kernel.  Block b of a launch uses table slot (b - batch0).
struct ModelArgs {
    const u8*  fq;          // FASTQ text (encode) -- device
    const u64* line_off;    // line_off[k] = offset of line k; line_off[nlines] = nbytes
    BlockDesc* blocks;
    u32 nblocks;            // total
    u32 batch0, nbatch;     // this launch covers blocks [batch0, batch0+nbatch)
    u8* arena;              // per-block stream regions (BlockDesc::out_off/out_cap)
    i32 level;
    u32 lossless;           // the block format's rules (dev_common.h): string fallback for header numbers, "gen.lc", the '+' line check
    u32 epoch_base;         // block b runs with epoch epoch_base + b + 1
    // tables
    u32* q_slots; RowHdr* q_hdr; u32 q_rows;     // Log64 rows per slot: 4096 (level 1) or 65536
    u32* p_slots; RowHdr* p_hdr;                 // PR_ROWS PowerRanger rows per slot
    u32* g_tab;   u32 g_bits;                    // (1 << g_bits) Base2 dwords per slot
    // quality warm start (prior.hip); all null = cold rows, the reference's behaviour
    const u32* prior_w; const u32* prior_wovf;   // wave layout [q_rows][64] + overflow [q_rows][4]
    const u32* prior_ls; const RowHdr* prior_lh; // lane-per-block layout
    // format 6 with oversize records (frame.hip): the model kernels see the text without them; rec_map[k] = the file number
    // (0-based) of the k-th record they see -- the exception streams count records in file numbers (g_record_count).  Null: k itself.
    const u32* rec_map;
};
// g_record_count (config.cpp:43) of record r of a block that starts at base: block-relative, 1-based
__device__ __forceinline__ u64 rec_count_of(const ModelArgs& a, u64 r, u64 base) { return a.rec_map ? (u64)a.rec_map[r] + 1 : r - base + 1; }

// Decode-side extras.
struct DecodeArgs {
    ModelArgs m;
    const u8*  streams;                     // compact streams (device)
    const u64* blk_stream_off;              // [nblocks][SFQ_NSTREAMS] absolute offsets into streams
    const u8*  first_hdrs;                  // blob (device)
    u32* slen; u32* qlen;                   // per record (device), filled by the usr kernel
    u8*  pfg;  u8* pfq;                     // per record solid prefixes
    const u64* soff; const u64* qoff;       // exclusive scans of slen/qlen
    const u64* boff;                        // match model (gm.hip): soff[r] = boff[r] + r -- a '\n' behind every staged base line; null: soff is the plain scan
    u8*  seq_stage; u8* qual_stage;         // decoded bases / qualities
    u8*  hdr_stage; u32* hlen;              // decoded headers, back to back per block
    const u64* hdr_stage_off;               // per block base into hdr_stage
    const u32* hdr_stage_cap;
    u64* hoff;                              // per record offset into hdr_stage (filled by rec decode)
    u32  block_reads;                       // records per block (uniform; the last block may be short)
    u32  version;                           // archive format version (recs.cpp:400: < 5 takes load_pre5)
    u32  max_line;                          // a base / quality line longer than this is a corrupt stream (the caller's output capacity bounds it)
};

// ---- lane-per-chain kernels with frozen tables (chains.hip, dev_chain.h) ----------------------------------------
struct ChainGeoArgs { u32 chain_reads, cpb, nchains; };     // chain c = chain c % cpb of block c / cpb
#define RDEC_LDS_ROWS 16u              // header rows (decoder's form) a workgroup of the fast header decoder stages in LDS
#define GEN_MAX_GENERATIONS 40
struct ChainArgs {
    ModelArgs m;
    ChainGeoArgs geo;           // chains of the quality and base streams
    ChainGeoArgs rgeo;          // chains of the header stream (longer: each starts from the block's first header)
    u32* rhb;                   // [rgeo.nchains] header bytes of each header chain (encode: written)
    u64 nbytes;                 // size of the FASTQ text (encode)
    u32 block_reads;            // records per block (uniform; the last block may be short)
    u32* csz;                   // [nchains] sizes of the stream being coded (encode: written; decode: read)
    const u64* coff;            // decode: [nchains] absolute offsets of the chains' streams
    // quality: frozen rows, total 2^16 each
    const u32* qrows;           // [q_rows][64] cum | freq << 16, in symbol order, indexed by the context
    const u16* qdec;            // decode: [q_rows][72] u16: the cum of every 8th symbol, then the cums of symbols 1 .. 64 (chains.hip QDEC_ROW)
    u32 q_rows;                 // quality contexts: 4096 (level 1) or 65536
    u32 q_hot;                  // room for that many rows in the LDS image of the quality chains' workgroups (0 = no staging)
    const u8* qh_img;           // the hot image: map, then rows (chains.hip k_hot_select)
    const u32* qh_info;         // [0] = rows staged
    const u32* qesc;            // escape row (256 entries), qlts.cpp:80-86
    // headers: frozen PowerRanger rows, total 2^16 each
    const u32* rrows;           // [PR_REC_ROWS][256] cum | freq << 16
    const u16* rdec;            // decode: [PR_REC_ROWS][272] u16: the cum at every 16th symbol, then of all 256 (chains.hip RDEC_ROW)
    const u16* rmap; const u16* rhot; u32 r_hot;   // rows staged in LDS: row -> place by weight (0xFFFF = none; a kernel stages the first few), place -> row, how many to stage
    u8* exc_flag;               // encode: [records] set to 1 by the quality / base chains where a record holds a '!' / an N (null = not wanted)
    // chains that are PARTS of one record (long reads; chains.hip "segments"): seg_len != 0, geo.chain_reads = 1
    u32 seg_len;                // symbols per segment asked for; a record of M = max(bases, qualities) symbols is cut into
                                // n = ceil(M / seg_len) segments of ceil(M / n) symbols, a chain each, in both streams
    const u64* seg_off;         // [records + 1] chains before each record
    const u32* seg_rec;         // [nchains] the record a chain belongs to
    // bases: where a counting pass reads them (decode: the staged bases; null = the FASTQ text through line_off)
    const u8* st_buf; u64 st_bytes; const u64* st_off; const u32* st_len;
    // bases: generation tables
    u32 g_ngen;                 // 0 = every chain codes with the initial row
    u32 g_bound[GEN_MAX_GENERATIONS + 1];      // generation g = blocks [g_bound[g], g_bound[g + 1])
    const u32* g_rows[GEN_MAX_GENERATIONS];    // its rows (null = the initial row)
    const u32* g_init;          // encode: one dword holding the initial row (3, 3, 3, 3), read where a generation has no rows
    u32 flat_quads;             // bases without a model (round 5, "chn.idx" flag bit 6): a line's bases four at a time, one symbol of 4^k equally likely ones
    u32 flat_raw;               // bases without a model (round 5b, "chn.idx" flag bit 7, block format 10): a chain's bases two bits each, four a byte, no coder
                                // (k = 4, fewer at a line's end) instead of 3 of 12 a base
};
void launch_hot_rows(const u32* hist, const u32* rows66, const u32* qrows, u32 q_rows, u32 want, u32* ctot /* [q_rows] */,
                     u8* img /* q_rows / 4 bytes of map + want x 100 bytes of rows */, u32* info, hipStream_t st);
void launch_hot_rows_dec(const u32* rows66, const u16* qdec, u32 q_rows, u32 want, u32* ctot /* [q_rows] */,
                         u8* img /* q_rows / 4 bytes of map + want x 16 bytes of coarse lists */, u32* info, hipStream_t st);
u32 hot_rows_dec_max(void);     // how many coarse lists a workgroup's LDS holds beside the map
void launch_qlt_frozen_rows(const u32* rows66, u32 q_rows, u32* qrows, u16* qdec /* decode; may be null */, hipStream_t st);
void launch_qlt_encode_c(const ChainArgs& a, hipStream_t st);
void launch_gen_count(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line /* the longest base line (picks lane per record / per stretch) */,
                      u32* cnt, const u32* rows, const u16* log2fp, u64* cost, hipStream_t st,
                      u32 sub = 0 /* 0 every selected record, 1 every GEN_PRE-th of them, 2 the others */, u32 do_count = 1 /* 0: the cost only */);
void launch_gen_rows(const u32* cnt, u32* rows, u64 nctx, u32 step, hipStream_t st);
// ---- bases under the generation MATCH model (gm.hip, round 5) ----
void launch_gm_lens(const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 n, u32* slen, hipStream_t st);       // base-line lengths of records [0, n)
void launch_gm_soff(const u64* boff, u64 n, u64* soff, hipStream_t st);                                                       // soff[r] = boff[r] + r, r in [0, n]
void launch_gm_sentinels(u8* stage, const u64* soff, const u32* slen, u64 n, hipStream_t st);
void launch_gm_stage(const ModelArgs& m, u32 block_reads, u64 r0, u64 r1, u64 nbytes /* of the text */, u8* stage, const u64* soff, const u32* slen, u8* exc_flag /* or null */, hipStream_t st);
// (a: ChainArgs whose st_buf / st_off / st_len describe the stage)
void launch_gm_insert(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line, u64* T, u32 tb, hipStream_t st);
void launch_gm_plan(const ChainArgs& a, u64 nlanes /* records, or chains where they are segments */, const u8* stage, u64 stage_bytes, const u64* soff, const u32* slen,
                    const u64* T, u32 tb, u8* tok, hipStream_t st);
void launch_gm_price(const ChainArgs& a, u64 r0, u64 r1, u64 step, u32 max_line /* the call's longest base line: a lane per 256 bases of a record */, u64 lim_rec, const u8* stage, u64 stage_bytes, const u64* soff, const u32* slen, const u64* T, u32 tb,
                     const u16* costs /* hit[4], miss[4] in 1/1024 bit */, u64* cost /* [0] += cost, [1] += bases */, hipStream_t st);
void launch_gm_code(const ChainArgs& a, const u8* tok, hipStream_t st);
void launch_gm_decode_c(const ChainArgs& a, const struct DecodeArgs& da, u32 c0, u32 c1, u64 lim_rec, const u64* T, u32 tb, u64 stage_bytes, hipStream_t st);
// the same counts through bins (chains.hip "counting through bins": two streaming passes instead of a global atomic per base);
// rows_out != null: the rows of the sums are written as well (launch_gen_rows folded in).  bins / fill: scratch, fill zeroed by the caller
#define GEN_BIN_BATCH (1ull << 27)      /* keys a pair of passes takes at most (the bins hold twice that, 2 bytes a key) */
struct GenBins { u16* bins; u32* fill; u32 cap, g_bits; u64 batch, bins_bytes, fill_bytes; };
GenBins gen_bins_plan(u32 g_bits, u64 max_keys /* the bases a counting pass can meet (the call's text is a bound) */);
void launch_gen_count_binned(const ChainArgs& a, u32 b0, u32 b1, u64 nrec_range, u32 max_line, const GenBins& gb, u32* cnt, u32* rows_out, u32 step,
                             hipStream_t st, u32 sub = 0);
void launch_gen_encode_c(const ChainArgs& a, hipStream_t st, u32 c0 = 0, u32 c1 = 0 /* chains [c0, c1); 0, 0 = all */, bool flat = false /* every chain: the initial row */);
// gen.Ns / gen.Nn side streams, a wave per block (models_w.hip); flags: the records that may hold an exception (null = look at all)
// "chn.idx": the size lists csz[0 .. n) (lists [0, b1), [b1, b2), [b2, b3), [b3, n)) as zigzag-difference varints, back to back in
// out; len[n], off[n + 1], scan_tmp: scratch; info[0] = bytes before list b2, info[1] = all
void launch_chain_index_bytes(const u32* csz, u32 n, u32 b1, u32 b2, u32 b3, u32* len, u64* off, u64* scan_tmp, u8* out, u64* info, hipStream_t st);
void launch_gen_exc_w(const ModelArgs& a, const u8* flags, u32* ticket, hipStream_t st);
// the same lists as adaptive Rice codes (models_w.hip k_gen_exc_w<true>, dev_rice.h; frozen tables, "chn.idx" flag bit 4) and the way back, a lane per block (exc.hip)
void launch_gen_exc_r(const ModelArgs& a, const u8* flags, u32* ticket, hipStream_t st);
void launch_gen_exc_decode_r(const struct DecodeArgs& a, u32 nblocks, hipStream_t st);
#define REC_COUNT_COPIES 32u        // the header prior's counting pass counts into this many copies of the table (chains.hip k_rec_count_sum)
void launch_rec_count(const ModelArgs& a, u64 nrec, u64 stride, u32 run, u32 nruns, u32* cnt /* [REC_COUNT_COPIES][PR_REC_ROWS][256], zeroed; the sums end up in copy 0 */,
                      u32* flags /* [nruns], zeroed */, hipStream_t st);
void launch_rec_frozen_rows(const u32* f, u32 nrows, u32* rrows, u16* rdec /* the decoder's form; may be null */, hipStream_t st);
// counts -> the transmitted prior's frequencies f[nrows][256], the rows' sums rtot[nrows], and the nh heaviest rows (map[nrows], hot[nh])
void launch_rec_prior_freqs(const u32* cnt, u32 nrows, u32* f, u32* rtot, u32 nh, u16* map, u16* hot, hipStream_t st);
// one header chain per lane; a.csz / a.rhb per chain; max_hdr = the call's longest header (picks the LDS image of the fast kernel)
void launch_rec_encode_c(const ChainArgs& a, u32* flags /* [rgeo.nchains], zeroed */, u32* flags2 /* the same */, u32* tok /* rec_token_bytes(records) */, u32* ntok /* [rgeo.nchains] */,
                         u32 n_hot, u32 max_hdr, hipStream_t st,
                         u32 min_hdr = 0 /* the call's shortest header: over 127 = every chain to the general kernel, nothing else launched */,
                         hipEvent_t after_tokens = nullptr /* recorded behind the token step (the longest kernel of the five), for sfq_result.coder_ms */);
u64 rec_token_bytes(u64 nrec);
// segments: chains per record (from the text's line index, or from the decoder's line lengths), then the chains' records
void launch_seg_count(const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 nrec, u32 seg_len, u32* nseg, hipStream_t st);
void launch_seg_count_dec(const u32* slen, const u32* qlen, u64 nrec, u32 seg_len, u32* nseg, hipStream_t st);
void launch_seg_fill(const u64* seg_off, u64 nrec, u32* seg_rec, hipStream_t st);
void launch_chain_block_sizes(const ChainArgs& a, const ChainGeoArgs& geo, int stream, const u32* csz, const u32* rhb /* or null */, hipStream_t st);
void launch_compact_chains(const ChainArgs& a, const ChainGeoArgs& geo, int stream, u32 num, u32 den, const u32* csz, const u64* blk_stream_off,
                           const u64* stream_base, u8* out, hipStream_t st, const u32* gate = nullptr /* frame.hip k_stream_gate */);
#define GEN_STEP 4u             // a counted base adds GEN_STEP to its row entry (chains.hip)
// A generation of n records (its blocks x block_reads: the last block of a call may be short) is counted through every
// s-th record, s = ceil(n / GEN_COUNT_CAP): half a million records tell a row's shape, and the counting -- a
// scattered atomic per base into a table of 2^gen_bits x 16 bytes -- was most of the time of inputs whose bases can be
// learned (half of a 10 M-read call: 50 ms).  The decoder counts the same records.
#ifndef GEN_COUNT_CAP
#define GEN_COUNT_CAP 524288ull
#endif
#define GEN_PRE 8u                      /* the pre-verdict looks at every 8th of the records a counting pass takes (api.cpp gen_tables_begin) */
static inline u32 gen_count_stride(u64 n) { return (u32)((n + GEN_COUNT_CAP - 1) / GEN_COUNT_CAP ? (n + GEN_COUNT_CAP - 1) / GEN_COUNT_CAP : 1); }

// framing
void launch_max_u32(const u32* v, u64 n, u32* out /* raised to the largest of v */, hipStream_t st);
// format 6's oversize records (frame.hip, models_w.hip)
void launch_over_first(const u64* line_off, u64 nrec, u32* first /* 0xFFFFFFFF */, hipStream_t st);
void launch_over_solid(const u8* fq, const u64* line_off, u64 r, u32* out, hipStream_t st);
void launch_over_flags(const u64* line_off, u64 nrec, u32 solid, u32* flags, u32* kbytes, u32* status, hipStream_t st);
void launch_over_split(const u8* fq, const u64* line_off, u64 nrec, const u32* flags, const u64* fpos, const u64* koff, u8* filt, u32* rec_map, u32* over_list, hipStream_t st);
void launch_over_encode_w(const ModelArgs& a, const u8* fq, const u64* line_off, const u32* over_list, u32 n_over, const u64 out_off[3], const u32 out_cap[3], hipStream_t st);
void launch_over_mark(const u64* over_no, u32 n_over, u64 total, u32* flags /* zeroed */, u32* status, hipStream_t st);
void launch_over_map(const u32* flags, const u64* fpos, u64 total, u32* rec_map, hipStream_t st);
void launch_over_decode_w(const ModelArgs& a, const u8* stream, u32 size, u32 which, u32 store, u32 n_over, u64* cnt, u64* no, u64* piece, u8* txt, u64 cap, hipStream_t st);
void launch_over_place(u32 n_over, const u64* no, const u64* piece, const u8* lrec_txt, const u8* lgen_txt, const u8* lqlt_txt, const u64* roff_all, u8* out, hipStream_t st);
void launch_over_sizes(const u32* rec_map, const u32* rsize, u64 n_kept, const u64* no, const u64* piece, u32 n_over, u32* size_all, hipStream_t st);
void launch_gather_u64(const u64* src, const u32* idx, u64 n, u64* dst, hipStream_t st);
void launch_block_prepare(const u8* fq, const u64* line_off, u64 nrec, u32 block_reads, BlockDesc* blocks, u32 nblocks,
                          u64 nbytes, i32 level, i32 gen_bits_req, hipStream_t st);
// framing in one pass (frame.hip k_frame): the line index (at most cap entries behind entry 0 are written), the '@' / '+' checks into
// status[0], the marks of the exception pass; frame_out: { u64 lines, u32 look-back guard tripped }
u32 frame_tiles(u64 n);
void launch_frame(const u8* fq, u64 n, u64* tstat /* [frame_tiles(n)], zeroed */, u64* line_off, u64 cap, u32* status, u8* exc_flag /* or null */, u64 ecap /* its entries */,
                  void* frame_out /* 16 bytes, zeroed */, hipStream_t st);
void launch_validate_lines(const u64* line_off, u64 nrec, u32 max_hdr, u32 max_line, u32* status, hipStream_t st);
void launch_text_fingerprint(const u8* fq, u64 n, u64* out /* zeroed */, hipStream_t st);
#define FRAME_CHUNK 16384u

// generic exclusive scan u32 -> u64 (out has n+1 entries)
void launch_scan_u32(const u32* in, u64* out, u64 n, u64* tmp /* >= n/1024+2 */, hipStream_t st, u32 pad = 0 /* 2^k - 1: the values rounded up to multiples of 2^k */);

// models, lane-per-block reference kernels
void launch_qlt_encode_l(const ModelArgs& a, hipStream_t st);
void launch_gen_encode_l(const ModelArgs& a, hipStream_t st);
void launch_rec_encode_l(const ModelArgs& a, hipStream_t st);
void launch_usr_encode_l(const ModelArgs& a, hipStream_t st);
void launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t st);

// models, wave-per-block throughput kernels (persistent: one workgroup per pair of table slots -- a.nbatch is even --
// blocks 0..a.nblocks-1 are handed out through *ticket, which must be 0)
void launch_qlt_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_gen_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_rec_encode_w(const ModelArgs& a, u32* ticket_fast, u32* ticket_slow, hipStream_t st);
void launch_usr_encode_w(const ModelArgs& a, hipStream_t st);      // framing exceptions, a wave per block; blocks [batch0, batch0 + nbatch), slot = workgroup

void launch_qlt_decode_c(const ChainArgs& a, const DecodeArgs& da, hipStream_t st);
void launch_gen_decode_c(const ChainArgs& a, const DecodeArgs& da, u32 c0, u32 c1 /* chains [c0, c1) */, hipStream_t st);
void launch_rec_decode_c(const ChainArgs& a, const DecodeArgs& da, u32* flags /* [rgeo.nchains], zeroed; null = general path only */, hipStream_t st,
                         u32* dtok = nullptr /* rec_dtok_bytes(records); null = the lane kernels alone */, u32* dtoff = nullptr /* [records] */, u32* dflags = nullptr /* [rgeo.nchains], zeroed */);
u64 rec_dtok_bytes(u64 nrec);
void launch_gen_exc_decode_l(const DecodeArgs& a, hipStream_t st);          // applies gen.Ns / gen.Nn to the staged bases
void launch_gen_exc_decode_w(const DecodeArgs& a, hipStream_t st);          // the same, a wave per block (models_w.hip); blocks [batch0, batch0 + nbatch), slot = workgroup
void launch_usr_decode_l(const DecodeArgs& a, hipStream_t st, u32 prefilled = 0);          // prefilled: launch_usr_fill has written the blocks without framing exceptions
void launch_usr_decode_w(const DecodeArgs& a, hipStream_t st, u32 prefilled);     // the same, a wave per block (decode_w.hip); blocks [batch0, batch0 + nbatch), slot = workgroup
void launch_usr_fill(const DecodeArgs& a, u64 nrec, hipStream_t st);                                // block format: the records of blocks whose four usr.* streams are empty
void launch_qlt_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_gen_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_rec_decode_l(const DecodeArgs& a, hipStream_t st);
// the same chains, a wavefront per block (decode_w.hip): blocks [batch0, batch0 + nbatch), slot = workgroup
void launch_qlt_decode_w(const DecodeArgs& a, hipStream_t st);
void launch_gen_decode_w(const DecodeArgs& a, hipStream_t st);          // (gen_bits >= 6)
void launch_rec_decode_w(const DecodeArgs& a, hipStream_t st);

// packing
void launch_block_stream_offsets(BlockDesc* blocks, u32 nblocks, u64* blk_stream_off, u64* stream_total, u32 s0, u32 s1 /* streams [s0, s1) */, hipStream_t st);
void launch_compact(const BlockDesc* blocks, u32 nblocks, const u8* arena, const u64* blk_stream_off,
                    const u64* stream_base, u8* out, u32 skip_streams /* bit s: stream s is packed by launch_compact_chains */, hipStream_t st, const u32* gate = nullptr);
void launch_stream_gate(const BlockDesc* blocks, u32 nblocks, const u64* stream_total, u64 out_cap, u64* stream_base /* [SFQ_NSTREAMS] */, u32* gate /* [2]: go, worst status */, hipStream_t st);
void launch_record_sizes(const DecodeArgs& a, u64 nrec, u32* rsize, hipStream_t st);
void launch_assemble(const DecodeArgs& a, u64 nrec, const u64* roff, u8* out, hipStream_t st);
void launch_first_hdr_lens(const BlockDesc* blocks, u32 nblocks, u32* lens, hipStream_t st);
void launch_gather_first_hdrs(const BlockDesc* blocks, u32 nblocks, const u8* fq, const u64* blob_off, u8* blob, u64 cap, hipStream_t st);

// quality prior (prior.hip)
void launch_qlt_hist(const u8* fq, u64 nbytes, const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 nrec, u32 step,
                     int level, u32 cap /* symbols counted per sampled record */, u32* hist, hipStream_t st);
#define PRIOR_SYMBOLS 4096u
void launch_prior_rows(const u32* hist, u32 q_rows, u32* rows66, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st);
// the listed rows of rows66 back to back: slot[q_rows] scratch; list[4 + 67 n]: [0] = n, then per row its context and 66 words
void launch_prior_list(const u32* rows66, u32 q_rows, u32* slot, u32* list, hipStream_t st);
void launch_prior_scatter(const u32* ctxs, const u32* rows /* [n][66] */, u32 n, u32* rows66 /* zeroed */, hipStream_t st);
void launch_prior_spread(const u32* rows66, u32 q_rows, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st);
