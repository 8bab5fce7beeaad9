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
    u32 epoch_base;         // block b runs with epoch epoch_base + b + 1
    // tables
    u32* q_slots; RowHdr* q_hdr; u32 q_rows;     // Log64 rows per slot: 4096 (level 1) or 65536
    u32* p_slots; RowHdr* p_hdr;                 // PR_ROWS PowerRanger rows per slot
    u32* g_tab;   u32 g_bits;                    // (1 << g_bits) Base2 dwords per slot
    // quality warm start (prior.hip); all null = cold rows, the reference's behaviour
    const u32* prior_w; const u32* prior_wovf;   // wave layout [q_rows][64] + overflow [q_rows][4]
    const u32* prior_ls; const RowHdr* prior_lh; // lane-per-block layout
    // parked triples of the split kernels (dev_common.h TRIP_*): entries per block at trip_base(); counts per block
    u64* trip_q; u32* trip_g; u32* ntrip_q; u32* ntrip_g;
};

// Decode-side extras.
struct DecodeArgs {
    ModelArgs m;
    const u8*  streams;                     // compact streams (device)
    const u64* blk_stream_off;              // [nblocks][SFQ_NSTREAMS] absolute offsets into streams
    const u8*  first_hdrs;                  // blob (device)
    u32* slen; u32* qlen;                   // per record (device), filled by the usr kernel
    u8*  pfg;  u8* pfq;                     // per record solid prefixes
    const u64* soff; const u64* qoff;       // exclusive scans of slen/qlen
    u8*  seq_stage; u8* qual_stage;         // decoded bases / qualities
    u8*  hdr_stage; u32* hlen;              // decoded headers, back to back per block
    const u64* hdr_stage_off;               // per block base into hdr_stage
    const u32* hdr_stage_cap;
    u64* hoff;                              // per record offset into hdr_stage (filled by rec decode)
    u32  block_reads;                       // records per block (uniform; the last block may be short)
    u32  version;                           // archive format version (recs.cpp:400: < 5 takes load_pre5)
};

// framing
void launch_count_newlines(const u8* fq, u64 n, u32* chunk_counts, u32 nchunks, hipStream_t st);
void launch_write_newlines(const u8* fq, u64 n, const u64* chunk_base, u64* line_off, u32 nchunks, hipStream_t st);
void launch_validate_records(const u8* fq, const u64* line_off, u64 nrec, u32* status, hipStream_t st);
void launch_block_prepare(const u8* fq, const u64* line_off, u64 nrec, u32 block_reads, BlockDesc* blocks, u32 nblocks,
                          u64 nbytes, i32 level, i32 gen_bits_req, hipStream_t st);
#define FRAME_CHUNK 16384u

// generic exclusive scan u32 -> u64 (out has n+1 entries)
void launch_scan_u32(const u32* in, u64* out, u64 n, u64* tmp /* >= n/1024+2 */, hipStream_t st);

// models, lane-per-block reference kernels
void launch_qlt_encode_l(const ModelArgs& a, hipStream_t st);
void launch_gen_encode_l(const ModelArgs& a, hipStream_t st);
void launch_rec_encode_l(const ModelArgs& a, hipStream_t st);
void launch_usr_encode_l(const ModelArgs& a, hipStream_t st);
void launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t st);

// models, wave-per-block throughput kernels
// (persistent: grid = a.nbatch table slots; blocks 0..a.nblocks-1 are handed out through *ticket, which must be 0)
void launch_qlt_encode_w(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_qlt_encode_s(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_qlt_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st);   // 2 blocks per wave; a.nbatch even
void launch_gen_encode_w(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_gen_encode_k(const ModelArgs& a, u32* ticket, hipStream_t st);
int  gen_chains();                                                            // blocks per wave of that kernel (2)   // K blocks per wave (SFQ_GEN_CHAINS = 2/4/8); a.nbatch a multiple of 8
void launch_rec_encode_w(const ModelArgs& a, u32* ticket_fast, u32* ticket_slow, hipStream_t st);
// split form (default): the model kernels park (cum, freq, tot) triples, launch_rc_lanes codes them, one block per lane
void launch_qlt_model_s(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_gen_model_w(const ModelArgs& a, u32* ticket, hipStream_t st);
void launch_rc_lanes(const ModelArgs& a, bool quality, hipStream_t st);

void launch_usr_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_qlt_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_gen_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_rec_decode_l(const DecodeArgs& a, hipStream_t st);
void launch_gen_fixup(const DecodeArgs& a, u64 nrec, hipStream_t st);

// packing
void launch_block_stream_offsets(BlockDesc* blocks, u32 nblocks, u64* blk_stream_off, u64* stream_total, hipStream_t st);
void launch_compact(const BlockDesc* blocks, u32 nblocks, const u8* arena, const u64* blk_stream_off,
                    const u64* stream_base, u8* out, hipStream_t st);
void launch_record_sizes(const DecodeArgs& a, u64 nrec, u32* rsize, hipStream_t st);
void launch_assemble(const DecodeArgs& a, u64 nrec, const u64* roff, u8* out, hipStream_t st);
void launch_first_hdr_lens(const BlockDesc* blocks, u32 nblocks, u32* lens, hipStream_t st);
void launch_gather_first_hdrs(const BlockDesc* blocks, u32 nblocks, const u8* fq, const u64* blob_off, u8* blob, u64 cap, hipStream_t st);

// quality prior (prior.hip)
void launch_qlt_hist(const u8* fq, u64 nbytes, const u64* line_off, const BlockDesc* blocks, u32 block_reads, u64 nrec, u32 step,
                     int level, u32 cap /* symbols counted per sampled record */, u32* hist, hipStream_t st);
#define PRIOR_SYMBOLS 4096u
void launch_prior_rows(const u32* hist, u32 q_rows, u32* rows66, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st);
void launch_prior_spread(const u32* rows66, u32 q_rows, u32* w_rows, u32* w_ovf, u32* l_slots, RowHdr* l_hdr, hipStream_t st);
