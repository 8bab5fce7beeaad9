/*
 * slimfastq_amd.h -- C ABI of the MI355X-native slimfastq hot path.
 *
 * This is the drop-in boundary: everything the reference does between `UsrSave::encode()` /
 * `UsrLoad::decode()` (usrs.cpp:392-407 / 539-574) and `FilerSave::put()` / `FilerLoad::get()`
 * (filer.hpp:70-75, 94-97) -- i.e. the stream models qlts.cpp / gens.cpp / recs.cpp, the rangers
 * (base2_ranger.hpp, log64_ranger.hpp, power_ranger.hpp), the range coder (coder.hpp) and the
 * exception side streams (xfile.cpp) -- runs behind these entry points as hand-written gfx950 HIP
 * kernels.  The reference has no FFI of its own; each entry point names the C++ call it replaces.
 *
 * Conventions: extern "C", plain pointers and sizes, int status (0 = ok, <0 = error, never exit(),
 * no exceptions across the boundary).  The caller owns every buffer it passes; the context owns
 * device tables, scratch and its HIP stream.  One context per host thread per GPU; calls on
 * different contexts are concurrent-safe.  Pointers named d_* are DEVICE pointers, h_* are host
 * pointers.  There is no CPU fallback: without a HIP device sfq_ctx_create fails.
 */
#ifndef SLIMFASTQ_AMD_H
#define SLIMFASTQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFQ_ABI_VERSION 3

/* status codes */
#define SFQ_OK              0
#define SFQ_E_ARG          -1   /* bad argument                                              */
#define SFQ_E_HIP          -2   /* HIP runtime error / no device                             */
#define SFQ_E_NOMEM        -3   /* device or host allocation failed                          */
#define SFQ_E_FORMAT       -4   /* input is not 4-line FASTQ (reference: croak, usrs.cpp:162-172) */
#define SFQ_E_OVERFLOW     -5   /* an output arena was too small                             */
#define SFQ_E_CORRUPT      -6   /* compressed stream inconsistent                            */
#define SFQ_E_UNSUPPORTED  -7   /* e.g. the block format and a '+' line that is neither empty nor the record's header (the reference
                                   would silently replace it, usrs.cpp:236-239 / 518-523)                                       */
#define SFQ_E_GENCHAR      -8   /* unexpected genome char (gens.cpp:125-126) / switched N byte (gens.cpp:107-108) */

/* Stream ids: the reference's stream names (FilerSave(name) call sites). */
enum sfq_stream {
    SFQ_S_REC = 0,      /* "rec"     recs.cpp:40      */
    SFQ_S_GEN = 1,      /* "gen"     gens.cpp:63      */
    SFQ_S_QLT = 2,      /* "qlt"     qlts.cpp:46      */
    SFQ_S_GEN_NS = 3,   /* "gen.Ns"  gens.cpp:69      */
    SFQ_S_GEN_NN = 4,   /* "gen.Nn"  gens.cpp:70      */
    SFQ_S_REC_X = 5,    /* "rec.x"   recs.cpp:44      */
    SFQ_S_USR_X = 6,    /* "usr.x"   usrs.cpp:48      */
    SFQ_S_USR_XQ = 7,   /* "usr.x.q" usrs.cpp:49      */
    SFQ_S_USR_PFG = 8,  /* "usr.pfg" usrs.cpp:50      */
    SFQ_S_USR_PFQ = 9,  /* "usr.pfq" usrs.cpp:51      */
    SFQ_S_GEN_LC = 10,  /* "gen.lc"  block format only: positions of the lowercase bases (the reference accepts them,
                           gens.cpp:73-77, and gives them back uppercase, gens.cpp:171-178), gap-coded like "gen.Ns"     */
    SFQ_S_USR_LREC = 11,/* "usr.lrec" usrs.cpp:55: oversize records -- gap, header line, '+' line, raw            */
    SFQ_S_USR_LGEN = 12,/* "usr.lgen" usrs.cpp:53: their base lines                                              */
    SFQ_S_USR_LQLT = 13,/* "usr.lqlt" usrs.cpp:54: their quality lines                                           */
    SFQ_NSTREAMS = 14
};
const char* sfq_stream_name(int stream);

/* Which models to run (config C2 of BASELINE.json runs SFQ_M_QLT alone). */
#define SFQ_M_REC 1u
#define SFQ_M_GEN 2u
#define SFQ_M_QLT 4u
#define SFQ_M_USR 8u
#define SFQ_M_ALL 15u

typedef struct sfq_ctx sfq_ctx;

typedef struct sfq_params {
    int32_t  level;        /* 1..4 : conf.level (config.cpp:260-263, clamped like config.cpp:232-237) */
    uint32_t block_reads;  /* records per independent block; 0 = a single block, i.e. streams that are
                              byte-identical to the reference's own (format 6), its quirks included: lowercase bases
                              come back uppercase, an emptied header field comes back "0" (SURVEY H7);
                              SFQ_BLOCK_AUTO = about 376 KiB of text per block (1024 records of 150 bp, a handful of
                              long reads).  The block format is LOSSLESS: where the reference would alter the text,
                              its blocks depart from the reference's bytes (a header field that would not print back
                              is coded as a string, lowercase bases are listed in "gen.lc") or refuse the input    */
    int32_t  gen_bits;     /* base-model context bits; 0 = the level's (gens.hpp:43-53) capped by block size.  (The table they size is
                              the adaptive modes' and kernel = 2's; the default frozen mode's match model has no table of rows) */
    uint32_t models;       /* SFQ_M_* mask; 0 = SFQ_M_ALL                                             */
    uint32_t kernel;       /* 0 = default kernels; 1 = lane-per-block reference kernels (the reference loop on one lane:
                              slow, the cross-check of the default kernels; adaptive tables); 2 = the default kernels, but
                              with frozen tables the base exceptions (gen.Ns / gen.Nn / gen.lc) keep the reference's own
                              coding -- XFile streams through adaptive PowerRanger rows, what archives written before
                              round 4 hold -- instead of Rice-coded gap lists ("chn.idx" flag bit 4), and the
                              BASES keep round 4's coding: generation tables of Base2 rows where they pay, the initial row's 3 of 12
                              a base where they do not -- instead of the generation match model ("chn.idx" flag bit 5: a chain
                              follows a pointer into the earlier generations' bases, gm.hip) and, where they have nothing to learn, two bits a
                              base without a coder (bit 7, block format 10)   */
    uint32_t version;      /* decode only: archive "version" info key (config.cpp:373); 0 = current (6).
                              Versions < 5 take RecLoad::load_pre5 (recs.cpp:400-401)                      */
    uint32_t prior_step;   /* encode, block mode only: 0 = cold blocks (each block == the reference run on that block);
                              N > 0 = warm start: quality rows start from a prior counted over every N-th record
                              (its first 4096 quality symbols)
                              (SFQ_PRIOR_AUTO picks N from the input size; frozen tables always have a prior)       */
    uint32_t tables;       /* block mode only: SFQ_TABLES_ADAPTIVE (0) = every block runs the reference's adaptive rows
                              (Log64Ranger / Base2Ranger updated per symbol), a wavefront per block;
                              SFQ_TABLES_FROZEN (1) = the rows are built by counting passes and frozen while a chain
                              is coded -- qualities from the transmitted prior, bases from the counts of the earlier
                              generations of the same call (round 5: read as such -- the match model, DESIGN.md 4.3 --, not
                              counted into a table) -- one chain per LANE                              */
    uint32_t chain_reads;  /* frozen tables: records per chain; 0 = automatic (about 205 000 chains a call, of 4 KiB of text or more; long
                              reads -- fewer than 204 800 records, each a chain's worth or more -- are cut into SEGMENTS of one record);
                              SFQ_CHAIN_SEGMENT(n): chains of (at most) n quality symbols / bases of ONE record                     */
    uint32_t lds_rows;     /* frozen tables: quality rows staged in LDS by every workgroup of the quality chains.  Encode: the N most used
                              rows (up to 1024); 0 = automatic (800 rows where the call has 150 000 chains or more: 17.3 -> 15.9 ms per 3.7 GB
                              call), SFQ_LDS_ROWS_NONE = none.  Decode: the coarse lists (16 bytes) of the N contexts the prior gives the most
                              weight (up to 6000); 0 = automatic (6000 where the call has 150 000 chains or more: 23.0 -> 20.0 ms),
                              SFQ_LDS_ROWS_NONE = none.  Where the rows come from never shows in the streams.                          */
} sfq_params;
#define SFQ_CHAIN_SEGMENT_FLAG 0x80000000u
#define SFQ_CHAIN_SEGMENT(n) (SFQ_CHAIN_SEGMENT_FLAG | (uint32_t)(n))
#define SFQ_TABLES_ADAPTIVE 0u
#define SFQ_TABLES_FROZEN   1u
#define SFQ_TABLES_AUTO     2u   /* by the size of the text: frozen tables from 64 MiB on; below that their transmitted priors weigh too
                                    much (samples/tst7.fq, 3.9 MB: 1.15 x the reference's bytes) and a GPU has nothing to win on so
                                    little text -- adaptive tables in blocks of 65536 records (SFQ_BLOCK_AUTO), i.e. the reference's
                                    own streams for a file of up to that many records */
#define SFQ_LDS_ROWS_NONE  0xFFFFFFFFu
#define SFQ_PRIOR_AUTO  0xFFFFFFFFu
#define SFQ_PRIOR_GIVEN 0xFFFFFFFEu  /* prior_step: the priors installed with sfq_set_qlt_prior / sfq_set_rec_prior */
#define SFQ_PRIOR_COUNTS 0xFFFFFFFDu /* prior_step: the sample COUNTS installed with sfq_set_prior_counts stand for this call's own sample */
#define SFQ_BLOCK_AUTO 0xFFFFFFFFu

/* One entry per block: what a decoder needs besides the stream bytes (the "block index").
 * The per-block stream bytes are the reference's streams for a FASTQ consisting of that block alone,
 * with g_record_count / g_genofs_count (config.cpp:43-44) restarting at the block. */
typedef struct sfq_block_info {
    uint64_t first_record;                 /* 0-based index of the block's first record           */
    uint32_t n_records;                    /* num_records (usrs.cpp:405)                          */
    uint32_t llen;                         /* "llen"      (usrs.cpp:265)                          */
    uint8_t  solid;                        /* "usr.solid" (usrs.cpp:262)                          */
    uint8_t  two_id;                       /* "usr.2id"   (usrs.cpp:266)                          */
    uint8_t  n_byte;                       /* "gen.N_byte" (gens.cpp:104), 0 = none seen          */
    uint8_t  gen_bits;                     /* context bits used by the base model of this block   */
    uint32_t extra_hi;                     /* "qlt.extra.hi" (qlts.cpp:57-61)                     */
    uint32_t first_hdr_len;                /* "rec.first" length (recs.cpp:68-75)                 */
    uint64_t first_hdr_off;                /* its offset in the first-header blob                 */
    uint64_t size[SFQ_NSTREAMS];           /* bytes of each stream of this block (0 = absent); 64 bit: a format-6
                                              archive is ONE block, and the reference writes streams over 4 GiB  */
    uint32_t status;                       /* 0 or -SFQ_E_* for this block                        */
    uint32_t hdr_bytes;                    /* sum of header-line lengths (sizes the decoder's staging; 0 = unknown) */
} sfq_block_info;

typedef struct sfq_result {
    uint64_t n_records;
    uint32_t n_blocks;
    uint32_t abi_version;
    uint64_t stream_bytes[SFQ_NSTREAMS];   /* per stream: sum over blocks                          */
    uint64_t stream_offset[SFQ_NSTREAMS];  /* per stream: where its block-concatenation starts in d_out */
    uint64_t total_bytes;                  /* bytes used in d_out                                  */
    uint64_t first_hdr_bytes;              /* size of the first-header blob (see sfq_get_first_headers) */
    uint32_t n_chains;                     /* frozen tables: chains per chain-coded stream (else 0)   */
    uint32_t reserved;
    double   kernel_ms[8];                 /* device time of the last call, by phase (see SFQ_T_*): a model's phase is
                                              everything on its stream -- counting passes, row building, the coding kernel */
    double   coder_ms[4];                  /* encode: the coding kernel alone (HIP events around its launch on its stream):
                                              [0] quality, [1] bases, [2] headers, [3] the framing kernel (k_frame).  What a kernel trace shows as
                                              k_qlt_encode_c / k_gen_encode_c (k_gm_code under the match model) / k_rec_tokens -- the longest of the
                                              header chains' kernels; the coder behind it, k_rec_code, is in the phase -- (k_*_encode_k / _w with
                                              adaptive tables) */
} sfq_result;

#define SFQ_T_FRAME   0   /* line index + block descriptors                     */
#define SFQ_T_QLT     1   /* quality model kernel(s)                            */
#define SFQ_T_GEN     2   /* base model kernel(s)                               */
#define SFQ_T_REC     3   /* header model kernel(s)                             */
#define SFQ_T_USR     4   /* framing-exception kernel(s)                        */
#define SFQ_T_PACK    5   /* size scan + compaction / FASTQ assembly            */
#define SFQ_T_TOTAL   6   /* first launch -> last launch of the call            */

/* ---- context ------------------------------------------------------------------------------- */
/* Replaces the reference's process-wide singletons (conf, onef, g_record_count, g_genofs_count:
 * config.hpp:62-64) with an explicit, re-entrant object bound to one HIP device. */
int  sfq_ctx_create(sfq_ctx** out, int hip_device);
void sfq_ctx_destroy(sfq_ctx* ctx);
const char* sfq_last_error(const sfq_ctx* ctx);         /* replaces croak() text (config.cpp:54-68) */
/* Upper bound on device bytes the context may hold for model tables (default: 70 % of the device). */
int  sfq_ctx_set_table_budget(sfq_ctx* ctx, uint64_t bytes);
/* Total memory of the context's device, in bytes (to share a GPU between contexts: budget = a fraction of it). */
uint64_t sfq_ctx_device_memory(const sfq_ctx* ctx);
/* The HIP stream the context launches on (a hipStream_t), for callers that order work against it. */
void* sfq_ctx_stream(sfq_ctx* ctx);
int  sfq_ctx_synchronize(sfq_ctx* ctx);
/* Page-locked host memory for the buffers handed to the *_host entry points: the copy engine reads / writes it at the
 * link's rate (a pageable buffer is staged through the driver page by page).  The reference has no counterpart: its
 * I/O is fread / fwrite into static buffers (usrs.cpp:95-118, filer.cpp).  NULL when the allocation fails. */
void* sfq_host_alloc(sfq_ctx* ctx, uint64_t bytes);
void  sfq_host_free(sfq_ctx* ctx, void* p);

/* ---- compress ------------------------------------------------------------------------------
 * Replaces the body of UsrSave::encode()'s record loop (usrs.cpp:400-404: gen.save / rec.save /
 * qlt.save per record) plus UsrSave::get_record()'s framing (usrs.cpp:303-390) for a whole buffer
 * of FASTQ text resident on the device.  d_out receives, stream after stream, the concatenation of
 * every block's bytes for that stream; sfq_get_block_index() tells the per-block sizes.
 * Worst-case d_out size: sfq_encode_bound(nbytes). Asynchronous errors are reported at return
 * (the call synchronizes the context's stream once, at the end). */
uint64_t sfq_encode_bound(uint64_t fastq_bytes);
int sfq_encode_blocks(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params,
                      uint8_t* d_out, uint64_t out_cap, sfq_result* result);
/* BASELINE.json config C2: the quality model alone == QltSave::save over all records (qlts.hpp:82-90). */
int sfq_encode_qlt_blocks(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params,
                          uint8_t* d_out, uint64_t out_cap, sfq_result* result);
/* The priors of a text without coding it: afterwards sfq_get_qlt_prior (and, with frozen tables, sfq_get_rec_prior) hold what
 * an sfq_encode_blocks call with the same parameters would have built.  For several contexts / GPUs that compress parts
 * of one file from ONE prior: build it once, install it everywhere (sfq_set_*_prior), encode with prior_step = SFQ_PRIOR_GIVEN. */
int sfq_build_priors(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params);
/* Several GPUs, one file, NO serial head: every rank counts the sample of ITS share of the file (sfq_count_priors: every
 * sample_scale-th record of what one call alone would sample -- sample_scale = the number of ranks keeps the job's sample the size
 * of one call's), the ranks add their counts up (an all-reduce of two u32 arrays: sfq_get_prior_counts -> the caller's device
 * buffers -> sfq_set_prior_counts), and every rank codes with prior_step = SFQ_PRIOR_COUNTS: identical priors everywhere, nobody
 * waits for a rank that builds them.  Array sizes in u32 words: sfq_prior_counts_words.
 * An sfq_encode_blocks with SFQ_PRIOR_COUNTS that follows sfq_count_priors on the same context, buffer and size keeps that
 * call's line index (the text is framed once per step, not twice): the text must not change between the two calls. */
int  sfq_count_priors(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params, uint32_t sample_scale);
void sfq_prior_counts_words(int level, uint64_t* qlt_words, uint64_t* rec_words);
int  sfq_get_prior_counts(sfq_ctx* ctx, int level, uint32_t* d_qlt, uint32_t* d_rec);
int  sfq_set_prior_counts(sfq_ctx* ctx, int level, const uint32_t* d_qlt, const uint32_t* d_rec);
/* Same with host buffers (stages through the context's device memory; PCIe-inclusive). */
int sfq_encode_blocks_host(sfq_ctx* ctx, const uint8_t* h_fastq, uint64_t nbytes, const sfq_params* params,
                           uint8_t* h_out, uint64_t out_cap, sfq_result* result);

/* Block index / first headers of the LAST encode call on this context (host copies). */
int sfq_get_block_index(sfq_ctx* ctx, sfq_block_info* h_blocks, uint32_t cap);
int sfq_get_first_headers(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap);

/* Quality warm-start prior of the LAST encode call (the "qlt.pri" stream): returns its size, copies it if
 * cap allows; 0 = the call used cold blocks. */
int64_t sfq_get_qlt_prior(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap);
/* Install the prior the next sfq_decode_blocks call must start its quality rows from (n = 0: cold). */
int sfq_set_qlt_prior(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n);
/* Frozen tables: the header prior of the LAST encode call (the "rec.pri" stream: scaled symbol counts of the header
 * model's rows); same conventions.  0 = the call coded its headers with adaptive rows. */
int64_t sfq_get_rec_prior(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap);
int sfq_set_rec_prior(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n);
/* Frozen tables: the chain index of the LAST encode call (the "chn.idx" stream: records per chain, flags, and the size
 * of every chain's qlt and gen stream); returns its size, copies it if cap allows; 0 = the call used adaptive tables.
 * A decoder installs it before sfq_decode_blocks (n = 0: the archive has none, i.e. adaptive tables). */
int64_t sfq_get_chain_index(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap);
int sfq_set_chain_index(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n);

/* ---- decompress ----------------------------------------------------------------------------
 * Replaces UsrLoad::decode()'s loop (usrs.cpp:555-571: rec.load / qlt.load / gen.load / save).
 * d_streams holds the streams laid out as sfq_encode_blocks wrote them (stream_offset[] + running
 * sum of sfq_block_info.size[]); a reference-written format-6 archive is the one-block case with
 * gen_bits = the level's.  h_first_hdrs is the first-header blob (one "rec.first" per block).
 * Writes the FASTQ text to d_fastq_out; *out_bytes = its length. */
int sfq_decode_blocks(sfq_ctx* ctx, const sfq_params* params, const sfq_block_info* h_blocks, uint32_t n_blocks,
                      const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                      const uint8_t* d_streams, const uint64_t stream_offset[SFQ_NSTREAMS],
                      uint8_t* d_fastq_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* result);
int sfq_decode_blocks_host(sfq_ctx* ctx, const sfq_params* params, const sfq_block_info* h_blocks, uint32_t n_blocks,
                           const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                           const uint8_t* h_streams, uint64_t streams_bytes, const uint64_t stream_offset[SFQ_NSTREAMS],
                           uint8_t* h_fastq_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* result);

/* ---- the ".sfq" container (host only) ----------------------------------------------------------
 * Replaces FilerSave + the info page (filer.cpp:217-242, config.cpp:334-347) for hosts that assemble an archive
 * themselves, e.g. the writer rank of a multi-GPU job (slimfastq_amd/dist_compress.py): info_text is the info
 * page ("key=value\n" lines), then n_streams named byte streams (names of at most 8 bytes: filer.cpp:42-47). */
int sfq_archive_write(const char* path, const char* info_text, uint32_t n_streams,
                      const char* const* names, const uint8_t* const* data, const uint64_t* sizes);
/* The "blk.idx" stream of a block-format archive from a block index; returns its size (out == NULL: size only). */
int64_t sfq_pack_block_index(const sfq_block_info* blocks, uint32_t n, uint8_t* out, uint64_t cap);

/* ---- utilities (host only, no GPU) ----------------------------------------------------------- */
/* Deterministic synthetic FASTQ. kind 0 = 150 bp-style Illumina reads of
 * read_len bases; kind 1 = long reads, lengths log-uniform in [10000, 50000] (read_len ignored);
 * kind 2 = kind 0 with NovaSeq-style 4-level binned qualities; kind 3 = kind 0 with the bases sampled (either
 * strand, 0.5 % substitutions) from a seeded random 10 Mbp genome: 30x coverage at 2 M reads of 150 bp.
 * Returns bytes written, or the required size when h_out == NULL, or <0. */
int64_t sfq_synth_fastq(uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed, int kind,
                        uint8_t* h_out, uint64_t cap);
int sfq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLIMFASTQ_AMD_H */
