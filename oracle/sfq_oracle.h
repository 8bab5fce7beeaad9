/*
 * sfq_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the slimfastq (Infinidat/slimfastq, format version 6) hot path and of
 * the container around it.  It exists to CHECK the HIP product path (tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg); nothing in slimfastq_amd/ may include, link or call it.
 *
 * Parity pinning: oracle/_ref/slimfastq_ref is the real reference compiled from /root/reference by
 * oracle/Makefile (`make ref`).  tests/test_oracle_vs_ref.py checks, for every reference sample and
 * level 1..4, that this restatement produces stream-byte-identical archives and decodes the
 * reference's archives bit-exactly; tests/golden/ holds reference-produced vectors for the GPU box.
 *
 * Citations are file:line into the reference tree.
 */
#ifndef SFQ_ORACLE_H
#define SFQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfqo_archive sfqo_archive;   /* an in-memory .sfq container (page image + directory) */

/* Options beyond the reference CLI; all-zero == exactly the reference behaviour. */
typedef struct sfqo_opts {
    int level;          /* 1..4 (config.cpp:329-336)                                            */
    int quiet;          /* -q : suppress log.* / qlt.extra.hi info keys (config.cpp:267)        */
    int gen_bits;       /* 0 = by level (gens.hpp:43-53: 18/22/24/26); else context bits        */
    const char* orig_filename;   /* NULL -> "<< stdin >>" (config.cpp:346)                      */
    long long   orig_size;       /* <0 -> key omitted (stdin mode)                              */
    int lossless;       /* this project's block format (NOT reference behaviour; slimfastq_amd/csrc/dev_common.h): a header
                           field the reference would not print back is coded as a string, a header with a NUL inside goes
                           whole, lowercase bases are listed in a "gen.lc" stream, an irregular '+' line is an error  */
} sfqo_opts;

/* Whole-file compress: restates main.cpp:51-60 -> UsrSave::encode (usrs.cpp:392-407).
 * Returns NULL on error (message via sfqo_last_error). */
sfqo_archive* sfqo_compress(const uint8_t* fastq, size_t n, const sfqo_opts* opts);

/* Whole-file decompress: restates UsrLoad::decode (usrs.cpp:539-574).  *out is malloc'ed. */
int sfqo_decompress(const sfqo_archive* a, uint8_t** out, size_t* out_len);

/* Container I/O (filer.cpp). */
sfqo_archive* sfqo_archive_from_image(const uint8_t* image, size_t n);   /* parse a .sfq file image */
sfqo_archive* sfqo_archive_read(const char* path);
int           sfqo_archive_write(const sfqo_archive* a, const char* path);
const uint8_t* sfqo_archive_image(const sfqo_archive* a, size_t* n);      /* full page image */
void          sfqo_archive_free(sfqo_archive* a);

/* Directory access: stream 0 is the info page text. */
int            sfqo_nstreams(const sfqo_archive* a);
const char*    sfqo_stream_name(const sfqo_archive* a, int i);
size_t         sfqo_stream_size(const sfqo_archive* a, int i);
/* copies the stream's bytes (walking the page chain) into buf (>= size); returns size or -1 */
long long      sfqo_stream_read(const sfqo_archive* a, int i, uint8_t* buf, size_t cap);
int            sfqo_stream_find(const sfqo_archive* a, const char* name);  /* -1 if absent */
const char*    sfqo_info_get(const sfqo_archive* a, const char* key);      /* "" if absent */

const char*    sfqo_last_error(void);

/* ---- stream-level entry points (what the HIP kernels are compared with) -------------------------
 * A "record table" addresses lines inside a caller buffer.  All encoders start from the cold model
 * state (exactly the state of a fresh reference process) and return a malloc'ed byte stream
 * INCLUDING the 8 flush bytes of RCoder::done (coder.hpp:52-61).                                  */

/* qlts.cpp:74-136 -- qualities of nrec records; level selects save_1/2/3. */
int sfqo_qlt_encode(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec,
                    int level, uint8_t** out, size_t* out_len, uint32_t* extra_hi);
/* qlts.cpp:163-234 -- inverse; writes len[i] bytes at dst+off[i]. */
int sfqo_qlt_decode(const uint8_t* stream, size_t n, uint8_t* dst, const uint64_t* off,
                    const uint32_t* len, size_t nrec, int level);

/* gens.cpp:116-159 -- bases (with their qualities for the N/Q0 rules). gen_bits as in sfqo_opts.
 * Exception gap lists (gens.cpp:91-114) are returned as absolute 1-based base indices.            */
typedef struct sfqo_gen_side {
    uint64_t* ns; size_t n_ns;     /* N with quality != '!'      (gen.Ns) */
    uint64_t* nn; size_t n_nn;     /* real base with quality '!' (gen.Nn) */
    int       n_byte;              /* first N-like char seen, 0 if none   */
} sfqo_gen_side;
int sfqo_gen_encode(const uint8_t* base, const uint64_t* goff, const uint32_t* glen,
                    const uint64_t* qoff, const uint32_t* qlen, size_t nrec, int gen_bits,
                    uint8_t** out, size_t* out_len, sfqo_gen_side* side);
/* gens.cpp:215-249 without the N substitution: emits "ACGT"/"0123" codes only. */
int sfqo_gen_decode_raw(const uint8_t* stream, size_t n, uint8_t* dst, const uint64_t* goff,
                        const uint32_t* glen, size_t nrec, int gen_bits, int solid);

/* recs.cpp:277-372 -- headers (text after '@', without '\n') of nrec records.  The first header is
 * not coded (it goes to info key rec.first, recs.cpp:68-75).  Shape-change exceptions (rec.x) are
 * returned as 1-based record numbers in xrec (malloc'ed).                                         */
int sfqo_rec_encode(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec,
                    uint8_t** out, size_t* out_len, uint64_t** xrec, size_t* n_xrec);

/* xfile.cpp + power_ranger.hpp:133-192 -- a PowerRangerU gap stream: put_u of each value, then the
 * terminator 0 and the flush (xfile.cpp:40-47).                                                   */
int sfqo_xfile_encode_u(const uint64_t* vals, size_t n, uint8_t** out, size_t* out_len);
int sfqo_xfile_decode_u(const uint8_t* stream, size_t n, uint64_t* vals, size_t nvals);

/* ---- block format 7 extensions (this project's own, restated for the tests; see sfq_oracle.c) ----------------- */
int sfqo_qlt_histogram(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, int level,
                       size_t first, size_t step, uint32_t* counts);
int sfqo_qlt_prior_rows(const uint32_t* counts, size_t q_rows, uint32_t* rows);
int sfqo_qlt_encode_blocks(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, int level, size_t block_reads,
                           const uint32_t* prior_rows, uint8_t** out, size_t* out_len, uint32_t* sizes);
/* frozen tables */
int sfqo_qlt_frozen_rows(const uint32_t* rows66, size_t q_rows, uint32_t* out);
long long sfqo_qlt_encode_chains(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, int level, size_t block_reads,
                                 size_t chain_reads, const uint32_t* frozen_rows, uint8_t** out, size_t* out_len, uint32_t* sizes, uint32_t* extra_hi);
long long sfqo_gen_encode_chains(const uint8_t* base, const uint64_t* goff, const uint32_t* glen, size_t nrec, int gen_bits, size_t block_reads,
                                 size_t chain_reads, uint32_t step, uint8_t** out, size_t* out_len, uint32_t* sizes, int* gen_on);
/* the same for chains that are SEGMENTS of one record (long reads): other_len = the lengths of the record's other line */
long long sfqo_qlt_encode_segs(const uint8_t* base, const uint64_t* off, const uint32_t* len, const uint32_t* other_len, size_t nrec, int level, uint32_t seg_len,
                               const uint32_t* frozen_rows, uint8_t** out, size_t* out_len, uint32_t* sizes, uint32_t* extra_hi);
long long sfqo_gen_encode_segs(const uint8_t* base, const uint64_t* goff, const uint32_t* glen, const uint32_t* other_len, size_t nrec, int gen_bits, size_t block_reads,
                               uint32_t seg_len, uint32_t step, uint8_t** out, size_t* out_len, uint32_t* sizes, int* gen_on);
/* bases, the generation MATCH model (round 5, gm.hip; sfq_oracle.c "the generation MATCH model"): same shapes as the two above,
   table_bits = log2 of the index's entries ("chn.idx") */
long long sfqo_gm_encode_chains(const uint8_t* base, const uint64_t* goff, const uint32_t* glen, size_t nrec, int table_bits, size_t block_reads,
                                size_t chain_reads, uint8_t** out, size_t* out_len, uint32_t* sizes, int* gen_on);
long long sfqo_gm_encode_segs(const uint8_t* base, const uint64_t* goff, const uint32_t* glen, const uint32_t* other_len, size_t nrec, int table_bits, size_t block_reads,
                              uint32_t seg_len, uint8_t** out, size_t* out_len, uint32_t* sizes, int* gen_on);
/* the way back on the CPU (whole-record chains of a call that took the model): the bases' codes 0..3, codes[sum glen]; 0 or -1 */
int sfqo_gm_decode_chains(const uint8_t* streams, const uint32_t* sizes, const uint32_t* glen, size_t nrec, int table_bits, size_t block_reads, size_t chain_reads, uint8_t* codes);
/* frozen tables, round 4: a block's three base-exception lists ("gen.Ns", "gen.Nn", "gen.lc") as adaptive Rice codes
   (chains.hip k_gen_exc_r; NOT the reference's XFile coding: DESIGN.md 4.10) and the way back */
int sfqo_exc_rice_block(const uint8_t* base, const uint64_t* goff, const uint32_t* glen, const uint64_t* qoff, const uint32_t* qlen, size_t nrec,
                        uint8_t** out, size_t* out_len, uint32_t* n_byte_out);
long long sfqo_exc_rice_decode(const uint8_t* p, size_t n, uint64_t* pos, size_t cap);
int sfqo_rec_count(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, size_t stride, size_t run, size_t nruns, uint32_t* counts);
int sfqo_rec_prior_freqs(const uint32_t* counts, uint32_t* f);
int sfqo_rec_frozen_rows(const uint32_t* f, uint32_t* rows);
long long sfqo_rec_encode_chains_frozen(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, size_t block_reads, size_t chain_reads,
                                        const uint32_t* frozen_rows, uint8_t** out, size_t* out_len, uint32_t* sizes, uint32_t* hdr_bytes);

/* a "rec" stream in the pre-version-5 layout (what RecLoad::load_pre5, recs.cpp:463-510, reads): test input only */
int sfqo_rec_encode_pre5(const uint8_t* base, const uint64_t* off, const uint32_t* len, size_t nrec, uint8_t** out, size_t* out_len);

void sfqo_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
