// api.cpp -- host side of the C ABI (include/slimfastq_amd.h): context, device memory, launch
// sequencing.  No entropy-coding arithmetic lives here: every model / coder step runs in the HIP
// kernels (models_*.hip, decode_l.hip); this file only frames, sizes, launches and packs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <functional>
#include <future>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>

#include "kernels.h"

namespace {

// where a call's HOST time goes (SFQ_HOST_TIMING=1 in the environment: marks printed to stderr when the call returns)
struct HostTimes {
    bool on; std::chrono::steady_clock::time_point t0; std::vector<std::pair<const char*, double>> marks;
    HostTimes() : on(getenv("SFQ_HOST_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) { if (on) marks.push_back({ what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() }); }
    ~HostTimes() { if (on) { for (auto& m : marks) fprintf(stderr, "  [host] %-28s %8.3f ms\n", m.first, m.second); } }
};
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
};

struct Tables {
    DevBuf q_slots, q_hdr, p_slots, p_hdr, g_tab;
    u32 slots = 0;      // block slots the tables were sized for
    u32 q_rows = 0;
    u32 g_bits = 0;
};

}  // namespace

// One host thread that stays with the context and runs a job beside the calling thread (a decode reads the chain lists of
// "chn.idx" on it while the caller uploads priors and queues kernels).  Started with the first job -- a thread created per call
// came up cold and took half as long again for the same work --, joined when the context goes.
struct HostWorker {
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::function<int()> job; bool has_job = false, done = true, quit = false; int result = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> j = std::move(job); has_job = false;
            lk.unlock();
            int r;
            try { r = j(); }                                  // (a job that throws fails its call; the thread and the process live)
            catch (const std::bad_alloc&) { r = SFQ_E_NOMEM; }
            catch (...) { r = SFQ_E_CORRUPT; }
            lk.lock();
            result = r; done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> j) {
        std::unique_lock<std::mutex> lk(mu);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        job = std::move(j); has_job = true; done = false;
        cv.notify_all();
    }
    bool busy() { std::unique_lock<std::mutex> lk(mu); return !done; }
    int wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); return result; }
    ~HostWorker() {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};
struct sfq_ctx {
    int dev = 0;
    HostWorker worker;
    hipStream_t st = nullptr;
    hipStream_t st_aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[28] = {};
    bool rec_copy_pending = false;     // the header prior's frequencies are still to be copied to the host (rec_prior_copy_back)
    bool rec_blob_pending = false;     // "rec.pri" is still to be packed from the frequencies behind ev[24]
    std::string err;
    u64 table_budget = 0;
    u64 dev_total = 0;
    u32 wave_slots = 8192;                 // wavefronts the device holds: CUs x 4 SIMDs x 8
    u32 epoch_base = 0;
    Tables tab;
    // scratch (grow-only)
    DevBuf chunk_counts, chunk_base, scan_tmp, line_off, status, blocks, arena, blk_stream_off, stream_total,
           lens, blob_off, blob, in_stage, out_stage;
    // decode scratch
    DevBuf slen, qlen, pfg, pfq, soff, qoff, seq_stage, qual_stage, hdr_stage, hlen, hoff, hso, hsc, rsize, roff, d_first;
    // quality warm start
    DevBuf hist, rows66, ptmp, prior_w, prior_wovf, prior_ls, prior_lh, tickets;
    // frozen tables (sfq_params.tables = SFQ_TABLES_FROZEN): dense quality rows, chain sizes, generation tables of the bases
    DevBuf gm_T, gm_slen, gm_boff, gm_soff, gm_scan, gm_stage, gm_tok, gm_csz, gm_idx;     // bases under the match model (gm.hip): index, stage, tokens
    DevBuf qrows, qdec, qesc, qw, csz, coff, gcnt, grows, glog, gcost, gbins, gfill, hcnt, hfreq, rrows, rdec, rmap, rflags, rtok, excf, cflags;
    DevBuf chn_len, chn_off, chn_out;      // "chn.idx": the size lists as bytes (chains.hip launch_chain_index_bytes)
    DevBuf pslot, plist;                   // the quality prior's listed rows, back to back (prior.hip launch_prior_list)
    DevBuf segn, segoff, segrec;           // chains that are segments of one record (long reads): segments per record, their scan, a chain's record
    // format 6's oversize records (frame.hip): flags, kept bytes, their scans, the text without them, kept record -> file number, the list;
    // the file's own line index; decode: numbers, pieces, raw text of the three streams, sizes / offsets in file order
    DevBuf oflags, okbytes, ofpos, okoff, ofilt, orecmap, olist, line_off_o, ono, opiece, otxt[3], osize_all, oroff_all, oroff_k, ocnt;
    u32 r_hot = 0, r_hot_dec = 0;
    bool blobs_from_encode = false;        // prior_blob / rec_prior_blob / chain_blob are what the last ENCODE left for sfq_get_*:
                                           // a decode never reads those (only what sfq_set_* installed)
    bool unsettled = false;                // a call returned with an error: its side streams may still be running
    bool counts_only = false; u32 sample_scale = 1;      // sfq_count_priors: stop once the sample is counted; every sample_scale-th sampled record
    // the line index sfq_count_priors left, for the encode of the SAME text that follows it (SFQ_PRIOR_COUNTS): a rank of a
    // multi-GPU job frames its shard once per step, not twice (3 ms of a 21 ms step)
    struct { const u8* ptr = nullptr; u64 nbytes = 0, nrec = 0, print = 0; bool marks = false, valid = false; } framed;       // print: frame.hip k_text_fingerprint of the text it indexes
    // what `hist` / `hcnt` hold for an SFQ_PRIOR_COUNTS encode: set by sfq_count_priors and sfq_set_prior_counts, dropped by any
    // call that samples into those buffers itself (an ordinary encode's leftovers are not "installed counts")
    struct { bool valid = false, rec = false; u32 q_rows = 0; } counts;
    void* pin = nullptr; size_t pin_cap = 0;
    void* pin2 = nullptr; size_t pin2_cap = 0;     // the same for the end of an encode: block descriptors, chain sizes       // pinned host scratch: device -> host copies that must not block the launching thread
    std::vector<u8> chain_blob;            // "chn.idx" of the last encode / installed for the next decode
    std::vector<u8> rec_prior_blob;        // "rec.pri" likewise
    std::vector<u8> chain_tmp;             // scratch the chain index is written into
    bool prior_on = false;                 // the device prior tables are valid for the running call
    std::vector<u8> prior_blob;            // packed prior of the last encode / installed for the next decode
    // last encode, host copies
    std::vector<sfq_block_info> index;
    std::vector<u8> first_hdrs;
};

namespace {

int fail(sfq_ctx* c, int code, const char* fmt, ...) {
    char b[512];
    va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
    if (c) c->err = b;
    return code;
}
#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(ctx, e_ == hipErrorOutOfMemory ? SFQ_E_NOMEM : SFQ_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

int reserve(sfq_ctx* ctx, DevBuf& b, size_t bytes, bool zero_new = false) {
    if (bytes <= b.cap && b.p) return SFQ_OK;
    if (b.p) { HIPC(hipStreamSynchronize(ctx->st)); HIPC(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIPC(hipMalloc(&b.p, want));
    b.cap = want;
    if (zero_new) HIPC(hipMemsetAsync(b.p, 0, want, ctx->st));
    return SFQ_OK;
}
void release(DevBuf& b) { if (b.p) { (void)hipDeviceSynchronize(); (void)hipFree(b.p); } b.p = nullptr; b.cap = 0; }

int level_gen_bits(int level) {   // gens.hpp:43-53
    switch (level) { case 1: return 18; case 2: return 22; case 3: return 24; default: return 26; }
}
int clamp_level(int level) { return level > 4 ? 4 : level < 1 ? 1 : level; }   // config.cpp:232-237

// Size the model tables for `want` concurrent block slots (or fewer if the budget says so).
int reserve_pinned_buf(sfq_ctx* ctx, void*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap && p) return SFQ_OK;
    if (p) { HIPC(hipDeviceSynchronize()); HIPC(hipHostFree(p)); p = nullptr; cap = 0; }
    bytes += bytes / 4;                                    // (grow-only, with headroom: re-pinning costs milliseconds)
    HIPC(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return SFQ_OK;
}
int reserve_pinned(sfq_ctx* ctx, size_t bytes) { return reserve_pinned_buf(ctx, ctx->pin, ctx->pin_cap, bytes); }

int ensure_tables(sfq_ctx* ctx, u32 want, u32 q_rows, u32 g_bits, u32 models, u32* got) {
    const u64 per_q = (models & SFQ_M_QLT) ? (u64)q_rows * (L64_NSYM * 4 + sizeof(RowHdr)) : 0;
    const u64 per_p = (u64)PR_ROWS * (PW_NSYM * 4 + sizeof(RowHdr));
    const u64 per_g = (models & SFQ_M_GEN) ? ((u64)4 << g_bits) : 0;
    const u64 per = per_q + per_p + per_g;
    u64 fit = ctx->table_budget / per;
    if (fit == 0) return fail(ctx, SFQ_E_NOMEM, "table budget %llu B too small for one block slot (%llu B)",
                              (unsigned long long)ctx->table_budget, (unsigned long long)per);
    const u32 cap = 12288;                                              // the chip holds 8192 waves; the two-block kernels take 2 slots per wave
    u32 slots = (u32)std::min<u64>(std::min<u64>(want, fit), cap);
    Tables& t = ctx->tab;
    // Row tables are epoch-tagged and epochs only grow, so stale rows of any earlier geometry can never
    // match: slot storage needs no clearing, and a header array is zeroed only when it is (re)allocated.
    int rc;
    if (per_q) {
        const size_t need_s = (size_t)slots * q_rows * L64_NSYM * 4, need_h = (size_t)slots * q_rows * sizeof(RowHdr);
        if ((rc = reserve(ctx, t.q_slots, need_s, true))) return rc;      // zeroed: the wave kernel keeps its row tags inside
        if (need_h > t.q_hdr.cap || !t.q_hdr.p) { release(t.q_hdr); if ((rc = reserve(ctx, t.q_hdr, need_h, true))) return rc; }
        t.q_rows = q_rows;
    }
    {
        const size_t need_s = (size_t)slots * PR_ROWS * PW_NSYM * 4, need_h = (size_t)slots * PR_ROWS * sizeof(RowHdr);
        if ((rc = reserve(ctx, t.p_slots, need_s))) return rc;
        if (need_h > t.p_hdr.cap || !t.p_hdr.p) { release(t.p_hdr); if ((rc = reserve(ctx, t.p_hdr, need_h, true))) return rc; }
    }
    if (per_g) { if ((rc = reserve(ctx, t.g_tab, (size_t)slots * per_g))) return rc; t.g_bits = g_bits; }
    t.slots = slots;
    *got = slots;
    return SFQ_OK;
}

void fill_model_args(sfq_ctx* ctx, ModelArgs& a, u32 nblocks, int level, u32 g_bits, bool lossless) {
    memset(&a, 0, sizeof a);
    Tables& t = ctx->tab;
    a.line_off = (const u64*)ctx->line_off.p;
    a.blocks = (BlockDesc*)ctx->blocks.p;
    a.nblocks = nblocks;
    a.arena = (u8*)ctx->arena.p;
    a.level = level;
    a.lossless = lossless ? 1u : 0u;
    a.epoch_base = ctx->epoch_base;
    ctx->epoch_base += nblocks;            // taken now: a call that fails half-way has used them all the same
    a.q_slots = (u32*)t.q_slots.p; a.q_hdr = (RowHdr*)t.q_hdr.p; a.q_rows = t.q_rows;
    a.p_slots = (u32*)t.p_slots.p; a.p_hdr = (RowHdr*)t.p_hdr.p;
    a.g_tab = (u32*)t.g_tab.p; a.g_bits = g_bits;
    if (ctx->prior_on) {
        a.prior_w = (const u32*)ctx->prior_w.p; a.prior_wovf = (const u32*)ctx->prior_wovf.p;
        a.prior_ls = (const u32*)ctx->prior_ls.p; a.prior_lh = (const RowHdr*)ctx->prior_lh.p;
    }
}

// epochs are unique per (context lifetime, block); on wrap the row headers are cleared
int advance_epoch(sfq_ctx* ctx, u32 nblocks) {
    if ((u64)ctx->epoch_base + nblocks + 2 >= 0x3FFFFFFFull) {
        Tables& t = ctx->tab;
        if (t.q_slots.p) HIPC(hipMemsetAsync(t.q_slots.p, 0, t.q_slots.cap, ctx->st));
        if (t.q_hdr.p) HIPC(hipMemsetAsync(t.q_hdr.p, 0, t.q_hdr.cap, ctx->st));
        if (t.p_hdr.p) HIPC(hipMemsetAsync(t.p_hdr.p, 0, t.p_hdr.cap, ctx->st));
        ctx->epoch_base = 0;
    }
    return SFQ_OK;
}

// ---- "qlt.pri": the prior's exchange form (66 dwords per context: slot[64], total, iend) <-> bytes.
// Per non-empty context, in order: varint(ctx delta), iend, nnz, then nnz x { symbol byte, varint(freq) } for
// the leading non-zero slots; the remaining slots are the unused symbols below iend in ascending order.
void put_v(std::vector<u8>& o, u64 v) { while (v >= 0x80) { o.push_back((u8)(v | 0x80)); v >>= 7; } o.push_back((u8)v); }
bool get_v(const u8* b, size_t n, size_t& p, u64& v) {
    v = 0;
    for (int sh = 0; sh < 64; sh += 7) { if (p >= n) return false; u8 c = b[p++]; v |= (u64)(c & 0x7f) << sh; if (!(c & 0x80)) return true; }
    return false;
}
// n sizes of a "chn.idx" list into out[]: plain varints, or (deltas) zigzag differences to the entry before -- nearly all of
// those one byte: eight at a time where eight bytes have no continuation bit (a call has half a million)
bool read_sizes(const u8* b, size_t nb, size_t& p, bool deltas, u32* out, size_t n) {
    u64 prev = 0;
    size_t i = 0;
    while (i < n) {
        if (deltas && p + 8 <= nb && n - i >= 8) {
            u64 w; memcpy(&w, b + p, 8);
            if (!(w & 0x8080808080808080ull)) {
                for (int j = 0; j < 8; j++) {
                    const u32 x = (u32)(w >> (8 * j)) & 0x7fu;
                    const i64 v = (i64)prev + ((i64)(x >> 1) ^ -(i64)(x & 1));
                    if (v < 0 || v > 0xFFFFFFFFll) return false;
                    prev = (u64)v; out[i + j] = (u32)v;
                }
                p += 8; i += 8;
                continue;
            }
        }
        u64 raw;
        // (one or two bytes, nearly always: decoded in line -- get_v's loop and call were most of what an entry cost where the eight-at-a-time path does not apply,
        //  a group of eight with a two-byte entry among them: more than half of the groups)
        if (p < nb && b[p] < 0x80u) raw = b[p++];
        else if (p + 1 < nb && b[p + 1] < 0x80u) { raw = (u64)(b[p] & 0x7fu) | ((u64)b[p + 1] << 7); p += 2; }
        else if (!get_v(b, nb, p, raw)) return false;
        if (deltas) {
            if (raw > 0x1FFFFFFFFull) return false;
            const i64 v = (i64)prev + ((i64)(raw >> 1) ^ -(i64)(raw & 1));
            if (v < 0 || v > 0xFFFFFFFFll) return false;
            raw = (u64)v;
        }
        if (raw > 0xFFFFFFFFull) return false;
        prev = raw; out[i++] = (u32)raw;
    }
    return true;
}
// list: [0] = n, then per listed row (ascending contexts) 67 words: the context, 64 slots freq | sym << 16, total, iend
#define PRIOR_LIST_ROW 67u
#define PRIOR_LIST_HEAD 4u
#define PRIOR_LIST_EAGER 16384u          /* rows copied to the host before their number is known (4.4 MB) */
std::vector<u8> pack_prior(const u32* list, u32 q_rows) {
    const u32 n = list[0];
    std::vector<u8> o((size_t)n * (5 + 2 + 64 * 4) + 16);                  // (the most a row can take)
    u8* w = o.data();
    auto put = [&](u64 v) { while (v >= 0x80) { *w++ = (u8)(v | 0x80); v >>= 7; } *w++ = (u8)v; };       // put_v
    put(q_rows);
    u32 prev = 0;
    for (u32 i = 0; i < n; i++) {
        const u32* r = list + PRIOR_LIST_HEAD + (size_t)i * PRIOR_LIST_ROW;
        const u32 c = r[0], iend = r[66];
        r++;
        u32 nnz = 0;
        while (nnz < iend && (r[nnz] & 0xffff)) nnz++;
        put(c - prev + 1); prev = c;
        *w++ = (u8)iend; *w++ = (u8)nnz;
        for (u32 j = 0; j < nnz; j++) { *w++ = (u8)(r[j] >> 16); put(r[j] & 0xffff); }
    }
    put(0);
    o.resize((size_t)(w - o.data()));
    return o;
}
// "qlt.pri" -> the rows it lists, back to back (66 words each: 64 slots freq | sym << 16, total, iend), and their contexts
bool unpack_prior(const u8* b, size_t n, u32 q_rows, std::vector<u32>& ctxs, std::vector<u32>& rows) {
    size_t p = 0; u64 v;
    if (!get_v(b, n, p, v) || v != q_rows) return false;
    ctxs.clear(); rows.clear();
    u64 c = 0; bool first = true;
    for (;;) {
        if (!get_v(b, n, p, v)) return false;
        if (v == 0) break;
        // deltas are stored +1 so that 0 terminates; each is relative to the previous context (the first to 0)
        if (!first && v == 1) return false;                 // (the same context twice)
        c = first ? v - 1 : c + v - 1; first = false;
        if (c >= q_rows || p + 2 > n) return false;
        ctxs.push_back((u32)c);
        rows.resize(rows.size() + 66, 0);
        u32* r = rows.data() + rows.size() - 66;
        const u32 iend = b[p++], nnz = b[p++];
        if (iend > 64 || nnz > iend) return false;
        bool used[64] = {false};
        u32 total = 0;
        for (u32 j = 0; j < nnz; j++) {
            if (p >= n) return false;
            const u32 sym = b[p++];
            if (sym >= iend || used[sym] || !get_v(b, n, p, v) || v > 0xffff) return false;
            used[sym] = true; r[j] = (u32)v | (sym << 16); total += (u32)v;
        }
        u32 j = nnz;
        for (u32 s = 0; s < iend; s++) if (!used[s]) r[j++] = s << 16;
        r[64] = total; r[65] = iend;
    }
    return true;
}
int ensure_prior_buffers(sfq_ctx* ctx, u32 q_rows) {
    int rc;
    if ((rc = reserve(ctx, ctx->hist, (size_t)q_rows * 64 * 4))) return rc;
    if ((rc = reserve(ctx, ctx->rows66, (size_t)q_rows * 66 * 4))) return rc;
    if ((rc = reserve(ctx, ctx->prior_w, (size_t)q_rows * 64 * 4))) return rc;
    if ((rc = reserve(ctx, ctx->prior_wovf, (size_t)q_rows * 4 * 4))) return rc;
    if ((rc = reserve(ctx, ctx->prior_ls, (size_t)q_rows * 64 * 4))) return rc;
    if ((rc = reserve(ctx, ctx->prior_lh, (size_t)q_rows * sizeof(RowHdr)))) return rc;
    return SFQ_OK;
}

// the quality prior of a call that was handed one ("qlt.pri"): rows66 on the device = zeros + the listed rows, scattered by
// a kernel (the dense form is 17 MB, of which a file lists a few thousand rows), and the adaptive kernels' spread of it
// spread: also the adaptive kernels' forms of the rows (a block's first touch of a row copies the prior's; frozen tables do not use them)
int upload_prior(sfq_ctx* ctx, u32 q_rows, hipStream_t st, bool spread = true) {
    std::vector<u32> ctxs, rows;
    if (!unpack_prior(ctx->prior_blob.data(), ctx->prior_blob.size(), q_rows, ctxs, rows)) return fail(ctx, SFQ_E_CORRUPT, "bad quality prior (qlt.pri)");
    int rc;
    if ((rc = ensure_prior_buffers(ctx, q_rows))) return rc;
    HIPC(hipMemsetAsync(ctx->rows66.p, 0, (size_t)q_rows * 66 * 4, st));
    if (!ctxs.empty()) {
        if ((rc = reserve(ctx, ctx->ptmp, (ctxs.size() + rows.size()) * 4))) return rc;
        u32* d_ctx = (u32*)ctx->ptmp.p; u32* d_rows = d_ctx + ctxs.size();
        HIPC(hipMemcpyAsync(d_ctx, ctxs.data(), ctxs.size() * 4, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        launch_prior_scatter(d_ctx, d_rows, (u32)ctxs.size(), (u32*)ctx->rows66.p, st);
    }
    if (spread) launch_prior_spread((const u32*)ctx->rows66.p, q_rows, (u32*)ctx->prior_w.p, (u32*)ctx->prior_wovf.p, (u32*)ctx->prior_ls.p, (RowHdr*)ctx->prior_lh.p, st);
    HIPC(hipStreamSynchronize(st));            // the vectors are locals
    return SFQ_OK;
}

// ---- frozen tables: host-side pieces -------------------------------------------------------------------------
// ---- "rec.pri": the header prior.  Counts -> scaled frequencies f = (14 * count) >> s with the smallest s that brings
// the row's largest to <= 32000 (the PowerRanger adds 14 per hit, power_ranger.hpp:37-41); the blob lists, per non-empty row,
// varint(row - previous row + 1), varint(entries), entries x { byte value, varint(f) }, then a 0.
// "rec.pri" from the prior's frequencies f[PR_REC_ROWS][256] (chains.hip k_rec_prior_freqs makes them from the sample's counts):
// per row that has any, its distance from the row before + 1, the number of symbols, then (symbol, frequency) pairs
std::vector<u8> pack_rec_prior_f(const u32* f) {
    std::vector<u8> o;
    put_v(o, PR_REC_ROWS);
    u32 prev = 0;
    for (u32 r = 0; r < PR_REC_ROWS; r++) {
        const u32* fr = f + (size_t)r * 256;
        u32 nnz = 0;
        for (u32 s = 0; s < 256; s++) nnz += fr[s] != 0;
        if (!nnz) continue;
        put_v(o, r - prev + 1); prev = r;
        put_v(o, nnz);
        for (u32 s = 0; s < 256; s++) if (fr[s]) { o.push_back((u8)s); put_v(o, fr[s]); }
    }
    put_v(o, 0);
    return o;
}
bool unpack_rec_prior(const u8* b, size_t n, std::vector<u32>& f) {
    size_t p = 0; u64 v;
    if (!get_v(b, n, p, v) || v != PR_REC_ROWS) return false;
    f.assign((size_t)PR_REC_ROWS * 256, 0);
    u64 r = 0; bool first = true;
    for (;;) {
        if (!get_v(b, n, p, v)) return false;
        if (v == 0) break;
        r = first ? v - 1 : r + v - 1; first = false;
        if (r >= PR_REC_ROWS) return false;
        u64 nnz;
        if (!get_v(b, n, p, nnz) || nnz == 0 || nnz > 256) return false;
        int last = -1;
        for (u64 j = 0; j < nnz; j++) {
            if (p >= n) return false;
            const int sym = b[p++];
            if (sym <= last || !get_v(b, n, p, v) || v == 0 || v > 32000) return false;
            last = sym; f[(size_t)r * 256 + sym] = (u32)v;
        }
    }
    return true;
}
int upload_rec_rows(sfq_ctx* ctx, const std::vector<u32>& f, hipStream_t st) {
    int rc;
    {   // the rows worth staging in LDS: the ones the prior saw most
        std::vector<std::pair<u64, u32>> w;
        for (u32 r = 0; r < PR_REC_ROWS; r++) { u64 t = 0; for (u32 sx = 0; sx < 256; sx++) t += f[(size_t)r * 256 + sx]; if (t) w.push_back({ t, r }); }
        std::stable_sort(w.begin(), w.end(), [](const std::pair<u64, u32>& x, const std::pair<u64, u32>& y) { return x.first > y.first; });
        u16 map[PR_REC_ROWS], hot[64];
        for (u32 r = 0; r < PR_REC_ROWS; r++) map[r] = 0xFFFFu;
        const u32 nh = (u32)std::min<size_t>(w.size(), RDEC_LDS_ROWS);        // the encoders stage the first 8 of them (REC_LDS_ROWS, chains.hip)
        for (u32 i = 0; i < nh; i++) { hot[i] = (u16)w[i].second; map[w[i].second] = (u16)i; }
        if ((rc = reserve(ctx, ctx->rmap, sizeof map + sizeof hot))) return rc;
        HIPC(hipMemcpyAsync(ctx->rmap.p, map, sizeof map, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync((u8*)ctx->rmap.p + sizeof map, hot, sizeof hot, hipMemcpyHostToDevice, st));
        HIPC(hipStreamSynchronize(st));
        ctx->r_hot = std::min<u32>(nh, 8); ctx->r_hot_dec = nh;
    }
    if ((rc = reserve(ctx, ctx->hfreq, f.size() * 4))) return rc;
    if ((rc = reserve(ctx, ctx->rrows, f.size() * 4))) return rc;
    if ((rc = reserve(ctx, ctx->rdec, (size_t)PR_REC_ROWS * 272 * 2))) return rc;          // chains.hip RDEC_ROW
    HIPC(hipMemcpyAsync(ctx->hfreq.p, f.data(), f.size() * 4, hipMemcpyHostToDevice, st));
    launch_rec_frozen_rows((const u32*)ctx->hfreq.p, PR_REC_ROWS, (u32*)ctx->rrows.p, (u16*)ctx->rdec.p, st);
    HIPC(hipStreamSynchronize(st));            // `f` may be a local
    return SFQ_OK;
}
// the escape row (qualities >= 63, qlts.cpp:80-86): all 256 values equally likely
int build_qesc(sfq_ctx* ctx, hipStream_t st) {
    int rc;
    if (ctx->qesc.p) return SFQ_OK;                      // the same for every call: built once per context
    if ((rc = reserve(ctx, ctx->qesc, 256 * 4))) return rc;
    u32 row[256];
    for (u32 i = 0; i < 256; i++) row[i] = (i << 8) | (256u << 16);          // cum | freq << 16, total 2^16
    HIPC(hipMemcpyAsync(ctx->qesc.p, row, sizeof row, hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return SFQ_OK;
}
// floor(1024 * log2(x)) for x in 1..1023, by integer arithmetic alone (squaring a 1.31 fixed-point mantissa)
void log2_table(u16* t) {
    t[0] = 0;
    for (u32 x = 1; x < 1024; x++) {
        u32 e = 31 - (u32)__builtin_clz(x);
        u64 m = (u64)x << (31 - e);                       // mantissa in [2^31, 2^32)
        u32 frac = 0;
        for (int i = 0; i < 10; i++) {
            m = (m * m) >> 31;                            // in [2^31, 2^33)
            frac <<= 1;
            if (m >> 32) { frac |= 1; m >>= 1; }
        }
        t[x] = (u16)(e * 1024 + frac);
    }
}
// generations of the base model: blocks [bound[g], bound[g + 1]); the first is 1/64 of the blocks, each next one as long as
// all before it (GEN_GROW_NUM / GEN_GROW_DEN of them)
#define GEN_GROW_NUM 2
#define GEN_GROW_DEN 1
u32 gen_bounds(u32 nblocks, u32* bound) {
    u32 n = 0; bound[0] = 0;
    u64 b = std::max<u32>(1, (nblocks + 63) / 64);
    while (b < nblocks && n + 2 < GEN_MAX_GENERATIONS) { bound[++n] = (u32)b; b = std::max<u64>(b + 1, b * GEN_GROW_NUM / GEN_GROW_DEN); }
    bound[++n] = nblocks;
    return n;                                              // number of generations
}
#define SFQ_CHAINS_WANT 262144ull
int default_chain_reads(u64 nrec, u64 nbytes) {
    // Chains are the unit of parallelism (64 per wavefront) and a lane's walk through its chain is the floor of a call's time, so
    // the COUNT of chains is held, not their length: as many as 256 workgroups of the quality chains' image kernel hold (1024 lanes,
    // one per CU, which its 120 registers fill: SFQ_CHAINS_WANT, and encode_blocks lengthens the chains until the blocks' chains of
    // equal length stay within it) -- down to chains of 4 KiB of text (12 records of 150 bp), below which a chain's flush and index
    // entry start to show (round 4 size sweep, 0.25 / 0.5 / 1 / 2 / 3.7 GB of 150 bp reads: 12 / 12 / 12 / 24 / 49 records per
    // chain; the streams of the 1 GB prefix are 0.14 % larger at 12 records than at 49).  Round 3 floored a chain at 16 KiB: a 740 MB
    // call then had 45 k chains -- a sixth of the chip -- each as long as a 3.7 GB call's, and took longer than that call.
    // (Rounds 4 and 5 held 205 k chains, 200 workgroups: "the other models' kernels get their work done on the 56 CUs it leaves".  They
    //  do -- and then the quality chains run on alone for 5 of the call's 13 ms on 200 CUs.  Round 5, the default call at 49 / 40 / 32 /
    //  24 / 16 records a chain = 200 / 248 / 305 / 407 / 610 workgroups: 13.06 / 12.65 / 14.50 / 13.37 / 13.78 ms, decode 18.5 / 17.2 /
    //  21.7 / 20.4 / 20.1 -- a 257th workgroup waits for a CU and then walks its chains alone.)
    const u64 per_rec = std::max<u64>(1, nbytes / std::max<u64>(1, nrec));
    u64 cr = std::max<u64>(1, (nrec + SFQ_CHAINS_WANT - 1) / SFQ_CHAINS_WANT);
    cr = std::max<u64>(cr, (4096 + per_rec - 1) / per_rec);
    return (int)std::min<u64>(cr, 4096);
}
// The header prior of an encode with frozen tables: counted over this call's text -- the header model run over short runs
// of records spread over the call -- or the installed one (SFQ_PRIOR_GIVEN); leaves the frozen rows on the device.
// Two halves, so that the launching thread can queue other streams' work while the counting pass runs: _begin queues the
// pass, _finish the kernels that make the rows from its counts (no wait in either).
#define REC_PRIOR_RUN 6u          // records per run of the header prior's counting pass: the base, one that warms the field types up, four counted
#define REC_PRIOR_RUNS 32768u
#define PIN_GEN_OFF 0u
#define PIN_REC_OFF 64u
#define PIN_BYTES (PIN_REC_OFF + (size_t)PR_REC_ROWS * 256 * 4)
int rec_prior_begin(sfq_ctx* ctx, const ModelArgs& a, u64 nrec, bool given, bool counted, hipStream_t st, u32 max_hdr) {
    if (given) return SFQ_OK;
    int rc;
    if ((rc = reserve(ctx, ctx->hcnt, (size_t)REC_COUNT_COPIES * PR_REC_ROWS * 256 * 4))) return rc;
    if (counted) return SFQ_OK;                          // SFQ_PRIOR_COUNTS: the counts are there (sfq_set_prior_counts)
    HIPC(hipMemsetAsync(ctx->hcnt.p, 0, (size_t)REC_COUNT_COPIES * PR_REC_ROWS * 256 * 4, st));
    // (short runs, many of them: the pass's time is one lane's walk through its run -- 8192 runs of 18 records took 2.9 ms of
    //  every call on 128 wavefronts; the sample is the same 131 k counted records)
    const u32 run = REC_PRIOR_RUN;
    u32 nruns = (u32)std::min<u64>(std::max<u32>(1u, REC_PRIOR_RUNS / std::max<u32>(1u, ctx->sample_scale)), std::max<u64>(1, nrec / run));
    // (headers beyond the fast counting kernel's 127 bytes are walked by the general one, a lane per run: a sample of at most 4 MiB
    //  of header text -- 60 k long reads' headers, all of them, were 4.2 ms with every chain of the call waiting)
    if (max_hdr > 127) nruns = (u32)std::min<u64>(nruns, std::max<u64>(1, (4ull << 20) / ((u64)run * max_hdr)));
    const u64 stride = std::max<u64>(run, nrec / nruns);
    if ((rc = reserve(ctx, ctx->cflags, (size_t)REC_PRIOR_RUNS * 4))) return rc;
    HIPC(hipMemsetAsync(ctx->cflags.p, 0, (size_t)nruns * 4, st));
    launch_rec_count(a, nrec, stride, run, nruns, (u32*)ctx->hcnt.p, (u32*)ctx->cflags.p, st);
    return SFQ_OK;
}
// The header prior's rows.  Given ("rec.pri" handed in): unpacked and uploaded.  Else from the counts of the pass above, all on
// the device and without a wait: frequencies, the rows to stage, the frozen rows; the frequencies come back to page-locked memory
// behind them and rec_prior_blob_now() packs "rec.pri" from them when the host has nothing better to do.
// "qlt.pri" from the listed rows in page-locked memory (h: the list's head; the rows behind the first PRIOR_LIST_EAGER are
// fetched here, if there are any)
int qlt_prior_blob_now(sfq_ctx* ctx, u32* h, u32 q_rows, hipStream_t st) {
    HIPC(hipEventSynchronize(ctx->ev[20]));
    const u32 n = h[0];
    if (n > q_rows) return fail(ctx, SFQ_E_HIP, "quality prior: %u rows listed of %u", n, q_rows);
    if (n > PRIOR_LIST_EAGER) {
        const size_t done = (size_t)PRIOR_LIST_HEAD + (size_t)PRIOR_LIST_EAGER * PRIOR_LIST_ROW;
        HIPC(hipMemcpyAsync(h + done, (const u32*)ctx->plist.p + done, (size_t)(n - PRIOR_LIST_EAGER) * PRIOR_LIST_ROW * 4, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    ctx->prior_blob = pack_prior(h, q_rows);
    return SFQ_OK;
}
int rec_prior_blob_now(sfq_ctx* ctx) {
    if (!ctx->rec_blob_pending) return SFQ_OK;
    ctx->rec_blob_pending = false;
    HIPC(hipEventSynchronize(ctx->ev[24]));
    ctx->rec_prior_blob = pack_rec_prior_f((const u32*)((const u8*)ctx->pin + PIN_REC_OFF));
    return SFQ_OK;
}
int rec_prior_finish(sfq_ctx* ctx, bool given, hipStream_t st) {
    if (given) {
        std::vector<u32> hf;
        if (!unpack_rec_prior(ctx->rec_prior_blob.data(), ctx->rec_prior_blob.size(), hf)) return fail(ctx, SFQ_E_CORRUPT, "bad header prior (rec.pri)");
        return upload_rec_rows(ctx, hf, st);
    }
    int rc;
    const size_t nf = (size_t)PR_REC_ROWS * 256;
    if ((rc = reserve(ctx, ctx->rmap, (size_t)PR_REC_ROWS * 2 + 64 * 2))) return rc;
    if ((rc = reserve(ctx, ctx->hfreq, nf * 4 + (size_t)PR_REC_ROWS * 4))) return rc;           // frequencies, then the rows' sums
    if ((rc = reserve(ctx, ctx->rrows, nf * 4))) return rc;
    if ((rc = reserve(ctx, ctx->rdec, (size_t)PR_REC_ROWS * 272 * 2))) return rc;          // chains.hip RDEC_ROW
    u16* map = (u16*)ctx->rmap.p;
    launch_rec_prior_freqs((const u32*)ctx->hcnt.p, PR_REC_ROWS, (u32*)ctx->hfreq.p, (u32*)ctx->hfreq.p + nf, RDEC_LDS_ROWS, map, map + PR_REC_ROWS, st);
    launch_rec_frozen_rows((const u32*)ctx->hfreq.p, PR_REC_ROWS, (u32*)ctx->rrows.p, (u16*)ctx->rdec.p, st);
    ctx->r_hot = std::min<u32>(RDEC_LDS_ROWS, 8); ctx->r_hot_dec = RDEC_LDS_ROWS;          // (rows the sample never saw fill the list up: staging one is harmless)
    ctx->rec_copy_pending = true;
    return SFQ_OK;
}
// the frequencies on their way to the host, for rec_prior_blob_now().  Queued BEHIND the header chains on their stream: in front
// of them the copy -- a blit kernel that has to find room beside the base chains and the exception pass -- took 1.9 ms, with the
// header chains and the quality chains waiting behind it
int rec_prior_copy_back(sfq_ctx* ctx, hipStream_t st) {
    if (!ctx->rec_copy_pending) return SFQ_OK;
    ctx->rec_copy_pending = false;
    HIPC(hipMemcpyAsync((u8*)ctx->pin + PIN_REC_OFF, ctx->hfreq.p, (size_t)PR_REC_ROWS * 256 * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipEventRecord(ctx->ev[24], st));
    ctx->rec_blob_pending = true;
    return SFQ_OK;
}

// Base-model generation tables for an encode: counts gen 0, 1; decides from generation 1's would-be cost under the rows
// of generation 0 whether the tables pay (a >= 1 % gain over the initial row's 2 bits per base); if so counts on.
// Leaves ca.g_* describing which rows every generation codes with.
struct GenPlan { u32 ngen = 0; u32 bound[GEN_MAX_GENERATIONS + 1]; bool pre = false; GenBins gb; };
// the bins of the counting passes (chains.hip "counting through bins"): scratch for the largest batch of keys this call can meet
int gen_bins_reserve(sfq_ctx* ctx, u32 g_bits, u64 max_keys, hipStream_t st, GenBins& gb) {
    gb = gen_bins_plan(g_bits, max_keys);
    int rc;
    if ((rc = reserve(ctx, ctx->gbins, (size_t)gb.bins_bytes + 64))) return rc;
    if ((rc = reserve(ctx, ctx->gfill, (size_t)gb.fill_bytes))) return rc;
    gb.bins = (u16*)ctx->gbins.p; gb.fill = (u32*)ctx->gfill.p;
    HIPC(hipMemsetAsync(gb.fill, 0, (size_t)gb.fill_bytes, st));
    return SFQ_OK;
}
int gen_tables_begin(sfq_ctx* ctx, const ChainArgs& ca, u32 nblocks, u32 g_bits, u32 max_line, hipStream_t st, GenPlan& gp) {
    u32* bound = gp.bound;
    const u32 ngen = gp.ngen = gen_bounds(nblocks, bound);
    if (ngen < 3) return SFQ_OK;                                   // too few blocks to learn from
    const u64 nctx = 1ull << g_bits;
    int rc;
    if ((rc = reserve(ctx, ctx->gcnt, (size_t)nctx * 16))) return rc;
    if ((rc = reserve(ctx, ctx->grows, (size_t)nctx * 4 * ngen))) return rc;
    if ((rc = reserve(ctx, ctx->gcost, 64))) return rc;
    if (!ctx->glog.p) {
        if ((rc = reserve(ctx, ctx->glog, 1024 * 2 + 64))) return rc;
        u16 t[1024 + 32] = {0}; log2_table(t);
        t[1024] = t[1025] = 0x0303;                               // the initial row, for ChainArgs::g_init
        HIPC(hipMemcpyAsync(ctx->glog.p, t, sizeof t, hipMemcpyHostToDevice, st));
        HIPC(hipStreamSynchronize(st));
    }
    HIPC(hipMemsetAsync(ctx->gcnt.p, 0, (size_t)nctx * 16, st));
    HIPC(hipMemsetAsync(ctx->gcost.p, 0, 64, st));
    if ((rc = gen_bins_reserve(ctx, g_bits, ca.nbytes, st, gp.gb))) return rc;
    const u64 br = ca.block_reads;
    auto recs = [&](u32 b0, u32 b1) { return (u64)(b1 - b0) * br; };          // an upper bound (the last block may be short): lanes past the end idle
    u32* rows = (u32*)ctx->grows.p;
    // The pre-verdict (generations of 16384 counted records or more): every GEN_PRE-th record of generation 0 counted, then over every
    // GEN_PRE-th of generation 1 the sum, over the bases whose context that sample has seen m times, of 4 x (times it saw this base) - m.
    // For bases that do not depend on their context that sum is 0 +- sqrt(3 sum m); five deviations above 0, or a generation too
    // small to tell, and the full passes follow (finish); else the call has no tables and the base chains start an eighth of two
    // small passes into it.  (The COST under the sample's rows cannot tell: a context seen once prices the next base at 1.19 or
    // 2.42 bits, and at low coverage the chance repeats' penalty hides the true repeats' gain.)
    gp.pre = recs(bound[0], bound[1]) / gen_count_stride(recs(bound[0], bound[1])) >= 16384;
    if (gp.pre) {
        launch_gen_count_binned(ca, bound[0], bound[1], recs(bound[0], bound[1]), max_line, gp.gb, (u32*)ctx->gcnt.p, rows + nctx * 1, GEN_STEP, st, 1);
        launch_gen_count(ca, bound[1], bound[2], recs(bound[1], bound[2]), max_line, (u32*)ctx->gcnt.p, rows + nctx * 1, (const u16*)ctx->glog.p, (u64*)ctx->gcost.p, st, 1, 0);
    }
    HIPC(hipMemcpyAsync((u8*)ctx->pin + PIN_GEN_OFF, ctx->gcost.p, 16, hipMemcpyDeviceToHost, st));
    return SFQ_OK;
}
int gen_tables_finish(sfq_ctx* ctx, ChainArgs& ca, u32 g_bits, u32 max_line, hipStream_t st, const GenPlan& gp, u32* gen_on) {
    *gen_on = 0;
    ca.g_ngen = 0;
    const u32 ngen = gp.ngen; const u32* bound = gp.bound;
    if (ngen < 3) return SFQ_OK;
    const u64 nctx = 1ull << g_bits;
    const u64 br = ca.block_reads;
    auto recs = [&](u32 b0, u32 b1) { return (u64)(b1 - b0) * br; };
    u32* rows = (u32*)ctx->grows.p;
    HIPC(hipStreamSynchronize(st));
    const u64* h = (const u64*)((const u8*)ctx->pin + PIN_GEN_OFF);
    if (gp.pre) {
        const long long s4 = (long long)h[0]; const u64 m = h[1];
        if (!(s4 > 0 && (u64)s4 * (u64)s4 > 75ull * m)) return SFQ_OK;        // nothing to learn: every chain codes with the initial row
    }
    // the verdict proper: (the rest of) generation 0 counted (on top of the sample's counts), then generation 1 counted and priced
    // under generation 0's rows (the initial row would cost 2 bits = 2048 units a base)
    HIPC(hipMemsetAsync(ctx->gcost.p, 0, 64, st));
    // (round 5: the counting goes through bins -- chains.hip -- and the pass that sums a generation's bins writes the next one's rows)
    launch_gen_count_binned(ca, bound[0], bound[1], recs(bound[0], bound[1]), max_line, gp.gb, (u32*)ctx->gcnt.p, rows + nctx * 1, GEN_STEP, st, gp.pre ? 2 : 0);
    launch_gen_count(ca, bound[1], bound[2], recs(bound[1], bound[2]), max_line, (u32*)ctx->gcnt.p, rows + nctx * 1, (const u16*)ctx->glog.p, (u64*)ctx->gcost.p, st, 0, 2);
    HIPC(hipMemcpyAsync((u8*)ctx->pin + PIN_GEN_OFF, ctx->gcost.p, 16, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    const u64 cost = h[0], nbases = h[1];
    if (!nbases || cost * 100 >= nbases * 2048 * 99) return SFQ_OK;           // no gain: every chain codes with the initial row
    *gen_on = 1;
    ca.g_ngen = ngen;
    ca.g_init = (const u32*)((const u8*)ctx->glog.p + 2048);
    for (u32 g = 0; g <= ngen; g++) ca.g_bound[g] = bound[g];
    ca.g_rows[0] = nullptr; ca.g_rows[1] = nullptr;                             // generations 0 and 1: the initial row
    for (u32 g = 1; g + 1 < ngen; g++) {                                        // generation g counted -> the rows of generation g + 1
        launch_gen_count_binned(ca, bound[g], bound[g + 1], recs(bound[g], bound[g + 1]), max_line, gp.gb, (u32*)ctx->gcnt.p, rows + nctx * (g + 1), GEN_STEP, st);
        ca.g_rows[g + 1] = rows + nctx * (g + 1);
    }
    return SFQ_OK;
}

// ---- bases under the generation MATCH model (gm.hip; round 5) -----------------------------------------------------------------
// floor(1024 * log2(x)) for any x >= 1 (log2_table's arithmetic)
u32 log2fp_u32(u32 x) {
    const u32 e = 31 - (u32)__builtin_clz(x);
    u64 m = (u64)x << (31 - e);
    u32 frac = 0;
    for (int i = 0; i < 10; i++) { m = (m * m) >> 31; frac <<= 1; if (m >> 32) { frac |= 1; m >>= 1; } }
    return e * 1024 + frac;
}
u32 gm_table_bits(u64 nbytes) {          // an entry per eight bases of the call, 2^16 .. 2^24 entries
    u32 tb = 16;
    while (tb < 24 && (1ull << tb) < nbytes / 16) tb++;
    return tb;
}
#ifndef GM_CHAIN_WANT
#define GM_CHAIN_WANT 819200ull       /* base chains a call under the match model aims at ... */
#define GM_CHAIN_FLOOR 2048ull        /* ... of this much text or more each */
#endif
struct GmPlan { u32 ngen = 0; u32 bound[GEN_MAX_GENERATIONS + 1]; u32 tb = 0; u64 cap = 0; u64 r2 = 0; ChainArgs cg; ChainGeoArgs ggeo; u32* gcsz = nullptr; };
// the stage of records [0, n): line lengths, their scan, the places, the letters
int gm_stage_upto(sfq_ctx* ctx, const ChainArgs& ca, u64 n, u64 from, hipStream_t st) {
    launch_gm_lens(ca.m.line_off, ca.m.blocks, ca.block_reads, n, (u32*)ctx->gm_slen.p, st);
    launch_scan_u32((const u32*)ctx->gm_slen.p, (u64*)ctx->gm_boff.p, n, (u64*)ctx->gm_scan.p, st);
    launch_gm_soff((const u64*)ctx->gm_boff.p, n, (u64*)ctx->gm_soff.p, st);
    launch_gm_stage(ca.m, ca.block_reads, from, n, ca.nbytes, (u8*)ctx->gm_stage.p, (const u64*)ctx->gm_soff.p, (const u32*)ctx->gm_slen.p, nullptr, st);
    return SFQ_OK;
}
// Begin: the first two generations staged, generation 0's counted records indexed, every GEN_PRE-th counted record of generation 1
// priced under that index; the sums on their way to the host.
int gm_begin(sfq_ctx* ctx, const ChainArgs& ca, u32 nblocks, u64 nrec, u32 max_line, hipStream_t st, GmPlan& gp) {
    const u32 ngen = gp.ngen = gen_bounds(nblocks, gp.bound);
    if (ngen < 3) return SFQ_OK;
    const u32* bound = gp.bound;
    int rc;
    gp.tb = gm_table_bits(ca.nbytes);
    const u64 br = ca.block_reads;
    gp.r2 = std::min<u64>(nrec, (u64)bound[2] * br);
    // The verdict needs the first two generations staged, no more: a call that turns the model down -- the default workload -- keeps
    // 50 MB of stage and offsets, not the 4 GB a stage of its whole text and every record's offsets would be.  (A staged byte is a
    // byte of the text -- a base, or its line's '\n' --, and a record has at most max_line of them and one.)
    gp.cap = std::min<u64>(ca.nbytes, gp.r2 * ((u64)max_line + 1));
    if ((rc = reserve(ctx, ctx->gm_T, (size_t)8 << gp.tb))) return rc;
    if ((rc = reserve(ctx, ctx->gm_slen, (size_t)gp.r2 * 4 + 64))) return rc;
    if ((rc = reserve(ctx, ctx->gm_boff, ((size_t)gp.r2 + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->gm_soff, ((size_t)gp.r2 + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->gm_scan, ((size_t)nrec / 1024 + 4) * 8 + 65536))) return rc;
    if ((rc = reserve(ctx, ctx->gm_stage, (size_t)gp.cap + 64))) return rc;
    if ((rc = reserve(ctx, ctx->gcost, 64))) return rc;
    HIPC(hipMemsetAsync(ctx->gm_T.p, 0xFF, (size_t)8 << gp.tb, st));
    HIPC(hipMemsetAsync(ctx->gcost.p, 0, 64, st));
    if ((rc = gm_stage_upto(ctx, ca, gp.r2, 0, st))) return rc;
    gp.cg = ca;
    gp.cg.st_buf = (const u8*)ctx->gm_stage.p; gp.cg.st_bytes = gp.cap; gp.cg.st_off = (const u64*)ctx->gm_soff.p; gp.cg.st_len = (const u32*)ctx->gm_slen.p;
    auto recs = [&](u32 b0, u32 b1) { return (u64)(b1 - b0) * br; };
    launch_gm_insert(gp.cg, bound[0], bound[1], recs(bound[0], bound[1]), max_line, (u64*)ctx->gm_T.p, gp.tb, st);
    u16 costs[8];
    for (u32 lv = 0; lv < 4; lv++) {
        const u32 fo = lv == 0 ? 64u : lv == 1 ? 40u : lv == 2 ? 24u : 16u;
        costs[lv] = (u16)(12 * 1024 - log2fp_u32(4096u - 3u * fo)); costs[4 + lv] = (u16)(12 * 1024 - log2fp_u32(fo));
    }
    const u64 r1 = (u64)bound[1] * br;
    launch_gm_price(gp.cg, r1, gp.r2, (u64)gen_count_stride(recs(bound[1], bound[2])) * GEN_PRE, max_line, r1, (const u8*)ctx->gm_stage.p, gp.cap, (const u64*)ctx->gm_soff.p,
                    (const u32*)ctx->gm_slen.p, (const u64*)ctx->gm_T.p, gp.tb, costs, (u64*)ctx->gcost.p, st);
    HIPC(hipMemcpyAsync((u8*)ctx->pin + PIN_GEN_OFF, ctx->gcost.p, 16, hipMemcpyDeviceToHost, st));
    return SFQ_OK;
}
// Finish: the verdict (on if generation 1's sample costs 1 % less than two bits a base); then the rest staged, the other
// generations' counted records indexed, the tokens planned -- the chains (launch_gm_code) are the caller's
int gm_finish(sfq_ctx* ctx, ChainArgs& ca, u64 nrec, u32 max_line, hipStream_t st, GmPlan& gp, u32* gen_on) {
    *gen_on = 0;
    const u32 ngen = gp.ngen; const u32* bound = gp.bound;
    if (ngen < 3) return SFQ_OK;
    HIPC(hipStreamSynchronize(st));
    const u64* h = (const u64*)((const u8*)ctx->pin + PIN_GEN_OFF);
    const u64 cost = h[0], nbases = h[1];
    if (!nbases || cost * 100 >= nbases * 2048 * 99) return SFQ_OK;           // nothing to gain: every chain codes with the initial row
    *gen_on = 1;
    int rc;
    // the whole call's stage now (buffers that grow are new ones: everything is staged again, the first two generations included --
    // their places, and so the index's entries, do not change)
    gp.cap = ca.nbytes;
    if ((rc = reserve(ctx, ctx->gm_slen, (size_t)nrec * 4 + 64))) return rc;
    if ((rc = reserve(ctx, ctx->gm_boff, ((size_t)nrec + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->gm_soff, ((size_t)nrec + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->gm_stage, (size_t)gp.cap + 64))) return rc;
    if ((rc = reserve(ctx, ctx->gm_tok, (size_t)gp.cap + 64))) return rc;
    if ((rc = gm_stage_upto(ctx, ca, nrec, 0, st))) return rc;
    gp.cg.st_buf = (const u8*)ctx->gm_stage.p; gp.cg.st_bytes = gp.cap; gp.cg.st_off = (const u64*)ctx->gm_soff.p; gp.cg.st_len = (const u32*)ctx->gm_slen.p;
    const u64 br = ca.block_reads;
    auto recs = [&](u32 b0, u32 b1) { return (u64)(b1 - b0) * br; };
    for (u32 g = 1; g + 1 < ngen; g++) launch_gm_insert(gp.cg, bound[g], bound[g + 1], recs(bound[g], bound[g + 1]), max_line, (u64*)ctx->gm_T.p, gp.tb, st);
    gp.cg.m = ca.m; gp.cg.csz = gp.gcsz ? gp.gcsz : ca.csz;
    gp.cg.geo = gp.ggeo;                                           // (the base chains' own geometry: api.cpp sfq_encode_blocks)
    gp.cg.g_ngen = ngen;
    for (u32 g = 0; g <= ngen; g++) gp.cg.g_bound[g] = bound[g];
    launch_gm_plan(gp.cg, ca.seg_len ? (u64)ca.geo.nchains : nrec, (const u8*)ctx->gm_stage.p, gp.cap, (const u64*)ctx->gm_soff.p, (const u32*)ctx->gm_slen.p,
                   (const u64*)ctx->gm_T.p, gp.tb, (u8*)ctx->gm_tok.p, st);
    return SFQ_OK;
}

// frame.hip k_text_fingerprint of a device-resident text (one small kernel and eight bytes back)
int text_fingerprint(sfq_ctx* ctx, const u8* d_text, u64 nbytes, hipStream_t st, u64* out) {
    int rc;
    if ((rc = reserve(ctx, ctx->status, 256))) return rc;
    u64* d = (u64*)ctx->status.p + 24;                    // (status words 48..49: nothing else lives there)
    HIPC(hipMemsetAsync(d, 0, 8, st));
    launch_text_fingerprint(d_text, nbytes, d, st);
    HIPC(hipMemcpyAsync(out, d, 8, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return SFQ_OK;
}

// pieces of a page-locked buffer, handed out front to back
struct Bump {
    u8* p; size_t off, cap;
    template <typename T> T* take(size_t n) { off = (off + 15) & ~(size_t)15; T* r = reinterpret_cast<T*>(p + off); off += n * sizeof(T); return off <= cap ? r : nullptr; }
};
float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

}  // namespace

extern "C" {

const char* sfq_stream_name(int s) {
    static const char* names[SFQ_NSTREAMS] = { "rec", "gen", "qlt", "gen.Ns", "gen.Nn", "rec.x", "usr.x", "usr.x.q", "usr.pfg", "usr.pfq",
                                               "gen.lc", "usr.lrec", "usr.lgen", "usr.lqlt" };
    return (s >= 0 && s < SFQ_NSTREAMS) ? names[s] : "";
}
int sfq_abi_version(void) { return SFQ_ABI_VERSION; }

int sfq_ctx_create(sfq_ctx** out, int hip_device) {
    if (!out) return SFQ_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SFQ_E_HIP;      // no CPU fallback, by design
    if (hip_device < 0 || hip_device >= n) return SFQ_E_ARG;
    if (hipSetDevice(hip_device) != hipSuccess) return SFQ_E_HIP;
    sfq_ctx* ctx = new sfq_ctx();
    ctx->dev = hip_device;
    // one stream per model (quality, bases, headers, framing)
    // (the context's stream carries the call's critical path -- framing, the quality sample, the quality chains -- and the
    //  first auxiliary stream the header model's two steps, the call's second-longest chain of kernels: both at the higher
    //  priority.  Round 4: with the EXCEPTION stream there instead -- rounds 2 and 3 -- a step of the distributed path, whose
    //  chains are queued earlier in the call, left the header tokens 9.6 ms behind the quality chains: 15.1 ms a step against
    //  13.9 this way; a call with priors of its own takes 13.7 either way)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if ((prio_hi != prio_lo ? hipStreamCreateWithPriority(&ctx->st, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking)) != hipSuccess) { delete ctx; return SFQ_E_HIP; }
    // (the runtime maps streams of one priority onto three hardware queues: the framing stream shares one with the base
    //  model's, and two streams that share a queue run one after the other -- keep long kernels off the framing stream)
    for (int i = 0; i < 3; i++) {
        const char* px = getenv("SFQ_EXP_PRIO");          /* scratch experiments: a mask of auxiliary streams created at the higher priority (default 1: the headers') */
        const int pm = px ? atoi(px) : 1;
        const hipError_t e = (((pm >> i) & 1) && prio_hi != prio_lo) ? hipStreamCreateWithPriority(&ctx->st_aux[i], hipStreamNonBlocking, prio_hi)
                                                              : hipStreamCreateWithFlags(&ctx->st_aux[i], hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return SFQ_E_HIP; }
    }
    for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { delete ctx; return SFQ_E_HIP; }
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    ctx->dev_total = tot;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->dev) == hipSuccess && cus > 0) ctx->wave_slots = (u32)cus * 32u; }
    ctx->table_budget = tot / 10 * 7;
    *out = ctx;
    return SFQ_OK;
}

void sfq_ctx_destroy(sfq_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->st);
    DevBuf* all[] = { &ctx->tab.q_slots, &ctx->tab.q_hdr, &ctx->tab.p_slots, &ctx->tab.p_hdr, &ctx->tab.g_tab,
        &ctx->chunk_counts, &ctx->chunk_base, &ctx->scan_tmp, &ctx->line_off, &ctx->status, &ctx->blocks, &ctx->arena,
        &ctx->blk_stream_off, &ctx->stream_total, &ctx->lens, &ctx->blob_off, &ctx->blob, &ctx->in_stage, &ctx->out_stage,
        &ctx->slen, &ctx->qlen, &ctx->pfg, &ctx->pfq, &ctx->soff, &ctx->qoff, &ctx->seq_stage, &ctx->qual_stage,
        &ctx->hdr_stage, &ctx->hlen, &ctx->hoff, &ctx->hso, &ctx->hsc, &ctx->rsize, &ctx->roff, &ctx->d_first,
        &ctx->hist, &ctx->rows66, &ctx->prior_w, &ctx->prior_wovf, &ctx->prior_ls, &ctx->prior_lh, &ctx->tickets,
        &ctx->hcnt, &ctx->hfreq, &ctx->rrows, &ctx->rdec, &ctx->rmap, &ctx->rflags, &ctx->rtok, &ctx->ptmp, &ctx->qrows, &ctx->qdec, &ctx->qesc, &ctx->qw, &ctx->csz, &ctx->coff, &ctx->gcnt, &ctx->grows, &ctx->glog, &ctx->gcost, &ctx->gbins, &ctx->gfill, &ctx->gm_T, &ctx->gm_slen, &ctx->gm_boff, &ctx->gm_soff, &ctx->gm_scan, &ctx->gm_stage, &ctx->gm_tok, &ctx->gm_csz, &ctx->gm_idx, &ctx->excf, &ctx->cflags, &ctx->segn, &ctx->segoff, &ctx->segrec, &ctx->pslot, &ctx->plist, &ctx->chn_len, &ctx->chn_off, &ctx->chn_out,
        &ctx->oflags, &ctx->okbytes, &ctx->ofpos, &ctx->okoff, &ctx->ofilt, &ctx->orecmap, &ctx->olist, &ctx->line_off_o, &ctx->ono, &ctx->opiece,
        &ctx->otxt[0], &ctx->otxt[1], &ctx->otxt[2], &ctx->osize_all, &ctx->oroff_all, &ctx->oroff_k, &ctx->ocnt };
    for (DevBuf* b : all) release(*b);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pin2) (void)hipHostFree(ctx->pin2);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto& s : ctx->st_aux) if (s) (void)hipStreamDestroy(s);
    if (ctx->st) (void)hipStreamDestroy(ctx->st);
    delete ctx;
}

const char* sfq_last_error(const sfq_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }
uint64_t sfq_ctx_device_memory(const sfq_ctx* ctx) { return ctx ? ctx->dev_total : 0; }
int sfq_ctx_set_table_budget(sfq_ctx* ctx, uint64_t bytes) { if (!ctx) return SFQ_E_ARG; ctx->table_budget = bytes; return SFQ_OK; }
void* sfq_ctx_stream(sfq_ctx* ctx) { return ctx ? (void*)ctx->st : nullptr; }
int sfq_ctx_synchronize(sfq_ctx* ctx) {
    if (!ctx) return SFQ_E_ARG;
    HIPC(hipStreamSynchronize(ctx->st));
    return SFQ_OK;
}

void* sfq_host_alloc(sfq_ctx* ctx, uint64_t bytes) {
    if (!ctx || !bytes) return nullptr;
    if (hipSetDevice(ctx->dev) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void sfq_host_free(sfq_ctx* ctx, void* p) {
    if (!ctx || !p) return;
    (void)hipSetDevice(ctx->dev);
    (void)hipHostFree(p);
}

uint64_t sfq_encode_bound(uint64_t n) { return n + n / 2 + 4096; }

// -------------------------------------------------------------------------------------------------
// compress
// -------------------------------------------------------------------------------------------------
// priors_only: stop once the call's priors ("qlt.pri", and "rec.pri" with frozen tables) are built (sfq_build_priors)
// A call that fails returns from wherever it is, possibly between the fork of the model streams and their join: the next
// call then starts by waiting for the device, so that nothing of the failed one still runs over the shared scratch.
struct Settle {
    sfq_ctx* ctx; bool ok = false;
    explicit Settle(sfq_ctx* c) : ctx(c) { if (ctx->unsettled) { (void)hipDeviceSynchronize(); ctx->unsettled = false; } }
    ~Settle() { if (!ok) ctx->unsettled = true; }
};
static int encode_body(sfq_ctx* ctx, const u8* d_fastq, u64 nbytes, const sfq_params* pp, u8* d_out, u64 out_cap,
                       sfq_result* res, u32 force_models, bool priors_only);
static int encode_impl(sfq_ctx* ctx, const u8* d_fastq, u64 nbytes, const sfq_params* pp, u8* d_out, u64 out_cap,
                       sfq_result* res, u32 force_models, bool priors_only = false) {
    if (!ctx || !d_fastq || !pp || (!d_out && !priors_only) || !res) return fail(ctx, SFQ_E_ARG, "null argument");
    if (nbytes == 0) return fail(ctx, SFQ_E_FORMAT, "empty input");
    HIPC(hipSetDevice(ctx->dev));
    Settle settle(ctx);
    const int rc = encode_body(ctx, d_fastq, nbytes, pp, d_out, out_cap, res, force_models, priors_only);
    settle.ok = rc == SFQ_OK;
    if (rc == SFQ_OK && !priors_only) ctx->blobs_from_encode = true;       // (sfq_build_priors leaves installed priors: SFQ_PRIOR_GIVEN reads them)
    return rc;
}
static int encode_body(sfq_ctx* ctx, const u8* d_fastq_in, u64 nbytes_in, const sfq_params* pp, u8* d_out, u64 out_cap,
                       sfq_result* res, u32 force_models, bool priors_only) {
    const u8* d_fastq = d_fastq_in; u64 nbytes = nbytes_in;       // (format 6 with oversize records: the text without them, below)
    sfq_params p = *pp;
    HostTimes ht;
    p.level = clamp_level(p.level);
    u32 models = force_models ? force_models : (p.models ? p.models : SFQ_M_ALL);
    if (p.kernel > 2) return fail(ctx, SFQ_E_ARG, "kernel %u: 0 = default kernels, 1 = lane-per-block cross-check kernels, 2 = default kernels with the reference's coding of the base exceptions", p.kernel);
    const bool exc_classic = p.kernel == 2;              // (everywhere below, kernel is 0 or 1)
    if (exc_classic) p.kernel = 0;
    if (p.tables > SFQ_TABLES_AUTO) return fail(ctx, SFQ_E_ARG, "tables %u: 0 = adaptive, 1 = frozen, 2 = by the size of the text", p.tables);
    bool small_auto = false;
    if (p.tables == SFQ_TABLES_AUTO) {                          // include/slimfastq_amd.h
        // (sfq_count_priors resolves it the way the SFQ_PRIOR_COUNTS encode that follows does: never by the size of one rank's share)
        small_auto = nbytes < (64ull << 20) && p.prior_step != SFQ_PRIOR_GIVEN && p.prior_step != SFQ_PRIOR_COUNTS && !ctx->counts_only;
        p.tables = small_auto ? SFQ_TABLES_ADAPTIVE : SFQ_TABLES_FROZEN;
        if (small_auto) { p.prior_step = 0; if (p.block_reads == SFQ_BLOCK_AUTO) p.block_reads = 65536; }
    }
    memset(res, 0, sizeof *res);
    res->abi_version = SFQ_ABI_VERSION;
    hipStream_t st = ctx->st;
    int rc;
    if (p.prior_step != SFQ_PRIOR_COUNTS) ctx->counts.valid = false;      // this call may sample into hist / hcnt itself

    // ---- framing -------------------------------------------------------------------------------
    HIPC(hipEventRecord(ctx->ev[0], st));
    if ((rc = reserve(ctx, ctx->status, 256))) return rc;
    // the quality sample's counters (16 MiB) are cleared here, ahead of the framing kernels: behind them the memset sat on
    // the quality model's critical path for a millisecond (it shares the chip with the counting passes by then)
    const u32 q_rows0 = p.level == 1 ? (1u << 12) : (1u << 16);
    bool hist_cleared = false;
    if (p.block_reads && (models & SFQ_M_QLT) && (p.prior_step || p.tables == SFQ_TABLES_FROZEN) && p.kernel == 0 && p.prior_step != SFQ_PRIOR_COUNTS) {
        if ((rc = ensure_prior_buffers(ctx, q_rows0))) return rc;
        HIPC(hipMemsetAsync(ctx->hist.p, 0, (size_t)q_rows0 * 64 * 4, st));
        hist_cleared = true;
    }
    const bool legacy = p.block_reads == 0;
    // frozen tables: the framing marks the records the pass over the N / quality-0 / case exceptions has to look at
    const bool want_marks = p.tables == SFQ_TABLES_FROZEN && p.block_reads != 0 && p.kernel == 0 && (models & SFQ_M_GEN) && (!priors_only || ctx->counts_only);
    u64 nrec = 0;
    hipStream_t vst = legacy ? st : ctx->st_aux[0];             // where k_validate_records runs
    // the line index of the current text (d_fastq, nbytes) and the per-record checks
    auto frame = [&]() -> int {
        int rc;
        const u32 tiles = frame_tiles(nbytes);
        if ((rc = reserve(ctx, ctx->chunk_base, (size_t)tiles * 8))) return rc;          // the tiles' look-back words
        if ((rc = reserve(ctx, ctx->scan_tmp, ((size_t)(nbytes / 16384) / 1024 + 4) * 8 + 65536))) return rc;      // (the scans of the packing: one entry per 1024 blocks)
        // One pass (frame.hip k_frame): the index is sized before the lines are counted -- room for a line every 64 bytes (150 bp
        // reads: one per 87; 0.125 + 0.03 bytes of index and marks per byte of text -- round 4 guessed a line per 32 bytes, 0.9 GB for
        // the default call: ADVICE), or what an earlier call left; a text of shorter lines is framed again with the count the pass returns
        u64 cap = std::max<u64>(ctx->line_off.cap / 8 > 2 ? ctx->line_off.cap / 8 - 2 : 0, nbytes / 64 + 1024);
        u64 nlines = 0;
        for (int attempt = 0; ; attempt++) {
            if ((rc = reserve(ctx, ctx->line_off, (size_t)(cap + 2) * 8))) return rc;
            const u64 ecap = cap / 4 + 2;
            if (want_marks) {
                if ((rc = reserve(ctx, ctx->excf, (size_t)ecap))) return rc;
                HIPC(hipMemsetAsync(ctx->excf.p, 0, (size_t)ecap, st));
            }
            HIPC(hipMemsetAsync(ctx->status.p, 0, 256, st));
            HIPC(hipMemsetAsync(ctx->chunk_base.p, 0, (size_t)tiles * 8, st));
            void* d_fo = (u8*)ctx->status.p + 224;                                   // (status bytes 224..239: nothing else lives there)
            HIPC(hipEventRecord(ctx->ev[12], st));
            launch_frame(d_fastq, nbytes, (u64*)ctx->chunk_base.p, (u64*)ctx->line_off.p, cap, (u32*)ctx->status.p, want_marks ? (u8*)ctx->excf.p : nullptr, ecap, d_fo, st);
            HIPC(hipEventRecord(ctx->ev[23], st));
            struct { u64 nlines; u32 tripped, pad; } fo = {0, 0, 0};
            u8 last_byte = 0;
            HIPC(hipMemcpyAsync(&fo, d_fo, 16, hipMemcpyDeviceToHost, st));
            HIPC(hipMemcpyAsync(&last_byte, d_fastq + nbytes - 1, 1, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (fo.tripped) return fail(ctx, SFQ_E_HIP, "framing: a tile waited in vain for the tiles before it");
            if (last_byte != '\n') return fail(ctx, SFQ_E_FORMAT, "fastq file: record seems truncated (no final newline)");   // usrs.cpp:169-172
            nlines = fo.nlines;
            if (nlines <= cap) break;
            if (attempt) return fail(ctx, SFQ_E_HIP, "framing: %llu lines after a pass that counted fewer", (unsigned long long)nlines);
            cap = nlines;
        }
        if (nlines == 0 || (nlines & 3)) return fail(ctx, SFQ_E_FORMAT, "fastq file: %llu lines is not a multiple of 4", (unsigned long long)nlines);
        nrec = nlines / 4;
        if (nrec >= 3000000000ULL) return fail(ctx, SFQ_E_UNSUPPORTED, "more than 3e9 records (usrs.cpp:394)");
        // headers up to 8190 bytes (usrs.hpp:34; format 6 sends longer ones to its oversize streams, below), base / quality lines
        // of any length: the block format codes them the usual way (a block's regions are sized by its text), format 6 has its
        // oversize streams
        // (the block format: the per-record checks run on a stream of their own, beside the block descriptors and the quality sample's
        //  histogram)
        if (vst != st) { HIPC(hipEventRecord(ctx->ev[22], st)); HIPC(hipStreamWaitEvent(vst, ctx->ev[22], 0)); }
        launch_validate_lines((const u64*)ctx->line_off.p, nrec, legacy ? 0x3ffffffeu : 0x1ffeu, 0x3ffffffeu, (u32*)ctx->status.p, vst);
        return SFQ_OK;
    };
    // (an encode from summed counts right behind sfq_count_priors on the same buffer: that call's line index, marks and checks stand)
    bool reframe = !(p.prior_step == SFQ_PRIOR_COUNTS && !legacy && ctx->framed.valid && ctx->framed.ptr == d_fastq && ctx->framed.nbytes == nbytes &&
                     (!want_marks || ctx->framed.marks));
    if (!reframe) {
        // the same address and size are not the same TEXT (a caller's buffer refilled in place): 65 536 sixteen-byte pieces spread
        // over the text must hash as they did when sfq_count_priors framed it, or it is framed again
        u64 print = 0;
        if ((rc = text_fingerprint(ctx, d_fastq, nbytes, st, &print))) return rc;
        if (print != ctx->framed.print) reframe = true;
    }
    ctx->framed.valid = false;
    if (!reframe) nrec = ctx->framed.nrec;
    else if ((rc = frame())) return rc;
    // ---- format 6: the reference's oversize records (usrs.cpp:269-301; frame.hip) ---------------------------------------
    u64 nrec_file = nrec;                       // records of the file, the oversize ones included ("num_records", usrs.cpp:405)
    u32 n_over = 0;
    const u8* d_file = d_fastq;                 // the whole text and its line index (ctx->line_off_o once the text is split)
    if (legacy) {
        u32 h4[4] = {0, 0, 0, 0};
        HIPC(hipMemcpyAsync(h4, ctx->status.p, 16, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        if (h4[0] == 0 && h4[3] != 0) {         // (a format error is reported below, from the same status words)
            if (nrec > 0xFFFFFFF0ull) return fail(ctx, SFQ_E_UNSUPPORTED, "oversize records in a file of more than 2^32 records");
            u32* d_first = (u32*)ctx->status.p + 8; u32* d_solid = d_first + 1; u32* d_bad = d_first + 2;
            u32 init3[3] = { 0xFFFFFFFFu, 0, 0 };
            HIPC(hipMemcpyAsync(d_first, init3, 12, hipMemcpyHostToDevice, st));
            launch_over_first((const u64*)ctx->line_off.p, nrec, d_first, st);
            u32 first = 0;
            HIPC(hipMemcpyAsync(&first, d_first, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (first == 0xFFFFFFFFu) return fail(ctx, SFQ_E_UNSUPPORTED, "every record is over the reference's line limits (usrs.cpp:190-197: \"all records were oversized\")");
            launch_over_solid(d_fastq, (const u64*)ctx->line_off.p, first, d_solid, st);
            u32 solid = 0;
            HIPC(hipMemcpyAsync(&solid, d_solid, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if ((rc = reserve(ctx, ctx->oflags, (size_t)nrec * 4))) return rc;
            if ((rc = reserve(ctx, ctx->okbytes, (size_t)nrec * 4))) return rc;
            if ((rc = reserve(ctx, ctx->ofpos, ((size_t)nrec + 1) * 8))) return rc;
            if ((rc = reserve(ctx, ctx->okoff, ((size_t)nrec + 1) * 8))) return rc;
            if ((rc = reserve(ctx, ctx->scan_tmp, ((size_t)nrec / 1024 + 4) * 8 + 65536))) return rc;
            launch_over_flags((const u64*)ctx->line_off.p, nrec, solid, (u32*)ctx->oflags.p, (u32*)ctx->okbytes.p, d_bad, st);
            launch_scan_u32((const u32*)ctx->oflags.p, (u64*)ctx->ofpos.p, nrec, (u64*)ctx->scan_tmp.p, st);
            launch_scan_u32((const u32*)ctx->okbytes.p, (u64*)ctx->okoff.p, nrec, (u64*)ctx->scan_tmp.p, st);
            u64 h_nover = 0, h_kbytes = 0; u32 h_bad = 0, first_flag = 0;
            HIPC(hipMemcpyAsync(&h_nover, (u64*)ctx->ofpos.p + nrec, 8, hipMemcpyDeviceToHost, st));
            HIPC(hipMemcpyAsync(&h_kbytes, (u64*)ctx->okoff.p + nrec, 8, hipMemcpyDeviceToHost, st));
            HIPC(hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipMemcpyAsync(&first_flag, (u32*)ctx->oflags.p + first, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (h_bad) return fail(ctx, SFQ_E_UNSUPPORTED, "a quality line over 65534 bytes beside a base line within the limit: the reference's own archive of such a record does not decode (usrs.cpp:340-345 run before 372-373)");
            if (first_flag) return fail(ctx, SFQ_E_UNSUPPORTED, "the first record within determine_record's limits (usrs.cpp:218-229) is oversize for get_record (usrs.cpp:333-338): a base line of exactly 65535 bytes");
            if (h_nover == nrec) return fail(ctx, SFQ_E_UNSUPPORTED, "every record is over the reference's line limits");
            if (h_nover) {
                n_over = (u32)h_nover;
                if ((rc = reserve(ctx, ctx->ofilt, (size_t)h_kbytes + 16))) return rc;
                if ((rc = reserve(ctx, ctx->orecmap, (size_t)(nrec - h_nover) * 4 + 16))) return rc;
                if ((rc = reserve(ctx, ctx->olist, (size_t)h_nover * 4 + 16))) return rc;
                launch_over_split(d_fastq, (const u64*)ctx->line_off.p, nrec, (const u32*)ctx->oflags.p, (const u64*)ctx->ofpos.p, (const u64*)ctx->okoff.p,
                                  (u8*)ctx->ofilt.p, (u32*)ctx->orecmap.p, (u32*)ctx->olist.p, st);
                std::swap(ctx->line_off, ctx->line_off_o);               // the file's line index stays for the oversize pass
                d_fastq = (const u8*)ctx->ofilt.p; nbytes = h_kbytes;
                if ((rc = frame())) return rc;                           // the text the models see
                if (nrec + h_nover != nrec_file) return fail(ctx, SFQ_E_HIP, "oversize split: %llu + %llu records of %llu", (unsigned long long)nrec, (unsigned long long)h_nover, (unsigned long long)nrec_file);
            }
        }
    }

    // SFQ_BLOCK_AUTO: blocks of about 376 KiB of text (1024 records of 150 bp; ~6 records of 60 kb): the unit of
    // parallelism is the block, and a fixed record count would leave long-read inputs with a handful of huge blocks
    if (p.block_reads == SFQ_BLOCK_AUTO) {
        const u64 r = std::max<u64>(1, (376ull << 10) / std::max<u64>(1, nbytes / nrec));
        p.block_reads = (u32)std::min<u64>(r >= 64 ? (r & ~63ull) : r, 4096);
    }
    const u32 block_reads = p.block_reads ? p.block_reads : (u32)std::min<u64>(nrec, 0xFFFFFFFFu);
    const u64 nblocks64 = (nrec + block_reads - 1) / block_reads;
    if (nblocks64 > (1u << 24)) return fail(ctx, SFQ_E_ARG, "too many blocks (%llu)", (unsigned long long)nblocks64);
    const u32 nblocks = (u32)nblocks64;
    // base-model context bits: the level's in single-block mode (reference parity); otherwise capped so a
    // block's table is not much larger than the block (a block cannot fill more contexts than it has bases)
    const bool frozen = p.tables == SFQ_TABLES_FROZEN && p.block_reads != 0 && p.kernel == 0;
    int g_bits = p.gen_bits ? p.gen_bits : level_gen_bits(p.level);
    if (!p.gen_bits && p.block_reads) {
        // (frozen tables: one table for the whole call, so the cap follows the call's bases, not the block's)
        const double bases = (frozen ? (double)nrec : (double)block_reads) * ((double)nbytes / (double)nrec) * 0.5;
        int cap = 12;
        while (cap < 26 && (double)(1u << cap) < bases) cap += 2;
        g_bits = std::min(g_bits, cap);
    }
    if (g_bits < 2 || g_bits > 26 ) return fail(ctx, SFQ_E_ARG, "gen_bits %d out of range", g_bits);
    if ((rc = reserve(ctx, ctx->blocks, (size_t)nblocks * sizeof(BlockDesc)))) return rc;
    launch_block_prepare(d_fastq, (const u64*)ctx->line_off.p, nrec, block_reads, (BlockDesc*)ctx->blocks.p, nblocks, nbytes, p.level, g_bits, st);
    // (format 6 with oversize records: three more regions behind the block's, for "usr.lrec" / "usr.lgen" / "usr.lqlt")
    const u64 arena_main = (((u64)nbytes * 17 / 2 + (u64)nblocks * 1024 + 4096) + 15) & ~15ull;      // frame.hip k_block_prepare
    const u64 over_bytes = n_over ? nbytes_in - nbytes : 0;
    const u64 over_cap = n_over ? ((over_bytes + over_bytes / 4 + 4096 + 15) & ~15ull) : 0;
    if (over_cap > 0xFFFFFFF0ull) return fail(ctx, SFQ_E_UNSUPPORTED, "more than 3.4 GB of oversize records");
    if ((rc = reserve(ctx, ctx->arena, (size_t)(arena_main + 3 * over_cap)))) return rc;
    // ---- models --------------------------------------------------------------------------------
    const u32 q_rows = p.level == 1 ? (1u << 12) : (1u << 16);       // qlts.hpp:36-40
    // warm start (format 7 only): count a sample, build the prior rows, keep a host copy for "qlt.pri"
    ctx->prior_on = false;
    u32 prior_step = p.block_reads ? p.prior_step : 0;
    // SFQ_PRIOR_GIVEN: the priors installed with sfq_set_qlt_prior / sfq_set_rec_prior (e.g. built once for a file that
    // several GPUs share, sfq_build_priors) instead of priors counted over this call's text
    const bool given = prior_step == SFQ_PRIOR_GIVEN;
    const bool counted = prior_step == SFQ_PRIOR_COUNTS;       // the sample's counts are installed (sfq_set_prior_counts): nothing is counted here
    if (counted) {
        const bool need_rec = frozen && (models & SFQ_M_REC);
        if (!ctx->counts.valid || !ctx->hist.p || (need_rec && !ctx->hcnt.p))
            return fail(ctx, SFQ_E_ARG, "SFQ_PRIOR_COUNTS: no counts installed (sfq_set_prior_counts / sfq_count_priors come first; any other encode drops them)");
        if (ctx->counts.q_rows != q_rows) return fail(ctx, SFQ_E_ARG, "SFQ_PRIOR_COUNTS: the counts were installed for %u quality contexts, level %d has %u", ctx->counts.q_rows, p.level, q_rows);
        if (need_rec && !ctx->counts.rec) return fail(ctx, SFQ_E_ARG, "SFQ_PRIOR_COUNTS: the installed counts hold no header sample (they were counted for adaptive tables), frozen tables need one");
    }
    if (given && ctx->prior_blob.empty()) return fail(ctx, SFQ_E_ARG, "SFQ_PRIOR_GIVEN: no quality prior installed (sfq_set_qlt_prior)");
    if (given && frozen && (models & SFQ_M_REC) && ctx->rec_prior_blob.empty()) return fail(ctx, SFQ_E_ARG, "SFQ_PRIOR_GIVEN: no header prior installed (sfq_set_rec_prior)");
    if (!given) { ctx->prior_blob.clear(); ctx->rec_prior_blob.clear(); }
    ctx->rec_blob_pending = false; ctx->rec_copy_pending = false;
    ctx->chain_blob.clear();
    if (frozen && !prior_step) prior_step = SFQ_PRIOR_AUTO;          // frozen rows ARE the prior
    u32 slots = 0;
    const u32 KR = 2;                                                  // the default kernels take table slots in pairs (two blocks per wave)
    const u32 nblocks_r = (nblocks + KR - 1) / KR * KR;
    ModelArgs a;
    u32* tickets = nullptr;
    auto setup_tables = [&]() -> int {
        int rc;
        if ((rc = ensure_tables(ctx, nblocks_r, q_rows, (u32)g_bits, frozen ? (models & ~(SFQ_M_QLT | SFQ_M_GEN)) : models, &slots))) return rc;
        if (p.kernel == 0) {
            if (slots < KR) return fail(ctx, SFQ_E_NOMEM, "table budget %llu B holds %u block slot(s); the kernels need %u", (unsigned long long)ctx->table_budget, slots, KR);
            slots &= ~(KR - 1);
        }
        if ((rc = advance_epoch(ctx, nblocks))) return rc;
        fill_model_args(ctx, a, nblocks, p.level, (u32)g_bits, p.block_reads != 0);          // the block format is lossless (dev_common.h)
        a.fq = d_fastq;
        a.rec_map = n_over ? (const u32*)ctx->orecmap.p : nullptr;
        // Default kernels are persistent: one workgroup per pair of table slots, blocks handed out through ticket counters.
        if ((rc = reserve(ctx, ctx->tickets, 64))) return rc;
        HIPC(hipMemsetAsync(ctx->tickets.p, 0, 64, st));
        tickets = (u32*)ctx->tickets.p;
        return SFQ_OK;
    };
    // The four models are independent chains over the same text: each runs on its own HIP stream, forked
    // from / joined to the context's stream with events, so their kernels overlap on the chip.
    // (frozen tables: the header and base models each need one host decision in the middle -- their counting passes are
    //  queued first, ahead of the quality prior, and finished in the order they come back)
    const u32 order[4] = { SFQ_M_QLT, frozen ? SFQ_M_REC : SFQ_M_GEN, SFQ_M_USR, frozen ? SFQ_M_GEN : SFQ_M_REC };
    const int tslot[4] = { SFQ_T_QLT, frozen ? SFQ_T_REC : SFQ_T_GEN, SFQ_T_USR, frozen ? SFQ_T_GEN : SFQ_T_REC };
    hipStream_t mst[4] = { st, ctx->st_aux[0], ctx->st_aux[1], ctx->st_aux[2] };
    // auto: sample about 24 M quality symbols (~160 k records of 150 bp: one per lane of the histogram kernel, all of them on
    // the chip at once; 60 M symbols code 0.06 % smaller; for long reads far fewer records --
    // the histogram walks a record on one lane, so its time is set by the longest record, not the sample size)
    // (of a long record only the first PRIOR_SYMBOLS count: one lane walks a record, so the sample's time is set by
    //  the longest walk)
    if (prior_step == SFQ_PRIOR_AUTO || (ctx->counts_only && prior_step == 0)) {
        const u64 per_rec = std::min<u64>(std::max<u64>(1, nbytes / nrec / 2), PRIOR_SYMBOLS);
        prior_step = (u32)std::min<u64>(std::max<u64>(1, nrec * per_rec / 24000000ull), 0x7FFFFFFFull);
        // (short records: not more of them than the histogram kernel's workgroups hold on the chip at once -- 512 of 256 lanes, a
        //  record per lane: a second round of workgroups waits for the first while the other models' kernels take the chip)
        if (nbytes / nrec <= 4000) prior_step = (u32)std::max<u64>(prior_step, (nrec + 131071) / 131072);
    }
    if (ctx->counts_only && prior_step && prior_step < SFQ_PRIOR_COUNTS)        // sfq_count_priors: a share of the job's sample
        prior_step = (u32)std::min<u64>((u64)prior_step * std::max<u32>(1u, ctx->sample_scale), 0x7FFFFFFFull);
    // The histogram of the quality sample goes FIRST and alone: its workgroups take 64 KiB of LDS each, and beside the other models'
    // early passes -- thousands of small workgroups that keep every CU's LDS in use -- they wait for room: 0.5 ms alone, 6.7 ms
    // beside them, and the quality chains wait for it.  Everything else forks behind it (ev[13] below).
    bool hist_launched = false;
    if (frozen && !given && !counted && prior_step && (models & SFQ_M_QLT)) {
        if ((rc = ensure_prior_buffers(ctx, q_rows))) return rc;
        if (!hist_cleared) { HIPC(hipMemsetAsync(ctx->hist.p, 0, (size_t)q_rows * 64 * 4, st)); hist_cleared = true; }
        launch_qlt_hist(d_fastq, nbytes, (const u64*)ctx->line_off.p, (const BlockDesc*)ctx->blocks.p, block_reads, nrec, prior_step, p.level, PRIOR_SYMBOLS, (u32*)ctx->hist.p, st);
        hist_launched = true;
    }
    u32 h_status2[5] = {0, 0, 0, 0, 0};
    HIPC(hipMemcpyAsync(h_status2, ctx->status.p, 20, hipMemcpyDeviceToHost, vst));
    HIPC(hipEventRecord(ctx->ev[1], st));
    if (vst != st) HIPC(hipStreamSynchronize(vst));
    else HIPC(hipStreamSynchronize(st));
    const u32 h_status = h_status2[0], max_hdr = h_status2[1], max_line = h_status2[2], min_hdr = ~h_status2[4];      // (k_validate_lines keeps the shortest header as a maximum)
    if (h_status == (u32)(-SFQ_E_FORMAT)) return fail(ctx, SFQ_E_FORMAT, "fastq file: expecting '@' / '+' line prefixes (usrs.cpp:162-167)");
    if (h_status) return fail(ctx, -(int)h_status, "record over the line limits: headers up to 8190 bytes; base and quality lines up to 65534 in format 6 (-B 0; the reference would write oversize side streams, usrs.cpp:269-301, which this library does not), up to 1 Gi in the block format; or an empty base line");

    // frozen tables: chain geometry; then everything that reads only the text starts now, beside the quality prior
    ChainArgs ca;
    memset(&ca, 0, sizeof ca);
    u32 nchains = 0, nsub = 0;
    GenPlan gplan; GmPlan gmplan;
    ChainGeoArgs ggeo; memset(&ggeo, 0, sizeof ggeo);   // the base chains' geometry: the quality chains', or shorter ones (below)
    const u32* gen_csz = nullptr;                      // ... and where their sizes are
    const bool flat_quads_req = getenv("SFQ_FLAT_QUADS") != nullptr;     // (a test hook: bases without a model as block format 9 wrote them, four a symbol through the coder)
    const bool gm = frozen && !exc_classic;            // bases: the match model (gm.hip); sfq_params.kernel = 2 keeps round 4's generation tables
    u32 seg_len = 0;                                   // chains that are segments of one record (chains.hip "segments")
    std::vector<u32> seg_blk;                          // ... and how many each block has ("chn.idx")
    if (frozen) {
        u32 cr = p.chain_reads ? p.chain_reads : (u32)default_chain_reads(nrec, nbytes);
        if (p.chain_reads & SFQ_CHAIN_SEGMENT_FLAG) { seg_len = p.chain_reads & ~SFQ_CHAIN_SEGMENT_FLAG; cr = 1; if (!seg_len) return fail(ctx, SFQ_E_ARG, "SFQ_CHAIN_SEGMENT(0)"); }
        else if (!p.chain_reads && cr == 1 && nrec < 204800) {
            // long records, few of them: a lane that walks a 50 kb read alone takes as long as the rest of the call -- the call's
            // symbols in about 250 000 segments (default_chain_reads), of 4096 symbols or more (round 5; 2048 before: a segment costs
            // seven to eight bytes -- its context's warm-up, its flush, its index entries -- and 6000 reads in segments of 2048 came out
            // 1.014 x the reference's at -l 4, profiles/r05_ratio_table.json)
            // (every record's last segment is a short one: about half a segment per record over the symbols' share -- the count stays within
            //  the 256 workgroups of default_chain_reads; 60 000 reads of 10-50 kb: 232.9 k segments of 7297 -> 251 k of 6673, 258 -> 263 GB/s, decode 15.4 -> 14.9 ms)
            const u64 segs_want = SFQ_CHAINS_WANT - 8192 - std::min<u64>(nrec / 2, 100000);
            const u64 want = std::min<u64>(std::max<u64>(4096, nbytes / 2 / segs_want), 1u << 20);
            if (max_line > want) seg_len = (u32)want;
        }
        if (!p.chain_reads && !seg_len)             // (the blocks' chains are of equal length: the count is blocks x chains per block, and rounds up)
            while (cr < block_reads && (u64)nblocks * ((block_reads + cr - 1) / cr) > SFQ_CHAINS_WANT) cr++;
        ca.geo.chain_reads = (u32)std::min<u64>(std::min(cr, block_reads), nrec);     // (a decoder sees min(block_reads, nrec) as the block size)
        ca.geo.cpb = (block_reads + ca.geo.chain_reads - 1) / ca.geo.chain_reads;
        // the automatic choice: the block's chains of equal length (1024 records in chains of 50 would leave a last chain of
        // 24 -- a lane that idles half of its wave's time)
        if (!p.chain_reads) ca.geo.chain_reads = std::max<u32>(1u, (u32)((std::min<u64>(block_reads, nrec) + ca.geo.cpb - 1) / ca.geo.cpb));
        const u32 last_nrec = (u32)(nrec - (u64)(nblocks - 1) * block_reads);
        const u64 nc = (u64)(nblocks - 1) * ca.geo.cpb + (last_nrec + ca.geo.chain_reads - 1) / ca.geo.chain_reads;
        if (nc > 0x7FFFFFFFull) return fail(ctx, SFQ_E_ARG, "too many chains (%llu)", (unsigned long long)nc);
        nchains = ca.geo.nchains = (u32)nc;
        ca.nbytes = nbytes; ca.block_reads = block_reads;
        if (seg_len) {
            // segments per record (from the line index), their scan, the chains' records; the blocks' shares come back for "chn.idx"
            if ((rc = reserve(ctx, ctx->segn, (size_t)nrec * 4))) return rc;
            if ((rc = reserve(ctx, ctx->segoff, ((size_t)nrec + 1) * 8))) return rc;
            if ((rc = reserve(ctx, ctx->scan_tmp, ((size_t)nrec / 1024 + 4) * 8 + 65536))) return rc;
            launch_seg_count((const u64*)ctx->line_off.p, (const BlockDesc*)ctx->blocks.p, block_reads, nrec, seg_len, (u32*)ctx->segn.p, st);
            launch_scan_u32((const u32*)ctx->segn.p, (u64*)ctx->segoff.p, nrec, (u64*)ctx->scan_tmp.p, st);
            std::vector<u64> h_off((size_t)nrec + 1);
            HIPC(hipMemcpyAsync(h_off.data(), ctx->segoff.p, ((size_t)nrec + 1) * 8, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (h_off[nrec] > 0x7FFFFFFFull) return fail(ctx, SFQ_E_ARG, "too many chains (%llu segments)", (unsigned long long)h_off[nrec]);
            nchains = ca.geo.nchains = (u32)h_off[nrec];
            seg_blk.resize(nblocks);
            for (u32 b = 0; b < nblocks; b++) seg_blk[b] = (u32)(h_off[std::min<u64>((u64)(b + 1) * block_reads, nrec)] - h_off[(u64)b * block_reads]);
            if ((rc = reserve(ctx, ctx->segrec, (size_t)nchains * 4 + 16))) return rc;
            launch_seg_fill((const u64*)ctx->segoff.p, nrec, (u32*)ctx->segrec.p, st);
            ca.seg_len = seg_len; ca.seg_off = (const u64*)ctx->segoff.p; ca.seg_rec = (const u32*)ctx->segrec.p;
        }
        // header chains: longer than the quality / base chains (each starts from the block's first header with cold field
        // types, which costs it a few bytes)
        {
            // (at least 128 records; above that as many chains as the chip holds at once -- 60 k lanes of the LDS-heavy
            //  header kernel -- since a second round of chains would double its time)
            // (a floor of 32 to 64 for small calls -- 600 k reads are 4700 lanes walking 128 headers each, the longest kernels of both
            //  directions -- was tried in round 4 and taken back: the output of a 200 k-read call grew past 1.01 x the reference's)
            const u32 cpb_want = std::max<u32>(1u, 61440u / nblocks);
            // (long reads: a header is a thousandth of its record, so a chain's few bytes of overhead do not show -- and 60 k records in chains
            //  of 128 were 470 lanes, the longest kernels of both directions: 16 records a chain there)
            const u32 rfloor = nbytes / nrec > 4000 ? 16u : 128u;
            const u32 rcr0 = std::max<u32>(rfloor, (block_reads + cpb_want - 1) / cpb_want);
            const u32 rcr = (u32)std::min<u64>(std::min<u32>(std::max<u32>(rcr0, ca.geo.chain_reads), block_reads), nrec);
            ca.rgeo.chain_reads = rcr;
            ca.rgeo.cpb = (block_reads + rcr - 1) / rcr;
            nsub = ca.rgeo.nchains = (u32)((u64)(nblocks - 1) * ca.rgeo.cpb + (last_nrec + rcr - 1) / rcr);
        }
        if ((rc = reserve(ctx, ctx->csz, ((size_t)nchains * 2 + (size_t)nsub * 2) * 4))) return rc;
        HIPC(hipMemsetAsync(ctx->csz.p, 0, ((size_t)nchains * 2 + (size_t)nsub * 2) * 4, st));
        // The base chains of a call that takes the match model (gm.hip) are SHORTER than the quality chains.  A decoder walks the generations one after the other and each takes as long as ONE lane needs for
        // its chain -- 6.6 ms at 49 records whatever the generation's size, seven times over; at 12 records the same call decodes in
        // 30 ms instead of 60 and is 0.13 % larger (bench.py --kind 3 --chain-reads 12 / 49, profiles/r05g).  Known only with the
        // verdict: their sizes go to a list of their own.
        ggeo = ca.geo;
        if (gm && !seg_len) {
            // (about 819 200 of them a call -- four times the quality chains: the last generation, half the call, then has six waves a
            //  SIMD and runs at the chip's rate, not at one lane's -- down to 2 KiB of text a chain: a 2 M-read call decodes in 10 ms
            //  instead of 17 for 0.4 % more bytes)
            const u64 per_rec = std::max<u64>(1, nbytes / std::max<u64>(1, nrec));
            const u64 gwant = std::max<u64>((GM_CHAIN_FLOOR + per_rec - 1) / per_rec, (nrec + GM_CHAIN_WANT - 1) / GM_CHAIN_WANT);
            const u32 gfloor = (u32)std::min<u64>(gwant, ca.geo.chain_reads);
            ggeo.cpb = (block_reads + gfloor - 1) / gfloor;
            ggeo.chain_reads = std::max<u32>(1u, (u32)((std::min<u64>(block_reads, nrec) + ggeo.cpb - 1) / ggeo.cpb));
            ggeo.cpb = (block_reads + ggeo.chain_reads - 1) / ggeo.chain_reads;
            const u64 ngc = (u64)(nblocks - 1) * ggeo.cpb + (last_nrec + ggeo.chain_reads - 1) / ggeo.chain_reads;
            if (ngc > 0x7FFFFFFFull || ggeo.chain_reads >= ca.geo.chain_reads) ggeo = ca.geo;
            else {
                ggeo.nchains = (u32)ngc;
                if ((rc = reserve(ctx, ctx->gm_csz, (size_t)ggeo.nchains * 4))) return rc;
                HIPC(hipMemsetAsync(ctx->gm_csz.p, 0, (size_t)ggeo.nchains * 4, st));
            }
        }
        if ((rc = reserve_pinned(ctx, PIN_BYTES + (size_t)q_rows * 66 * 4))) return rc;
        if ((rc = setup_tables())) return rc;
        // the pass over the N / quality-0 / case exceptions looks only at the records the framing has marked (frame.hip
        // k_write_newlines: the text is in registers there anyway) and runs beside the counting passes, while the chip is
        // mostly idle.  (Round 2 had the quality and base chains mark them -- the pass then ran BEHIND the chains, 2.5 ms at
        // the call's tail; a marking pass of its own over the text was measured too: it costs more than those 2.5 ms.)
        HIPC(hipEventRecord(ctx->ev[13], st));
        if (!priors_only) {
            for (int m = 1; m < 4; m++) { HIPC(hipStreamWaitEvent(mst[m], ctx->ev[13], 0)); HIPC(hipEventRecord(ctx->ev[2 + 2 * m], mst[m])); }
            a.batch0 = 0; a.nbatch = std::min(slots, nblocks_r);
            ca.m = a;
            if (models & SFQ_M_REC) { if ((rc = rec_prior_begin(ctx, a, nrec, given, counted, mst[1], max_hdr))) return rc; }
            if (models & SFQ_M_GEN) {
                if (gm) { if ((rc = gm_begin(ctx, ca, nblocks, nrec, max_line, mst[3], gmplan))) return rc; }
                else if ((rc = gen_tables_begin(ctx, ca, nblocks, (u32)g_bits, max_line, mst[3], gplan))) return rc;
            }
            // (the pass over the exceptions and the framing exceptions are queued behind the chains' launches below: beside the two
            //  counting passes the host waits for they made those take 1.7-1.9 ms instead of 0.3)
        } else if (models & SFQ_M_REC) {                    // sfq_build_priors: the header sample beside the quality sample
            HIPC(hipStreamWaitEvent(mst[1], ctx->ev[13], 0));
            if ((rc = rec_prior_begin(ctx, a, nrec, given, counted, mst[1], max_hdr))) return rc;
        }
    }
    u32* h_rows66 = nullptr;
    if (given && (models & SFQ_M_QLT)) {
        if ((rc = ensure_prior_buffers(ctx, q_rows))) return rc;
        if (!hist_cleared) HIPC(hipMemsetAsync(ctx->hist.p, 0, (size_t)q_rows * 64 * 4, st));          // (no sample of its own: LDS staging has nothing to rank by)
        if ((rc = upload_prior(ctx, q_rows, st, !frozen))) return rc;
        HIPC(hipEventRecord(ctx->ev[1], st));
        ctx->prior_on = true;
    } else if (prior_step && (models & SFQ_M_QLT)) {
        if ((rc = ensure_prior_buffers(ctx, q_rows))) return rc;
        if (!counted && !hist_launched) {
            if (!hist_cleared) HIPC(hipMemsetAsync(ctx->hist.p, 0, (size_t)q_rows * 64 * 4, st));
            launch_qlt_hist(d_fastq, nbytes, (const u64*)ctx->line_off.p, (const BlockDesc*)ctx->blocks.p, block_reads, nrec, prior_step, p.level, PRIOR_SYMBOLS, (u32*)ctx->hist.p, st);
        }
        if (ctx->counts_only) {                              // sfq_count_priors: the sample is counted (the header sample beside it)
            HIPC(hipStreamSynchronize(mst[1]));
            HIPC(hipStreamSynchronize(st));
            res->n_records = nrec; res->n_blocks = nblocks;
            ctx->counts.valid = true; ctx->counts.q_rows = q_rows; ctx->counts.rec = frozen && (models & SFQ_M_REC);
            if (!legacy && !n_over) {
                u64 print = 0;
                if ((rc = text_fingerprint(ctx, d_fastq, nbytes, st, &print))) return rc;
                ctx->framed.ptr = d_fastq; ctx->framed.nbytes = nbytes; ctx->framed.nrec = nrec; ctx->framed.marks = want_marks; ctx->framed.print = print; ctx->framed.valid = true;
            }
            return SFQ_OK;
        }

        launch_prior_rows((const u32*)ctx->hist.p, q_rows, (u32*)ctx->rows66.p, (u32*)ctx->prior_w.p, (u32*)ctx->prior_wovf.p,
                          (u32*)ctx->prior_ls.p, (RowHdr*)ctx->prior_lh.p, st);
        // "qlt.pri" is packed on the host from the rows the prior lists: gathered back to back on the device (a few thousand of
        // the 65 536 -- round 3 copied the dense 17 MB table, which held a hardware queue for half a millisecond and had to wait
        // behind the header chains), the first PRIOR_LIST_EAGER of them copied before their number is known
        if ((rc = reserve(ctx, ctx->pslot, (size_t)q_rows * 4))) return rc;
        if ((rc = reserve(ctx, ctx->plist, ((size_t)PRIOR_LIST_HEAD + (size_t)q_rows * PRIOR_LIST_ROW) * 4))) return rc;
        if ((rc = reserve_pinned(ctx, PIN_BYTES + ((size_t)PRIOR_LIST_HEAD + (size_t)q_rows * PRIOR_LIST_ROW) * 4))) return rc;
        h_rows66 = (u32*)((u8*)ctx->pin + PIN_BYTES);
        // (on a stream of its own: nothing on the device waits for the list)
        HIPC(hipEventRecord(ctx->ev[21], st));
        HIPC(hipStreamWaitEvent(mst[2], ctx->ev[21], 0));
        launch_prior_list((const u32*)ctx->rows66.p, q_rows, (u32*)ctx->pslot.p, (u32*)ctx->plist.p, mst[2]);
        HIPC(hipMemcpyAsync(h_rows66, ctx->plist.p, ((size_t)PRIOR_LIST_HEAD + (size_t)std::min<u32>(q_rows, PRIOR_LIST_EAGER) * PRIOR_LIST_ROW) * 4, hipMemcpyDeviceToHost, mst[2]));
        HIPC(hipEventRecord(ctx->ev[20], mst[2]));
        HIPC(hipEventRecord(ctx->ev[1], st));      // the model streams fork after the prior is built
        ctx->prior_on = true;
    }
    // frozen tables: dense quality rows
    if (frozen) {
        if (models & SFQ_M_QLT) {
            if ((rc = reserve(ctx, ctx->qrows, (size_t)q_rows * 64 * 4))) return rc;
            if ((rc = build_qesc(ctx, st))) return rc;
            launch_qlt_frozen_rows((const u32*)ctx->rows66.p, q_rows, (u32*)ctx->qrows.p, nullptr, st);
            ca.qrows = (const u32*)ctx->qrows.p; ca.qesc = (const u32*)ctx->qesc.p;
            ca.q_hot = 0; ca.q_rows = q_rows;
            // LDS staging of the rows the sample saw most (picked on the device, chains.hip k_hot_select); without a sample of
            // this call's own there is nothing to rank by
            // (0 = automatic: 800 rows.  Beside round 2's header kernel, which took 147 KiB of every CU's LDS, the image bought nothing in
            //  the full call; with the headers' token step it takes the call from 17.3 to 15.9 ms -- the chains it serves from LDS are row
            //  gathers the CU's vector memory path does not carry for the kernels beside the quality chains)
            //  (the image is shared by a workgroup of 1024 chains, one per CU: automatic only where the call has chains for every CU several times
            //   over -- 60 k long reads are 59 such workgroups, and coded at half the speed)
            const u32 want_hot = (p.lds_rows == SFQ_LDS_ROWS_NONE || given || !prior_step) ? 0u : p.lds_rows ? std::min<u32>(p.lds_rows, 1024u)
                                 : nchains >= 150000u ? 800u : 0u;
            if (want_hot) {
                const size_t img_bytes = (size_t)q_rows / 4 + (size_t)want_hot * 100 + 64;
                if ((rc = reserve(ctx, ctx->qw, img_bytes + (size_t)q_rows * 4 + 256))) return rc;
                u8* img = (u8*)ctx->qw.p; u32* info = (u32*)(img + ((img_bytes + 15) & ~(size_t)15)); u32* ctot = info + 16;
                launch_hot_rows((const u32*)ctx->hist.p, (const u32*)ctx->rows66.p, (const u32*)ctx->qrows.p, q_rows, want_hot, ctot, img, info, st);
                ca.q_hot = want_hot; ca.qh_img = img; ca.qh_info = info;
            }
            HIPC(hipEventRecord(ctx->ev[1], st));
        }
    }
    if (!frozen) { if ((rc = setup_tables())) return rc; }
    if (priors_only) {
        if (ctx->prior_on && !given) {                      // (packed while the header sample is still being counted)
            if ((rc = qlt_prior_blob_now(ctx, h_rows66, q_rows, st))) return rc;
        }
        if (frozen && (models & SFQ_M_REC)) { if ((rc = rec_prior_finish(ctx, given, mst[1]))) return rc; if ((rc = rec_prior_copy_back(ctx, mst[1]))) return rc; if ((rc = rec_prior_blob_now(ctx))) return rc; }
        HIPC(hipStreamSynchronize(mst[1]));
        HIPC(hipStreamSynchronize(st));
        ctx->prior_on = false;
        res->n_records = nrec; res->n_blocks = nblocks;
        return SFQ_OK;
    }
    u32 gen_on = 0;
    const bool exc_rice = frozen && !exc_classic;
    bool side_late = false;                            // the exception pass is still running when the packing starts
    if (frozen) {
        // The two host decisions first (the shorter counting pass, the headers', before the base tables' verdict), each followed by
        // its chains; the quality chains LAST: their row gathers keep every CU's vector memory path full, and whatever small pass
        // runs beside them crawls -- with the quality chains ahead of them the base tables' 0.1 ms row kernel took 3.5 ms and the
        // base chains started 8 ms into the call.
        a.batch0 = 0; a.nbatch = std::min(slots, nblocks_r);
        HIPC(hipEventRecord(ctx->ev[2], st));
        if (models & SFQ_M_REC) {
            if ((rc = rec_prior_finish(ctx, given, mst[1]))) return rc;
            ca.m = a; ca.rrows = (const u32*)ctx->rrows.p; ca.rdec = (const u16*)ctx->rdec.p;
            ca.rmap = (const u16*)ctx->rmap.p; ca.rhot = ca.rmap + PR_REC_ROWS; ca.r_hot = ctx->r_hot;
            if ((rc = reserve(ctx, ctx->rflags, (size_t)nsub * 4 * 3))) return rc;                     // flags, flags2, token counts
            if ((rc = reserve(ctx, ctx->rtok, rec_token_bytes(nrec)))) return rc;
            HIPC(hipMemsetAsync(ctx->rflags.p, 0, (size_t)nsub * 4 * 2, mst[1]));
            ca.csz = (u32*)ctx->csz.p + 2 * (size_t)nchains; ca.rhb = ca.csz + nsub;
            HIPC(hipEventRecord(ctx->ev[18], mst[1]));
            if (nbytes / nrec <= 4000) HIPC(hipStreamWaitEvent(st, ctx->ev[18], 0));   // (the quality chains behind the header prior's passes, as when the host waited for those; not where records are long: few headers, long ones, and the chains have better things to do than wait for their sample)
            launch_rec_encode_c(ca, (u32*)ctx->rflags.p, (u32*)ctx->rflags.p + nsub, (u32*)ctx->rtok.p, (u32*)ctx->rflags.p + 2 * (size_t)nsub, ctx->r_hot_dec, max_hdr, mst[1], min_hdr, min_hdr > 127 ? nullptr : ctx->ev[25]);
            HIPC(hipEventRecord(ctx->ev[min_hdr > 127 ? 25 : 19], mst[1]));
            HIPC(hipEventRecord(ctx->ev[3 + 2 * 1], mst[1]));          // (the header chains are through here; the copy below is not part of the model's phase)
            if ((rc = rec_prior_copy_back(ctx, mst[1]))) return rc;
        } else HIPC(hipEventRecord(ctx->ev[3 + 2 * 1], mst[1]));
        if (models & SFQ_M_GEN) {
            ca.m = a; ca.csz = (u32*)ctx->csz.p + nchains;
            if (gm) {
                gmplan.ggeo = ggeo; gmplan.gcsz = ggeo.nchains != ca.geo.nchains ? (u32*)ctx->gm_csz.p : nullptr;
                if ((rc = gm_finish(ctx, ca, nrec, max_line, mst[3], gmplan, &gen_on))) return rc;
                if (!gen_on) ggeo = ca.geo;                  // (no match model: the base chains are the quality chains' records)
            }
            else if ((rc = gen_tables_finish(ctx, ca, (u32)g_bits, max_line, mst[3], gplan, &gen_on))) return rc;
            gen_csz = (gm && gen_on && ggeo.nchains != ca.geo.nchains) ? (const u32*)ctx->gm_csz.p : (const u32*)ctx->csz.p + nchains;
            // (Round 4 measured a generation's chains started as soon as its rows were there, on a stream of their own beside the counting
            //  passes of the generations behind it: 57.4 ms per 10 M genome-sampled reads against 58.8 -- the passes' atomics and the chains'
            //  row gathers wait for the same thing, random 64-byte sectors of tables larger than the caches, and their times add up.)
            HIPC(hipEventRecord(ctx->ev[16], mst[3]));
            if (gm && gen_on) launch_gm_code(gmplan.cg, (const u8*)ctx->gm_tok.p, mst[3]);
            else { ca.flat_quads = (gm && !gen_on && flat_quads_req) ? 1u : 0u; ca.flat_raw = (gm && !gen_on && !flat_quads_req) ? 1u : 0u; launch_gen_encode_c(ca, mst[3], 0, 0, !gen_on); }
            HIPC(hipEventRecord(ctx->ev[17], mst[3]));
        }
        HIPC(hipEventRecord(ctx->ev[3 + 2 * 3], mst[3]));
        if (models & SFQ_M_QLT) {
            ca.m = a; ca.csz = (u32*)ctx->csz.p;
            // (round 5, measured and dropped: with the match model on, the quality chains held back until the bases' plan is through -- beside
            //  them the stage, the index and the plan take 3.7 + 6.0 + 5.8 ms against 1 + 3 + 4 alone.  The bases' phase 20.8 -> 17.0 ms, the
            //  quality chains end at 21.9 instead of 11.8, the call 24.3 -> 25.9 ms: the chip's work is conserved, whoever goes first.
            //  The other way round -- the quality chains queued FIRST, ahead of the two host decisions, in the default call: they take 6.4-8 ms
            //  instead of 9.1, the base and header chains 8 instead of 4.8, the call 12.4-13.8 ms against 12.6-12.8.)
            HIPC(hipEventRecord(ctx->ev[14], st)); launch_qlt_encode_c(ca, st); HIPC(hipEventRecord(ctx->ev[15], st));
        }
        HIPC(hipEventRecord(ctx->ev[3], st));
        // the base exceptions: Rice-coded gap lists (dev_rice.h; the pass itself is models_w.hip k_gen_exc_w); sfq_params.kernel = 2 keeps the reference's own
        // coding of them (adaptive PowerRanger rows, a wave per block: what rounds 2 and 3 wrote)
        if (models & SFQ_M_GEN) {
            if (exc_rice) launch_gen_exc_r(a, want_marks ? (const u8*)ctx->excf.p : nullptr, tickets + 1, mst[2]);
            else launch_gen_exc_w(a, want_marks ? (const u8*)ctx->excf.p : nullptr, tickets + 1, mst[2]);
        }
        if (models & SFQ_M_USR)
            for (u32 b0 = 0; b0 < nblocks; b0 += slots) { ModelArgs ua = a; ua.batch0 = b0; ua.nbatch = std::min(slots, nblocks - b0); launch_usr_encode_w(ua, mst[2]); }
        HIPC(hipEventRecord(ctx->ev[3 + 2 * 2], mst[2]));
        side_late = false;
        HIPC(hipStreamWaitEvent(st, ctx->ev[3 + 2 * 1], 0));        // header chains
        HIPC(hipStreamWaitEvent(st, ctx->ev[3 + 2 * 3], 0));        // base chains
        if (!side_late) HIPC(hipStreamWaitEvent(st, ctx->ev[3 + 2 * 2], 0));
    } else {
    // The lane-per-block reference kernels (kernel = 1, and usr) run in batches of `slots` blocks.
    if ((models & SFQ_M_GEN) && p.kernel == 1)   // first batch's Base2 tables (base2_ranger.hpp:68-71), while the chip is idle
        launch_fill_u32((u32*)ctx->tab.g_tab.p, (u64)std::min(slots, nblocks) << g_bits, 0x03030303u, st);
    HIPC(hipEventRecord(ctx->ev[1], st));
    for (int m = 0; m < 4; m++) {
        if (m) HIPC(hipStreamWaitEvent(mst[m], ctx->ev[1], 0));
        HIPC(hipEventRecord(ctx->ev[2 + 2 * m], mst[m]));
        if (models & order[m]) {
            const bool batched = p.kernel == 1 || order[m] == SFQ_M_USR;
            if (!batched) {
                a.batch0 = 0; a.nbatch = std::min(slots, nblocks_r);
                // Every kernel is persistent and would take all the chip's wave slots if launched alone, so the three
                // would run one after the other.  When all three models run, the two-block quality and base kernels
                // keep to a third of the wave slots each (2 table slots per wave) and the header kernel's workgroups
                // fill whatever is free: the three overlap from the start (measured: 225 -> 207 ms at 10 M reads).
                if ((models & (SFQ_M_QLT | SFQ_M_GEN | SFQ_M_REC)) == (SFQ_M_QLT | SFQ_M_GEN | SFQ_M_REC) &&
                    (order[m] == SFQ_M_QLT || order[m] == SFQ_M_GEN))
                {
#ifndef ADAPT_GEN_PCT
#define ADAPT_GEN_PCT 33
#define ADAPT_QLT_PCT 33
#endif
                    const u32 pct = order[m] == SFQ_M_GEN ? ADAPT_GEN_PCT : ADAPT_QLT_PCT;
                    a.nbatch = std::min<u32>(a.nbatch, std::max<u32>(KR, ((u32)((u64)ctx->wave_slots * pct / 100) * KR) & ~(KR - 1)));
                }
                switch (order[m]) {
                case SFQ_M_QLT: launch_qlt_encode_k(a, tickets + 0, mst[m]); break;
                case SFQ_M_GEN: launch_gen_encode_k(a, tickets + 1, mst[m]); break;
                case SFQ_M_REC: launch_rec_encode_w(a, tickets + 2, tickets + 3, mst[m]); break;
                }
            } else {
                for (u32 b0 = 0; b0 < nblocks; b0 += slots) {
                    a.batch0 = b0; a.nbatch = std::min(slots, nblocks - b0);
                    switch (order[m]) {
                    case SFQ_M_QLT: launch_qlt_encode_l(a, mst[m]); break;
                    case SFQ_M_GEN:
                        if (b0) launch_fill_u32((u32*)ctx->tab.g_tab.p, (u64)a.nbatch << g_bits, 0x03030303u, mst[m]);
                        launch_gen_encode_l(a, mst[m]);
                        break;
                    case SFQ_M_REC: launch_rec_encode_l(a, mst[m]); break;
                    case SFQ_M_USR: if (p.kernel == 1) launch_usr_encode_l(a, mst[m]); else launch_usr_encode_w(a, mst[m]); break;
                    }
                }
            }
        }
        if (order[m] == SFQ_M_USR && n_over) {                     // the oversize records' raw lines: three waves beside everything else
            const u64 ooff[3] = { arena_main, arena_main + over_cap, arena_main + 2 * over_cap };
            const u32 ocap[3] = { (u32)over_cap, (u32)over_cap, (u32)over_cap };
            launch_over_encode_w(a, d_file, (const u64*)ctx->line_off_o.p, (const u32*)ctx->olist.p, n_over, ooff, ocap, mst[m]);
        }
        HIPC(hipEventRecord(ctx->ev[3 + 2 * m], mst[m]));
        if (m) HIPC(hipStreamWaitEvent(st, ctx->ev[3 + 2 * m], 0));
    }
    }
    ht.mark("chains queued");
    HIPC(hipEventRecord(ctx->ev[10], st));

    // ---- pack ----------------------------------------------------------------------------------
    if ((rc = reserve(ctx, ctx->blk_stream_off, (size_t)nblocks * SFQ_NSTREAMS * 8))) return rc;
    if ((rc = reserve(ctx, ctx->stream_total, 2 * SFQ_NSTREAMS * 8 + 64))) return rc;                  // totals, bases, the packing's gate
    u32 chain_streams = 0;                             // streams packed chain by chain
    if (frozen) {
        ca.m = a;
        if (models & SFQ_M_QLT) { launch_chain_block_sizes(ca, ca.geo, SFQ_S_QLT, (const u32*)ctx->csz.p, nullptr, st); chain_streams |= 1u << SFQ_S_QLT; }
        if (models & SFQ_M_GEN) { launch_chain_block_sizes(ca, ggeo, SFQ_S_GEN, gen_csz, nullptr, st); chain_streams |= 1u << SFQ_S_GEN; }
        if (models & SFQ_M_REC) {
            const u32* rs = (const u32*)ctx->csz.p + 2 * (size_t)nchains;
            launch_chain_block_sizes(ca, ca.rgeo, SFQ_S_REC, rs, rs + nsub, st); chain_streams |= 1u << SFQ_S_REC;
        }
    }
    // The three chain-coded streams come first in the output (rec, gen, qlt: enum sfq_stream), so their offsets need
    // nothing of the side streams: with the exception pass still running (side_late) they are packed beside it, and
    // the side streams -- a few MB -- when it is through.
    const bool two_halves = side_late && chain_streams == 7u;
    launch_block_stream_offsets((BlockDesc*)ctx->blocks.p, nblocks, (u64*)ctx->blk_stream_off.p, (u64*)ctx->stream_total.p, 0, two_halves ? 3 : SFQ_NSTREAMS, st);
    // first headers -> blob
    if ((rc = reserve(ctx, ctx->lens, (size_t)nblocks * 4))) return rc;
    if ((rc = reserve(ctx, ctx->blob_off, ((size_t)nblocks + 1) * 8))) return rc;
    const u64 blob_cap = std::min<u64>(nbytes, (u64)nblocks * SFQ_MAX_ID_LLEN);
    if ((rc = reserve(ctx, ctx->blob, blob_cap))) return rc;
    launch_first_hdr_lens((const BlockDesc*)ctx->blocks.p, nblocks, (u32*)ctx->lens.p, st);
    launch_scan_u32((const u32*)ctx->lens.p, (u64*)ctx->blob_off.p, nblocks, (u64*)ctx->scan_tmp.p, st);
    launch_gather_first_hdrs((const BlockDesc*)ctx->blocks.p, nblocks, d_fastq, (const u64*)ctx->blob_off.p, (u8*)ctx->blob.p, blob_cap, st);
    // what comes back to the host at the end lands in page-locked memory: [totals][block descriptors][blob offsets][chain sizes]
    const size_t p2_hb = 128, p2_off = p2_hb + (((size_t)nblocks * sizeof(BlockDesc) + 63) & ~(size_t)63);
    const size_t p2_csz = p2_off + ((((size_t)nblocks + 1) * 8 + 63) & ~(size_t)63);
    const u32 ngc = frozen ? ggeo.nchains : 0;             // base chains (the quality chains' number unless the match model cut its own)
    const bool gsplit = frozen && ngc != nchains;
    const size_t n_sizes = (size_t)nchains + ngc + (size_t)nsub * 2;
    const u32* d_sizes = (const u32*)ctx->csz.p;            // the size lists back to back: quality, bases, headers, header bytes
    if (gsplit) {
        if ((rc = reserve(ctx, ctx->gm_idx, n_sizes * 4 + 64))) return rc;
        u32* di = (u32*)ctx->gm_idx.p;
        HIPC(hipMemcpyAsync(di, ctx->csz.p, (size_t)nchains * 4, hipMemcpyDeviceToDevice, st));
        HIPC(hipMemcpyAsync(di + nchains, ctx->gm_csz.p, (size_t)ngc * 4, hipMemcpyDeviceToDevice, st));
        HIPC(hipMemcpyAsync(di + nchains + ngc, (const u32*)ctx->csz.p + 2 * (size_t)nchains, (size_t)nsub * 2 * 4, hipMemcpyDeviceToDevice, st));
        d_sizes = di;
    }
    const size_t p2_end = p2_csz + n_sizes * 4 + 64;
    if ((rc = reserve_pinned_buf(ctx, ctx->pin2, ctx->pin2_cap, p2_end))) return rc;
    u64* totals = (u64*)ctx->pin2;
    HIPC(hipMemcpyAsync(totals, ctx->stream_total.p, SFQ_NSTREAMS * 8, hipMemcpyDeviceToHost, st));
    // host work that needs nothing of what is still running -- "qlt.pri", 0.7 ms -- is done while the chains are coded, and behind
    // the LAUNCHES of everything that follows them: a small call's chains are through before the host is, and the packing must
    // not wait for it (round 4: 0.73 ms of idle GPU in a 4.1 ms call of 600 k reads)
    auto pack_prior_now = [&]() -> int {
        if (ctx->prior_on && !given && h_rows66) {
            if ((rc = qlt_prior_blob_now(ctx, h_rows66, q_rows, mst[2]))) return rc;
            h_rows66 = nullptr;
        }
        return rec_prior_blob_now(ctx);
    };
    BlockDesc* hb = (BlockDesc*)((u8*)ctx->pin2 + p2_hb);
    u64* hboff = (u64*)((u8*)ctx->pin2 + p2_off);
    u32* h_csz = (u32*)((u8*)ctx->pin2 + p2_csz);
    bool chn_on_device = false; u32 chn_n = 0; size_t chn_eager = 0;       // "chn.idx": the size lists arrive as bytes ([info 16 B][bytes] where h_csz points)
    u64 bases[SFQ_NSTREAMS], run = 0;
    if (two_halves) {
        if ((rc = pack_prior_now())) return rc;
        HIPC(hipStreamSynchronize(st));                                 // the chains are through; totals[0..2] are here
        for (int s = 0; s < 3; s++) { bases[s] = run; run += totals[s]; }
        if (run > out_cap) return fail(ctx, SFQ_E_OVERFLOW, "output needs more than %llu bytes, caller gave %llu", (unsigned long long)run, (unsigned long long)out_cap);
        HIPC(hipMemcpyAsync((u64*)ctx->stream_total.p + SFQ_NSTREAMS, bases, 3 * 8, hipMemcpyHostToDevice, st));
        launch_compact_chains(ca, ca.geo, SFQ_S_QLT, 2, 1, (const u32*)ctx->csz.p, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st);
        launch_compact_chains(ca, ggeo, SFQ_S_GEN, 3, 4, gen_csz, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st);
        launch_compact_chains(ca, ca.rgeo, SFQ_S_REC, 3, 2, (const u32*)ctx->csz.p + 2 * (size_t)nchains, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st);
        HIPC(hipMemcpyAsync(h_csz, d_sizes, n_sizes * 4, hipMemcpyDeviceToHost, st));
        // the side streams: behind the exception pass and the framing exceptions
        HIPC(hipStreamWaitEvent(st, ctx->ev[12], 0));
        HIPC(hipStreamWaitEvent(st, ctx->ev[3 + 2 * 2], 0));
        launch_block_stream_offsets((BlockDesc*)ctx->blocks.p, nblocks, (u64*)ctx->blk_stream_off.p, (u64*)ctx->stream_total.p, 3, SFQ_NSTREAMS, st);
        HIPC(hipMemcpyAsync(totals + 3, (u64*)ctx->stream_total.p + 3, (SFQ_NSTREAMS - 3) * 8, hipMemcpyDeviceToHost, st));
    }
    // The packing follows the sizes on the device: k_stream_gate places the streams and holds the packing back where a block has
    // failed or the caller's buffer is too small (the host reports that below, from the same numbers) -- no host round trip between
    // the chains and the packing.  Everything the host wants comes back behind it in one go; the first headers too, as far as a
    // guess at their size reaches.
    u32* d_gate = (u32*)((u64*)ctx->stream_total.p + 2 * SFQ_NSTREAMS);
    const u64 blob_guess = std::min<u64>(blob_cap, (u64)nblocks * 256);
    if (!two_halves) {
        launch_stream_gate((const BlockDesc*)ctx->blocks.p, nblocks, (const u64*)ctx->stream_total.p, out_cap, (u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_gate, st);
        launch_compact((const BlockDesc*)ctx->blocks.p, nblocks, (const u8*)ctx->arena.p, (const u64*)ctx->blk_stream_off.p,
                       (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, chain_streams, st, d_gate);
        if (frozen) {
            if (chain_streams & (1u << SFQ_S_QLT))
                launch_compact_chains(ca, ca.geo, SFQ_S_QLT, 2, 1, (const u32*)ctx->csz.p, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st, d_gate);
            if (chain_streams & (1u << SFQ_S_GEN))
                launch_compact_chains(ca, ggeo, SFQ_S_GEN, 3, 4, gen_csz, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st, d_gate);
            if (chain_streams & (1u << SFQ_S_REC))
                launch_compact_chains(ca, ca.rgeo, SFQ_S_REC, 3, 2, (const u32*)ctx->csz.p + 2 * (size_t)nchains, (const u64*)ctx->blk_stream_off.p, (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, st, d_gate);
            // "chn.idx": the size lists come back as the bytes the blob holds (chains.hip launch_chain_index_bytes: made from half a
            // million sizes on the host they were 1.0 ms of every call's tail), as far as a guess at their number reaches
            {
                const bool recl = (chain_streams >> SFQ_S_REC) & 1;
                chn_n = nchains + ngc + (recl ? nsub * 2 : 0u);
                const u32 b1 = nchains, b2 = nchains + ngc, b3 = recl ? nchains + ngc + nsub : chn_n;
                if ((rc = reserve(ctx, ctx->chn_len, (size_t)chn_n * 4 + 16))) return rc;
                if ((rc = reserve(ctx, ctx->chn_off, ((size_t)chn_n + 4) * 8))) return rc;
                if ((rc = reserve(ctx, ctx->chn_out, (size_t)chn_n * 5 + 64))) return rc;
                if ((rc = reserve(ctx, ctx->scan_tmp, ((size_t)chn_n / 1024 + 4) * 8 + 65536))) return rc;
                u64* d_info = (u64*)ctx->chn_off.p + chn_n + 1;
                launch_chain_index_bytes(d_sizes, chn_n, b1, b2 > chn_n ? chn_n : b2, b3, (u32*)ctx->chn_len.p, (u64*)ctx->chn_off.p, (u64*)ctx->scan_tmp.p,
                                         (u8*)ctx->chn_out.p, d_info, st);
                chn_eager = std::min<size_t>((size_t)chn_n * 5, n_sizes * 4 + 64 - 16);
                HIPC(hipMemcpyAsync(h_csz, d_info, 16, hipMemcpyDeviceToHost, st));
                if (chn_eager) HIPC(hipMemcpyAsync((u8*)h_csz + 16, ctx->chn_out.p, chn_eager, hipMemcpyDeviceToHost, st));
                chn_on_device = true;
            }
        }
        HIPC(hipEventRecord(ctx->ev[11], st));
        if ((rc = pack_prior_now())) return rc;               // (before the copy into pageable memory below: that one returns when the stream has reached it)
        ctx->first_hdrs.resize((size_t)blob_guess);
        if (blob_guess) HIPC(hipMemcpyAsync(ctx->first_hdrs.data(), ctx->blob.p, (size_t)blob_guess, hipMemcpyDeviceToHost, st));
    }
    HIPC(hipMemcpyAsync(hb, ctx->blocks.p, (size_t)nblocks * sizeof(BlockDesc), hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(hboff, ctx->blob_off.p, ((size_t)nblocks + 1) * 8, hipMemcpyDeviceToHost, st));
    ht.mark("packing queued");
    if ((rc = pack_prior_now())) return rc;
    ht.mark("priors packed");
    HIPC(hipStreamSynchronize(st));
    ht.mark("device through");
    run = 0;
    for (int s = 0; s < SFQ_NSTREAMS; s++) { bases[s] = run; run += totals[s]; res->stream_bytes[s] = totals[s]; res->stream_offset[s] = bases[s]; }
    res->total_bytes = run;
    // per-block status first: an overflowed block has a meaningless size
    int worst = 0;
    for (u32 b = 0; b < nblocks; b++) if (hb[b].status) worst = std::max<int>(worst, (int)hb[b].status);
    if (worst) return fail(ctx, -worst, "block kernel reported error %d (%s)", -worst,
                           -worst == SFQ_E_OVERFLOW ? "stream arena too small" : -worst == SFQ_E_GENCHAR ? "unexpected genome char / switched N byte" :
                           -worst == SFQ_E_UNSUPPORTED ? "a '+' line that is neither empty nor its record's header: the block format refuses what it could not give back (usrs.cpp:236-239)" : "see status codes");
    ht.mark("  statuses looked at");
    if (run > out_cap) return fail(ctx, SFQ_E_OVERFLOW, "output needs %llu bytes, caller gave %llu", (unsigned long long)run, (unsigned long long)out_cap);
    if (hboff[nblocks] > blob_cap) return fail(ctx, SFQ_E_OVERFLOW, "first-header blob overflow");
    if (two_halves) {
        HIPC(hipMemcpyAsync((u64*)ctx->stream_total.p + SFQ_NSTREAMS, bases, sizeof bases, hipMemcpyHostToDevice, st));
        launch_compact((const BlockDesc*)ctx->blocks.p, nblocks, (const u8*)ctx->arena.p, (const u64*)ctx->blk_stream_off.p,
                       (const u64*)ctx->stream_total.p + SFQ_NSTREAMS, d_out, chain_streams, st);
        HIPC(hipEventRecord(ctx->ev[11], st));
        ctx->first_hdrs.resize((size_t)hboff[nblocks]);
        if (hboff[nblocks]) HIPC(hipMemcpyAsync(ctx->first_hdrs.data(), ctx->blob.p, (size_t)hboff[nblocks], hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    } else {
        if (hboff[nblocks] > blob_guess) {                               // (long first headers: the rest of them)
            ctx->first_hdrs.resize((size_t)hboff[nblocks]);
            HIPC(hipMemcpyAsync(ctx->first_hdrs.data() + blob_guess, (const u8*)ctx->blob.p + blob_guess, (size_t)(hboff[nblocks] - blob_guess), hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
        } else ctx->first_hdrs.resize((size_t)hboff[nblocks]);
    }

    ht.mark("  first headers home");
    ctx->prior_on = false;
    ctx->chain_blob.clear();
    if (frozen) {            // "chn.idx": chain_reads, flags (bit 0: generation tables of the bases in use), nchains, sizes
        // (half a million varints, written through a pointer into room for the longest: pushed byte by byte into the vector they
        //  were 1.0 ms of every call, after the GPU had finished)
        // (written into a scratch vector that keeps its size from call to call -- growing the blob itself zero-fills 2.5 MB every call --
        //  and copied out at its final size)
        std::vector<u8>& o = ctx->chain_tmp;
        const size_t o_need = ((size_t)nchains + ngc + (size_t)nsub * 2 + (seg_len ? nblocks : 0)) * 5 + 64;
        if (o.size() < o_need) o.resize(o_need);
        u8* w = o.data();
        auto put = [&w](u32 v) { while (v >= 0x80) { *w++ = (u8)(v | 0x80); v >>= 7; } *w++ = (u8)v; };
        // a list of sizes: each as the zigzag difference to the one before it (neighbouring chains hold as many symbols of the same
        // statistics: a byte a chain instead of two -- at 12 records a chain the index was 0.5 % of the archive)
        // (half a million sizes: eight at a time where all eight differences take one byte -- nearly always; 0.86 -> 0.3 ms of the call's tail)
        auto put_list = [&put, &w](const u32* v, size_t n) {
            u32 prev = 0; size_t i = 0;
            for (; i + 8 <= n; i += 8) {
                u32 z[8], any = 0;
                for (int j = 0; j < 8; j++) { const i32 d = (i32)(v[i + j] - (j ? v[i + j - 1] : prev)); z[j] = ((u32)d << 1) ^ (u32)(d >> 31); any |= z[j]; }
                if (any < 0x80u) { u64 b8 = 0; for (int j = 0; j < 8; j++) b8 |= (u64)z[j] << (8 * j); memcpy(w, &b8, 8); w += 8; }
                else for (int j = 0; j < 8; j++) put(z[j]);
                prev = v[i + 7];
            }
            for (; i < n; i++) { const i32 d = (i32)(v[i] - prev); put(((u32)d << 1) ^ (u32)(d >> 31)); prev = v[i]; }
        };
        const bool rec_chains = (chain_streams >> SFQ_S_REC) & 1;
        put(ca.geo.chain_reads); put(gen_on | (rec_chains ? 2u : 0u) | 4u /* sizes as differences */ | (seg_len ? 8u : 0u) | (exc_rice ? 16u : 0u) | ((gm && gen_on) ? 32u : 0u) /* bases: the match model */
            | ((gm && !gen_on) ? (flat_quads_req ? 64u : 128u) : 0u) /* bases without a model: two bits each, no coder (block format 10; 64 = four a symbol through the coder: format 9's, still read) */);
        if (gm && gen_on) { put(gmplan.tb); put(ggeo.chain_reads); put(ngc); }      // (the index's bits; the base chains' records, their number)
        put(nchains);
        if (seg_len) { put(seg_len); for (u32 b = 0; b < nblocks; b++) put(seg_blk[b]); }      // segments: their length, every block's share of the chains
        if (chn_on_device) {           // the lists as the device wrote them
            const u64* info = (const u64*)h_csz;
            const u8* lb = (const u8*)h_csz + 16;
            const u64 la = info[0], lall = info[1];
            if (la > lall || lall > (u64)chn_n * 5) return fail(ctx, SFQ_E_HIP, "chain index: %llu / %llu bytes of lists for %u sizes", (unsigned long long)la, (unsigned long long)lall, chn_n);
            std::vector<u8> rest;
            if (lall > chn_eager) {                                      // (sizes that take more bytes than the guess: the rest of them)
                rest.resize((size_t)lall);
                HIPC(hipMemcpyAsync(rest.data(), ctx->chn_out.p, (size_t)lall, hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                lb = rest.data();
            }
            if (o.size() < o_need + (size_t)lall) { const size_t used = (size_t)(w - o.data()); o.resize(o_need + (size_t)lall); w = o.data() + used; }
            memcpy(w, lb, (size_t)la); w += la;
            if (rec_chains) { put(ca.rgeo.chain_reads); put(nsub); memcpy(w, lb + la, (size_t)(lall - la)); w += lall - la; }
        } else {
        put_list(h_csz, nchains); put_list(h_csz + nchains, ngc);
        if (rec_chains) {              // header chains: records per chain, their number, stream sizes, header bytes
            put(ca.rgeo.chain_reads); put(nsub);
            put_list(h_csz + (size_t)nchains + ngc, nsub); put_list(h_csz + (size_t)nchains + ngc + nsub, nsub);
        }
        }
        ctx->chain_blob.assign(o.data(), w);
    }
    ht.mark("chain index written");
    res->n_chains = nchains;
    ctx->index.resize(nblocks);
    for (u32 b = 0; b < nblocks; b++) {
        sfq_block_info& bi = ctx->index[b];
        const BlockDesc& d = hb[b];
        memset(&bi, 0, sizeof bi);
        bi.first_record = d.rec0; bi.n_records = n_over ? (u32)nrec_file : d.nrec; bi.llen = d.llen;      // (num_records counts the oversize records too, usrs.cpp:405)
        bi.solid = d.solid; bi.two_id = d.two_id; bi.n_byte = (u8)d.n_byte; bi.gen_bits = d.gen_bits;
        bi.extra_hi = d.extra_hi; bi.first_hdr_len = d.first_hdr_len; bi.first_hdr_off = hboff[b];
        for (int s = 0; s < SFQ_NSTREAMS; s++) bi.size[s] = d.size[s];
        bi.status = 0; bi.hdr_bytes = d.hdr_bytes;
    }
    res->n_records = nrec_file; res->n_blocks = nblocks; res->first_hdr_bytes = hboff[nblocks];
    res->kernel_ms[SFQ_T_FRAME] = ev_ms(ctx->ev[0], ctx->ev[1]);
    for (int m = 0; m < 4; m++) res->kernel_ms[tslot[m]] = ev_ms(ctx->ev[2 + 2 * m], ctx->ev[3 + 2 * m]);     // the models overlap: these do not add up
    res->kernel_ms[SFQ_T_PACK] = ev_ms(ctx->ev[10], ctx->ev[11]);
    res->kernel_ms[SFQ_T_TOTAL] = ev_ms(ctx->ev[0], ctx->ev[11]);
    if (reframe) res->coder_ms[3] = ev_ms(ctx->ev[12], ctx->ev[23]);          // the framing kernel (frame.hip k_frame)
    ht.mark("block index built");
    if (frozen) {
        if (models & SFQ_M_QLT) res->coder_ms[0] = ev_ms(ctx->ev[14], ctx->ev[15]);
        if (models & SFQ_M_GEN) res->coder_ms[1] = ev_ms(ctx->ev[16], ctx->ev[17]);
        if (models & SFQ_M_REC) res->coder_ms[2] = ev_ms(ctx->ev[18], ctx->ev[25]);      // (the token step; where every header is long, the general kernel)
    } else {                                       // one persistent kernel per model: the phase is the kernel
        res->coder_ms[0] = res->kernel_ms[SFQ_T_QLT]; res->coder_ms[1] = res->kernel_ms[SFQ_T_GEN]; res->coder_ms[2] = res->kernel_ms[SFQ_T_REC];
    }
    return SFQ_OK;
}

int sfq_encode_blocks(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params,
                      uint8_t* d_out, uint64_t out_cap, sfq_result* result) {
    return encode_impl(ctx, d_fastq, nbytes, params, d_out, out_cap, result, 0);
}
int sfq_encode_qlt_blocks(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params,
                          uint8_t* d_out, uint64_t out_cap, sfq_result* result) {
    return encode_impl(ctx, d_fastq, nbytes, params, d_out, out_cap, result, SFQ_M_QLT);
}
int sfq_build_priors(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params) {
    sfq_result res;
    return encode_impl(ctx, d_fastq, nbytes, params, nullptr, 0, &res, 0, true);
}
int sfq_count_priors(sfq_ctx* ctx, const uint8_t* d_fastq, uint64_t nbytes, const sfq_params* params, uint32_t sample_scale) {
    if (!ctx || !params) return fail(ctx, SFQ_E_ARG, "null argument");
    if (params->prior_step == SFQ_PRIOR_GIVEN || params->prior_step == SFQ_PRIOR_COUNTS || !params->block_reads)
        return fail(ctx, SFQ_E_ARG, "sfq_count_priors: a sample of this text's own, in the block format (prior_step: a step or SFQ_PRIOR_AUTO)");
    sfq_result res;
    ctx->counts_only = true; ctx->sample_scale = sample_scale ? sample_scale : 1;
    const int rc = encode_impl(ctx, d_fastq, nbytes, params, nullptr, 0, &res, 0, true);
    ctx->counts_only = false; ctx->sample_scale = 1;
    return rc;
}
void sfq_prior_counts_words(int level, uint64_t* qlt_words, uint64_t* rec_words) {
    if (qlt_words) *qlt_words = (uint64_t)(clamp_level(level) == 1 ? (1u << 12) : (1u << 16)) * 64;
    if (rec_words) *rec_words = (uint64_t)PR_REC_ROWS * 256;
}
int sfq_get_prior_counts(sfq_ctx* ctx, int level, uint32_t* d_qlt, uint32_t* d_rec) {
    if (!ctx || !d_qlt || !d_rec) return fail(ctx, SFQ_E_ARG, "null argument");
    uint64_t nq, nr; sfq_prior_counts_words(level, &nq, &nr);
    if (!ctx->counts.valid || !ctx->hist.p || ctx->counts.q_rows != (u32)(nq / 64)) return fail(ctx, SFQ_E_ARG, "no counts for level %d (sfq_count_priors comes first)", level);
    HIPC(hipSetDevice(ctx->dev));
    HIPC(hipMemcpyAsync(d_qlt, ctx->hist.p, nq * 4, hipMemcpyDeviceToDevice, ctx->st));
    // (counted for adaptive tables: there is no header sample -- zeros, so that the ranks' sums stay sums)
    if (ctx->counts.rec && ctx->hcnt.p) HIPC(hipMemcpyAsync(d_rec, ctx->hcnt.p, nr * 4, hipMemcpyDeviceToDevice, ctx->st));
    else HIPC(hipMemsetAsync(d_rec, 0, nr * 4, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
    return SFQ_OK;
}
int sfq_set_prior_counts(sfq_ctx* ctx, int level, const uint32_t* d_qlt, const uint32_t* d_rec) {
    if (!ctx || !d_qlt || !d_rec) return fail(ctx, SFQ_E_ARG, "null argument");
    uint64_t nq, nr; sfq_prior_counts_words(level, &nq, &nr);
    HIPC(hipSetDevice(ctx->dev));
    int rc;
    const bool own_rec = ctx->counts.valid && ctx->counts.q_rows == (u32)(nq / 64) ? ctx->counts.rec : true;     // (sums of counts taken for adaptive tables stay "no header sample")
    ctx->counts.valid = false;
    if ((rc = ensure_prior_buffers(ctx, (u32)(nq / 64)))) return rc;
    if ((rc = reserve(ctx, ctx->hcnt, (size_t)REC_COUNT_COPIES * PR_REC_ROWS * 256 * 4))) return rc;
    HIPC(hipMemcpyAsync(ctx->hist.p, d_qlt, nq * 4, hipMemcpyDeviceToDevice, ctx->st));
    HIPC(hipMemcpyAsync(ctx->hcnt.p, d_rec, nr * 4, hipMemcpyDeviceToDevice, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
    ctx->counts.valid = true; ctx->counts.q_rows = (u32)(nq / 64); ctx->counts.rec = own_rec;
    return SFQ_OK;
}
int sfq_encode_blocks_host(sfq_ctx* ctx, const uint8_t* h_fastq, uint64_t nbytes, const sfq_params* params,
                           uint8_t* h_out, uint64_t out_cap, sfq_result* result) {
    if (!ctx || !h_fastq || !h_out) return fail(ctx, SFQ_E_ARG, "null argument");
    HIPC(hipSetDevice(ctx->dev));
    int rc;
    if ((rc = reserve(ctx, ctx->in_stage, (size_t)nbytes + 16))) return rc;
    const u64 bound = sfq_encode_bound(nbytes);
    if ((rc = reserve(ctx, ctx->out_stage, (size_t)bound))) return rc;
    HIPC(hipMemcpyAsync(ctx->in_stage.p, h_fastq, (size_t)nbytes, hipMemcpyHostToDevice, ctx->st));
    rc = encode_impl(ctx, (const u8*)ctx->in_stage.p, nbytes, params, (u8*)ctx->out_stage.p, bound, result, 0);
    if (rc) return rc;
    if (result->total_bytes > out_cap) return fail(ctx, SFQ_E_OVERFLOW, "output needs %llu bytes", (unsigned long long)result->total_bytes);
    HIPC(hipMemcpyAsync(h_out, ctx->out_stage.p, (size_t)result->total_bytes, hipMemcpyDeviceToHost, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
    return SFQ_OK;
}

int sfq_get_block_index(sfq_ctx* ctx, sfq_block_info* h_blocks, uint32_t cap) {
    if (!ctx || !h_blocks) return SFQ_E_ARG;
    if (cap < ctx->index.size()) return fail(ctx, SFQ_E_ARG, "index has %zu blocks", ctx->index.size());
    if (!ctx->index.empty()) memcpy(h_blocks, ctx->index.data(), ctx->index.size() * sizeof(sfq_block_info));
    return (int)ctx->index.size();
}
int sfq_get_first_headers(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap) {
    if (!ctx || !h_blob) return SFQ_E_ARG;
    if (cap < ctx->first_hdrs.size()) return fail(ctx, SFQ_E_ARG, "blob has %zu bytes", ctx->first_hdrs.size());
    if (!ctx->first_hdrs.empty()) memcpy(h_blob, ctx->first_hdrs.data(), ctx->first_hdrs.size());
    return SFQ_OK;
}

int64_t sfq_get_qlt_prior(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap) {
    if (!ctx) return SFQ_E_ARG;
    if (h_blob && cap >= ctx->prior_blob.size() && !ctx->prior_blob.empty()) memcpy(h_blob, ctx->prior_blob.data(), ctx->prior_blob.size());
    return (int64_t)ctx->prior_blob.size();
}
static void drop_encode_blobs(sfq_ctx* ctx) {
    if (!ctx->blobs_from_encode) return;
    ctx->prior_blob.clear(); ctx->rec_prior_blob.clear(); ctx->chain_blob.clear();
    ctx->blobs_from_encode = false;
}
int sfq_set_qlt_prior(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n) {
    if (!ctx || (n && !h_blob)) return SFQ_E_ARG;
    drop_encode_blobs(ctx);
    ctx->prior_blob.assign(h_blob, h_blob + n);
    return SFQ_OK;
}
int64_t sfq_get_rec_prior(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap) {
    if (!ctx) return SFQ_E_ARG;
    if (h_blob && cap >= ctx->rec_prior_blob.size() && !ctx->rec_prior_blob.empty()) memcpy(h_blob, ctx->rec_prior_blob.data(), ctx->rec_prior_blob.size());
    return (int64_t)ctx->rec_prior_blob.size();
}
int sfq_set_rec_prior(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n) {
    if (!ctx || (n && !h_blob)) return SFQ_E_ARG;
    drop_encode_blobs(ctx);
    ctx->rec_prior_blob.assign(h_blob, h_blob + n);
    return SFQ_OK;
}
int64_t sfq_get_chain_index(sfq_ctx* ctx, uint8_t* h_blob, uint64_t cap) {
    if (!ctx) return SFQ_E_ARG;
    if (h_blob && cap >= ctx->chain_blob.size() && !ctx->chain_blob.empty()) memcpy(h_blob, ctx->chain_blob.data(), ctx->chain_blob.size());
    return (int64_t)ctx->chain_blob.size();
}
int sfq_set_chain_index(sfq_ctx* ctx, const uint8_t* h_blob, uint64_t n) {
    if (!ctx || (n && !h_blob)) return SFQ_E_ARG;
    drop_encode_blobs(ctx);
    ctx->chain_blob.assign(h_blob, h_blob + n);
    return SFQ_OK;
}

// -------------------------------------------------------------------------------------------------
// decompress
// -------------------------------------------------------------------------------------------------
static int decode_body(sfq_ctx* ctx, const sfq_params* pp, const sfq_block_info* h_blocks, uint32_t nblocks,
                       const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                       const uint8_t* d_streams, const uint64_t stream_offset[SFQ_NSTREAMS],
                       uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* res);
int sfq_decode_blocks(sfq_ctx* ctx, const sfq_params* pp, const sfq_block_info* h_blocks, uint32_t nblocks,
                      const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                      const uint8_t* d_streams, const uint64_t stream_offset[SFQ_NSTREAMS],
                      uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* res) {
    if (!ctx || !pp || !h_blocks || !nblocks || !d_streams || !stream_offset || !d_out || !out_bytes)
        return fail(ctx, SFQ_E_ARG, "null argument");
    HIPC(hipSetDevice(ctx->dev));
    Settle settle(ctx);
    drop_encode_blobs(ctx);                             // whatever an encode left behind is not this archive's
    const int rc = decode_body(ctx, pp, h_blocks, nblocks, h_first_hdrs, first_hdr_bytes, d_streams, stream_offset, d_out, out_cap, out_bytes, res);
    settle.ok = rc == SFQ_OK;
    return rc;
}
static int decode_body(sfq_ctx* ctx, const sfq_params* pp, const sfq_block_info* h_blocks, uint32_t nblocks,
                       const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                       const uint8_t* d_streams, const uint64_t stream_offset[SFQ_NSTREAMS],
                       uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* res) {
    sfq_params p = *pp;
    HostTimes ht;
    p.level = clamp_level(p.level);
    ctx->framed.valid = false;                     // (a decode's scratch is not an encode's line index)
    const u32 version = p.version ? p.version : 6;
    if (p.kernel == 2) p.kernel = 0;                     // (an encoder's choice; a decoder reads what "chn.idx" says)
    if (p.kernel > 1) return fail(ctx, SFQ_E_ARG, "kernel %u: 0 = default kernels, 1 = lane-per-block cross-check kernels", p.kernel);
    if (version > 6) return fail(ctx, SFQ_E_UNSUPPORTED, "archive version %u is newer than 6 (config.cpp:373-377)", version);
    hipStream_t st = ctx->st;
    int rc;
    sfq_result local; if (!res) res = &local;
    memset(res, 0, sizeof *res);
    res->abi_version = SFQ_ABI_VERSION;
    HIPC(hipEventRecord(ctx->ev[0], st));

    // What the host builds for the device -- block descriptors, stream offsets, the chains' sizes and offsets, the header
    // staging slices -- is built in page-locked memory (the encoder's end-of-call scratch): a copy out of pageable memory goes
    // through the driver's bounce buffer at a few GB/s, and these are megabytes per call.
    Bump bump;
    {
        const size_t cn = ctx->chain_blob.size();              // (a chain takes at least a byte of "chn.idx")
        const size_t need = (size_t)nblocks * (sizeof(BlockDesc) + 8 * SFQ_NSTREAMS + 12) + (cn + 64) * 24 + 4096;
        if ((rc = reserve_pinned_buf(ctx, ctx->pin2, ctx->pin2_cap, need))) return rc;
        bump.p = (u8*)ctx->pin2; bump.off = 0; bump.cap = need;
    }
    // device block descriptors + per-block stream offsets
    BlockDesc* hb = bump.take<BlockDesc>(nblocks);
    const size_t nbso = (size_t)nblocks * SFQ_NSTREAMS;
    u64* bso = bump.take<u64>(nbso);
    if (!hb || !bso) return fail(ctx, SFQ_E_NOMEM, "decode: host scratch");
    u64 run[SFQ_NSTREAMS];
    for (int s = 0; s < SFQ_NSTREAMS; s++) run[s] = stream_offset[s];
    u64 nrec = 0; u32 block_reads = h_blocks[0].n_records; int g_bits = 0, g_bits_min = 64;
    for (u32 b = 0; b < nblocks; b++) {
        const sfq_block_info& bi = h_blocks[b];
        BlockDesc& d = hb[b];
        memset(&d, 0, sizeof d);
        if (bi.first_record != nrec) return fail(ctx, SFQ_E_ARG, "block %u: first_record %llu, expected %llu", b, (unsigned long long)bi.first_record, (unsigned long long)nrec);
        if (b + 1 < nblocks && bi.n_records != block_reads) return fail(ctx, SFQ_E_ARG, "block %u: non-uniform block size", b);
        if (bi.n_records == 0 || (b + 1 == nblocks && bi.n_records > block_reads)) return fail(ctx, SFQ_E_ARG, "block %u: bad record count", b);
        if (bi.first_hdr_off + bi.first_hdr_len > first_hdr_bytes) return fail(ctx, SFQ_E_ARG, "block %u: first header outside the blob", b);
        if (bi.first_hdr_len > SFQ_MAX_ID_LLEN) return fail(ctx, SFQ_E_CORRUPT, "block %u: first header of %u bytes (limit %u, usrs.hpp:34)", b, bi.first_hdr_len, (u32)SFQ_MAX_ID_LLEN);
        if (bi.gen_bits < 2 || bi.gen_bits > 26) return fail(ctx, SFQ_E_ARG, "block %u: gen_bits %u", b, bi.gen_bits);
        d.rec0 = nrec; d.nrec = bi.n_records; d.llen = bi.llen; d.solid = bi.solid; d.two_id = bi.two_id;
        d.gen_bits = bi.gen_bits; d.n_byte = bi.n_byte; d.first_hdr_off = bi.first_hdr_off; d.first_hdr_len = bi.first_hdr_len;
        g_bits = std::max<int>(g_bits, bi.gen_bits); g_bits_min = std::min<int>(g_bits_min, bi.gen_bits);
        for (int s = 0; s < SFQ_NSTREAMS; s++) {
            if (bi.size[s] > 0xFFFFFFFFull) return fail(ctx, SFQ_E_UNSUPPORTED, "block %u: stream %s has %llu bytes (the kernels take streams below 4 GiB)", b, sfq_stream_name(s), (unsigned long long)bi.size[s]);
            d.size[s] = (u32)bi.size[s]; bso[(size_t)b * SFQ_NSTREAMS + s] = run[s]; run[s] += bi.size[s];
        }
        nrec += bi.n_records;
    }
    // format 6's oversize records (usrs.cpp:269-301): one block, its "usr.lrec" / "usr.lgen" / "usr.lqlt" streams
    const bool has_over = h_blocks[0].size[SFQ_S_USR_LREC] != 0;
    for (u32 b = has_over ? 1u : 0u; b < nblocks; b++)
        if (h_blocks[b].size[SFQ_S_USR_LREC] | h_blocks[b].size[SFQ_S_USR_LGEN] | h_blocks[b].size[SFQ_S_USR_LQLT])
            return fail(ctx, SFQ_E_UNSUPPORTED, "block %u holds oversize records (usr.lrec): only a one-block (format 6) archive does", b);
    if (has_over && nblocks != 1) return fail(ctx, SFQ_E_UNSUPPORTED, "oversize records (usr.lrec) in an archive of %u blocks: only a one-block (format 6) archive has them", nblocks);
    ht.mark("blocks checked");
    // frozen tables: the chain index ("chn.idx")
    const bool frozen = !ctx->chain_blob.empty();
    u32 chain_reads = 0, cpb = 0, nchains = 0, gen_on = 0, rec_chains = 0, rchain_reads = 0, rcpb = 0, nsub = 0, gm_on = 0, gm_tb = 0;
    u32 flat_quads = 0, flat_raw = 0;
    u32 gchain_reads = 0, gcpb = 0, ngc = 0; u64 gm_ngc = 0;      // the base chains (the quality chains' geometry unless the match model has its own)
    bool exc_rice = false;
    u32 seg_len = 0; std::vector<u32> seg_c0;            // segments (chains.hip): their length, the first chain of every block
    std::vector<u32> h_rsz, h_rhb, rec_prior_f;
    u32* h_csz = nullptr; u64* h_coff = nullptr; size_t ncs = 0;
    // the chain lists of "chn.idx" are read beside the head of the call (ctx->worker); whatever way this function is left, the job is
    // through before the locals it writes to are gone
    std::string lists_err; bool lists_pending = false;
    std::atomic<int> lists_stage{0};                    // 1: the quality chains' list stands in h_csz / h_coff (the decoder that needs it is the call's longest); -1: the job has failed
    struct ListsGuard { sfq_ctx* c; bool* pending; ~ListsGuard() { if (*pending) (void)c->worker.wait(); } } lists_guard{ ctx, &lists_pending };
    if (frozen) {
        const u8* cb = ctx->chain_blob.data(); const size_t cn = ctx->chain_blob.size();
        size_t cp = 0; u64 v = 0;
        if (!get_v(cb, cn, cp, v) || v == 0 || v > 0xFFFFFFFFull) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx)");
        chain_reads = (u32)v;
        if (!get_v(cb, cn, cp, v)) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx)");
        gen_on = (u32)v & 1u; rec_chains = ((u32)v >> 1) & 1u;
        const bool deltas = ((u32)v >> 2) & 1u;              // sizes as zigzag differences to the entry before (round 4; version-8 archives of round 3: plain)
        const bool segs = ((u32)v >> 3) & 1u;                // chains are segments of one record (long reads): their length and the blocks' shares follow
        exc_rice = ((u32)v >> 4) & 1u;                       // the base exceptions are Rice-coded gap lists (exc.hip; round 4)
        gm_on = ((u32)v >> 5) & 1u;                          // the bases are coded under the match model (gm.hip; round 5): the index's bits follow
        flat_quads = ((u32)v >> 6) & 1u;                     // bases without a model are coded four a symbol (round 5)
        flat_raw = ((u32)v >> 7) & 1u;                       // ... or not coded at all: two bits each, four a byte (round 5b, block format 10)
        if (v >> 8 || ((flat_quads | flat_raw) && gen_on) || (flat_quads && flat_raw)) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: unknown flags)");
        if (gm_on) {
            u64 t = 0;
            if (!gen_on || !exc_rice || !get_v(cb, cn, cp, t) || t < 8 || t > 26) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: match model)");
            gm_tb = (u32)t;
            // the base chains' own geometry: records per chain (at most the quality chains'), their number
            u64 gc = 0, gn = 0;
            if (!get_v(cb, cn, cp, gc) || !get_v(cb, cn, cp, gn) || gc == 0 || gc > chain_reads) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: base chains)");
            gchain_reads = (u32)gc; gm_ngc = gn;
        }
        if (!get_v(cb, cn, cp, v)) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx)");
        if (chain_reads > block_reads) return fail(ctx, SFQ_E_CORRUPT, "chain index: %u records per chain, %u per block", chain_reads, block_reads);
        cpb = (block_reads + chain_reads - 1) / chain_reads;
        const u32 last_nrec = h_blocks[nblocks - 1].n_records;
        u64 want = (u64)(nblocks - 1) * cpb + (last_nrec + chain_reads - 1) / chain_reads;
        if (segs) {
            if (chain_reads != 1 || v > 0x7FFFFFFFull || nblocks > cn) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: segments)");
            want = v;
            u64 sl = 0;
            if (!get_v(cb, cn, cp, sl) || sl == 0 || sl > 0x7FFFFFFFull) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: segments)");
            seg_len = (u32)sl;
            seg_c0.resize((size_t)nblocks + 1);
            u64 run_c = 0;
            for (u32 b = 0; b < nblocks; b++) {
                u64 k = 0;
                if (!get_v(cb, cn, cp, k) || k < h_blocks[b].n_records || k > want) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: a block's segments)");
                seg_c0[b] = (u32)run_c; run_c += k;
            }
            seg_c0[nblocks] = (u32)run_c;
            if (run_c != want) return fail(ctx, SFQ_E_CORRUPT, "chain index: the blocks' segments do not add up");
        }
        if (v != want || want > 0x7FFFFFFFull) return fail(ctx, SFQ_E_CORRUPT, "chain index: %llu chains, the blocks have %llu", (unsigned long long)v, (unsigned long long)want);
        nchains = (u32)want;
        gcpb = cpb; ngc = nchains;
        if (gm_on) {
            if (segs ? (gchain_reads != 1 || gm_ngc != want) : false) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx: base chains of a call in segments)");
            if (!segs) {
                gcpb = (block_reads + gchain_reads - 1) / gchain_reads;
                const u64 wantg = (u64)(nblocks - 1) * gcpb + (last_nrec + gchain_reads - 1) / gchain_reads;
                if (gm_ngc != wantg || wantg > 0x7FFFFFFFull) return fail(ctx, SFQ_E_CORRUPT, "chain index: %llu base chains, the blocks have %llu", (unsigned long long)gm_ngc, (unsigned long long)wantg);
                ngc = (u32)wantg;
            }
        } else gchain_reads = chain_reads;
        h_csz = bump.take<u32>(cn + 16);
        h_coff = bump.take<u64>(cn + 16);
        if (!h_csz || !h_coff || (size_t)nchains + ngc > cn) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx)");
        ncs = (size_t)nchains + ngc;
        // what the call needs besides the streams is checked here, on host state alone, before any device work is queued
        if (ctx->prior_blob.empty()) return fail(ctx, SFQ_E_ARG, "frozen tables need the quality prior (qlt.pri)");
        if (rec_chains) {
            if (ctx->rec_prior_blob.empty()) return fail(ctx, SFQ_E_ARG, "header chains need the header prior (rec.pri)");
            if (!unpack_rec_prior(ctx->rec_prior_blob.data(), ctx->rec_prior_blob.size(), rec_prior_f)) return fail(ctx, SFQ_E_CORRUPT, "bad header prior (rec.pri)");
        }
        // The lists themselves -- half a million sizes, 1.2 ms of host time -- are read by a thread of their own (pure host work: no HIP
        // call, no ctx state) while this one uploads the priors and queues the head of the call; joined before the chains' decoders
        // are queued (lists_job)
        lists_pending = true;
        ctx->worker.submit([&, cb, cn, cp, deltas]() mutable -> int {
        u64 v = 0;
        auto parse_rec_chains = [&]() -> bool {
            if (!get_v(cb, cn, cp, v) || v == 0 || v > block_reads) return false;
            rchain_reads = (u32)v;
            rcpb = (block_reads + rchain_reads - 1) / rchain_reads;
            const u64 wantr = (u64)(nblocks - 1) * rcpb + (h_blocks[nblocks - 1].n_records + rchain_reads - 1) / rchain_reads;
            if (!get_v(cb, cn, cp, v) || v != wantr || wantr > 0x7FFFFFFFull) return false;
            // (every listed size takes a byte of the index or more: a count the index cannot hold is refused BEFORE anything is sized
            //  by it -- a damaged archive could ask for 2^31 entries here, on a thread where bad_alloc would end the process)
            if (wantr > cn - std::min(cn, cp) || ncs + wantr > cn + 16) return false;
            nsub = (u32)wantr;
            h_rsz.resize(nsub); h_rhb.resize(nsub);
            if (!read_sizes(cb, cn, cp, deltas, h_rsz.data(), nsub) || !read_sizes(cb, cn, cp, deltas, h_rhb.data(), nsub)) return false;
            for (u32 b = 0; b < nblocks; b++) {
                u64 sum = 0;
                for (u32 j = 0; j < rcpb && (u64)b * rcpb + j < nsub; j++) sum += h_rsz[(size_t)b * rcpb + j];
                if (sum != h_blocks[b].size[SFQ_S_REC]) return false;
            }
            return true;
        };
        struct Fail { std::atomic<int>& st; bool ok = false; ~Fail() { if (!ok) st.store(-1, std::memory_order_release); } } on_exit{ lists_stage };
        for (int k = 0; k < 2; k++) {
            const int sid = k ? SFQ_S_GEN : SFQ_S_QLT;
            const u32 kn = k ? ngc : nchains, kcpb = k ? gcpb : cpb;
            u32* const sz = h_csz + (size_t)k * nchains; u64* const off = h_coff + (size_t)k * nchains;
            if (k == 1) lists_stage.store(1, std::memory_order_release);
            if (!read_sizes(cb, cn, cp, deltas, sz, kn)) { lists_err = "bad chain index (chn.idx)"; return SFQ_E_CORRUPT; }
            u64 at = stream_offset[sid];
            for (u32 b = 0; b < nblocks; b++) {
                u64 sum = 0;
                const u64 bc0 = seg_len ? seg_c0[b] : (u64)b * kcpb, bc1 = seg_len ? seg_c0[b + 1] : std::min<u64>(bc0 + kcpb, kn);
                for (u64 cc = bc0; cc < bc1; cc++) { off[cc] = at + sum; sum += sz[cc]; }
                at += sum;
                if (sum != h_blocks[b].size[sid]) { char m[160]; snprintf(m, sizeof m, "chain index: block %u's chains do not add up to its %s stream", b, sfq_stream_name(sid)); lists_err = m; return SFQ_E_CORRUPT; }
            }
        }
        if (rec_chains) {
            if (!parse_rec_chains()) { lists_err = "bad chain index (chn.idx: header chains)"; return SFQ_E_CORRUPT; }
            u64 at = stream_offset[SFQ_S_REC];
            if (ncs + nsub > cn + 16) { lists_err = "bad chain index (chn.idx: header chains)"; return SFQ_E_CORRUPT; }
            for (u32 c = 0; c < nsub; c++) { h_csz[ncs] = h_rsz[c]; h_coff[ncs] = at; ncs++; at += h_rsz[c]; }
        }
        on_exit.ok = true;
        return SFQ_OK;
        });
    }
    ht.mark("  header prior, chain sizes up");
    if ((rc = reserve(ctx, ctx->blocks, (size_t)nblocks * sizeof(BlockDesc)))) return rc;
    if ((rc = reserve(ctx, ctx->blk_stream_off, nbso * 8))) return rc;
    if ((rc = reserve(ctx, ctx->d_first, (size_t)first_hdr_bytes + 16))) return rc;
    HIPC(hipMemcpyAsync(ctx->blocks.p, hb, (size_t)nblocks * sizeof(BlockDesc), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(ctx->blk_stream_off.p, bso, nbso * 8, hipMemcpyHostToDevice, st));
    if (first_hdr_bytes) HIPC(hipMemcpyAsync(ctx->d_first.p, h_first_hdrs, (size_t)first_hdr_bytes, hipMemcpyHostToDevice, st));

    const u32 q_rows = p.level == 1 ? (1u << 12) : (1u << 16);
    u32 slots = 0;
    if ((rc = ensure_tables(ctx, nblocks, q_rows, (u32)g_bits, frozen ? (SFQ_M_REC | SFQ_M_USR) : SFQ_M_ALL, &slots))) return rc;
    if ((rc = advance_epoch(ctx, nblocks))) return rc;

    if ((rc = reserve(ctx, ctx->slen, (size_t)nrec * 4))) return rc;
    if ((rc = reserve(ctx, ctx->qlen, (size_t)nrec * 4))) return rc;
    if ((rc = reserve(ctx, ctx->pfg, (size_t)nrec))) return rc;
    if ((rc = reserve(ctx, ctx->pfq, (size_t)nrec))) return rc;
    if ((rc = reserve(ctx, ctx->soff, ((size_t)nrec + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->qoff, ((size_t)nrec + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->hlen, (size_t)nrec * 4))) return rc;
    if ((rc = reserve(ctx, ctx->hoff, (size_t)nrec * 8))) return rc;
    if ((rc = reserve(ctx, ctx->rsize, (size_t)nrec * 4))) return rc;
    if ((rc = reserve(ctx, ctx->roff, ((size_t)nrec + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->scan_tmp, ((size_t)nrec / 1024 + 4) * 8 + 65536))) return rc;

    ctx->prior_on = false;
    if (!ctx->prior_blob.empty()) {
        ht.mark("  buffers reserved");
        if ((rc = upload_prior(ctx, q_rows, st, !frozen))) return rc;
        ht.mark("  quality prior up");
        ctx->prior_on = true;
    }
    DecodeArgs da;
    memset(&da, 0, sizeof da);
    fill_model_args(ctx, da.m, nblocks, p.level, (u32)g_bits, false);                      // (a decoder follows what the streams say)
    ctx->prior_on = false;
    da.streams = d_streams;
    da.blk_stream_off = (const u64*)ctx->blk_stream_off.p;
    da.first_hdrs = (const u8*)ctx->d_first.p;
    da.slen = (u32*)ctx->slen.p; da.qlen = (u32*)ctx->qlen.p; da.pfg = (u8*)ctx->pfg.p; da.pfq = (u8*)ctx->pfq.p;
    da.soff = (const u64*)ctx->soff.p; da.qoff = (const u64*)ctx->qoff.p;
    da.hlen = (u32*)ctx->hlen.p; da.hoff = (u64*)ctx->hoff.p;
    da.block_reads = block_reads; da.version = version; da.max_line = (u32)std::min<u64>(out_cap, 0x3ffffffeull);

    // 0. format 6's oversize records (UsrLoad::update, usrs.cpp:473-485): their numbers and raw lines first -- the other
    //    streams count records in file numbers, the models never saw them
    const u64 nrec_file = nrec; u32 n_over = 0;
    if (has_over) {
        ctx->epoch_base += 2;                              // (the passes below run under epochs of their own: models_w.hip k_over_decode_w)
        const sfq_block_info& bi = h_blocks[0];
        const u8* s_l[3] = { d_streams + stream_offset[SFQ_S_USR_LREC], d_streams + stream_offset[SFQ_S_USR_LGEN], d_streams + stream_offset[SFQ_S_USR_LQLT] };
        if ((rc = reserve(ctx, ctx->ocnt, 64))) return rc;
        HIPC(hipMemsetAsync(ctx->ocnt.p, 0, 64, st));
        da.m.batch0 = 0; da.m.nbatch = 1;
        launch_over_decode_w(da.m, s_l[0], (u32)bi.size[SFQ_S_USR_LREC], 0, 0, 0, (u64*)ctx->ocnt.p, nullptr, nullptr, nullptr, out_cap, st);
        u64 h_cnt[2] = {0, 0}; u32 h_st = 0;
        u32* d_status0 = &((BlockDesc*)ctx->blocks.p)[0].status;
        HIPC(hipMemcpyAsync(h_cnt, ctx->ocnt.p, 16, hipMemcpyDeviceToHost, st));
        HIPC(hipMemcpyAsync(&h_st, d_status0, 4, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        if (h_st || h_cnt[0] == 0 || h_cnt[0] >= nrec_file || h_cnt[1] > out_cap) return fail(ctx, SFQ_E_CORRUPT, "damaged oversize stream (usr.lrec)");
        n_over = (u32)h_cnt[0];
        if ((rc = reserve(ctx, ctx->ono, (size_t)n_over * 8))) return rc;
        if ((rc = reserve(ctx, ctx->opiece, (size_t)n_over * 64))) return rc;
        if ((rc = reserve(ctx, ctx->otxt[0], (size_t)h_cnt[1] + 16))) return rc;
        if ((rc = reserve(ctx, ctx->otxt[1], (size_t)out_cap + 16))) return rc;
        if ((rc = reserve(ctx, ctx->otxt[2], (size_t)out_cap + 16))) return rc;
        HIPC(hipMemsetAsync(ctx->opiece.p, 0, (size_t)n_over * 64, st));
        HIPC(hipEventRecord(ctx->ev[8], st));
        launch_over_decode_w(da.m, s_l[0], (u32)bi.size[SFQ_S_USR_LREC], 0, 1, n_over, nullptr, (u64*)ctx->ono.p, (u64*)ctx->opiece.p, (u8*)ctx->otxt[0].p, h_cnt[1], st);
        // the base and quality lines: a wave each, beside everything else (joined before the records are laid out)
        HIPC(hipStreamWaitEvent(ctx->st_aux[2], ctx->ev[8], 0));
        launch_over_decode_w(da.m, s_l[1], (u32)bi.size[SFQ_S_USR_LGEN], 1, 1, n_over, nullptr, nullptr, (u64*)ctx->opiece.p, (u8*)ctx->otxt[1].p, out_cap, ctx->st_aux[2]);
        launch_over_decode_w(da.m, s_l[2], (u32)bi.size[SFQ_S_USR_LQLT], 2, 1, n_over, nullptr, nullptr, (u64*)ctx->opiece.p, (u8*)ctx->otxt[2].p, out_cap, ctx->st_aux[2]);
        HIPC(hipEventRecord(ctx->ev[9], ctx->st_aux[2]));
        // kept record -> file number
        if ((rc = reserve(ctx, ctx->oflags, (size_t)nrec_file * 4))) return rc;
        if ((rc = reserve(ctx, ctx->ofpos, ((size_t)nrec_file + 1) * 8))) return rc;
        if ((rc = reserve(ctx, ctx->orecmap, (size_t)nrec_file * 4 + 16))) return rc;
        HIPC(hipMemsetAsync(ctx->oflags.p, 0, (size_t)nrec_file * 4, st));
        launch_over_mark((const u64*)ctx->ono.p, n_over, nrec_file, (u32*)ctx->oflags.p, d_status0, st);
        launch_scan_u32((const u32*)ctx->oflags.p, (u64*)ctx->ofpos.p, nrec_file, (u64*)ctx->scan_tmp.p, st);
        launch_over_map((const u32*)ctx->oflags.p, (const u64*)ctx->ofpos.p, nrec_file, (u32*)ctx->orecmap.p, st);
        nrec = nrec_file - n_over;
        hb[0].nrec = (u32)nrec;
        HIPC(hipMemcpyAsync(&((BlockDesc*)ctx->blocks.p)[0].nrec, &hb[0].nrec, 4, hipMemcpyHostToDevice, st));
        da.m.rec_map = (const u32*)ctx->orecmap.p;
    }
    ht.mark("index, priors parsed; buffers");
    // 1. framing exceptions -> per-record line lengths
    // the header rows from "rec.pri", while the chip is still idle (behind the decoders' launches k_rec_frozen_rows took 0.6 ms, with
    // the header decoder waiting for it)
    if (frozen && rec_chains != 0) { if ((rc = upload_rec_rows(ctx, rec_prior_f, st))) return rc; }
    // "no header yet" for every record, while the chip is still idle: behind the decoders' launches these two fills -- 120 MB, a fill
    // kernel that must find room beside them -- took 2.4 ms on the header decoder's stream (hdr_marks_cleared)
    HIPC(hipMemsetAsync(ctx->hoff.p, 0xFF, (size_t)nrec * 8, st));
    HIPC(hipMemsetAsync(ctx->hlen.p, 0, (size_t)nrec * 4, st));
    const u32 usr_prefilled = (block_reads != 0 && !has_over && nblocks > 1) ? 1u : 0u;
    if (usr_prefilled) launch_usr_fill(da, nrec, st);
    // (a wave per block -- decode_w.hip -- unless the cross-check kernels are asked for or format 6's oversize records break the count of the records)
    for (u32 b0 = 0; b0 < nblocks; b0 += slots) {
        da.m.batch0 = b0; da.m.nbatch = std::min(slots, nblocks - b0);
        if (p.kernel == 0 && !has_over) launch_usr_decode_w(da, st, usr_prefilled); else launch_usr_decode_l(da, st, usr_prefilled);
    }
    const u32 spad = (frozen && !gm_on && !gen_on && exc_rice && !seg_len && nrec && out_cap / nrec >= 128) ? 31u : 0u;
    if (gm_on) {
        // the match model's stage: a '\n' behind every base line (gm.hip) -- soff[r] = (bases before r) + r
        if ((rc = reserve(ctx, ctx->gm_boff, ((size_t)nrec + 1) * 8))) return rc;
        launch_scan_u32(da.slen, (u64*)ctx->gm_boff.p, nrec, (u64*)ctx->scan_tmp.p, st);
        launch_gm_soff((const u64*)ctx->gm_boff.p, nrec, (u64*)ctx->soff.p, st);
        da.boff = (const u64*)ctx->gm_boff.p;
    } else if (spad) {
        // flat bases (no model of either kind), Rice-coded exception lists: the lines start on 32-byte sectors, as the quality lines do; the
        // exception lists' positions count bases, so the bases before every record go along (exc.hip)
        if ((rc = reserve(ctx, ctx->gm_boff, ((size_t)nrec + 1) * 8))) return rc;
        launch_scan_u32(da.slen, (u64*)ctx->gm_boff.p, nrec, (u64*)ctx->scan_tmp.p, st);
        launch_scan_u32(da.slen, (u64*)ctx->soff.p, nrec, (u64*)ctx->scan_tmp.p, st, spad);
        da.boff = (const u64*)ctx->gm_boff.p;
    } else
    launch_scan_u32(da.slen, (u64*)ctx->soff.p, nrec, (u64*)ctx->scan_tmp.p, st);
    // the quality lines' places in their stage: on 32-byte sectors where the lines are long enough for that to cost little (dev_chain.h LaneOut32)
    const u32 qpad = (frozen && nrec && out_cap / nrec >= 128) ? 31u : 0u;
    launch_scan_u32(da.qlen, (u64*)ctx->qoff.p, nrec, (u64*)ctx->scan_tmp.p, st, qpad);
    u64 tot_s = 0, tot_q = 0;
    u32 dec_max_line = 0;                              // the longest base line (how the base tables' counting passes split their work)
    if (frozen && gen_on) {                            // (only the generation tables' counting passes ask: 0.12 ms of the head otherwise)
        if ((rc = reserve(ctx, ctx->status, 256))) return rc;
        HIPC(hipMemsetAsync(ctx->status.p, 0, 256, st));
        launch_max_u32(da.slen, nrec, (u32*)ctx->status.p, st);
        HIPC(hipMemcpyAsync(&dec_max_line, ctx->status.p, 4, hipMemcpyDeviceToHost, st));
    }
    HIPC(hipMemcpyAsync(&tot_s, (u64*)ctx->soff.p + nrec, 8, hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(&tot_q, (u64*)ctx->qoff.p + nrec, 8, hipMemcpyDeviceToHost, st));
    HIPC(hipEventRecord(ctx->ev[1], st));
    HIPC(hipStreamSynchronize(st));
    // (a damaged usr stream can claim any lengths: what cannot fit the caller's buffer is refused before anything is decoded)
    if (tot_s > out_cap + (u64)spad * nrec || tot_q > out_cap + (u64)qpad * nrec) return fail(ctx, SFQ_E_CORRUPT, "line lengths add up to %llu bases / %llu qualities, the output buffer holds %llu bytes",
                                                           (unsigned long long)tot_s, (unsigned long long)tot_q, (unsigned long long)out_cap);
    if ((rc = reserve(ctx, ctx->seq_stage, (size_t)tot_s + 64))) return rc;          // (gm.hip's windows read sixteen bytes at any place up to tot_s)
    if ((rc = reserve(ctx, ctx->qual_stage, (size_t)tot_q + 16))) return rc;
    da.seq_stage = (u8*)ctx->seq_stage.p; da.qual_stage = (u8*)ctx->qual_stage.p;
    if (gm_on && seg_len) launch_gm_sentinels(da.seq_stage, da.soff, da.slen, nrec, st);        // (whole-record chains write their lines' sentinels themselves)

    ht.mark("head queued");
    const size_t lists_cap = lists_pending ? ctx->chain_blob.size() + 16 : 0;       // (every listed size is a byte of the index or more: what h_csz / h_coff were sized by)
    if (lists_pending) {
        // frozen tables: the chain lists.  The QUALITY chains' comes first in the index and its decoder is the call's longest kernel: as soon as that
        // list stands it goes to the device and the decoder is queued; the base and header chains' lists follow behind it (below) -- the reading thread
        // takes 1.2-1.4 ms for the half million sizes, the quality decoder used to wait for all of them.
        if ((rc = reserve(ctx, ctx->csz, lists_cap * 4))) return rc;
        if ((rc = reserve(ctx, ctx->coff, lists_cap * 8))) return rc;
        while (lists_stage.load(std::memory_order_acquire) == 0 && ctx->worker.busy()) std::this_thread::yield();
        if (lists_stage.load(std::memory_order_acquire) != 1) {
            rc = ctx->worker.wait(); lists_pending = false;
            if (rc) return fail(ctx, rc, "%s", lists_err.empty() ? "chain index (chn.idx): out of memory or damaged" : lists_err.c_str());
        }
        HIPC(hipMemcpyAsync(ctx->csz.p, h_csz, (size_t)nchains * 4, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(ctx->coff.p, h_coff, (size_t)nchains * 8, hipMemcpyHostToDevice, st));
    }
    ht.mark("quality chain list read, copied");
    // the other lists: behind the quality decoder's launch (frozen tables), on the base decoder's stream; the header decoder's waits for them
    auto rest_of_lists = [&](hipStream_t sg, hipStream_t sr) -> int {
        if (!lists_cap) return SFQ_OK;
        if (lists_pending) {
            rc = ctx->worker.wait(); lists_pending = false;
            if (rc) return fail(ctx, rc, "%s", lists_err.empty() ? "chain index (chn.idx): out of memory or damaged" : lists_err.c_str());
        }
        if (ncs > lists_cap) return fail(ctx, SFQ_E_CORRUPT, "bad chain index (chn.idx)");
        if (ncs > nchains) {
            HIPC(hipMemcpyAsync((u32*)ctx->csz.p + nchains, h_csz + nchains, (ncs - nchains) * 4, hipMemcpyHostToDevice, sg));
            HIPC(hipMemcpyAsync((u64*)ctx->coff.p + nchains, h_coff + nchains, (ncs - nchains) * 8, hipMemcpyHostToDevice, sg));
        }
        HIPC(hipEventRecord(ctx->ev[27], sg));
        HIPC(hipStreamWaitEvent(sr, ctx->ev[27], 0));
        ht.mark("chain lists read, copied");
        return SFQ_OK;
    };
    // 2. quality, bases and (3.) headers are independent chains: three streams.  (The one rule that ties bases to
    //    qualities -- quality '!' means N -- is applied when the records are assembled.)
    //    The fork comes BEHIND the quality rows and the hot image's helper kernels (round 5): forked in front of them, the base and
    //    header decoders' 205 k lanes filled the chip and k_hot_select -- one workgroup of 1024 threads -- waited 3.2 ms for a CU
    //    with sixteen free wave slots, the quality decoder (the call's critical path) behind it.
    hipStream_t st_rec = ctx->st_aux[0], st_gen = ctx->st_aux[1];
    auto fork_streams = [&]() -> int {
        HIPC(hipEventRecord(ctx->ev[2], st));
        HIPC(hipStreamWaitEvent(ctx->st_aux[0], ctx->ev[2], 0));
        HIPC(hipStreamWaitEvent(ctx->st_aux[1], ctx->ev[2], 0));
        return SFQ_OK;
    };
    if (frozen) {
        // quality: dense frozen rows from the prior, one chain per lane
        ChainArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.m = da.m; ca.geo.chain_reads = chain_reads; ca.geo.cpb = cpb; ca.geo.nchains = nchains; ca.block_reads = block_reads;
        if (seg_len) {
            // segments: a record's share of the chains follows from its line lengths (the usr streams have given them); what the
            // index says of the blocks must agree
            if ((rc = reserve(ctx, ctx->segn, (size_t)nrec * 4))) return rc;
            if ((rc = reserve(ctx, ctx->segoff, ((size_t)nrec + 1) * 8))) return rc;
            launch_seg_count_dec(da.slen, da.qlen, nrec, seg_len, (u32*)ctx->segn.p, st);
            launch_scan_u32((const u32*)ctx->segn.p, (u64*)ctx->segoff.p, nrec, (u64*)ctx->scan_tmp.p, st);
            std::vector<u64> h_off((size_t)nrec + 1);
            HIPC(hipMemcpyAsync(h_off.data(), ctx->segoff.p, ((size_t)nrec + 1) * 8, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            for (u32 b = 0; b <= nblocks; b++)
                if (h_off[std::min<u64>((u64)b * block_reads, nrec)] != seg_c0[b]) return fail(ctx, SFQ_E_CORRUPT, "chain index: block %u's segments disagree with its records' line lengths", b);
            if ((rc = reserve(ctx, ctx->segrec, (size_t)nchains * 4 + 16))) return rc;
            launch_seg_fill((const u64*)ctx->segoff.p, nrec, (u32*)ctx->segrec.p, st);
            ca.seg_len = seg_len; ca.seg_off = (const u64*)ctx->segoff.p; ca.seg_rec = (const u32*)ctx->segrec.p;
        }
        auto chain0_of = [&](u32 b) -> u32 { return seg_len ? seg_c0[b] : (u32)std::min<u64>((u64)b * cpb, nchains); };
        if ((rc = reserve(ctx, ctx->qrows, (size_t)q_rows * 64 * 4))) return rc;
        if ((rc = build_qesc(ctx, st))) return rc;
        if ((rc = reserve(ctx, ctx->qdec, (size_t)q_rows * 72 * 2 + 64))) return rc;            // chains.hip QDEC_ROW
        launch_qlt_frozen_rows((const u32*)ctx->rows66.p, q_rows, (u32*)ctx->qrows.p, (u16*)ctx->qdec.p, st);
        ca.qrows = (const u32*)ctx->qrows.p; ca.qesc = (const u32*)ctx->qesc.p; ca.qdec = (const u16*)ctx->qdec.p;
        ca.q_hot = 0; ca.q_rows = q_rows;
        // the coarse lists of the contexts that carry the most weight live in LDS (chains.hip launch_hot_rows_dec); automatic where the
        // call has chains for a 1024-lane workgroup on most CUs (as the encoder's image); sfq_params.lds_rows: a number, or none
        const u32 want_hot = p.lds_rows == SFQ_LDS_ROWS_NONE ? 0u : p.lds_rows ? std::min<u32>(p.lds_rows, hot_rows_dec_max())
                             : nchains >= 150000u ? hot_rows_dec_max() : 0u;
        if (want_hot) {
            const size_t img_bytes = (size_t)q_rows / 4 + (size_t)want_hot * 16 + 64;       // chains.hip QHD_ROW_U16
            if ((rc = reserve(ctx, ctx->qw, img_bytes + (size_t)q_rows * 4 + 256))) return rc;
            u8* img = (u8*)ctx->qw.p; u32* info = (u32*)(img + ((img_bytes + 15) & ~(size_t)15)); u32* ctot = info + 16;
            launch_hot_rows_dec((const u32*)ctx->rows66.p, (const u16*)ctx->qdec.p, q_rows, want_hot, ctot, img, info, st);
            ca.q_hot = want_hot; ca.qh_img = img; ca.qh_info = info;
        }
        ca.csz = (u32*)ctx->csz.p; ca.coff = (const u64*)ctx->coff.p;
        if ((rc = fork_streams())) return rc;
        launch_qlt_decode_c(ca, da, st);
        HIPC(hipEventRecord(ctx->ev[3], st));
#ifdef SFQ_EXP_QDEC_ALONE          /* scratch experiments only: the quality decoder by itself, timed; the call then fails */
        HIPC(hipStreamSynchronize(st));
        fprintf(stderr, "EXP qdec alone: %.3f ms\n", ev_ms(ctx->ev[2], ctx->ev[3]));
        return fail(ctx, SFQ_E_ARG, "experiment build");
#endif
        if ((rc = rest_of_lists(st_gen, st_rec))) return rc;
        HIPC(hipEventRecord(ctx->ev[7], st_gen));
        // bases: generation by generation -- a generation's rows come from the counts of everything decoded before it
        ca.csz = (u32*)ctx->csz.p + nchains; ca.coff = (const u64*)ctx->coff.p + nchains;
        ca.st_buf = da.seq_stage; ca.st_bytes = tot_s; ca.st_off = da.soff; ca.st_len = da.slen;
        u32 bound[GEN_MAX_GENERATIONS + 1];
        const u32 ngen = gen_bounds(nblocks, bound);
        if (gm_on) {
            // the match model: generation by generation, each indexed behind its chains (gm.hip)
            if (ngen < 3) return fail(ctx, SFQ_E_CORRUPT, "chain index: the match model on a call of %u generations", ngen);
            if ((rc = reserve(ctx, ctx->gm_T, (size_t)8 << gm_tb))) return rc;
            HIPC(hipMemsetAsync(ctx->gm_T.p, 0xFF, (size_t)8 << gm_tb, st_gen));
            ca.geo.chain_reads = gchain_reads; ca.geo.cpb = gcpb; ca.geo.nchains = ngc;        // (the base chains' own geometry; nothing behind this reads the quality chains')
            auto gchain0_of = [&](u32 b) -> u32 { return seg_len ? seg_c0[b] : (u32)std::min<u64>((u64)b * gcpb, ngc); };
            for (u32 g = 0; g < ngen; g++) {
                launch_gm_decode_c(ca, da, gchain0_of(bound[g]), gchain0_of(bound[g + 1]), (u64)bound[g] * block_reads, (const u64*)ctx->gm_T.p, gm_tb, tot_s, st_gen);
                if (g + 1 < ngen) launch_gm_insert(ca, bound[g], bound[g + 1], (u64)(bound[g + 1] - bound[g]) * block_reads, dec_max_line, (u64*)ctx->gm_T.p, gm_tb, st_gen);
            }
        } else
        if (!gen_on || ngen < 3) { ca.flat_quads = flat_quads; ca.flat_raw = flat_raw; launch_gen_decode_c(ca, da, 0, nchains, st_gen); }
        else {
            const u64 nctx = 1ull << g_bits;
            if ((rc = reserve(ctx, ctx->gcnt, (size_t)nctx * 16))) return rc;
            if ((rc = reserve(ctx, ctx->grows, (size_t)nctx * 4 * 2))) return rc;
            HIPC(hipMemsetAsync(ctx->gcnt.p, 0, (size_t)nctx * 16, st_gen));
            GenBins gbins;
            if ((rc = gen_bins_reserve(ctx, (u32)g_bits, tot_s, st_gen, gbins))) return rc;
            ca.g_ngen = ngen;
            for (u32 g = 0; g <= ngen; g++) ca.g_bound[g] = bound[g];
            u32* rows = (u32*)ctx->grows.p;
            const u64 br = block_reads;
            for (u32 g = 0; g < ngen; g++) {
                // rows of generation g (two buffers in turn: a generation's rows are dead once it is decoded)
                // (the pass that sums generation g - 1's bins has written them, below)
                ca.g_rows[g] = g >= 2 ? rows + nctx * (g & 1) : nullptr;
                // (generations 0 and 1 both code with the initial row: one launch -- each is a sixty-fourth of the call, a launch of
                //  its own runs as long as ONE lane takes for its chain, 3 ms at 10 M reads)
                if (g == 0) { ca.g_rows[1] = nullptr; launch_gen_decode_c(ca, da, chain0_of(bound[0]), chain0_of(bound[2]), st_gen); }
                else if (g >= 2) launch_gen_decode_c(ca, da, chain0_of(bound[g]), chain0_of(bound[g + 1]), st_gen);
                if (g + 1 < ngen) launch_gen_count_binned(ca, bound[g], bound[g + 1], (u64)(bound[g + 1] - bound[g]) * br, dec_max_line, gbins, (u32*)ctx->gcnt.p,
                                                          g >= 1 ? rows + nctx * ((g + 1) & 1) : nullptr, GEN_STEP, st_gen);
            }
        }
        if (exc_rice) launch_gen_exc_decode_r(da, nblocks, st_gen);
        else for (u32 b0 = 0; b0 < nblocks; b0 += slots) { da.m.batch0 = b0; da.m.nbatch = std::min(slots, nblocks - b0); launch_gen_exc_decode_w(da, st_gen); }
    } else {
    // adaptive tables: a wavefront per block (decode_w.hip); sfq_params.kernel = 1: the lane-per-block cross-check kernels
    const bool wave_dec = p.kernel == 0;
    if ((rc = fork_streams())) return rc;
    for (u32 b0 = 0; b0 < nblocks; b0 += slots) {
        da.m.batch0 = b0; da.m.nbatch = std::min(slots, nblocks - b0);
        if (wave_dec) launch_qlt_decode_w(da, st); else launch_qlt_decode_l(da, st);
    }
    HIPC(hipEventRecord(ctx->ev[3], st));
    HIPC(hipEventRecord(ctx->ev[7], st_gen));
    for (u32 b0 = 0; b0 < nblocks; b0 += slots) {
        da.m.batch0 = b0; da.m.nbatch = std::min(slots, nblocks - b0);
        launch_fill_u32((u32*)ctx->tab.g_tab.p, (u64)da.m.nbatch << g_bits, 0x03030303u, st_gen);
        // (the wave kernel's 64-entry window assumes every block of the call has the context bits it was picked for: an index
        //  that mixes them -- no encoder writes one -- goes to the lane kernel, which follows each block's own)
        if (wave_dec && g_bits_min >= 6 && g_bits_min == g_bits) launch_gen_decode_w(da, st_gen); else launch_gen_decode_l(da, st_gen);
    }
    }
    HIPC(hipEventRecord(ctx->ev[4], st_gen));
    HIPC(hipStreamWaitEvent(st, ctx->ev[4], 0));

    // 3. headers; the staging size comes from the index when known, else grows on overflow
    ht.mark("quality + base decoders queued");
    const bool frozen_rec = frozen && rec_chains != 0;
    const u32 nstage = frozen_rec ? nsub : nblocks;            // staging slices: one per header chain / per block
    u64* hso = bump.take<u64>((size_t)nstage + 1);
    u32* hsc = bump.take<u32>(nstage);
    if (!hso || !hsc) return fail(ctx, SFQ_E_NOMEM, "decode: host scratch");
    if ((rc = reserve(ctx, ctx->hso, ((size_t)nstage + 1) * 8))) return rc;
    if ((rc = reserve(ctx, ctx->hsc, (size_t)nstage * 4))) return rc;
    for (int attempt = 0; ; attempt++) {
        u64 o = 0;
        if (frozen_rec) {
            for (u32 c = 0; c < nsub; c++) {
                const u32 b = c / rcpb, j = c - b * rcpb;
                const u32 nr = std::min<u32>(rchain_reads, h_blocks[b].n_records - std::min(h_blocks[b].n_records, j * rchain_reads));
                u64 cap = (u64)h_rhb[c] + nr + SFQ_MAX_ID_LLEN + 64;
                if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
                hso[c] = o; hsc[c] = (u32)cap; o += (cap + 15) & ~15ull;
            }
        } else
        for (u32 b = 0; b < nblocks; b++) {
            const sfq_block_info& bi = h_blocks[b];
            // (hdr_bytes is what the ENCODER saw: where the reference's restoration differs from the text -- an emptied field
            //  comes back "0", SURVEY H7 -- the headers of a format-6 block come back longer, so an overflow is retried with more)
            u64 cap = bi.hdr_bytes ? (((u64)bi.hdr_bytes + bi.n_records) << attempt) + SFQ_MAX_ID_LLEN + 64
                                   : ((u64)bi.n_records * 96 << attempt) + 2 * SFQ_MAX_ID_LLEN + 64;
            if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
            hso[b] = o; hsc[b] = (u32)cap; o += (cap + 15) & ~15ull;
        }
        hso[nstage] = o;
        // (all the headers together cannot outgrow the caller's buffer for the text: a damaged stream that keeps asking for more ends here)
        if (attempt && o > 2 * out_cap + (u64)nstage * (SFQ_MAX_ID_LLEN + 96)) return fail(ctx, SFQ_E_CORRUPT, "decode: the headers do not fit the output buffer (corrupt stream)");
        if ((rc = reserve(ctx, ctx->hdr_stage, (size_t)o + 16))) return rc;
        HIPC(hipMemcpyAsync(ctx->hso.p, hso, ((size_t)nstage + 1) * 8, hipMemcpyHostToDevice, st_rec));
        HIPC(hipMemcpyAsync(ctx->hsc.p, hsc, (size_t)nstage * 4, hipMemcpyHostToDevice, st_rec));
        if (attempt) {                                     // (the first time round they were cleared at the head of the call: hdr_marks_cleared)
            HIPC(hipMemsetAsync(ctx->hoff.p, 0xFF, (size_t)nrec * 8, st_rec));
            HIPC(hipMemsetAsync(ctx->hlen.p, 0, (size_t)nrec * 4, st_rec));
        }
        da.hdr_stage = (u8*)ctx->hdr_stage.p; da.hdr_stage_off = (const u64*)ctx->hso.p; da.hdr_stage_cap = (const u32*)ctx->hsc.p;
        if (attempt) { if ((rc = advance_epoch(ctx, nblocks))) return rc; da.m.epoch_base = ctx->epoch_base; ctx->epoch_base += nblocks; }
        if (frozen_rec) {
            ChainArgs cr; memset(&cr, 0, sizeof cr);
            cr.m = da.m; cr.rrows = (const u32*)ctx->rrows.p; cr.rdec = (const u16*)ctx->rdec.p;
            cr.rgeo.chain_reads = rchain_reads; cr.rgeo.cpb = rcpb; cr.rgeo.nchains = nsub;
            cr.csz = (u32*)ctx->csz.p + (size_t)nchains + ngc; cr.coff = (const u64*)ctx->coff.p + (size_t)nchains + ngc;
            cr.rmap = (const u16*)ctx->rmap.p; cr.rhot = cr.rmap + PR_REC_ROWS; cr.r_hot = ctx->r_hot_dec;
            u32* rflags = nullptr; u32* dtok = nullptr; u32* dtoff = nullptr; u32* dflags = nullptr; bool all_pre = false;
            if (version >= 5) {                                        // (load_pre5 archives: the general path)
                if ((rc = reserve(ctx, ctx->rflags, (size_t)nsub * 4 * 2))) return rc;                 // the lane kernels' flags, the two-step decoder's
                // A chain that restores more than 127 bytes per header on average holds a header neither the two steps nor the fast lane
                // kernel take: marked here for the general kernel (flags 1, dflags 2) -- 60 k long reads with 230-byte headers spent
                // 3.6 + 7.8 ms in kernels that found that out record by record, the longest path of their decode
                std::vector<u32> pre_v; u32* pre = nullptr; bool any_pre = false;
                if (!attempt) {
                    for (u32 c = 0; c < nsub && !any_pre; c++) any_pre = (u64)h_rhb[c] > 127ull * rchain_reads;      // (a first look: usually none)
                    if (any_pre) { pre_v.assign((size_t)nsub * 2, 0u); pre = pre_v.data(); any_pre = false; }
                }
                u32 n_pre = 0;
                if (pre) {
                    for (u32 c = 0; c < nsub; c++) {
                        const u32 b = c / rcpb, k0 = (c - b * rcpb) * rchain_reads, bn = h_blocks[b].n_records;
                        const u64 n = k0 < bn ? std::min<u64>(rchain_reads, bn - k0) : 0;
                        if (n && (u64)h_rhb[c] > 127ull * n) { pre[c] = 1; pre[nsub + c] = 2; any_pre = true; n_pre++; }
                    }
                }
                all_pre = n_pre == nsub;                   // every chain: the general kernel alone is launched (below)
                if (any_pre) { HIPC(hipMemcpyAsync(ctx->rflags.p, pre, (size_t)nsub * 8, hipMemcpyHostToDevice, st_rec)); HIPC(hipStreamSynchronize(st_rec)); }      // (pre_v is a local)
                else HIPC(hipMemsetAsync(ctx->rflags.p, 0, (size_t)nsub * 4 * 2, st_rec));
                rflags = (u32*)ctx->rflags.p; dflags = rflags + nsub;
                const u64 tb = (rec_dtok_bytes(nrec) + 15) & ~15ull;
                if ((rc = reserve(ctx, ctx->rtok, (size_t)(tb + nrec * 4 + 16)))) return rc;          // tokens, then the records' places among them
                dtok = (u32*)ctx->rtok.p; dtoff = (u32*)((u8*)ctx->rtok.p + tb);
            }
            if (all_pre) launch_rec_decode_c(cr, da, nullptr, st_rec, nullptr, nullptr, nullptr);
            else launch_rec_decode_c(cr, da, rflags, st_rec, dtok, dtoff, dflags);
        } else
        for (u32 b0 = 0; b0 < nblocks; b0 += slots) {
            da.m.batch0 = b0; da.m.nbatch = std::min(slots, nblocks - b0);
            if (p.kernel == 0) launch_rec_decode_w(da, st_rec); else launch_rec_decode_l(da, st_rec);
        }
        HIPC(hipStreamSynchronize(st_rec));
        HIPC(hipStreamSynchronize(st));
        HIPC(hipMemcpyAsync(hb, ctx->blocks.p, (size_t)nblocks * sizeof(BlockDesc), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        bool overflow = false; int worst = 0;
        for (u32 b = 0; b < nblocks; b++) {
            if (hb[b].status == (u32)(-SFQ_E_OVERFLOW) && !frozen_rec) overflow = true;
            else if (hb[b].status) worst = std::max<int>(worst, (int)hb[b].status);
        }
        if (worst) return fail(ctx, -worst, "decode: block kernel reported error %d (corrupt or truncated stream)", -worst);
        if (!overflow) break;
        if (attempt >= 8) return fail(ctx, SFQ_E_OVERFLOW, "decode: header staging overflow");
        for (u32 b = 0; b < nblocks; b++) hb[b].status = 0;
        HIPC(hipMemcpyAsync(ctx->blocks.p, hb, (size_t)nblocks * sizeof(BlockDesc), hipMemcpyHostToDevice, st));
    }
    ht.mark("decoders through (host waited)");
    HIPC(hipEventRecord(ctx->ev[5], st));

    // 4. lay the records out
    launch_record_sizes(da, nrec, (u32*)ctx->rsize.p, st);
    u64 total = 0;
    const u64* d_roff = (const u64*)ctx->roff.p;
    if (!n_over) {
        launch_scan_u32((const u32*)ctx->rsize.p, (u64*)ctx->roff.p, nrec, (u64*)ctx->scan_tmp.p, st);
        HIPC(hipMemcpyAsync(&total, (u64*)ctx->roff.p + nrec, 8, hipMemcpyDeviceToHost, st));
    } else {                                               // every record of the file in file order: the kept ones' sizes, the oversize ones' raw lines
        HIPC(hipStreamWaitEvent(st, ctx->ev[9], 0));
        if ((rc = reserve(ctx, ctx->osize_all, (size_t)nrec_file * 4))) return rc;
        if ((rc = reserve(ctx, ctx->oroff_all, ((size_t)nrec_file + 1) * 8))) return rc;
        if ((rc = reserve(ctx, ctx->oroff_k, (size_t)nrec * 8 + 16))) return rc;
        launch_over_sizes((const u32*)ctx->orecmap.p, (const u32*)ctx->rsize.p, nrec, (const u64*)ctx->ono.p, (const u64*)ctx->opiece.p, n_over, (u32*)ctx->osize_all.p, st);
        launch_scan_u32((const u32*)ctx->osize_all.p, (u64*)ctx->oroff_all.p, nrec_file, (u64*)ctx->scan_tmp.p, st);
        launch_gather_u64((const u64*)ctx->oroff_all.p, (const u32*)ctx->orecmap.p, nrec, (u64*)ctx->oroff_k.p, st);
        HIPC(hipMemcpyAsync(&total, (u64*)ctx->oroff_all.p + nrec_file, 8, hipMemcpyDeviceToHost, st));
        d_roff = (const u64*)ctx->oroff_k.p;
    }
    u32 h_st_end = 0;
    if (n_over) HIPC(hipMemcpyAsync(&h_st_end, &((BlockDesc*)ctx->blocks.p)[0].status, 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (h_st_end) return fail(ctx, SFQ_E_CORRUPT, "damaged oversize streams (usr.lgen / usr.lqlt)");
    *out_bytes = total;
    if (total > out_cap) return fail(ctx, SFQ_E_OVERFLOW, "decoded text needs %llu bytes, caller gave %llu", (unsigned long long)total, (unsigned long long)out_cap);
    ht.mark("sizes known");
    launch_assemble(da, nrec, d_roff, d_out, st);
    if (n_over) launch_over_place(n_over, (const u64*)ctx->ono.p, (const u64*)ctx->opiece.p, (const u8*)ctx->otxt[0].p, (const u8*)ctx->otxt[1].p, (const u8*)ctx->otxt[2].p,
                                  (const u64*)ctx->oroff_all.p, d_out, st);
    HIPC(hipEventRecord(ctx->ev[6], st));
    HIPC(hipStreamSynchronize(st));
    ht.mark("assembled");
    res->n_records = nrec_file; res->n_blocks = nblocks; res->total_bytes = total;
    res->kernel_ms[SFQ_T_USR] = ev_ms(ctx->ev[0], ctx->ev[1]);
    res->kernel_ms[SFQ_T_QLT] = ev_ms(ctx->ev[2], ctx->ev[3]);
    res->kernel_ms[SFQ_T_GEN] = ev_ms(ctx->ev[7], ctx->ev[4]);
    res->kernel_ms[SFQ_T_REC] = ev_ms(ctx->ev[2], ctx->ev[5]);          // the chains overlap: these do not add up
    res->kernel_ms[SFQ_T_PACK] = ev_ms(ctx->ev[5], ctx->ev[6]);
    res->kernel_ms[SFQ_T_TOTAL] = ev_ms(ctx->ev[0], ctx->ev[6]);
    return SFQ_OK;
}

int sfq_decode_blocks_host(sfq_ctx* ctx, const sfq_params* params, const sfq_block_info* h_blocks, uint32_t n_blocks,
                           const uint8_t* h_first_hdrs, uint64_t first_hdr_bytes,
                           const uint8_t* h_streams, uint64_t streams_bytes, const uint64_t stream_offset[SFQ_NSTREAMS],
                           uint8_t* h_out, uint64_t out_cap, uint64_t* out_bytes, sfq_result* result) {
    if (!ctx || !h_streams || !h_out || !h_blocks || !stream_offset) return fail(ctx, SFQ_E_ARG, "null argument");
    HIPC(hipSetDevice(ctx->dev));
    int rc;
    // the index is untrusted: every stream's blocks must lie inside the bytes handed over (the device entry point cannot
    // know the size of the caller's buffer; there the caller vouches for stream_offset[s] + the blocks' sizes)
    for (int sx = 0; sx < SFQ_NSTREAMS; sx++) {
        u64 tot = 0;
        for (u32 b = 0; b < n_blocks; b++) tot += h_blocks[b].size[sx];
        if (stream_offset[sx] > streams_bytes || tot > streams_bytes - stream_offset[sx])
            return fail(ctx, SFQ_E_CORRUPT, "block index: stream %s needs %llu bytes at offset %llu, %llu were given", sfq_stream_name(sx),
                        (unsigned long long)tot, (unsigned long long)stream_offset[sx], (unsigned long long)streams_bytes);
    }
    if ((rc = reserve(ctx, ctx->in_stage, (size_t)streams_bytes + 16))) return rc;
    if ((rc = reserve(ctx, ctx->out_stage, (size_t)out_cap + 16))) return rc;
    HIPC(hipMemcpyAsync(ctx->in_stage.p, h_streams, (size_t)streams_bytes, hipMemcpyHostToDevice, ctx->st));
    rc = sfq_decode_blocks(ctx, params, h_blocks, n_blocks, h_first_hdrs, first_hdr_bytes, (const u8*)ctx->in_stage.p,
                           stream_offset, (u8*)ctx->out_stage.p, out_cap, out_bytes, result);
    if (rc) return rc;
    HIPC(hipMemcpyAsync(h_out, ctx->out_stage.p, (size_t)*out_bytes, hipMemcpyDeviceToHost, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
    return SFQ_OK;
}

}  // extern "C"
