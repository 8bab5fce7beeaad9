/*
 * sfq_oracle.c -- TEST INFRASTRUCTURE ONLY (see sfq_oracle.h).
 *
 * CPU restatement, in plain C, of slimfastq's format-6 algorithm.  Every function cites the
 * reference file:line it follows.  The code is organised bottom-up like the reference's layers:
 *   L1 container (filer.cpp)  ->  L2 coder + rangers (coder.hpp, *_ranger.hpp)  ->  L2x xfile
 *   ->  L3 models (qlts/gens/recs)  ->  L4 framing (usrs.cpp)  ->  info page (config.cpp).
 *
 * Parity status: PINNED -- checked against oracle/_ref/slimfastq_ref (the compiled reference) on
 * all 16 reference samples x 4 levels, both directions (tests/test_oracle_vs_ref.py).
 */
#define _GNU_SOURCE
#include "sfq_oracle.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------------
 * errors: the reference croak()s and exits (config.cpp:54-68); the oracle records and unwinds.
 * ---------------------------------------------------------------------------------------------- */
static __thread char g_err[512];
static __thread int  g_failed;

static void fail(const char* fmt, ...) {
    if (g_failed) return;
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_failed = 1;
}
const char* sfqo_last_error(void) { return g_err; }
void sfqo_free(void* p) { free(p); }

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s ? s : 1); if (!p) abort(); return p; }
static void* xrealloc(void* q, size_t n) { void* p = realloc(q, n ? n : 1); if (!p) abort(); return p; }

/* ================================================================================================
 * L1  container  (filer.hpp / filer.cpp)
 * ============================================================================================== */
#define PAGE 0x2000                         /* filer.hpp:34  */
#define MAXI_NODES (PAGE / 4 - 1)           /* filer.hpp:39  = 2047 page ids per node page */
#define MAX_ROOT_FILES (PAGE / 24)          /* filer.cpp:49  = 341 */

typedef struct {                            /* filer.cpp:42-47, packed 24 bytes */
    u64 name; u64 size; u32 first; u32 node;
} __attribute__((packed)) fattr;

typedef struct { char* key; char* val; } kv;

struct sfqo_archive {
    u8*    img;  size_t cap_pages;          /* page image */
    u64    num_pages;                       /* filer.cpp:54 allocator / image size in pages */
    fattr  files[MAX_ROOT_FILES + 1];       /* filer.cpp:52 */
    u64    next_findex;                     /* filer.cpp:53 */
    kv*    info; size_t n_info;             /* parsed info page, insertion order (config.cpp:47-49) */
};

static void img_reserve(sfqo_archive* a, u64 pages) {
    if (pages <= a->cap_pages) return;
    size_t ncap = a->cap_pages ? a->cap_pages : 16;
    while (ncap < pages) ncap *= 2;
    a->img = xrealloc(a->img, ncap * PAGE);
    memset(a->img + a->cap_pages * PAGE, 0, (ncap - a->cap_pages) * PAGE);
    a->cap_pages = ncap;
}
static void write_page(sfqo_archive* a, u32 idx, const void* page) {  /* filer.cpp:80-87 */
    img_reserve(a, (u64)idx + 1);
    memcpy(a->img + (size_t)idx * PAGE, page, PAGE);
}
static u32 allocate(sfqo_archive* a) { return (u32)(a->num_pages++); }  /* filer.cpp:68-70 */

static u64 name2u(const char* name) {       /* filer.cpp:149-154 */
    u64 u = 0; size_t n = strlen(name); memcpy(&u, name, n < 8 ? n : 8); return u;
}

/* ---- byte sink: a FilerSave (paged, filer.cpp:185-242) or a plain vector ---- */
typedef struct wr {
    sfqo_archive* a;                        /* NULL -> plain vector mode */
    u8* data; size_t n, cap;                /* plain */
    u8  buf[PAGE + 10]; size_t cur;         /* filer.hpp:45-46 */
    u64 page_count;
    u32 node[MAXI_NODES + 1];
    u32 node_i, node_p, onef_i;
    int valid;
} wr;

static wr* wr_new_plain(void) {
    wr* w = xcalloc(1, sizeof *w); w->valid = 1; return w;
}
static wr* wr_new_named(sfqo_archive* a, const char* name) {   /* FilerSave::FilerSave(name) filer.cpp:185-191 */
    wr* w = xcalloc(1, sizeof *w);
    w->a = a; w->valid = 1;
    if (a->next_findex >= MAX_ROOT_FILES) { fail("Internal error: Too many open files"); }
    u32 fi = w->onef_i = (u32)(a->next_findex++);
    a->files[fi].name = name2u(name);
    a->files[fi].size = 0;
    a->files[fi].node = 0;
    a->files[fi].first = allocate(a);
    return w;
}
static wr* wr_new_info(sfqo_archive* a) {   /* FilerSave::FilerSave(42) filer.cpp:193-198 */
    wr* w = xcalloc(1, sizeof *w);
    w->a = a; w->valid = 1;
    memset(&a->files[0], 0, sizeof a->files[0]);
    w->onef_i = 0;
    return w;
}
static void wr_save_node(wr* w, u32 next_node) {   /* filer.cpp:208-215 */
    w->node[w->node_i] = next_node;
    write_page(w->a, w->node_p, w->node);
    w->node_p = next_node;
    w->node_i = 0;
}
static void wr_save_page(wr* w, int finit) {       /* filer.cpp:217-242 */
    if (!w->valid || !w->cur) return;
    sfqo_archive* a = w->a;
    if (!w->node_p) {
        write_page(a, a->files[w->onef_i].first, w->buf);
        if (!finit) a->files[w->onef_i].node = w->node_p = allocate(a);
    } else {
        write_page(a, w->node[w->node_i++], w->buf);
        if (w->node_i == MAXI_NODES && !finit) wr_save_node(w, allocate(a));
    }
    if (!finit) w->node[w->node_i] = allocate(a);
    a->files[w->onef_i].size += w->cur;
    w->cur = 0;
    w->page_count++;
    /* like the reference, the page buffer is not cleared: the tail of a stream's last page repeats
       stale bytes of its previous page (page tails are not part of the parity contract) */
}
static inline void wr_put(wr* w, u8 c) {           /* FilerSave::put filer.hpp:70-75 */
    if (!w->a) {
        if (w->n == w->cap) { w->cap = w->cap ? w->cap * 2 : 4096; w->data = xrealloc(w->data, w->cap); }
        w->data[w->n++] = c;
        return;
    }
    if (w->cur >= PAGE) wr_save_page(w, 0);
    w->buf[w->cur++] = c;
}
static size_t wr_tell(const wr* w) {               /* FilerBase::tell filer.cpp:166-173 */
    if (!w->a) return w->n;
    return w->page_count ? (size_t)((w->page_count - 1) * PAGE + w->cur) : w->cur;
}
static void wr_close(wr* w) {                      /* FilerSave::~FilerSave filer.cpp:200-206 */
    if (!w) return;
    if (w->a) {
        wr_save_page(w, 1);
        w->valid = 0;
        if (w->node_p) wr_save_node(w, 0);
    }
    free(w->data);
    free(w);
}

/* ---- byte source: FilerLoad::get returns 0 past EOF and clears *valid (filer.hpp:94-97, filer.cpp:273-280) ---- */
typedef struct rd { u8* data; size_t n, pos; int valid; } rd;

static inline u8 rd_get(rd* r) {
    if (!r->valid) return 0;
    if (r->pos >= r->n) { r->valid = 0; return 0; }
    return r->data[r->pos++];
}

int sfqo_nstreams(const sfqo_archive* a) { return (int)a->next_findex; }
const char* sfqo_stream_name(const sfqo_archive* a, int i) {
    static __thread char nm[9];
    if (i < 0 || (u64)i >= a->next_findex) return "";
    if (i == 0) return "<info>";
    memcpy(nm, &a->files[i].name, 8); nm[8] = 0;
    return nm;
}
size_t sfqo_stream_size(const sfqo_archive* a, int i) {
    if (i < 0 || (u64)i >= a->next_findex) return 0;
    return (size_t)a->files[i].size;
}
int sfqo_stream_find(const sfqo_archive* a, const char* name) {   /* OneFile::get_findex(name) filer.cpp:61-67 */
    u64 u = name2u(name);
    for (u64 i = 1; i < a->next_findex; i++) if (a->files[i].name == u) return (int)i;
    return -1;
}
static const u8* page_ptr(const sfqo_archive* a, u32 idx) {
    if ((u64)idx >= a->num_pages) return NULL;
    return a->img + (size_t)idx * PAGE;
}
/* FilerLoad::load_page chain walk, filer.cpp:273-303 */
long long sfqo_stream_read(const sfqo_archive* a, int i, u8* out, size_t cap) {
    if (i < 0 || (u64)i >= a->next_findex) return -1;
    u64 size = a->files[i].size;
    if (cap < size) return -1;
    u64 done = 0; u32 node_p = 0, node_i = 0; const u32* node = NULL; int first = 1;
    while (done < size) {
        const u8* pg;
        if (first) {
            pg = page_ptr(a, a->files[i].first);
            node_p = a->files[i].node;
            if (node_p) node = (const u32*)page_ptr(a, node_p);
            node_i = 0; first = 0;
        } else {
            if (!node) return -1;
            if (node_i == MAXI_NODES) {
                node_p = node[MAXI_NODES];
                node = (const u32*)page_ptr(a, node_p);
                if (!node) return -1;
                node_i = 0;
            }
            pg = page_ptr(a, node[node_i++]);
        }
        if (!pg) return -1;
        u64 take = size - done < PAGE ? size - done : PAGE;
        memcpy(out + done, pg, take);
        done += take;
    }
    return (long long)size;
}
static rd rd_open(const sfqo_archive* a, const char* name) {      /* FilerLoad::FilerLoad(name) filer.cpp:246-256 */
    rd r; memset(&r, 0, sizeof r);
    int i = sfqo_stream_find(a, name);
    if (i <= 0) return r;
    r.n = a->files[i].size;
    r.data = xmalloc(r.n);
    if (sfqo_stream_read(a, i, r.data, r.n) < 0) { free(r.data); r.data = NULL; r.n = 0; return r; }
    r.valid = r.n > 0;
    return r;
}
static void rd_close(rd* r) { free(r->data); r->data = NULL; }

/* ---- info page (config.cpp:87-159) ---- */
static void info_insert(sfqo_archive* a, const char* key, const char* val) {  /* std::map::insert keeps the first */
    for (size_t i = 0; i < a->n_info; i++) if (!strcmp(a->info[i].key, key)) return;
    a->info = xrealloc(a->info, (a->n_info + 1) * sizeof(kv));
    a->info[a->n_info].key = strdup(key);
    a->info[a->n_info].val = strdup(val);
    a->n_info++;
}
const char* sfqo_info_get(const sfqo_archive* a, const char* key) {           /* config.cpp:113-120 */
    for (size_t i = 0; i < a->n_info; i++) if (!strcmp(a->info[i].key, key)) return a->info[i].val;
    return "";
}
static long long info_long(const sfqo_archive* a, const char* key, long long dflt) {  /* config.cpp:127-130 */
    const char* s = sfqo_info_get(a, key);
    return *s ? atoll(s) : dflt;
}
static int info_bool(const sfqo_archive* a, const char* key) {                /* config.cpp:122-125 */
    const char* s = sfqo_info_get(a, key);
    return *s && *s != '0';
}
static void info_put_str(wr* w, const char* s) {                              /* config.cpp:132-138 */
    int sanity = 0x200;
    while (*s && --sanity) wr_put(w, (u8)*s++);
    if (!sanity) fail("oversize string value");
}
static void set_info(sfqo_archive* a, wr* w, const char* key, const char* val) {  /* config.cpp:140-148 */
    info_put_str(w, key); wr_put(w, '='); info_put_str(w, val); wr_put(w, '\n');
    info_insert(a, key, val);
}
static void set_info_ll(sfqo_archive* a, wr* w, const char* key, long long num) { /* config.cpp:150-154 */
    char b[40]; sprintf(b, "%lld", num); set_info(a, w, key, b);
}
static void load_info(sfqo_archive* a) {                                      /* config.cpp:87-107 */
    size_t n = a->files[0].size;
    u8* txt = xmalloc(n + 1);
    /* stream 0 = page 0 (+ chain); files[0].first was repurposed as the entry count (filer.cpp:95-96,123) */
    if (sfqo_stream_read(a, 0, txt, n) < 0) { free(txt); return; }
    rd r = { txt, n, 0, n > 0 };
    char line[0x200];
    while (r.valid) {
        for (int i = 0; i < 0x200; i++) {
            line[i] = (char)rd_get(&r);
            if (!r.valid || line[i] == '\n') line[i] = 0;
            if (line[i] == 0) break;
        }
        line[0x1ff] = 0;
        char* pos = strchr(line, '=');
        if (pos) { *pos = 0; info_insert(a, line, pos + 1); }
    }
    free(txt);
}

static sfqo_archive* archive_new_write(void) {                                /* OneFile::init_write filer.cpp:112-120 */
    sfqo_archive* a = xcalloc(1, sizeof *a);
    a->next_findex = 1;
    a->num_pages = 2;
    img_reserve(a, 2);
    return a;
}
static void archive_finit_write(sfqo_archive* a) {                            /* OneFile::finit_write filer.cpp:121-128 */
    a->files[0].first = (u32)a->next_findex;
    img_reserve(a, a->num_pages);
    memcpy(a->img + PAGE, a->files, PAGE);
    a->files[0].first = 0;
}
sfqo_archive* sfqo_archive_from_image(const u8* image, size_t n) {            /* OneFile::init_read filer.cpp:88-97 */
    if (n < 2 * PAGE) { fail("container too small"); return NULL; }
    sfqo_archive* a = xcalloc(1, sizeof *a);
    a->num_pages = n / PAGE;
    img_reserve(a, a->num_pages);
    memcpy(a->img, image, a->num_pages * PAGE);
    memcpy(a->files, a->img + PAGE, PAGE);
    a->next_findex = a->files[0].first;
    a->files[0].first = 0;
    if (a->next_findex > MAX_ROOT_FILES) { fail("bad directory"); sfqo_archive_free(a); return NULL; }
    load_info(a);
    return a;
}
sfqo_archive* sfqo_archive_read(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fail("cannot read %s", path); return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    u8* b = xmalloc((size_t)n);
    size_t got = fread(b, 1, (size_t)n, f); fclose(f);
    sfqo_archive* a = got == (size_t)n ? sfqo_archive_from_image(b, (size_t)n) : NULL;
    free(b);
    return a;
}
const u8* sfqo_archive_image(const sfqo_archive* a, size_t* n) { *n = (size_t)a->num_pages * PAGE; return a->img; }
int sfqo_archive_write(const sfqo_archive* a, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    size_t n = (size_t)a->num_pages * PAGE;
    int ok = fwrite(a->img, 1, n, f) == n;
    fclose(f);
    return ok ? 0 : -1;
}
void sfqo_archive_free(sfqo_archive* a) {
    if (!a) return;
    for (size_t i = 0; i < a->n_info; i++) { free(a->info[i].key); free(a->info[i].val); }
    free(a->info); free(a->img); free(a);
}

/* ================================================================================================
 * L2  range coder  (coder.hpp)
 * ============================================================================================== */
#define TOP (1ULL << 24)                    /* coder.hpp:24 */

typedef struct { u64 low, code; u32 range; wr* out; rd* in; } rcoder;

static void rc_init_save(rcoder* c, wr* out) { c->out = out; c->in = NULL; c->low = 0; c->range = (u32)-1; c->code = 0; }  /* coder.hpp:34-39 */
static void rc_init_load(rcoder* c, rd* in) {                                 /* coder.hpp:41-49 */
    c->in = in; c->out = NULL; c->low = 0; c->range = (u32)-1; c->code = 0;
    for (int i = 0; i < 8; i++) c->code = (c->code << 8) | rd_get(in);
}
static void rc_done(rcoder* c) {                                              /* coder.hpp:52-61 */
    if (c->out) {
        for (int i = 0; i < 8; i++) { wr_put(c->out, (u8)(c->low >> 56)); c->low <<= 8; }
        c->out = NULL;
    }
}
static inline void rc_encode(rcoder* c, u32 cum, u32 freq, u32 tot) {         /* coder.hpp:66-81 */
    c->range /= tot;
    c->low   += (u32)(cum * c->range);      /* UINT32*UINT32 -> UINT32, then widened */
    c->range *= freq;
    int guard = 0;
    while (c->range < TOP) {
        if ((c->low ^ (c->low + c->range)) & (0xffULL << 56))
            c->range = (((u32)c->low | (u32)(TOP - 1)) - (u32)c->low);
        wr_put(c->out, (u8)(c->low >> 56));
        c->range <<= 8;
        c->low   <<= 8;
        /* the reference loops forever if the clamp yields range 0 (probability ~2^-24 per clamp) */
        if (++guard > 64) { fail("coder stuck (range clamped to 0)"); return; }
    }
}
static inline u32 rc_get_freq(rcoder* c, u32 tot) {                           /* coder.hpp:83-86 */
    c->range /= tot;
    if (c->range == 0) { fail("corrupt stream (range 0)"); c->range = 1; }
    return (u32)(c->code / c->range);
}
static inline void rc_decode(rcoder* c, u32 cum, u32 freq, u32 tot) {         /* coder.hpp:88-102 */
    (void)tot;
    u32 temp = cum * c->range;
    c->low  += temp;
    c->code -= temp;
    c->range *= freq;
    int guard = 0;
    while (c->range < TOP) {
        if ((c->low ^ (c->low + c->range)) & (0xffULL << 56))
            c->range = (((u32)c->low | (u32)(TOP - 1)) - (u32)c->low);
        c->code <<= 8;
        c->code |= rd_get(c->in);
        c->range <<= 8;
        c->low   <<= 8;
        if (++guard > 64) { fail("corrupt stream (coder stuck)"); return; }
    }
}

/* ---- Base2Ranger (base2_ranger.hpp) ---- */
typedef union { u8 f[4]; u32 v; } base2;
#define B2_INIT 0x03030303u                 /* base2_ranger.hpp:39,70 */

static inline void b2_update(base2* r, int sym) {                             /* base2_ranger.hpp:60-66, 48-53 */
    if (r->f[sym] > 254) r->v = ((r->v & ~0x01010101u) >> 1) | (r->v & 0x01010101u);
    r->f[sym]++;
}
static inline void b2_put(base2* r, rcoder* c, u8 sym) {                      /* base2_ranger.hpp:74-84 */
    u16 total = (u16)((r->f[0] + r->f[1]) + (r->f[2] + r->f[3]));
    u16 offs = 0;
    switch (sym) {
    case 3: offs += r->f[2]; /* fallthrough */
    case 2: offs += r->f[1]; /* fallthrough */
    case 1: offs += r->f[0];
    }
    rc_encode(c, offs, r->f[sym], total);
    b2_update(r, sym);
}
static inline u8 b2_get(base2* r, rcoder* c) {                                /* base2_ranger.hpp:86-104 */
    u16 total = (u16)((r->f[0] + r->f[1]) + (r->f[2] + r->f[3]));
    u32 prob = rc_get_freq(c, total);
    u32 sumf = 0; int i;
    for (i = 0; i < 4; i++) {
        if (sumf + r->f[i] <= prob) sumf += r->f[i]; else break;
    }
    if (i >= 4) { fail("corrupt gen stream"); i = 3; sumf -= r->f[3]; }      /* reference: assert(i<4) */
    rc_decode(c, sumf, r->f[i], total);
    b2_update(r, i);
    return (u8)i;
}

/* ---- Log64Ranger (log64_ranger.hpp) ---- */
typedef struct { u16 freq[64]; u16 iend; u32 total; u8 count; u8 syms[64]; } log64;   /* all-zero initial state, :89-96 */
#define L64_STEP 6
#define L64_NSYM 64
#define L64_MAXF ((1 << 16) - 64)

static inline u8 l64_update(log64* r, int i) {                                /* log64_ranger.hpp:69-87 */
    if (r->freq[i] > (L64_MAXF - L64_STEP)) {
        if (i == 0 && r->freq[i] + 20U > r->total) return r->syms[i];
        u32 t = 0;                                                            /* normalize :51-54 */
        for (u32 k = 0; k < r->iend; k++) t += (r->freq[k] /= 2);
        r->total = t;
    }
    r->freq[i] += L64_STEP;
    r->total   += L64_STEP;
    if (i == 0 || (++r->count & 0xf) || r->freq[i] <= r->freq[i - 1]) return r->syms[i];
    u8 c = r->syms[i]; r->syms[i] = r->syms[i - 1]; r->syms[i - 1] = c;      /* down_level :56-67 */
    u16 f = r->freq[i]; r->freq[i] = r->freq[i - 1]; r->freq[i - 1] = f;
    return c;
}
static inline void l64_put(log64* r, rcoder* c, u8 sym) {                     /* log64_ranger.hpp:98-112 */
    u32 sumf = 0, i = 0;
    if (r->iend <= sym) for (; r->iend <= sym; r->iend++) r->syms[r->iend] = (u8)r->iend;
    for (; r->syms[i] != sym; sumf += r->freq[i++]);
    rc_encode(c, sumf + i, r->freq[i] + 1, r->total + L64_NSYM);
    l64_update(r, (int)i);
}
static inline u16 l64_get(log64* r, rcoder* c) {                              /* log64_ranger.hpp:114-138 */
    u32 vtot = r->total + L64_NSYM, sumf = 0, i;
    u32 prob = rc_get_freq(c, vtot);
    for (i = 0; i < L64_NSYM; i++) {
        if (r->iend == i) r->syms[r->iend++] = (u8)i;
        if (sumf + r->freq[i] + 1 <= prob) sumf += r->freq[i] + 1; else break;
    }
    if (i >= L64_NSYM) { fail("corrupt qlt stream"); i = L64_NSYM - 1; sumf -= r->freq[i] + 1; }
    rc_decode(c, sumf, r->freq[i] + 1, vtot);
    return l64_update(r, (int)i);
}

/* ---- PowerRanger (power_ranger.hpp:36-131) ---- */
typedef struct { u32 total; u16 freq[256]; u16 iend; u8 count; u8 syms[256]; } power;  /* zeroed, :87-89 */
#define PW_STEP 14
#define PW_NSYM 256
#define PW_MAXF ((1 << 15) - 32)

static inline u8 pw_update(power* r, int i) {                                 /* power_ranger.hpp:66-84 */
    if (r->freq[i] > (PW_MAXF - PW_STEP)) {
        if (i == 0 && r->freq[i] + 256U > r->total) return r->syms[i];
        u32 t = 0;                                                            /* normalize :49-52 */
        for (u32 k = 0; k < r->iend; k++) t += (r->freq[k] >>= 1);
        r->total = t;
    }
    r->freq[i] += PW_STEP;
    r->total   += PW_STEP;
    if (i == 0 || (++r->count & 0xf) || r->freq[i] <= r->freq[i - 1]) return r->syms[i];
    u8 t = r->syms[i]; r->syms[i] = r->syms[i - 1]; r->syms[i - 1] = t;      /* down_level :54-64 */
    u16 f = r->freq[i]; r->freq[i] = r->freq[i - 1]; r->freq[i - 1] = f;
    return t;
}
static void pw_put(power* r, rcoder* c, u8 sym) {                             /* power_ranger.hpp:91-104 */
    u32 sumf = 0, i = 0;
    if (r->iend <= sym) for (; r->iend <= sym; r->iend++) r->syms[r->iend] = (u8)r->iend;
    for (; r->syms[i] != sym; sumf += r->freq[i++]);
    rc_encode(c, sumf + i, r->freq[i] + 1, r->total + PW_NSYM);
    pw_update(r, (int)i);
}
static u16 pw_get(power* r, rcoder* c) {                                      /* power_ranger.hpp:106-130 */
    u32 vtot = r->total + PW_NSYM, sumf = 0, i;
    u32 prob = rc_get_freq(c, vtot);
    for (i = 0; i < PW_NSYM; i++) {
        if (r->iend == i) r->syms[r->iend++] = (u8)i;
        if (sumf + r->freq[i] + 1 <= prob) sumf += r->freq[i] + 1; else break;
    }
    if (i >= PW_NSYM) { fail("corrupt stream"); i = PW_NSYM - 1; sumf -= r->freq[i] + 1; }
    rc_decode(c, sumf, r->freq[i] + 1, vtot);
    return pw_update(r, (int)i);
}

/* ---- PowerRangerU (power_ranger.hpp:133-192) ---- */
typedef struct { power p[14]; } poweru;

static int pwu_put(poweru* u, rcoder* c, u64 num) {                           /* power_ranger.hpp:138-163 */
    if (num <= 0x7f) { pw_put(&u->p[0], c, (u8)(0xff & num)); return 0; }
    if (num < 0x7ffe) {
        pw_put(&u->p[0], c, (u8)(0xff & (0x80 | (num >> 8))));
        pw_put(&u->p[1], c, (u8)(0xff & num));
        return 0;
    }
    pw_put(&u->p[0], c, 0xff);
    if (num < 1ULL << 32) {
        pw_put(&u->p[1], c, 0xfe);
        for (int shift = 0, i = 2; shift < 32; shift += 8, i++) pw_put(&u->p[i], c, (u8)(0xff & (num >> shift)));
        return 1;
    }
    pw_put(&u->p[1], c, 0xff);
    for (int shift = 0, i = 6; shift < 64; shift += 8, i++) pw_put(&u->p[i], c, (u8)(0xff & (num >> shift)));
    return 1;
}
static u64 pwu_get(poweru* u, rcoder* c) {                                    /* power_ranger.hpp:165-190 */
    u64 num = pw_get(&u->p[0], c);
    if (num > 0x7f) {
        num <<= 8;
        num |= pw_get(&u->p[1], c);
        if (num < 0xfffe) num &= 0x7fff;
        else if (num == 0xfffe) {
            num = 0;
            for (int shift = 0, i = 2; shift < 32; shift += 8, i++) { u64 ch = pw_get(&u->p[i], c); num |= ch << shift; }
        } else {
            num = 0;
            for (int shift = 0, i = 6; shift < 64; shift += 8, i++) { u64 ch = pw_get(&u->p[i], c); num |= ch << shift; }
        }
    }
    return num;
}

/* ================================================================================================
 * L2x  exception side streams  (xfile.hpp / xfile.cpp)
 * ============================================================================================== */
typedef struct {
    sfqo_archive* a;            /* where a lazily created stream goes (NULL -> plain) */
    const char* name;
    wr* filer;                  /* created on first put, xfile.cpp:60-64 */
    rcoder rc;
    poweru ranger;
    power  ranger_str;
} xsave;

static xsave* xs_new(sfqo_archive* a, const char* name) {
    xsave* x = xcalloc(1, sizeof *x); x->a = a; x->name = name; return x;
}
static void xs_init(xsave* x) {                                               /* xfile.cpp:60-64 */
    x->filer = x->a ? wr_new_named(x->a, x->name) : wr_new_plain();
    rc_init_save(&x->rc, x->filer);
}
static int xs_put(xsave* x, u64 gap) { if (!x->filer) xs_init(x); return pwu_put(&x->ranger, &x->rc, gap); }  /* :66-69 */
static void xs_put_chr(xsave* x, u8 ch) { if (!x->filer) xs_init(x); pw_put(&x->ranger_str, &x->rc, ch); }   /* :71-74 */
static void xs_put_str(xsave* x, const u8* p, size_t len) {                   /* xfile.cpp:95-99 */
    xs_put(x, len);
    for (u32 j = 0; j < len; j++) pw_put(&x->ranger_str, &x->rc, p[j]);
}
static size_t xs_tell(const xsave* x) { return x->filer ? wr_tell(x->filer) : 0; }  /* xfile.cpp:108-110 */
/* XFileSave::~XFileSave xfile.cpp:40-47.  In plain mode the bytes are handed to the caller. */
static void xs_close(xsave* x, u8** out, size_t* out_len) {
    if (!x) return;
    if (x->filer) {
        xs_put(x, 0);
        rc_done(&x->rc);
        if (out) { *out = x->filer->data; *out_len = x->filer->n; x->filer->data = NULL; }
        wr_close(x->filer);
    } else if (out) { *out = NULL; *out_len = 0; }
    free(x);
}

typedef struct { rd r; int opened, valid; rcoder rc; poweru ranger; power ranger_str; } xload;

static xload* xl_new(const sfqo_archive* a, const char* name) {               /* XFileLoad::init xfile.cpp:81-88 (eager here: same result) */
    xload* x = xcalloc(1, sizeof *x);
    x->r = rd_open(a, name);
    x->valid = x->r.valid;
    if (x->valid) rc_init_load(&x->rc, &x->r);
    return x;
}
static u64 xl_get(xload* x) { return x->valid ? pwu_get(&x->ranger, &x->rc) : 0; }          /* xfile.cpp:90-93 */
static u8  xl_get_chr(xload* x) { return x->valid ? (u8)pw_get(&x->ranger_str, &x->rc) : 0; } /* xfile.cpp:76-79 */
static u8* xl_get_str(xload* x, u8* p) {                                      /* xfile.cpp:101-106 */
    size_t len = (size_t)xl_get(x);
    for (u32 j = 0; j < len; j++) p[j] = (u8)pw_get(&x->ranger_str, &x->rc);
    return p + len;
}
static void xl_close(xload* x) { if (!x) return; rd_close(&x->r); free(x); }

/* ================================================================================================
 * L3  quality model  (qlts.hpp / qlts.cpp)
 * ============================================================================================== */
#define LAST_QLT 63                                                           /* log64_ranger.hpp:34 */

typedef struct {
    log64* ranger; size_t cnt;                                                /* qlts.cpp:34-39 */
    power  exranger;                                                          /* qlts.hpp:45 */
    rcoder rc;
    int level;
    u32 extra_hi;
} qltm;

static void qlt_alloc(qltm* q, int level) {
    memset(q, 0, sizeof *q);
    q->level = level;
    q->cnt = level == 1 ? (1u << 12) : (1u << 16);                            /* qlts.hpp:36-40, qlts.cpp:34-39 */
    q->ranger = xcalloc(q->cnt, sizeof(log64));
}
static inline u32 calc_last_delta(u32* delta, u8 q, u8 q1, u8 q2) {           /* qlts.hpp:62-74 */
    if (q1 > q) *delta += (u32)(q1 - q);
    return ( (u32)q
           | ((u32)(q1 < q2 ? q2 : q1) << 6)
           | ((u32)(q1 == q2) << 12)
           | ((7 > (*delta >> 3) ? (*delta >> 3) : 7) << 13)
           ) & 0xFFFF;
}
static inline void qlt_put_sym(qltm* m, u32 last, u8 b) {                     /* qlts.cpp:79-86 (same in all three loops) */
    if (b < LAST_QLT) l64_put(&m->ranger[last], &m->rc, b);
    else {
        l64_put(&m->ranger[last], &m->rc, LAST_QLT);
        pw_put(&m->exranger, &m->rc, b);
        m->extra_hi++;
    }
}
static void qlt_save(qltm* m, const u8* buf, size_t size) {                   /* qlts.hpp:82-90 dispatch */
    if (m->level <= 2) {                                                      /* save_1 / save_2, qlts.cpp:74-106 */
        u32 mask = m->level == 1 ? 0xFFF : 0xFFFF, last = 0;
        for (const u8* p = buf; p < buf + size; p++) {
            u8 b = (u8)(*p - '!');
            qlt_put_sym(m, last, b);
            last = (b | (last << 6)) & mask;                                  /* qlts.hpp:52-57 */
        }
        return;
    }
    u32 last = 0, delta = 5, di = 0; u8 q1 = 0, q2 = 0;                       /* save_3, qlts.cpp:108-136 */
    for (const u8* p = buf; p < buf + size; p++) {
        u8 b = (u8)(*p - '!');
        qlt_put_sym(m, last, b);
        if (++di & 1) { last = calc_last_delta(&delta, b, q1, q2); q2 = b; }
        else          { last = calc_last_delta(&delta, b, q2, q1); q1 = b; }
    }
}
static inline u8 qlt_get_sym(qltm* m, u32 last) {                             /* qlts.cpp:168-171 */
    u8 b = (u8)l64_get(&m->ranger[last], &m->rc);
    if (b == LAST_QLT) b = (u8)pw_get(&m->exranger, &m->rc);
    return b;
}
static void qlt_load(qltm* m, u8* buf, size_t size) {                         /* qlts.cpp:163-234 */
    if (m->level <= 2) {
        u32 mask = m->level == 1 ? 0xFFF : 0xFFFF, last = 0;
        for (u8* p = buf; p < buf + size; p++) {
            u8 b = qlt_get_sym(m, last);
            *p = (u8)('!' + b);
            last = (b | (last << 6)) & mask;
        }
        return;
    }
    u32 last = 0, delta = 5, di = 0; u8 q1 = 0, q2 = 0;
    for (u8* p = buf; p < buf + size; p++) {
        u8 b = qlt_get_sym(m, last);
        *p = (u8)('!' + b);
        if (++di & 1) { last = calc_last_delta(&delta, b, q1, q2); q2 = b; }
        else          { last = calc_last_delta(&delta, b, q2, q1); q1 = b; }
    }
}

/* ================================================================================================
 * L3  base model  (gens.hpp / gens.cpp)
 * ============================================================================================== */
typedef struct {
    base2* ranger; u64 mask;
    rcoder rc;
    u64 genofs;                 /* g_genofs_count, config.cpp:44 */
    u64 ns_index, nn_index;     /* gens.hpp:59-63 */
    u8  n_byte;
    int lossless; u64 lc_index; /* block format: lowercase bases listed in "gen.lc" (not reference behaviour) */
    /* save side */
    xsave *x_ns, *x_nn, *x_lc;
    u64 *ns_list, *nn_list; size_t n_ns, n_nn, cap_ns, cap_nn; int keep_lists;
    sfqo_archive* a; wr* info;  /* for set_info("gen.N_byte") gens.cpp:104 */
    /* load side */
    xload *l_ns, *l_nn, *l_lc;
    const char* gencode;
} genm;

static int gen_bits_for_level(int level) {                                    /* gens.hpp:43-53 */
    switch (level) { case 1: return 18; case 2: return 22; case 3: return 24; default: return 26; }
}
static void gen_alloc(genm* g, int bits) {
    memset(g, 0, sizeof *g);
    size_t cnt = (size_t)1 << bits;
    g->mask = cnt - 1;
    g->ranger = xmalloc(cnt * sizeof(base2));
    for (size_t i = 0; i < cnt; i++) g->ranger[i].v = B2_INIT;                /* base2_ranger.hpp:68-71 */
}
static int gencode_of(u8 c) {                                                 /* gens.cpp:72-77 */
    switch (c) {
    case '0': case 'A': case 'a': return 0;
    case '1': case 'C': case 'c': return 1;
    case '2': case 'G': case 'g': return 2;
    case '3': case 'T': case 't': return 3;
    case '.': case 'N': case 'n': return 4;
    default: return 0x10;
    }
}
static void push64(u64** v, size_t* n, size_t* cap, u64 x) {
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 64; *v = xrealloc(*v, *cap * sizeof(u64)); }
    (*v)[(*n)++] = x;
}
static void gen_bad(genm* g, u8 gen, int bad_n, int bad_q) {                  /* bad_q_or_bad_n gens.cpp:91-114 */
    if (!bad_n) {
        xs_put(g->x_nn, g->genofs - g->nn_index);
        g->nn_index = g->genofs;
        if (g->keep_lists) push64(&g->nn_list, &g->n_nn, &g->cap_nn, g->genofs);
        return;
    }
    if (!g->n_byte) {
        g->n_byte = gen;
        if ('N' != gen && g->info) set_info_ll(g->a, g->info, "gen.N_byte", gen);
    }
    if (gen != g->n_byte) { fail("switched N_byte: %c", gen); return; }
    if (!bad_q) {
        xs_put(g->x_ns, g->genofs - g->ns_index);
        g->ns_index = g->genofs;
        if (g->keep_lists) push64(&g->ns_list, &g->n_ns, &g->cap_ns, g->genofs);
    }
}
static inline u8 gen_normalize_save(genm* g, u8 gen, u8 qlt) {                /* gens.cpp:116-136 */
    int bad_n; const int bad_q = qlt == '!';
    if (g->lossless && (gen == 'a' || gen == 'c' || gen == 'g' || gen == 't' || gen == 'n')) {   /* block format only */
        xs_put(g->x_lc, g->genofs + 1 - g->lc_index);
        g->lc_index = g->genofs + 1;
        if (gen == 'n') gen = 'N';
    }
    int n = gencode_of(gen);
    if (n <= 3) bad_n = 0;
    else {
        if (n > 4) { fail("unexpected genome char: %c", gen); return 0; }
        bad_n = 1; n = 0;
    }
    g->genofs++;
    if (bad_n || bad_q) gen_bad(g, gen, bad_n, bad_q);
    return (u8)n;
}
static void gen_save(genm* g, const u8* gen, const u8* qlt, u64 llen, u64 qlen) {  /* gens.hpp:89-93, gens.cpp:138-159 */
    u32 last = 0x007616c7;
    for (u32 i = 0; i < llen && !g_failed; i++) {
        u8 n = gen_normalize_save(g, gen[i], (llen == qlen || i < qlen) ? qlt[i] : 40);
        last &= (u32)g->mask;
        b2_put(&g->ranger[last], &g->rc, n);
        last = (last << 2) | n;
    }
}
static inline void gen_normalize_load(genm* g, u8* gen, u8 qlt) {             /* gens.cpp:200-213 */
    g->genofs++;
    if (g->nn_index == g->genofs) g->nn_index += xl_get(g->l_nn);
    else if (qlt == '!') *gen = g->n_byte;
    else if (g->ns_index == g->genofs) { *gen = g->n_byte; g->ns_index += xl_get(g->l_ns); }
}
static void gen_load(genm* g, u8* gen, const u8* qlt, u64 llen, u64 qlen) {   /* gens.hpp:111-115, gens.cpp:215-249 */
    u32 last = 0x007616c7;
    for (u32 i = 0; i < llen && !g_failed; i++) {
        last &= (u32)g->mask;
        u8 b = b2_get(&g->ranger[last], &g->rc);
        gen[i] = (u8)g->gencode[b];
        last = (last << 2) + b;
        gen_normalize_load(g, &gen[i], (llen == qlen || i < qlen) ? qlt[i] : 40);
        if (g->l_lc && g->lc_index == g->genofs) { gen[i] |= 0x20; g->lc_index += xl_get(g->l_lc); }   /* block format: "gen.lc" */
    }
}

/* ================================================================================================
 * L3  header model  (recs.hpp / recs.cpp)
 * ============================================================================================== */
typedef struct { power type; power str; poweru num; } rec_ranger;             /* recs.hpp:42-46 */
typedef struct { int off[66]; int wln[66]; u8 str[66]; int len; } space_map;  /* recs.hpp:68-73 (one spare slot) */

typedef struct {
    rec_ranger* ranger;         /* [66] recs.hpp:48 */
    rcoder rc;
    int initialized; u64 index; /* m_last recs.hpp:52-56 */
    space_map smap[2];
    u8  ctype[2][66];           /* 0=? 1=deci 2=hexa, recs.hpp:75 */
    u64 cnumb[2][66];
    int imap;
    xsave* x_file; xload* l_file;
    u64* xrec; size_t n_xrec, cap_xrec; int keep_lists;
    int comp_version;
    /* frozen-table mode (this project's format 7, not reference behaviour): when set, the symbols of the "rec" stream
       go to hook(arg, row, byte) instead of the adaptive PowerRanger rows; row = field * 16 + {0 type, 1 str, 2.. num} */
    void (*hook)(void* arg, int row, u8 sym); void* hook_arg; int no_x;
    int lossless;               /* block format (not reference behaviour): string fallback for numbers that would not print back */
} recm;

enum {  /* recs.cpp:159-190 */
    ST_DGT = 0, ST_DLT = 1, ST_STR = 2, ST_HGT = 3, ST_HLT = 4, ST_HGT_Z = 5, ST_HLT_Z = 6,
    ST_HGTC = 7, ST_HLTC = 8, ST_HGTC_Z = 9, ST_HLTC_Z = 10, ST_DGT_Z = 11, ST_DLT_Z = 12
};

static void rec_alloc(recm* r) {
    memset(r, 0, sizeof *r);
    r->ranger = xcalloc(66, sizeof(rec_ranger));
}
static int isword(u8 c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }  /* recs.cpp:139 ("C" locale) */

static void map_space(recm* r, const u8* p, int flip) {                       /* recs.cpp:141-157 */
    space_map* m = &r->smap[flip];
    m->len = 0; m->off[0] = 0;
    for (int i = 0; ; i++) {
        if (!isword(p[i])) {
            m->wln[m->len] = i - m->off[m->len];
            m->str[m->len++] = p[i];
            m->off[m->len] = i + 1;
            if (p[i] == 0 || p[i] == '\n') break;
            if (m->len > 64) { fail("ERROR: irregulal record (over 64 non alpha non digit). Is it a valid fastq file?"); return; }
        }
    }
    if (m->len > 64) fail("header with 65 separators: reference behaviour undefined (recs.hpp:69-71 overflow)");
}
static u8 numberwang(const u8* p, int len, u64* num, u8 pctype) {             /* recs.cpp:192-262 */
    int i = 0;
    int has_z = p[i] == '0';
    if (has_z) if (p[++i] == '0') return ST_STR;
    u8 caps = 0;
    *num = 0;
    while (pctype != 2) {
        if (i >= len) return has_z ? ST_DGT_Z : ST_DGT;
        if (p[i] >= '0' && p[i] <= '9') {
            u64 tnum = (*num << 3) + (*num << 1) + (p[i++]) - '0';
            if (tnum < *num) return ST_STR;
            *num = tnum;
            continue;
        }
        if ((p[i] | 0x20) < 'a' || (p[i] | 0x20) > 'f') return ST_STR;
        caps = (u8)(1 + (p[i] < 'a'));
        i = has_z;
        *num = 0;
        break;
    }
    if (len > 16) return ST_STR;
    for (; i < len; i++) {
        int nib;
        if (p[i] >= '0' && p[i] <= '9') nib = p[i] - '0';
        else if (p[i] >= 'a' && p[i] <= 'f') { if (caps == 2) return ST_STR; caps = 1; nib = 10 + (p[i] - 'a'); }
        else if (p[i] >= 'A' && p[i] <= 'F') { if (caps == 1) return ST_STR; caps = 2; nib = 10 + (p[i] - 'A'); }
        else return ST_STR;
        *num = (*num << 4) + (u64)nib;
    }
    return caps == 2 ? (has_z ? ST_HGTC_Z : ST_HGTC) : (has_z ? ST_HGT_Z : ST_HGT);
}
static void rec_sym(recm* r, int field, int which, u8 sym) {                   /* which: 0 type, 1 str */
    if (r->hook) r->hook(r->hook_arg, field * 16 + which, sym);
    else pw_put(which ? &r->ranger[field].str : &r->ranger[field].type, &r->rc, sym);
}
static void rec_num(recm* r, int field, u64 num) {                             /* PowerRangerU::put_u's byte sequence */
    if (!r->hook) { pwu_put(&r->ranger[field].num, &r->rc, num); return; }
    const int row0 = field * 16 + 2;
    if (num <= 0x7f) { r->hook(r->hook_arg, row0, (u8)num); return; }
    if (num < 0x7ffe) { r->hook(r->hook_arg, row0, (u8)(0x80 | (num >> 8))); r->hook(r->hook_arg, row0 + 1, (u8)num); return; }
    r->hook(r->hook_arg, row0, 0xff);
    if (num < 1ULL << 32) {
        r->hook(r->hook_arg, row0 + 1, 0xfe);
        for (int shift = 0, i = 2; shift < 32; shift += 8, i++) r->hook(r->hook_arg, row0 + i, (u8)(num >> shift));
        return;
    }
    r->hook(r->hook_arg, row0 + 1, 0xff);
    for (int shift = 0, i = 6; shift < 64; shift += 8, i++) r->hook(r->hook_arg, row0 + i, (u8)(num >> shift));
}
/* RecSave::save recs.cpp:277-372.  `first_out` receives the first header (recs.cpp:68-75). */
static void rec_save(recm* r, u64 record_count, const u8* buf, const u8* end, const u8* prev_buf,
                     sfqo_archive* a, wr* info) {
    if (!r->initialized) {
        r->imap = 0;
        size_t n = (size_t)(end - buf);
        if (n >= 400) { fail("first header >= 400 bytes: reference overflows MAX_LLINE (recs.cpp:30,68-70)"); return; }
        if (a) {
            char first[400]; size_t k = 0;
            for (; k < n && buf[k]; k++) first[k] = (char)buf[k];                 /* sncpy recs.cpp:59-65 */
            first[k] = 0;
            set_info(a, info, "rec.first", first);
        }
        r->initialized = 1;
        map_space(r, buf, r->imap);
        memset(r->ctype, 0, sizeof r->ctype);
        return;
    }
    int pmap = r->imap;
    r->imap = r->imap ? 0 : 1;
    int imap = r->imap;
    map_space(r, buf, imap);
    if (g_failed) return;
    space_map *mi = &r->smap[imap], *mp = &r->smap[pmap];
    int shape = mi->len != mp->len || memcmp(mi->str, mp->str, (size_t)mi->len);
    if (r->lossless && mi->len && mi->str[mi->len - 1] == 0) shape = 1;       /* a NUL inside: the fields behind it would be lost */
    if (r->hook) r->hook(r->hook_arg, 65 * 16, (u8)(shape ? 1 : 0));          /* frozen mode: every record starts with a flag symbol */
    if (shape) {
        if (r->hook) {                                                        /* ... and a header whose shape changed is coded in the chain itself */
            rec_num(r, 65, (u64)(end - buf));
            for (const u8* q = buf; q < end; q++) r->hook(r->hook_arg, 65 * 16 + 1, *q);
        } else {
            xs_put(r->x_file, record_count - r->index);
            r->index = record_count;
            xs_put_str(r->x_file, buf, (size_t)(end - buf));
        }
        memset(r->ctype[imap], 0, sizeof r->ctype[imap]);
        if (r->keep_lists) push64(&r->xrec, &r->n_xrec, &r->cap_xrec, record_count);
        return;
    }
    u64 map = 0;
    for (int i = 0; i < mi->len; i++)
        if (mi->wln[i] != mp->wln[i] || memcmp(buf + mi->off[i], prev_buf + mp->off[i], (size_t)mi->wln[i]))
            map |= 1ULL << i;
    rec_num(r, 0, map);                                                       /* put_num(0, map) */
    for (int i = 0; i < r->smap[0].len; i++) {
        if (map & (1ULL << i)) {
            const u8* b = buf + mi->off[i];
            u64 bnum;
            u8 type = numberwang(b, mi->wln[i], &bnum, r->ctype[pmap][i]);
            if (r->lossless && type != ST_STR) {                               /* would RecLoad::load (recs.cpp:430-456) print the same bytes back? */
                const int len = mi->wln[i], deci = type < ST_STR || type >= ST_DGT_Z;
                if (len == 0 || (deci && len - (b[0] == '0') > 18)) type = ST_STR;
            }
            if (type == ST_STR) {
                rec_sym(r, i + 1, 0, type);
                rec_num(r, i + 1, (u64)mi->wln[i]);                            /* put_str recs.cpp:77-81 */
                for (int j = 0; j < mi->wln[i]; j++) rec_sym(r, i + 1, 1, b[j]);
                r->ctype[imap][i] = 0;
                continue;
            }
            u64 pnum = r->ctype[pmap][i] ? r->cnumb[pmap][i] : 0;
            u64 gap;
            r->ctype[imap][i] = (type < ST_STR || type >= ST_DGT_Z) ? 1 : 2;
            r->cnumb[imap][i] = bnum;
            if (bnum < pnum) { gap = pnum - bnum; type++; }
            else gap = bnum - pnum;
            rec_sym(r, i + 1, 0, type);
            rec_num(r, i + 1, gap);
        } else {
            r->ctype[imap][i] = r->ctype[pmap][i];
            r->cnumb[imap][i] = r->cnumb[pmap][i];
        }
    }
}
static int is_number(const u8* p, int len, long long* num) {                  /* recs.cpp:265-275 */
    if (*p == '0') return 0;
    *num = 0;
    for (int i = 0; i < len; i++)
        if (p[i] >= '0' && p[i] <= '9') *num = (*num << 3) + (*num << 1) + (p[i]) - '0';
        else return 0;
    return 1;
}
static u8* rec_get_str(recm* r, int i, u8* p) {                               /* recs.cpp:129-134 */
    u32 len = (u32)pwu_get(&r->ranger[i].num, &r->rc);
    if (len > 0x2000) { fail("corrupt rec stream"); return p; }
    for (u32 j = 0; j < len; j++) p[j] = (u8)pw_get(&r->ranger[i].str, &r->rc);
    return p + len;
}
static size_t rec_load_pre5(recm* r, u8* buf, const u8* prev) {               /* recs.cpp:463-510 */
    u64 map = pwu_get(&r->ranger[0].num, &r->rc);
    u8* b = buf;
    space_map* m = &r->smap[0];
    for (int i = 0; i < m->len; i++) {
        if (map & (1ULL << i)) {
            u8 type = (u8)pw_get(&r->ranger[i + 1].type, &r->rc);
            switch (type) {
            case ST_DGT: case ST_DLT: {
                long long pval = 0;
                is_number(prev + m->off[i], m->wln[i], &pval);
                long long gap = (long long)pwu_get(&r->ranger[i + 1].num, &r->rc);
                long long val = type == ST_DGT ? pval + gap : pval - gap;
                b += sprintf((char*)b, "%lld", val);
            } break;
            case ST_STR: b = rec_get_str(r, i + 1, b); break;
            default: fail("REC: bad type value %d", type); return 0;
            }
        } else {
            memcpy(b, prev + m->off[i], (size_t)m->wln[i]);
            b += m->wln[i];
        }
        *b++ = m->str[i];
    }
    return (size_t)(b - buf - 1);
}
/* RecLoad::load recs.cpp:374-461 */
static size_t rec_load(recm* r, u64 record_count, u8* buf, const u8* prev, const sfqo_archive* a) {
    if (!r->initialized) {
        memset(r->ctype, 0, sizeof r->ctype);
        r->imap = 0;
        r->initialized = 1;
        const char* first = sfqo_info_get(a, "rec.first");                    /* load_first_line recs.cpp:113-119 */
        strcpy((char*)buf, first);
        return strlen(first);
    }
    int pmap = r->imap;
    r->imap = r->imap ? 0 : 1;
    int imap = r->imap;
    if (r->index == record_count) {
        u8* b = xl_get_str(r->l_file, buf);
        r->index += xl_get(r->l_file);
        memset(r->ctype[imap], 0, sizeof r->ctype[imap]);
        return (size_t)(b - buf);
    }
    map_space(r, prev, 0);
    if (g_failed) return 0;
    if (r->comp_version < 5) return rec_load_pre5(r, buf, prev);
    u64 map = pwu_get(&r->ranger[0].num, &r->rc);
    u8* b = buf;
    space_map* m = &r->smap[0];
    for (int i = 0; i < m->len && !g_failed; i++) {
        if (!(map & (1ULL << i))) {
            u32 count = (u32)m->wln[i];
            memcpy(b, prev + m->off[i], count);
            b += count;
            *b++ = m->str[i];
            r->ctype[imap][i] = r->ctype[pmap][i];
            r->cnumb[imap][i] = r->cnumb[pmap][i];
            continue;
        }
        u8 type = (u8)pw_get(&r->ranger[i + 1].type, &r->rc);
        if (type == ST_STR) {
            b = rec_get_str(r, i + 1, b);
            r->ctype[imap][i] = 0;
            *b++ = m->str[i];
            continue;
        }
        u64 pval = r->ctype[pmap][i] == 0 ? 0 : r->cnumb[pmap][i];
        u64 gap = pwu_get(&r->ranger[i + 1].num, &r->rc);
        u64 val; const char* fmt;
        switch (type) {
        case ST_DGT:    fmt = "%lld";  val = pval + gap; break;
        case ST_DLT:    fmt = "%lld";  val = pval - gap; break;
        case ST_HGT:    fmt = "%llx";  val = pval + gap; break;
        case ST_HLT:    fmt = "%llx";  val = pval - gap; break;
        case ST_HGT_Z:  fmt = "0%llx"; val = pval + gap; break;
        case ST_HLT_Z:  fmt = "0%llx"; val = pval - gap; break;
        case ST_HGTC:   fmt = "%llX";  val = pval + gap; break;
        case ST_HLTC:   fmt = "%llX";  val = pval - gap; break;
        case ST_HGTC_Z: fmt = "0%llX"; val = pval + gap; break;
        case ST_HLTC_Z: fmt = "0%llX"; val = pval - gap; break;
        case ST_DGT_Z:  fmt = "0%lld"; val = pval + gap; break;
        case ST_DLT_Z:  fmt = "0%lld"; val = pval - gap; break;
        default: fail("REC: bad type value %d", type); return 0;
        }
        r->ctype[imap][i] = (type < ST_STR || type >= ST_DGT_Z) ? 1 : 2;
        r->cnumb[imap][i] = val;
        if (val == 0) *b++ = '0';
        else b += sprintf((char*)b, fmt, (unsigned long long)val);
        *b++ = m->str[i];
    }
    return (size_t)(b - buf - 1);
}

/* ================================================================================================
 * L4  framing  (usrs.hpp / usrs.cpp)  +  whole-file drivers
 * ============================================================================================== */
#define MAX_ID_LLEN 0x2000                                                    /* usrs.hpp:34 */
#define MAX_GN_LLEN 0x10000                                                   /* usrs.hpp:35 */

typedef struct {
    const u8* buf; size_t cur, end;         /* whole input in memory (the reference pages it, usrs.cpp:98-124) */
    int valid;
    u64 record_count;                       /* g_record_count */
    int llen, qlen, solid;
    u64 i_llen, i_qlen, i_sgen, i_sqlt, i_long; u8 solid_pf_gen, solid_pf_qlt;   /* usrs.hpp:53-63 */
    xsave *x_llen, *x_qlen, *x_sgen, *x_sqlt, *x_lgen, *x_lqlt, *x_lrec;
    const u8 *rec, *rec_end, *prev_rec, *prev_rec_end, *gen, *qlt;
    int lossless, two_id;                   /* block format: an irregular '+' line is refused (not reference behaviour) */
} usrs;

static u8 us_at(const usrs* u, size_t i) { return i < u->end ? u->buf[i] : 0; }

static void us_update(usrs* u, int type, u16 dat) {                           /* usrs.cpp:126-160 */
    switch (type) {
    case 0: xs_put(u->x_llen, u->record_count - u->i_llen); xs_put(u->x_llen, dat); u->i_llen = u->record_count; u->llen = dat; break;
    case 1: xs_put(u->x_qlen, u->record_count - u->i_qlen); xs_put(u->x_qlen, dat); u->i_qlen = u->record_count; break;
    case 2: xs_put(u->x_sgen, u->record_count - u->i_sgen); xs_put_chr(u->x_sgen, (u8)dat); u->i_sgen = u->record_count; u->solid_pf_gen = (u8)dat; break;
    case 3: xs_put(u->x_sqlt, u->record_count - u->i_sqlt); xs_put_chr(u->x_sqlt, (u8)dat); u->i_sqlt = u->record_count; u->solid_pf_qlt = (u8)dat; break;
    }
}
static int us_expect(usrs* u, u8 chr) {                                       /* usrs.cpp:162-167 */
    u8 got = us_at(u, u->cur++);
    if (got == chr) return 1;
    fail("fastq file: expecting '%c', got '%c' after record %llu", chr, got, (unsigned long long)u->record_count);
    return 0;
}
static int us_get_record(usrs* u);
static int us_oversized(usrs* u, size_t cur, int from_get) {                  /* usrs.cpp:269-301 */
    xs_put(u->x_lrec, u->record_count - u->i_long);
    u->i_long = u->record_count;
    u->cur = cur;
    xsave* order[4] = { u->x_lrec, u->x_lgen, u->x_lrec, u->x_lqlt };
    if (us_at(u, u->cur++) != '@') { fail("record %llu: bad (long) record", (unsigned long long)u->record_count); return 0; }
    for (int k = 0; k < 4; k++) {
        u8 c;
        do {
            if (u->cur >= u->end) { fail("record %llu: seems truncated", (unsigned long long)u->record_count); return 0; }
            c = u->buf[u->cur++];
            xs_put_chr(order[k], c);
        } while (c != '\n');
    }
    if (from_get) { u->record_count++; return us_get_record(u); }
    return 1;
}
static void us_determine_record(usrs* u, sfqo_archive* a, wr* info) {         /* usrs.cpp:186-267 */
    size_t q = u->cur;
    if (u->cur >= u->end) return;
    if (us_at(u, q++) != '@') { fail("first record: Missing prefix '@', is it really a fastq format?"); return; }
    int sanity = MAX_ID_LLEN;
    while (--sanity && us_at(u, q) != '\n') q++;
    if (!sanity) { u->record_count++; if (!us_oversized(u, u->cur, 0)) return; us_determine_record(u, a, info); return; }
    if (us_at(u, q++) != '\n') { fail("first record: Expected newline"); return; }
    size_t qg = q;
    for (int i = 1; i < MAX_GN_LLEN && !u->llen; i++) if (us_at(u, q + (size_t)i) == '\n') u->llen = i;
    if (!u->llen) { u->record_count++; if (!us_oversized(u, u->cur, 0)) return; us_determine_record(u, a, info); return; }
    q += (size_t)u->llen + 1;
    if (us_at(u, q) != '+') { fail("first record: Missing 2nd prefix '+', is it really a fastq format?"); return; }
    int has_2nd_id = 0;
    while (us_at(u, ++q) != '\n') {
        if (q >= u->end) { fail("first record truncated"); return; }
        if (us_at(u, q) != ' ') has_2nd_id = 1;
    }
    int d_solid = 0;
    for (int i = 1; i < u->llen && !d_solid && !u->solid; i++)
        switch (us_at(u, qg + (size_t)i) | 0x20) {
        case '0': case '1': case '2': case '3': u->solid = 1; break;
        case 'a': case 'c': case 'g': case 't': d_solid = 1; break;
        default: break;
        }
    if (u->solid) { set_info_ll(a, info, "usr.solid", u->solid); u->llen--; }
    set_info_ll(a, info, "llen", u->llen);
    set_info_ll(a, info, "usr.2id", has_2nd_id);
    u->two_id = has_2nd_id;
}
static int us_get_record(usrs* u) {                                           /* usrs.cpp:303-390 */
#define CHECK_OVERFLOW if (u->cur >= u->end) { fail("fastq file: record seems truncated  after record %llu", (unsigned long long)u->record_count); return 0; }
    if (u->cur >= u->end) { u->valid = 0; return 0; }
    size_t currec = u->cur;
    if (!us_expect(u, '@')) return 0;
    int sanity = MAX_ID_LLEN;
    while (--sanity && us_at(u, u->cur) != '\n') u->cur++;
    if (!sanity) return us_oversized(u, currec, 1);
    const u8* rec_end = u->buf + u->cur;
    CHECK_OVERFLOW;
    if (!us_expect(u, '\n')) return 0;
    u8 update_solid_pf = 0;
    if (u->solid) {
        if (u->solid_pf_gen != us_at(u, u->cur)) update_solid_pf = us_at(u, u->cur);
        u->cur++;
    }
    const u8* gen = u->buf + u->cur;
    const size_t gi = u->cur;
    sanity = MAX_GN_LLEN;
    while (--sanity && us_at(u, u->cur) != '\n') u->cur++;
    if (!sanity) return us_oversized(u, currec, 1);
    if (update_solid_pf) us_update(u, 2, update_solid_pf);
    CHECK_OVERFLOW;
    if ((size_t)u->llen != u->cur - gi) us_update(u, 0, (u16)(u->cur - gi));
    if (!us_expect(u, '\n')) return 0;
    if (!us_expect(u, '+')) return 0;
    const size_t plus0 = u->cur;
    for (sanity = MAX_ID_LLEN; --sanity && us_at(u, u->cur) != '\n'; u->cur++);
    CHECK_OVERFLOW;
    if (!sanity) { fail("wierd second id at record %llu", (unsigned long long)u->record_count); return 0; }
    if (u->lossless) {
        const size_t pl = u->cur - plus0, hl = (size_t)(rec_end - (u->buf + currec + 1));
        if (u->two_id ? (pl != hl || memcmp(u->buf + plus0, u->buf + currec + 1, hl)) : pl != 0) {
            fail("record %llu: a '+' line that is neither empty nor the record's header", (unsigned long long)u->record_count); return 0;
        }
    }
    if (!us_expect(u, '\n')) return 0;
    if (u->solid) {
        if (u->solid_pf_qlt != us_at(u, u->cur)) us_update(u, 3, us_at(u, u->cur));
        u->cur++;
    }
    const u8* qlt = u->buf + u->cur;
    u->qlen = 0;
    for (sanity = MAX_GN_LLEN; --sanity && us_at(u, u->cur + (size_t)u->qlen) != '\n'; u->qlen++)
        if (u->cur + (size_t)u->qlen >= u->end) { fail("fastq file: record seems truncated  after record %llu", (unsigned long long)u->record_count); return 0; }
    if (!sanity) return us_oversized(u, currec, 1);
    if (u->qlen != u->llen) us_update(u, 1, (u16)u->qlen);
    u->cur += (size_t)u->qlen;
    CHECK_OVERFLOW;
    if (!us_expect(u, '\n')) return 0;
    u->prev_rec = u->rec; u->prev_rec_end = u->rec_end;
    u->rec = u->buf + currec + 1; u->rec_end = rec_end;
    u->gen = gen; u->qlt = qlt;
    return 1;
#undef CHECK_OVERFLOW
}

sfqo_archive* sfqo_compress(const u8* fastq, size_t n, const sfqo_opts* opts) {
    g_failed = 0; g_err[0] = 0;
    int level = opts->level;
    if (level < 1 || level > 4) { fail("level must be 1..4"); return NULL; }
    sfqo_archive* a = archive_new_write();                                    /* config.cpp:327-348 */
    wr* info = wr_new_info(a);
    set_info(a, info, "whoami", "slimfastq");
    set_info_ll(a, info, "version", 6);
    set_info_ll(a, info, "config.level", level);
    set_info(a, info, "orig.filename", opts->orig_filename ? opts->orig_filename : "<< stdin >>");
    if (opts->orig_filename && opts->orig_size >= 0) set_info_ll(a, info, "orig.size", opts->orig_size);

    /* zero-padded private copy: the reference may peek a few bytes past a line end */
    u8* buf = xcalloc(n + 16, 1);
    memcpy(buf, fastq, n);

    usrs u; memset(&u, 0, sizeof u);                                          /* UsrSave::UsrSave usrs.cpp:37-60 */
    u.buf = buf; u.end = n; u.valid = n > 0;
    u.x_llen = xs_new(a, "usr.x");   u.x_qlen = xs_new(a, "usr.x.q");
    u.x_sgen = xs_new(a, "usr.pfg"); u.x_sqlt = xs_new(a, "usr.pfq");
    u.x_lgen = xs_new(a, "usr.lgen"); u.x_lqlt = xs_new(a, "usr.lqlt"); u.x_lrec = xs_new(a, "usr.lrec");
    u.lossless = opts->lossless;
    if (u.valid) us_determine_record(&u, a, info);

    recm rec; rec_alloc(&rec);                                                /* UsrSave::encode usrs.cpp:392-407 */
    rec.lossless = opts->lossless;
    wr* f_rec = wr_new_named(a, "rec"); rc_init_save(&rec.rc, f_rec);
    rec.x_file = xs_new(a, "rec.x");
    genm gen; gen_alloc(&gen, opts->gen_bits ? opts->gen_bits : gen_bits_for_level(level));
    wr* f_gen = wr_new_named(a, "gen"); rc_init_save(&gen.rc, f_gen);
    gen.x_ns = xs_new(a, "gen.Ns"); gen.x_nn = xs_new(a, "gen.Nn");
    gen.lossless = opts->lossless; gen.x_lc = opts->lossless ? xs_new(a, "gen.lc") : NULL;
    gen.a = a; gen.info = info;
    qltm qlt; qlt_alloc(&qlt, level);
    wr* f_qlt = wr_new_named(a, "qlt"); rc_init_save(&qlt.rc, f_qlt);

    while (!g_failed && ++u.record_count < 3000000000ULL && us_get_record(&u)) {
        gen_save(&gen, u.gen, u.qlt, (u64)u.llen, (u64)u.qlen);
        rec_save(&rec, u.record_count, u.rec, u.rec_end, u.prev_rec, a, info);
        qlt_save(&qlt, u.qlt, (size_t)u.qlen);
    }
    set_info_ll(a, info, "num_records", (long long)u.record_count - 1);

    /* destructors, in the reference's order: ~QltSave, ~GenSave, ~RecSave (usrs.cpp:396-398 reversed), then ~UsrSave */
    rc_done(&qlt.rc);                                                         /* qlts.cpp:55-66 */
    if (!opts->quiet && qlt.extra_hi) set_info_ll(a, info, "qlt.extra.hi", qlt.extra_hi);
    free(qlt.ranger); wr_close(f_qlt);
    rc_done(&gen.rc);                                                         /* gens.cpp:80-89 */
    free(gen.ranger); wr_close(f_gen);
    xs_close(gen.x_ns, NULL, NULL); xs_close(gen.x_nn, NULL, NULL);
    if (gen.x_lc) xs_close(gen.x_lc, NULL, NULL);
    rc_done(&rec.rc);                                                         /* recs.cpp:50-57 */
    wr_close(f_rec); xs_close(rec.x_file, NULL, NULL);
    free(rec.ranger);
    if (!opts->quiet) {                                                       /* usrs.cpp:62-95 */
        char b[0x100];
        u64 lg = xs_tell(u.x_llen), lq = xs_tell(u.x_qlen);
        if (lg || lq) { sprintf(b, "gen:%llu qlt:%llu", (unsigned long long)lg, (unsigned long long)lq); set_info(a, info, "log.size.change", b); }
        u64 sg = xs_tell(u.x_sgen), sq = xs_tell(u.x_sqlt);
        if (sg || sq) { sprintf(b, "gen:%llu qlt:%llu", (unsigned long long)sg, (unsigned long long)sq); set_info(a, info, "log.solid.pf", b); }
        u64 orr = xs_tell(u.x_lrec), og = xs_tell(u.x_lgen), oq = xs_tell(u.x_lqlt);
        if (orr || og || oq) { sprintf(b, "rec:%llu gen:%llu qlt:%llu", (unsigned long long)orr, (unsigned long long)og, (unsigned long long)oq); set_info(a, info, "log.oversize", b); }
    }
    xs_close(u.x_llen, NULL, NULL); xs_close(u.x_qlen, NULL, NULL);
    xs_close(u.x_sgen, NULL, NULL); xs_close(u.x_sqlt, NULL, NULL);
    xs_close(u.x_lrec, NULL, NULL); xs_close(u.x_lgen, NULL, NULL); xs_close(u.x_lqlt, NULL, NULL);

    set_info_ll(a, info, "comp.size", (long long)(a->num_pages * PAGE));      /* Config::finit config.cpp:381-389 */
    wr_close(info);
    archive_finit_write(a);
    free(buf);
    if (g_failed) { sfqo_archive_free(a); return NULL; }
    return a;
}

/* ---- growing output ---- */
typedef struct { u8* p; size_t n, cap; } obuf;
static void ob_write(obuf* o, const void* s, size_t n) {
    if (o->n + n > o->cap) { while (o->n + n > o->cap) o->cap = o->cap ? o->cap * 2 : (1u << 16); o->p = xrealloc(o->p, o->cap); }
    memcpy(o->p + o->n, s, n); o->n += n;
}
static void ob_putc(obuf* o, u8 c) { ob_write(o, &c, 1); }
static void ob_putline(obuf* o, u8* buf, u32 size) { buf[size++] = '\n'; ob_write(o, buf, size); }   /* usrs.cpp:537-543 */

int sfqo_decompress(const sfqo_archive* a, u8** out, size_t* out_len) {      /* UsrLoad usrs.cpp:411-574 */
    g_failed = 0; g_err[0] = 0;
    obuf o = { 0, 0, 0 };
    int level = (int)info_long(a, "config.level", 2);                         /* config.cpp:359 */
    level = level > 4 ? 4 : level < 1 ? 1 : level;
    int version = (int)info_long(a, "version", 0);
    if (version > 6) { fail("compressed with a newer version %d", version); return -1; }

    u64 rc_count = 0;
    int two_id = (int)info_long(a, "usr.2id", 0) != 0;
    int solid  = info_bool(a, "usr.solid");
    size_t llen = (size_t)info_long(a, "llen", 0), qlen = llen;
    static __thread u8 m_rep[MAX_ID_LLEN + 8], m_rec[MAX_ID_LLEN + 8], m_qlt[MAX_GN_LLEN + 8], m_gen[MAX_GN_LLEN + 8];
    memset(m_rep, 0, sizeof m_rep); memset(m_rec, 0, sizeof m_rec);
    memset(m_qlt, 0, sizeof m_qlt); memset(m_gen, 0, sizeof m_gen);
    m_rec[0] = '@'; m_rep[0] = '@';
    if (!two_id) { m_gen[llen + 1] = '\n'; m_gen[llen + 2] = '+'; }
    u8 *gen_ptr, *qlt_ptr; size_t factor;
    if (solid) { gen_ptr = m_gen; qlt_ptr = m_qlt; factor = 1; }
    else       { gen_ptr = m_gen + 1; qlt_ptr = m_qlt + 1; factor = 0; }
    xload *x_llen = xl_new(a, "usr.x"), *x_qlen = xl_new(a, "usr.x.q");
    xload *x_sgen = xl_new(a, "usr.pfg"), *x_sqlt = xl_new(a, "usr.pfq");
    u64 i_llen = xl_get(x_llen), i_qlen = xl_get(x_qlen), i_sgen = xl_get(x_sgen), i_sqlt = xl_get(x_sqlt);
    xload* x_lrec = xl_new(a, "usr.lrec"); xload *x_lgen = NULL, *x_lqlt = NULL;
    u64 i_long = xl_get(x_lrec);
    if (i_long) { x_lgen = xl_new(a, "usr.lgen"); x_lqlt = xl_new(a, "usr.lqlt"); }

    size_t n_recs = (size_t)info_long(a, "num_records", 0);
    if (!n_recs && !i_long) fail("Zero records, what's going on?");

    recm rec; rec_alloc(&rec);                                                /* RecLoad::RecLoad recs.cpp:93-105 */
    rd r_rec = rd_open(a, "rec"); rc_init_load(&rec.rc, &r_rec);
    rec.l_file = xl_new(a, "rec.x"); rec.index = xl_get(rec.l_file);
    rec.comp_version = version;
    genm gen; gen_alloc(&gen, gen_bits_for_level(level));                     /* GenLoad::GenLoad gens.cpp:165-189 */
    gen.n_byte = (u8)info_long(a, "gen.N_byte", 'N');
    gen.gencode = solid ? "0123" : "ACGT";
    rd r_gen = rd_open(a, "gen"); rc_init_load(&gen.rc, &r_gen);
    gen.l_ns = xl_new(a, "gen.Ns"); gen.l_nn = xl_new(a, "gen.Nn");
    gen.ns_index = xl_get(gen.l_ns); gen.nn_index = xl_get(gen.l_nn);
    if (sfqo_stream_find(a, "gen.lc") >= 0) { gen.l_lc = xl_new(a, "gen.lc"); gen.lc_index = xl_get(gen.l_lc); }   /* block format only */
    qltm qlt; qlt_alloc(&qlt, level);                                         /* QltLoad::QltLoad qlts.cpp:142-148 */
    rd r_qlt = rd_open(a, "qlt"); rc_init_load(&qlt.rc, &r_qlt);

    u8 *b_qlt = m_qlt + 1, *b_gen = m_gen + 1;
    int flip = 0;
    while (!g_failed) {                                                       /* usrs.cpp:555-571 */
        rc_count++;
        for (;;) {                                                            /* UsrLoad::update usrs.cpp:471-510 */
            if (i_long == rc_count) {
                u8 c = '@'; ob_putc(&o, c);
                xload* order[4] = { x_lrec, x_lgen, x_lrec, x_lqlt };
                for (int k = 0; k < 4 && !g_failed; k++) {
                    size_t guard = 0;
                    do { c = xl_get_chr(order[k]); ob_putc(&o, c); if (++guard > (1u << 30)) fail("corrupt oversize stream"); } while (c != '\n' && !g_failed);
                }
                i_long += xl_get(x_lrec);
                rc_count++;
                continue;
            }
            break;
        }
        if (i_llen == rc_count) {
            llen = (size_t)xl_get(x_llen); qlen = llen;
            i_llen += xl_get(x_llen);
            if (llen + 3 >= sizeof m_gen) { fail("corrupt usr.x"); break; }
            m_gen[llen + 1] = '\n'; m_gen[llen + 2] = '+';
        }
        if (i_qlen == rc_count) { qlen = (size_t)xl_get(x_qlen); i_qlen += xl_get(x_qlen); if (qlen + 3 >= sizeof m_qlt) { fail("corrupt usr.x.q"); break; } }
        else if (qlen != llen) qlen = llen;
        if (solid && i_sgen == rc_count) { m_gen[0] = xl_get_chr(x_sgen); i_sgen += xl_get(x_sgen); }
        if (solid && i_sqlt == rc_count) { m_qlt[0] = xl_get_chr(x_sqlt); i_sqlt += xl_get(x_sqlt); }
        if (rc_count > n_recs) break;

        u8* b_rec = (flip ? m_rep : m_rec) + 1;
        u8* p_rec = (flip ? m_rec : m_rep) + 1;
        u32 rec_size = (u32)rec_load(&rec, rc_count, b_rec, p_rec, a);
        if (!rec_size) { if (!g_failed) fail("premature EOF - %llu records left", (unsigned long long)n_recs + 1); break; }
        qlt_load(&qlt, b_qlt, qlen); b_qlt[qlen] = '\n';
        gen_load(&gen, b_gen, b_qlt, llen, qlen);
        /* UsrLoad::save usrs.cpp:512-535 */
        u8* pr = flip ? m_rep : m_rec;
        flip = flip ? 0 : 1;
        ob_putline(&o, pr, rec_size + 1);
        if (two_id) {
            ob_putline(&o, gen_ptr, (u32)(llen + factor));
            pr[0] = '+'; ob_putline(&o, pr, rec_size + 1); pr[0] = '@';
        } else ob_putline(&o, gen_ptr, (u32)(llen + factor + 2));
        ob_putline(&o, qlt_ptr, (u32)(qlen + factor));
    }
    free(rec.ranger); free(gen.ranger); free(qlt.ranger);
    rd_close(&r_rec); rd_close(&r_gen); rd_close(&r_qlt);
    xl_close(rec.l_file); xl_close(gen.l_ns); xl_close(gen.l_nn); xl_close(gen.l_lc);
    xl_close(x_llen); xl_close(x_qlen); xl_close(x_sgen); xl_close(x_sqlt);
    xl_close(x_lrec); xl_close(x_lgen); xl_close(x_lqlt);
    if (g_failed) { free(o.p); return -1; }
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return 0;
}

/* ================================================================================================
 * stream-level entry points
 * ============================================================================================== */
static void take(wr* w, u8** out, size_t* out_len) { *out = w->data; *out_len = w->n; w->data = NULL; wr_close(w); }

int sfqo_qlt_encode(const u8* base, const u64* off, const u32* len, size_t nrec, int level,
                    u8** out, size_t* out_len, u32* extra_hi) {
    g_failed = 0; g_err[0] = 0;
    qltm q; qlt_alloc(&q, level);
    wr* w = wr_new_plain(); rc_init_save(&q.rc, w);
    for (size_t i = 0; i < nrec && !g_failed; i++) qlt_save(&q, base + off[i], len[i]);
    rc_done(&q.rc);
    if (extra_hi) *extra_hi = q.extra_hi;
    free(q.ranger);
    take(w, out, out_len);
    return g_failed ? -1 : 0;
}
int sfqo_qlt_decode(const u8* stream, size_t n, u8* dst, const u64* off, const u32* len, size_t nrec, int level) {
    g_failed = 0; g_err[0] = 0;
    qltm q; qlt_alloc(&q, level);
    rd r = { (u8*)stream, n, 0, n > 0 };
    rc_init_load(&q.rc, &r);
    for (size_t i = 0; i < nrec && !g_failed; i++) qlt_load(&q, dst + off[i], len[i]);
    free(q.ranger);
    return g_failed ? -1 : 0;
}
int sfqo_gen_encode(const u8* base, const u64* goff, const u32* glen, const u64* qoff, const u32* qlen,
                    size_t nrec, int gen_bits, u8** out, size_t* out_len, sfqo_gen_side* side) {
    g_failed = 0; g_err[0] = 0;
    genm g; gen_alloc(&g, gen_bits);
    wr* w = wr_new_plain(); rc_init_save(&g.rc, w);
    g.x_ns = xs_new(NULL, "gen.Ns"); g.x_nn = xs_new(NULL, "gen.Nn");
    g.keep_lists = 1;
    for (size_t i = 0; i < nrec && !g_failed; i++) gen_save(&g, base + goff[i], base + qoff[i], glen[i], qlen[i]);
    rc_done(&g.rc);
    free(g.ranger);
    xs_close(g.x_ns, NULL, NULL); xs_close(g.x_nn, NULL, NULL);
    if (side) { side->ns = g.ns_list; side->n_ns = g.n_ns; side->nn = g.nn_list; side->n_nn = g.n_nn; side->n_byte = g.n_byte; }
    else { free(g.ns_list); free(g.nn_list); }
    take(w, out, out_len);
    return g_failed ? -1 : 0;
}
int sfqo_gen_decode_raw(const u8* stream, size_t n, u8* dst, const u64* goff, const u32* glen, size_t nrec,
                        int gen_bits, int solid) {
    g_failed = 0; g_err[0] = 0;
    genm g; gen_alloc(&g, gen_bits);
    rd r = { (u8*)stream, n, 0, n > 0 };
    rc_init_load(&g.rc, &r);
    const char* code = solid ? "0123" : "ACGT";
    for (size_t k = 0; k < nrec && !g_failed; k++) {
        u32 last = 0x007616c7; u8* d = dst + goff[k];
        for (u32 i = 0; i < glen[k]; i++) {
            last &= (u32)g.mask;
            u8 b = b2_get(&g.ranger[last], &g.rc);
            d[i] = (u8)code[b];
            last = (last << 2) + b;
        }
    }
    free(g.ranger);
    return g_failed ? -1 : 0;
}
int sfqo_rec_encode(const u8* base, const u64* off, const u32* len, size_t nrec, u8** out, size_t* out_len,
                    u64** xrec, size_t* n_xrec) {
    g_failed = 0; g_err[0] = 0;
    recm r; rec_alloc(&r);
    wr* w = wr_new_plain(); rc_init_save(&r.rc, w);
    r.x_file = xs_new(NULL, "rec.x"); r.keep_lists = 1;
    const u8* prev = NULL;
    for (size_t i = 0; i < nrec && !g_failed; i++) {
        /* headers must be '\n' or NUL terminated in the caller's buffer, as in the reference (recs.cpp:148) */
        rec_save(&r, (u64)i + 1, base + off[i], base + off[i] + len[i], prev, NULL, NULL);
        prev = base + off[i];
    }
    rc_done(&r.rc);
    xs_close(r.x_file, NULL, NULL);
    free(r.ranger);
    if (xrec) { *xrec = r.xrec; *n_xrec = r.n_xrec; } else free(r.xrec);
    take(w, out, out_len);
    return g_failed ? -1 : 0;
}
int sfqo_xfile_encode_u(const u64* vals, size_t n, u8** out, size_t* out_len) {
    g_failed = 0; g_err[0] = 0;
    xsave* x = xs_new(NULL, "x");
    xs_init(x);
    for (size_t i = 0; i < n; i++) xs_put(x, vals[i]);
    xs_close(x, out, out_len);
    return g_failed ? -1 : 0;
}
int sfqo_xfile_decode_u(const u8* stream, size_t n, u64* vals, size_t nvals) {
    g_failed = 0; g_err[0] = 0;
    xload* x = xcalloc(1, sizeof *x);
    x->r.data = xmalloc(n); memcpy(x->r.data, stream, n); x->r.n = n; x->r.valid = n > 0;
    x->valid = x->r.valid;
    if (x->valid) rc_init_load(&x->rc, &x->r);
    for (size_t i = 0; i < nvals; i++) vals[i] = xl_get(x);
    xl_close(x);
    return g_failed ? -1 : 0;
}

/* ================================================================================================
 * Block format 7 extension: quality tables that start from a shared PRIOR instead of all-zero rows.
 * This is NOT reference behaviour (the reference has one cold adaptive state per file); it restates
 * this project's own rule (DESIGN.md "warm start") so that the GPU path has a CPU checker for it.
 * The arithmetic per symbol is the pinned Log64Ranger/RCoder code above; only the initial rows differ.
 * ============================================================================================== */

/* contexts of one quality line, exactly as qlt_save walks them; cb(ctx, sym) per symbol */
static void qlt_walk(const u8* buf, size_t size, int level, void (*cb)(void*, u32, u8), void* arg) {
    if (level <= 2) {
        u32 mask = level == 1 ? 0xFFF : 0xFFFF, last = 0;
        for (const u8* p = buf; p < buf + size; p++) { u8 b = (u8)(*p - '!'); cb(arg, last, b); last = (b | (last << 6)) & mask; }
        return;
    }
    u32 last = 0, delta = 5, di = 0; u8 q1 = 0, q2 = 0;
    for (const u8* p = buf; p < buf + size; p++) {
        u8 b = (u8)(*p - '!');
        cb(arg, last, b);
        if (++di & 1) { last = calc_last_delta(&delta, b, q1, q2); q2 = b; }
        else          { last = calc_last_delta(&delta, b, q2, q1); q1 = b; }
    }
}
static void hist_cb(void* arg, u32 ctx, u8 b) { ((u32*)arg)[(size_t)ctx * 64 + (b < LAST_QLT ? b : LAST_QLT)]++; }

/* counts[q_rows][64] over records first, first+step, ... */
int sfqo_qlt_histogram(const u8* base, const u64* off, const u32* len, size_t nrec, int level,
                       size_t first, size_t step, u32* counts) {
    if (!step) step = 1;
    for (size_t i = first; i < nrec; i += step) qlt_walk(base + off[i], len[i], level, hist_cb, counts);
    return 0;
}

/* The prior-row rule: iend = highest seen symbol + 1; freq = (6 * count) >> s with the smallest s
 * that brings the context's largest to <= 32000; symbols by (freq desc, symbol asc); total = sum; count = 0.
 * rows: q_rows x { u32 slot[64] (freq | sym << 16), u32 total, u32 iend } = 66 dwords per row. */
int sfqo_qlt_prior_rows(const u32* counts, size_t q_rows, u32* rows) {
    for (size_t c = 0; c < q_rows; c++) {
        const u32* cn = counts + c * 64;
        u32* r = rows + c * 66;
        memset(r, 0, 66 * sizeof(u32));
        int iend = 0; u32 mx = 0;
        for (int s = 0; s < 64; s++) if (cn[s]) { iend = s + 1; if (cn[s] > mx) mx = cn[s]; }
        if (!iend) continue;
        int sh = 0;
        while ((((u64)mx * 6) >> sh) > 32000) sh++;
        u32 fs[64]; int order[64];
        for (int s = 0; s < iend; s++) { fs[s] = (u32)(((u64)cn[s] * 6) >> sh); order[s] = s; }
        for (int i = 1; i < iend; i++) {                       /* stable insertion sort: scaled freq desc, symbol asc */
            int k = order[i], j = i;
            while (j > 0 && fs[order[j - 1]] < fs[k]) { order[j] = order[j - 1]; j--; }
            order[j] = k;
        }
        u32 total = 0;
        for (int j = 0; j < iend; j++) {
            r[j] = fs[order[j]] | ((u32)order[j] << 16);
            total += fs[order[j]];
        }
        r[64] = total; r[65] = (u32)iend;
    }
    return 0;
}

static void row_from_prior(log64* r, const u32* pr) {
    memset(r, 0, sizeof *r);
    r->total = pr[64]; r->iend = (u16)pr[65];
    for (int j = 0; j < 64; j++) { r->freq[j] = (u16)(pr[j] & 0xffff); r->syms[j] = (u8)(pr[j] >> 16); }
}

/* qlt streams of consecutive blocks of block_reads records, every block starting from `prior_rows`
 * (NULL = cold, i.e. the reference's own per-chunk result).  out = the blocks' streams back to back,
 * sizes[b] = each block's length. */
int sfqo_qlt_encode_blocks(const u8* base, const u64* off, const u32* len, size_t nrec, int level, size_t block_reads,
                           const u32* prior_rows, u8** out, size_t* out_len, u32* sizes) {
    g_failed = 0; g_err[0] = 0;
    obuf o = { 0, 0, 0 };
    qltm q; qlt_alloc(&q, level);
    size_t nb = 0;
    for (size_t r0 = 0; r0 < nrec; r0 += block_reads, nb++) {
        size_t r1 = r0 + block_reads < nrec ? r0 + block_reads : nrec;
        if (prior_rows) for (size_t c = 0; c < q.cnt; c++) row_from_prior(&q.ranger[c], prior_rows + c * 66);
        else memset(q.ranger, 0, q.cnt * sizeof(log64));
        memset(&q.exranger, 0, sizeof q.exranger);
        wr* w = wr_new_plain(); rc_init_save(&q.rc, w);
        for (size_t i = r0; i < r1 && !g_failed; i++) qlt_save(&q, base + off[i], len[i]);
        rc_done(&q.rc);
        ob_write(&o, w->data, w->n);
        if (sizes) sizes[nb] = (u32)w->n;
        wr_close(w);
    }
    free(q.ranger);
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : 0;
}


/* ================================================================================================
 * Block format 7, FROZEN tables (sfq_params.tables = 1).  NOT reference behaviour: this project's own mode, restated
 * here so that the GPU path has a CPU checker.  The learning is taken out of the per-symbol loop:
 * rows are built by counting passes and frozen while a "chain" (a run of records with its own range coder) is coded.
 * What stays the reference's: the context functions, the symbol alphabets, the header model, RCoder::Encode.
 * ============================================================================================== */

/* the chain's range coder: RCoder (coder.hpp) with the four always-zero leading bytes elided and a five-byte flush of
   the smallest multiple of 2^24 >= low, less the flush's own trailing zero bytes */
typedef struct { u64 low; u32 range; u8* out; size_t n, cap, emitted; } chenc;
static void ch_init(chenc* c) { c->low = 0; c->range = (u32)-1; c->out = NULL; c->n = c->cap = 0; c->emitted = 0; }
static void ch_put(chenc* c, u8 b) {
    if (c->emitted++ < 4) { if (b) fail("chain coder: a leading byte is not zero"); return; }
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 256; c->out = xrealloc(c->out, c->cap); }
    c->out[c->n++] = b;
}
static void ch_renorm(chenc* c) {                                             /* coder.hpp:74-80 */
    int guard = 0;
    while (c->range < TOP) {
        if ((c->low ^ (c->low + c->range)) & (0xffULL << 56)) c->range = (((u32)c->low | (u32)(TOP - 1)) - (u32)c->low);
        ch_put(c, (u8)(c->low >> 56));
        c->range <<= 8; c->low <<= 8;
        if (++guard > 64) { fail("coder stuck"); return; }
    }
}
static void ch_encode(chenc* c, u32 cum, u32 freq, u32 tot) {                 /* coder.hpp:66-73 */
    c->range /= tot;
    c->low += (u32)(cum * c->range);
    c->range *= freq;
    ch_renorm(c);
}
static size_t ch_finish(chenc* c) {
    u64 v = (c->low + 0xFFFFFFull) & ~0xFFFFFFull;
    for (int i = 0; i < 5; i++) { ch_put(c, (u8)(v >> 56)); v <<= 8; }
    for (int i = 0; i < 5 && c->n && c->out[c->n - 1] == 0; i++) c->n--;     /* the flush's own trailing zeros (a decoder reads zeros past the end) */
    return c->n;
}

/* rows with a total of exactly 2^16 from weights x[0..n): g = max(1, floor(x * 65536 / sum x)), the remainder to the
   largest g (the first of them); entry = cum | g << 16 */
static void frozen_row(const u32* x, int n, u32* out) {
    u64 S = 0;
    for (int i = 0; i < n; i++) S += x[i];
    u32 g[256]; u64 sum = 0; int best = 0;
    for (int i = 0; i < n; i++) { g[i] = (u32)(((u64)x[i] << 16) / S); if (!g[i]) g[i] = 1; sum += g[i]; if (g[i] > g[best]) best = i; }
    g[best] = (u32)(g[best] + 65536 - sum);
    u32 cum = 0;
    for (int i = 0; i < n; i++) { out[i] = cum | (g[i] << 16); cum += g[i]; }
}
/* quality rows: prior rows (sfqo_qlt_prior_rows, 66 dwords per context) -> [q_rows][64] frozen entries in symbol order
   with the Log64Ranger's weights freq + 1 (log64_ranger.hpp:109); an unseen context has the uniform row */
int sfqo_qlt_frozen_rows(const u32* rows66, size_t q_rows, u32* out) {
    for (size_t c = 0; c < q_rows; c++) {
        const u32* r = rows66 + c * 66;
        u32 x[64];
        for (int s2 = 0; s2 < 64; s2++) x[s2] = 1;
        for (u32 j = 0; j < r[65]; j++) x[r[j] >> 16] = (r[j] & 0xffff) + 1;
        if (!r[65]) { for (int s2 = 0; s2 < 64; s2++) out[c * 64 + s2] = ((u32)s2 << 10) | (1024u << 16); }
        else frozen_row(x, 64, out + c * 64);
    }
    return 0;
}
typedef struct { chenc* c; const u32* rows; u32 extra; } qfz;
static void qfz_cb(void* arg, u32 ctx, u8 b) {
    qfz* q = arg;
    const u32 sym = b < LAST_QLT ? b : LAST_QLT;
    const u32 e = q->rows[(size_t)ctx * 64 + sym];
    ch_encode(q->c, e & 0xffff, e >> 16, 65536);
    if (b >= LAST_QLT) { ch_encode(q->c, (u32)b << 8, 256, 65536); q->extra++; }     /* the escape row: all 256 values alike */
}
/* The chains of a call: blocks of block_reads records, each cut into chains of chain_reads records.  out = the chains'
   streams back to back, sizes[c] = each chain's length (c counts through the blocks).  Returns the number of chains. */
long long sfqo_qlt_encode_chains(const u8* base, const u64* off, const u32* len, size_t nrec, int level, size_t block_reads,
                                 size_t chain_reads, const u32* frozen_rows, u8** out, size_t* out_len, u32* sizes, u32* extra_hi) {
    g_failed = 0; g_err[0] = 0;
    obuf o = { 0, 0, 0 };
    size_t nc = 0; u32 extra = 0;
    for (size_t b0 = 0; b0 < nrec; b0 += block_reads) {
        const size_t b1 = b0 + block_reads < nrec ? b0 + block_reads : nrec;
        for (size_t r0 = b0; r0 < b1; r0 += chain_reads, nc++) {
            const size_t r1 = r0 + chain_reads < b1 ? r0 + chain_reads : b1;
            chenc c; ch_init(&c);
            qfz q = { &c, frozen_rows, 0 };
            for (size_t i = r0; i < r1 && !g_failed; i++) qlt_walk(base + off[i], len[i], level, qfz_cb, &q);
            const size_t n = ch_finish(&c);
            ob_write(&o, c.out, n);
            if (sizes) sizes[nc] = (u32)n;
            extra += q.extra;
            free(c.out);
        }
    }
    if (extra_hi) *extra_hi = extra;
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : (long long)nc;
}

/* Chains that are SEGMENTS of one record (long reads; chains.hip): a record of M = max(bases, qualities) symbols is cut into
   n = ceil(M / seg_len) segments of L = ceil(M / n) symbols; segment s covers symbols [s L, (s + 1) L) of the line (as far as the line
   goes) and is a chain of its own: its context starts as a line's does.  other_len: the other line's lengths (the bases'). */
static void seg_geometry(u32 len, u32 other, u32 seg_len, size_t* n, size_t* L) {
    const u64 M = len > other ? len : other;
    size_t k = (size_t)((M + seg_len - 1) / seg_len); if (!k) k = 1;
    size_t l = (size_t)((M + k - 1) / k); if (!l) l = 1;
    *n = k; *L = l;
}
long long sfqo_qlt_encode_segs(const u8* base, const u64* off, const u32* len, const u32* other_len, size_t nrec, int level, u32 seg_len,
                               const u32* frozen_rows, u8** out, size_t* out_len, u32* sizes, u32* extra_hi) {
    g_failed = 0; g_err[0] = 0;
    obuf o = { 0, 0, 0 };
    size_t nc = 0; u32 extra = 0;
    for (size_t r = 0; r < nrec; r++) {
        size_t n, L; seg_geometry(len[r], other_len[r], seg_len, &n, &L);
        for (size_t sg = 0; sg < n; sg++, nc++) {
            const size_t lo = sg * L < len[r] ? sg * L : len[r];
            const size_t cnt = len[r] - lo < L ? len[r] - lo : L;
            chenc c; ch_init(&c);
            qfz q = { &c, frozen_rows, 0 };
            if (!g_failed) qlt_walk(base + off[r] + lo, cnt, level, qfz_cb, &q);
            const size_t nb = ch_finish(&c);
            ob_write(&o, c.out, nb);
            if (sizes) sizes[nc] = (u32)nb;
            extra += q.extra;
            free(c.out);
        }
    }
    if (extra_hi) *extra_hi = extra;
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : (long long)nc;
}

/* ---- bases: generation tables ---- */
/* floor(1024 * log2(x)), x in 1..1023, by integer arithmetic (squaring a 1.31 fixed-point mantissa ten times) */
static u32 log2fp(u32 x) {
    u32 e = 0; while ((x >> (e + 1)) != 0) e++;
    u64 m = (u64)x << (31 - e);
    u32 frac = 0;
    for (int i = 0; i < 10; i++) { m = (m * m) >> 31; frac <<= 1; if (m >> 32) { frac |= 1; m >>= 1; } }
    return e * 1024 + frac;
}
static u32 gen_row(const u32* n, u32 step) {          /* f = 3 + step * n, halved the reference's way while any exceeds 255 */
    u64 f[4];
    for (int i = 0; i < 4; i++) f[i] = 3 + (u64)step * n[i];
    while ((f[0] | f[1] | f[2] | f[3]) > 255) for (int i = 0; i < 4; i++) f[i] = (f[i] >> 1) | (f[i] & 1);
    return (u32)(f[0] | f[1] << 8 | f[2] << 16 | f[3] << 24);
}
static size_t gen_bounds(size_t nblocks, size_t* bound) {   /* generation g = blocks [bound[g], bound[g + 1]) */
    size_t n = 0; bound[0] = 0;
    size_t b = (nblocks + 63) / 64; if (!b) b = 1;
    while (b < nblocks && n + 2 < 40) { bound[++n] = b; b = b * 2 > b + 1 ? b * 2 : b + 1; }
    bound[++n] = nblocks;
    return n;
}
/* walk the bases of records [r0, r1): cb(ctx, code) per base; N is coded as 0, the context restarts per record */
#define GEN_COUNT_CAP 524288ull      /* a generation of n records is counted through every ceil(n / cap)-th record (kernels.h) */
static size_t gen_count_stride(size_t n) { size_t s = (n + GEN_COUNT_CAP - 1) / GEN_COUNT_CAP; return s ? s : 1; }
static void gen_walk_s(const u8* base, const u64* goff, const u32* glen, size_t r0, size_t r1, size_t stride, u32 mask,
                       void (*cb)(void*, u32, int), void* arg);
static void gen_walk(const u8* base, const u64* goff, const u32* glen, size_t r0, size_t r1, u32 mask,
                     void (*cb)(void*, u32, int), void* arg) { gen_walk_s(base, goff, glen, r0, r1, 1, mask, cb, arg); }
static void gen_walk_s(const u8* base, const u64* goff, const u32* glen, size_t r0, size_t r1, size_t stride, u32 mask,
                       void (*cb)(void*, u32, int), void* arg) {
    for (size_t i = r0; i < r1; i += stride) {
        u32 last = 0x007616c7u;                                                 /* gens.cpp:139 */
        for (u32 k = 0; k < glen[i]; k++) {
            int code = gencode_of(base[goff[i] + k]) & 3;
            if (gencode_of(base[goff[i] + k]) > 4) code = 0;
            cb(arg, last & mask, code);
            last = (last << 2) | (u32)code;
        }
    }
}
typedef struct { u32* cnt; const u32* rows; u64 cost, nbases; int price_only; } gcount;
static void gcount_cb(void* arg, u32 ctx, int code) {
    gcount* g = arg;
    if (!g->price_only) g->cnt[(size_t)ctx * 4 + code]++;
    if (g->rows) {
        const u32 v = g->rows[ctx];
        const u32 tot = (v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + (v >> 24);
        if (!g->price_only) {
            g->cost += log2fp(tot) - log2fp((v >> (8 * code)) & 0xff);
            g->nbases++;
        } else if (v != 0x03030303u) {                       /* the pre-verdict's statistic (signed, in cost) and its sum of m (in nbases) */
            const u32 m = (tot - 12u) >> 2;
            g->cost += (u64)(long long)((int)((v >> (8 * code)) & 0xff) - 3 - (int)m);
            g->nbases += m;
        }
    }
}
typedef struct { chenc* c; const u32* rows; } gfz;
static void gfz_cb(void* arg, u32 ctx, int code) {
    gfz* g = arg;
    const u32 v = g->rows ? g->rows[ctx] : B2_INIT;
    const u32 f[4] = { v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff, v >> 24 };
    u32 cum = 0; for (int i = 0; i < code; i++) cum += f[i];
    ch_encode(g->c, cum, f[code], f[0] + f[1] + f[2] + f[3]);                   /* base2_ranger.hpp:74-84 without the update */
}
/* gen_on: 1 = the generation tables were used (decided from generation 1's cost under generation 0's rows) */
static int g_gen_force_off = 0;           /* sfqo_gm_*: the match model's verdict said no -- no table, and (round 5) every line's bases FOUR AT A
                                             TIME: the codes of bases 4 q .. 4 q + k - 1 of a line (k = 4, fewer at its end) as one symbol
                                             S = sum code[j] << 2 j of 4^k equally likely ones ("chn.idx" flag bit 6).  The same two bits a base as
                                             the initial row's 3 of 12, a quarter of the coder steps -- and each of them a shift */
/* Round 5b (block format 10, "chn.idx" flag bit 7; g_gen_force_off == 2): no coder at all -- a chain is its bases' codes, two bits each, four a byte, the
   first in the low bits, across its records' ends; the last byte padded with zero bits.  The codes are gen_flat_quads' (N-like: 0). */
typedef struct { u8* p; size_t n, cap; u32 acc, nb; } rawpk;
static void raw_line(rawpk* r, const u8* line, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int cd = gencode_of(line[i]); if (cd > 4) cd = 0;
        r->acc |= (u32)(cd & 3) << r->nb; r->nb += 2;
        if (r->nb == 8) {
            if (r->n == r->cap) { r->cap = r->cap ? r->cap * 2 : 256; r->p = xrealloc(r->p, r->cap); }
            r->p[r->n++] = (u8)r->acc; r->acc = 0; r->nb = 0;
        }
    }
}
static size_t raw_finish(rawpk* r) {
    if (r->nb) {
        if (r->n == r->cap) { r->cap = r->cap ? r->cap * 2 : 256; r->p = xrealloc(r->p, r->cap); }
        r->p[r->n++] = (u8)r->acc; r->acc = 0; r->nb = 0;
    }
    return r->n;
}
static int g_flat_raw = 1;                 /* what a call without a base model writes: 1 = format 10's two bits a base, 0 = format 9's quads through the coder */
void sfqo_set_flat_raw(int on) { g_flat_raw = on; }
static void gen_flat_quads(chenc* c, const u8* line, size_t n) {
    for (size_t i = 0; i < n; i += 4) {
        const size_t k = n - i < 4 ? n - i : 4;
        u32 S = 0;
        for (size_t j = 0; j < k; j++) { int cd = gencode_of(line[i + j]); if (cd > 4) cd = 0; S |= (u32)(cd & 3) << (2 * j); }
        ch_encode(c, S, 1, 1u << (2 * k));
    }
}
static long long gen_encode_chains_x(const u8* base, const u64* goff, const u32* glen, size_t nrec, int gen_bits, size_t block_reads,
                                 size_t chain_reads, u32 step, u8** out, size_t* out_len, u32* sizes, int* gen_on,
                                 u32 seg_len /* != 0: segments of one record */, const u32* other_len) {
    g_failed = 0; g_err[0] = 0;
    const size_t nblocks = (nrec + block_reads - 1) / block_reads;
    const size_t nctx = (size_t)1 << gen_bits; const u32 mask = (u32)nctx - 1;
    size_t bound[48];
    const size_t ngen = gen_bounds(nblocks, bound);
    u32* cnt = xcalloc(nctx * 4, 4);
    u32** rows = xcalloc(ngen + 1, sizeof(u32*));
    int on = 0;
#define REC_OF(b) ((b) * block_reads < nrec ? (b) * block_reads : nrec)
    int maybe = ngen >= 3 && !g_gen_force_off;
#define GSTRIDE(g) gen_count_stride((bound[(g) + 1] - bound[g]) * block_reads)
    if (ngen >= 3 && !g_gen_force_off && (bound[1] - bound[0]) * block_reads / GSTRIDE(0) >= 16384) {
        /* the pre-verdict (api.cpp gen_tables_begin): every 8th of generation 0's counted records counted; over every 8th of
           generation 1's, and the bases whose context that sample has seen m times: s4 += 4 x (times it saw this base) - m.
           Five deviations (sqrt(3 sum m)) above 0, or the full passes are skipped: no tables */
        gcount pc = { cnt, NULL, 0, 0, 0 };
        gen_walk_s(base, goff, glen, REC_OF(bound[0]), REC_OF(bound[1]), GSTRIDE(0) * 8, mask, gcount_cb, &pc);
        u32* r1 = xmalloc(nctx * 4);
        for (size_t c = 0; c < nctx; c++) r1[c] = gen_row(cnt + c * 4, step);
        gcount pq = { cnt, r1, 0, 0, 1 };
        gen_walk_s(base, goff, glen, REC_OF(bound[1]), REC_OF(bound[2]), GSTRIDE(1) * 8, mask, gcount_cb, &pq);
        free(r1);
        const long long s4 = (long long)pq.cost;
        maybe = s4 > 0 && (u64)s4 * (u64)s4 > 75ull * pq.nbases;
        memset(cnt, 0, nctx * 4 * sizeof(u32));              /* (the device counts the rest of generation 0 on top of the sample: the same counts) */
    }
    if (maybe) {
        gcount gc = { cnt, NULL, 0, 0, 0 };
        gen_walk_s(base, goff, glen, REC_OF(bound[0]), REC_OF(bound[1]), GSTRIDE(0), mask, gcount_cb, &gc);
        u32* r1 = xmalloc(nctx * 4);
        for (size_t c = 0; c < nctx; c++) r1[c] = gen_row(cnt + c * 4, step);
        gc.rows = r1;
        gen_walk_s(base, goff, glen, REC_OF(bound[1]), REC_OF(bound[2]), GSTRIDE(1), mask, gcount_cb, &gc);
        free(r1);
        on = gc.nbases && gc.cost * 100 < gc.nbases * 2048 * 99;
        if (on) for (size_t g = 2; g < ngen; g++) {
            rows[g] = xmalloc(nctx * 4);
            for (size_t c = 0; c < nctx; c++) rows[g][c] = gen_row(cnt + c * 4, step);
            if (g + 1 < ngen) { gcount g2 = { cnt, NULL, 0, 0, 0 }; gen_walk_s(base, goff, glen, REC_OF(bound[g]), REC_OF(bound[g + 1]), GSTRIDE(g), mask, gcount_cb, &g2); }
        }
    }
    obuf o = { 0, 0, 0 };
    size_t nc = 0;
    for (size_t b = 0; b < nblocks; b++) {
        size_t g = 0; while (g + 1 < ngen && b >= bound[g + 1]) g++;
        const size_t b0 = REC_OF(b), b1 = REC_OF(b + 1);
        if (seg_len) {
            for (size_t r = b0; r < b1; r++) {
                size_t n, L; seg_geometry(glen[r], other_len[r], seg_len, &n, &L);
                for (size_t sg = 0; sg < n; sg++, nc++) {
                    const size_t lo = sg * L < glen[r] ? sg * L : glen[r];
                    const u64 o1 = goff[r] + lo; const u32 l1 = (u32)(glen[r] - lo < L ? glen[r] - lo : L);
                    if (g_gen_force_off == 2) {
                        rawpk rp = { 0, 0, 0, 0, 0 };
                        raw_line(&rp, base + o1, l1);
                        const size_t nb = raw_finish(&rp);
                        ob_write(&o, rp.p, nb);
                        if (sizes) sizes[nc] = (u32)nb;
                        free(rp.p);
                        continue;
                    }
                    chenc c; ch_init(&c);
                    gfz gz = { &c, on ? rows[g] : NULL };
                    if (g_gen_force_off) gen_flat_quads(&c, base + o1, l1);
                    else
                    gen_walk(base, &o1, &l1, 0, 1, mask, gfz_cb, &gz);        /* (a segment starts from the seed, as a line) */
                    const size_t nb = ch_finish(&c);
                    ob_write(&o, c.out, nb);
                    if (sizes) sizes[nc] = (u32)nb;
                    free(c.out);
                }
            }
            continue;
        }
        for (size_t r0 = b0; r0 < b1; r0 += chain_reads, nc++) {
            const size_t r1 = r0 + chain_reads < b1 ? r0 + chain_reads : b1;
            if (g_gen_force_off == 2) {
                rawpk rp = { 0, 0, 0, 0, 0 };
                for (size_t r = r0; r < r1; r++) raw_line(&rp, base + goff[r], glen[r]);
                const size_t n = raw_finish(&rp);
                ob_write(&o, rp.p, n);
                if (sizes) sizes[nc] = (u32)n;
                free(rp.p);
                continue;
            }
            chenc c; ch_init(&c);
            gfz gz = { &c, on ? rows[g] : NULL };
            if (g_gen_force_off) for (size_t r = r0; r < r1; r++) gen_flat_quads(&c, base + goff[r], glen[r]);
            else
            gen_walk(base, goff, glen, r0, r1, mask, gfz_cb, &gz);
            const size_t n = ch_finish(&c);
            ob_write(&o, c.out, n);
            if (sizes) sizes[nc] = (u32)n;
            free(c.out);
        }
    }
#undef REC_OF
    for (size_t g = 0; g <= ngen; g++) free(rows[g]);
    free(rows); free(cnt);
    if (gen_on) *gen_on = on;
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : (long long)nc;
}

long long sfqo_gen_encode_chains(const u8* base, const u64* goff, const u32* glen, size_t nrec, int gen_bits, size_t block_reads,
                                 size_t chain_reads, u32 step, u8** out, size_t* out_len, u32* sizes, int* gen_on) {
    return gen_encode_chains_x(base, goff, glen, nrec, gen_bits, block_reads, chain_reads, step, out, out_len, sizes, gen_on, 0, NULL);
}
long long sfqo_gen_encode_segs(const u8* base, const u64* goff, const u32* glen, const u32* other_len, size_t nrec, int gen_bits, size_t block_reads,
                               u32 seg_len, u32 step, u8** out, size_t* out_len, u32* sizes, int* gen_on) {
    return gen_encode_chains_x(base, goff, glen, nrec, gen_bits, block_reads, 1, step, out, out_len, sizes, gen_on, seg_len, other_len);
}

/* ---- bases: the generation MATCH model (round 5; gm.hip) ---------------------------------------------------------------
   NOT the reference's model.  The reference's base model is a table of 2^24 rows addressed by the last twelve bases: one random
   memory access per base, which is what bounds it on a CPU and on a GPU alike.  What that table learns from a file whose reads
   overlap -- "after these bases comes that one" -- is also what an earlier read of the same place says outright.  So a chain
   of generation g follows a POINTER into the bases of the generations before it:
     * the call's base lines, with N coded as A (gens.cpp:116-136) and a sentinel behind every line, form the STAGE; position
       soff[r] + i = base i of record r, soff[r + 1] = soff[r] + len[r] + 1;
     * the INDEX is a table of 2^tb entries.  Over the records the counting passes take (every stride-th of a generation,
       gen_count_stride), the k-mer of GM_K = 16 bases ending at base i (i + 1 >= 16, i + 1 < len) is hashed,
       h = kmer * 0x9E3779B97F4A7C15 mod 2^64; where its top two bits are zero (a quarter of the k-mers) the entry
       (h >> (62 - tb)) & (2^tb - 1) receives min(entry, p << 24 | check) with p = the stage position of base i + 1 and
       check = (h >> (38 - tb)) & 0xFFFFFF: the EARLIEST occurrence stays, whatever the order of the insertions;
     * a line (or a segment of one) is walked with a state (kmer, pointer, m = bases matched in a row, at most 31):
       with a pointer, the base at it is the prediction e: the base is coded with e at 4096 - 3 Fo[m] of 4096 and the other three
       at Fo[m] each, the pointer moves on; a miss sets m to 0, a miss while m < 8 drops the pointer; the pointer is dropped at a sentinel.
       Without one the base is coded flat (1024 of 4096).  Without a pointer and none pending, after base i, with sixteen bases of
       the line seen and i + 1 + GM_D < n: if the k-mer is one of the quarter, its entry holds its check bits and a position p
       below the generation's first stage position, the pointer p + GM_D starts to predict at base i + 1 + GM_D -- unless a
       sentinel lies in [p, p + GM_D].  (GM_D = 1 base passes between lookup and use: a decoder reads the entry behind base i and
       looks at it behind base i + 1, with that base's arithmetic to hide the round trip.)
   Whether a call uses the model at all is decided once: generation 0's counted records indexed, every eighth counted record of
   generation 1 priced (integer log2 costs; in stretches of 256 bases, each walked as a line of its own); on if that beats two bits
   a base by 1 %.                                        */
#define GM_K 16
#define GM_D 1
#define GM_DROP 8
#define GM_MCAP 31
#define GM_EMPTY (~0ull)
static u32 gm_fo(u32 m) { return m < 4 ? 64u : m < 8 ? 40u : m < 16 ? 24u : 16u; }
typedef struct { const u8* st; const u64* T; int tb; u64 lim; chenc* c; u64 cost, nbases; } gmw;
static void gm_walk(gmw* w, u64 q0, size_t n) {
    u32 kmer = 0, seen = 0, m = 0; int have = 0; u64 ptr = 0; long long pend_at = -1; u64 pend_p = 0;
    const u8* st = w->st;
    for (size_t i = 0; i < n; i++) {
        const u32 b = st[q0 + i];
        if (pend_at == (long long)i) {
            pend_at = -1;
            int ok = 1; for (int d = 0; d <= GM_D; d++) if (st[pend_p + d] == 0xFF) { ok = 0; break; }
            if (ok) { have = 1; m = GM_K; ptr = pend_p + GM_D; }
        }
        if (have && st[ptr] == 0xFF) have = 0;
        if (have) {
            const u32 e = st[ptr], fo = gm_fo(m), fm = 4096u - 3u * fo;
            if (w->c) ch_encode(w->c, b * fo + (b > e ? fm - fo : 0u), b == e ? fm : fo, 4096);
            else w->cost += 12 * 1024 - log2fp(b == e ? fm : fo);
            if (b == e) { if (m < GM_MCAP) m++; ptr++; }
            else if (m < GM_DROP) have = 0;
            else { m = 0; ptr++; }
        } else {
            if (w->c) ch_encode(w->c, b * 1024u, 1024u, 4096);
            else w->cost += 2048;
        }
        w->nbases++;
        kmer = (kmer << 2) | b; seen++;
        if (!have && pend_at < 0 && seen >= GM_K && i + 1 + GM_D < n) {
            const u64 h = (u64)kmer * 0x9E3779B97F4A7C15ull;
            if ((h >> 62) == 0) {
                const u64 e = w->T[(h >> (62 - w->tb)) & ((1ull << w->tb) - 1)];
                if (e != GM_EMPTY && (e & 0xFFFFFF) == ((h >> (38 - w->tb)) & 0xFFFFFF) && (e >> 24) < w->lim) { pend_at = (long long)(i + 1 + GM_D); pend_p = e >> 24; }
            }
        }
    }
}
static void gm_insert(const u8* st, const u64* soff, const u32* glen, size_t r0, size_t r1, size_t stride, u64* T, int tb) {
    for (size_t r = r0; r < r1; r += stride) {
        u32 kmer = 0;
        for (u32 i = 0; i + 1 < glen[r]; i++) {
            kmer = (kmer << 2) | st[soff[r] + i];
            if (i + 1 < GM_K) continue;
            const u64 h = (u64)kmer * 0x9E3779B97F4A7C15ull;
            if ((h >> 62) != 0) continue;
            const u64 p = soff[r] + i + 1;
            if (p >> 40) continue;
            const u64 v = (p << 24) | ((h >> (38 - tb)) & 0xFFFFFF);
            u64* e = &T[(h >> (62 - tb)) & ((1ull << tb) - 1)];
            if (v < *e) *e = v;
        }
    }
}
static long long gm_encode_x(const u8* base, const u64* goff, const u32* glen, size_t nrec, int tb, size_t block_reads, size_t chain_reads,
                             u8** out, size_t* out_len, u32* sizes, int* gen_on, u32 seg_len, const u32* other_len) {
    g_failed = 0; g_err[0] = 0;
    if (tb < 8 || tb > 26) { fail("gm: table bits %d", tb); return -1; }
    const size_t nblocks = (nrec + block_reads - 1) / block_reads;
    size_t bound[48];
    const size_t ngen = gen_bounds(nblocks, bound);
#define REC_OF(b) ((b) * block_reads < nrec ? (b) * block_reads : nrec)
#define GSTRIDE(g) gen_count_stride((bound[(g) + 1] - bound[g]) * block_reads)
    int on = 0;
    u8* st = NULL; u64* soff = NULL; u64* T = NULL;
    if (ngen >= 3) {
        u64 tot = 0; for (size_t r = 0; r < nrec; r++) tot += (u64)glen[r] + 1;
        st = xmalloc(tot + 64); soff = xmalloc((nrec + 1) * sizeof(u64));
        u64 sp = 0;
        for (size_t r = 0; r < nrec; r++) {
            soff[r] = sp;
            for (u32 i = 0; i < glen[r]; i++) { const int c = gencode_of(base[goff[r] + i]); st[sp++] = (u8)(c > 4 ? 0 : c & 3); }
            st[sp++] = 0xFF;
        }
        soff[nrec] = sp; memset(st + sp, 0xFF, 64);
        T = xmalloc(sizeof(u64) << tb); memset(T, 0xFF, sizeof(u64) << tb);
        gm_insert(st, soff, glen, REC_OF(bound[0]), REC_OF(bound[1]), GSTRIDE(0), T, tb);
        gmw w = { st, T, tb, soff[REC_OF(bound[1])], NULL, 0, 0 };
        for (size_t r = REC_OF(bound[1]); r < REC_OF(bound[2]); r += GSTRIDE(1) * 8)                       /* (in stretches of 256 bases, each walked as a line) */
            for (u32 lo = 0; lo < glen[r]; lo += 256) gm_walk(&w, soff[r] + lo, glen[r] - lo < 256 ? glen[r] - lo : 256);
        on = w.nbases && w.cost * 100 < w.nbases * 2048 * 99;
        if (on) for (size_t g = 1; g + 1 < ngen; g++) gm_insert(st, soff, glen, REC_OF(bound[g]), REC_OF(bound[g + 1]), GSTRIDE(g), T, tb);
    }
    if (gen_on) *gen_on = on;
    if (!on) {
        free(st); free(soff); free(T);
        g_gen_force_off = g_flat_raw ? 2 : 1;
        const long long rc = gen_encode_chains_x(base, goff, glen, nrec, 12, block_reads, chain_reads, 4, out, out_len, sizes, NULL, seg_len, other_len);
        g_gen_force_off = 0;
        return rc;
    }
    obuf o = { 0, 0, 0 };
    size_t nc = 0;
    for (size_t b = 0; b < nblocks; b++) {
        size_t g = 0; while (g + 1 < ngen && b >= bound[g + 1]) g++;
        const size_t b0 = REC_OF(b), b1 = REC_OF(b + 1);
        const u64 lim = soff[REC_OF(bound[g])];
        if (seg_len) {
            for (size_t r = b0; r < b1; r++) {
                size_t n, L; seg_geometry(glen[r], other_len[r], seg_len, &n, &L);
                for (size_t sg = 0; sg < n; sg++, nc++) {
                    const size_t lo = sg * L < glen[r] ? sg * L : glen[r];
                    const size_t cnt = glen[r] - lo < L ? glen[r] - lo : L;
                    chenc c; ch_init(&c);
                    gmw w = { st, T, tb, lim, &c, 0, 0 };
                    gm_walk(&w, soff[r] + lo, cnt);                              /* (a segment starts as a line does) */
                    const size_t nb = ch_finish(&c);
                    ob_write(&o, c.out, nb);
                    if (sizes) sizes[nc] = (u32)nb;
                    free(c.out);
                }
            }
            continue;
        }
        for (size_t r0 = b0; r0 < b1; r0 += chain_reads, nc++) {
            const size_t r1 = r0 + chain_reads < b1 ? r0 + chain_reads : b1;
            chenc c; ch_init(&c);
            gmw w = { st, T, tb, lim, &c, 0, 0 };
            for (size_t r = r0; r < r1; r++) gm_walk(&w, soff[r], glen[r]);
            const size_t n = ch_finish(&c);
            ob_write(&o, c.out, n);
            if (sizes) sizes[nc] = (u32)n;
            free(c.out);
        }
    }
#undef REC_OF
#undef GSTRIDE
    free(st); free(soff); free(T);
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : (long long)nc;
}
/* ---- the way back (CPU; what tests/test_oracle.py checks the rule's decodability with -- the product's decoder is gm.hip) ----
   A chain's stream read as RCoder reads (coder.hpp:40-48, 83-102): the four elided zero bytes, then the stored ones, zeros behind the end. */
typedef struct { const u8* p; size_t n, pos; u64 low, code; u32 range; } chdec;
static u8 chd_get(chdec* d) { const u8 b = d->pos < d->n ? d->p[d->pos] : 0; d->pos++; return b; }
static void chd_init(chdec* d, const u8* p, size_t n) {
    d->p = p; d->n = n; d->pos = 0; d->low = 0; d->range = (u32)-1; d->code = 0;
    for (int i = 0; i < 4; i++) d->code = (d->code << 8) | chd_get(d);        /* (code's first four bytes are the elided zeros) */
}
static u32 chd_freq(chdec* d, u32 tot) { d->range /= tot; return (u32)(d->code / d->range); }
static void chd_decode(chdec* d, u32 cum, u32 freq) {
    const u32 temp = cum * d->range;
    d->low += temp; d->code -= temp; d->range *= freq;
    int guard = 0;
    while (d->range < TOP) {
        if ((d->low ^ (d->low + d->range)) & (0xffULL << 56)) d->range = (((u32)d->low | (u32)(TOP - 1)) - (u32)d->low);
        d->code = (d->code << 8) | chd_get(d);
        d->range <<= 8; d->low <<= 8;
        if (++guard > 64) { fail("decoder stuck"); return; }
    }
}
/* one line (or segment) of n bases decoded to stage positions q0 .. (the mirror of gm_walk) */
static void gm_unwalk(chdec* d, u8* st, const u64* T, int tb, u64 lim, u64 q0, size_t n) {
    u32 kmer = 0, seen = 0, m = 0; int have = 0; u64 ptr = 0; long long pend_at = -1; u64 pend_p = 0;
    for (size_t i = 0; i < n; i++) {
        if (pend_at == (long long)i) {
            pend_at = -1;
            int ok = 1; for (int k = 0; k <= GM_D; k++) if (st[pend_p + k] == 0xFF) { ok = 0; break; }
            if (ok) { have = 1; m = GM_K; ptr = pend_p + GM_D; }
        }
        if (have && st[ptr] == 0xFF) have = 0;
        u32 b;
        const u32 q = chd_freq(d, 4096);
        if (have) {
            const u32 e = st[ptr], fo = gm_fo(m), fm = 4096u - 3u * fo;
            u32 cum = 0; b = 0;
            for (;; b++) { const u32 f = b == e ? fm : fo; if (b == 3 || q < cum + f) break; cum += f; }
            chd_decode(d, cum, b == e ? fm : fo);
            if (b == e) { if (m < GM_MCAP) m++; ptr++; }
            else if (m < GM_DROP) have = 0;
            else { m = 0; ptr++; }
        } else { b = q >> 10; if (b > 3) b = 3; chd_decode(d, b * 1024u, 1024u); }
        st[q0 + i] = (u8)b;
        kmer = (kmer << 2) | b; seen++;
        if (!have && pend_at < 0 && seen >= GM_K && i + 1 + GM_D < n) {
            const u64 h = (u64)kmer * 0x9E3779B97F4A7C15ull;
            if ((h >> 62) == 0) {
                const u64 e = T[(h >> (62 - tb)) & ((1ull << tb) - 1)];
                if (e != GM_EMPTY && (e & 0xFFFFFF) == ((h >> (38 - tb)) & 0xFFFFFF) && (e >> 24) < lim) { pend_at = (long long)(i + 1 + GM_D); pend_p = e >> 24; }
            }
        }
    }
}
/* The chains of a call under the match model (gen_on = 1; whole-record chains) back to the bases' codes: codes[sum glen] in record order.
   The decoder knows the line lengths (the "usr" streams give them) and the chains' sizes ("chn.idx"); it indexes a generation
   when it has decoded it -- and must arrive at the bases the encoder started from.  Returns 0, or -1. */
int sfqo_gm_decode_chains(const u8* streams, const u32* sizes, const u32* glen, size_t nrec, int tb, size_t block_reads, size_t chain_reads, u8* codes) {
    g_failed = 0; g_err[0] = 0;
    const size_t nblocks = (nrec + block_reads - 1) / block_reads;
    size_t bound[48];
    const size_t ngen = gen_bounds(nblocks, bound);
    if (ngen < 3 || tb < 8 || tb > 26) { fail("gm decode: %zu generations, %d index bits", ngen, tb); return -1; }
#define REC_OF(b) ((b) * block_reads < nrec ? (b) * block_reads : nrec)
    u64* soff = xmalloc((nrec + 1) * sizeof(u64));
    u64 sp = 0; for (size_t r = 0; r < nrec; r++) { soff[r] = sp; sp += (u64)glen[r] + 1; } soff[nrec] = sp;
    u8* st = xmalloc(sp + 64); memset(st, 0, sp + 64);
    for (size_t r = 0; r < nrec; r++) st[soff[r] + glen[r]] = 0xFF;                    /* the sentinels, before anything is decoded */
    memset(st + sp, 0xFF, 64);
    u64* T = xmalloc(sizeof(u64) << tb); memset(T, 0xFF, sizeof(u64) << tb);
    size_t at = 0, nc = 0;
    for (size_t g = 0; g < ngen && !g_failed; g++) {
        const u64 lim = soff[REC_OF(bound[g])];
        for (size_t b = bound[g]; b < bound[g + 1]; b++) {
            const size_t b0 = REC_OF(b), b1 = REC_OF(b + 1);
            for (size_t r0 = b0; r0 < b1; r0 += chain_reads, nc++) {
                const size_t r1 = r0 + chain_reads < b1 ? r0 + chain_reads : b1;
                chdec d; chd_init(&d, streams + at, sizes[nc]);
                for (size_t r = r0; r < r1; r++) gm_unwalk(&d, st, T, tb, lim, soff[r], glen[r]);
                at += sizes[nc];
            }
        }
        if (g + 1 < ngen) gm_insert(st, soff, glen, REC_OF(bound[g]), REC_OF(bound[g + 1]), gen_count_stride((bound[g + 1] - bound[g]) * block_reads), T, tb);
    }
#undef REC_OF
    size_t o = 0;
    for (size_t r = 0; r < nrec; r++) { memcpy(codes + o, st + soff[r], glen[r]); o += glen[r]; }
    free(st); free(soff); free(T);
    return g_failed ? -1 : 0;
}

long long sfqo_gm_encode_chains(const u8* base, const u64* goff, const u32* glen, size_t nrec, int table_bits, size_t block_reads,
                                size_t chain_reads, u8** out, size_t* out_len, u32* sizes, int* gen_on) {
    return gm_encode_x(base, goff, glen, nrec, table_bits, block_reads, chain_reads, out, out_len, sizes, gen_on, 0, NULL);
}
long long sfqo_gm_encode_segs(const u8* base, const u64* goff, const u32* glen, const u32* other_len, size_t nrec, int table_bits, size_t block_reads,
                              u32 seg_len, u8** out, size_t* out_len, u32* sizes, int* gen_on) {
    return gm_encode_x(base, goff, glen, nrec, table_bits, block_reads, 1, out, out_len, sizes, gen_on, seg_len, other_len);
}

/* ---- frozen tables: the base exceptions as adaptive Rice codes (round 4; chains.hip k_gen_exc_r) -------------------------
   NOT the reference's coding: with frozen tables the three gap lists of a block -- "gen.Ns" (N-like bases whose quality is not
   '!'), "gen.Nn" (real bases under quality '!'), "gen.lc" (lowercase bases), the gaps as the reference's XFile streams define
   them (gens.cpp:91-114, the position of a base counted from 1 over the block's base lines, each gap against the list's
   previous entry) -- are written as bit streams instead of through adaptive PowerRanger rows:
     k = the smallest k <= 24 with (N << k) >= A;  q = v >> k;
     q < 32: q one bits, a zero bit, the low k bits of v;   else: 32 one bits, then v in 40 bits (low 20 first)
     A += v, N += 1; when N reaches 32 both are halved.   Start: A = 256, N = 1.
   Bits fill bytes from the low end.  A list ends with v = 0 (no gap is 0) and zero bits up to a byte; an empty list is an
   empty stream.                                                                                                         */
typedef struct { u8* p; size_t n, cap; u64 acc; u32 nbits; u64 A; u32 N; int opened; } rice_w;
static void rice_bits(rice_w* w, u32 v, u32 n) {                              /* n <= 32 */
    if (!n) return;
    w->acc |= (u64)(n == 32 ? v : (v & ((1u << n) - 1u))) << w->nbits;
    w->nbits += n;
    while (w->nbits >= 8) {
        if (w->n == w->cap) { w->cap = w->cap ? w->cap * 2 : 64; w->p = xrealloc(w->p, w->cap); }
        w->p[w->n++] = (u8)w->acc; w->acc >>= 8; w->nbits -= 8;
    }
}
static u32 rice_k(u64 A, u32 N) { u32 k = 0; while (k < 24 && ((u64)N << k) < A) k++; return k; }
static void rice_put(rice_w* w, u64 v) {
    w->opened = 1;
    const u32 k = rice_k(w->A, w->N);
    const u64 q = v >> k;
    if (q < 32) { rice_bits(w, (u32)((1ull << q) - 1), (u32)q); rice_bits(w, 0, 1); rice_bits(w, (u32)(v & ((1ull << k) - 1)), k); }
    else { rice_bits(w, 0xFFFFFFFFu, 32); rice_bits(w, (u32)(v & 0xFFFFF), 20); rice_bits(w, (u32)((v >> 20) & 0xFFFFF), 20); }
    w->A += v; w->N++;
    if (w->N >= 32) { w->A >>= 1; w->N >>= 1; }
}
static void rice_finish(rice_w* w) {
    if (!w->opened) return;
    rice_put(w, 0);
    if (w->nbits) rice_bits(w, 0, 8 - w->nbits);
}
/* One block's three lists: records [r0, r0 + n) of base lines (goff, glen) and quality lines (qoff, qlen), all offsets into
   base (behind a SOLiD prefix, if any).  out[3] / out_len[3]: gen.Ns, gen.Nn, gen.lc (malloc'ed; free with sfqo_free).
   n_byte: the block's N character (0: none).  Returns 0, or -1 for a base line the model refuses (an illegal character, two
   different N characters). */
int sfqo_exc_rice_block(const u8* base, const u64* goff, const u32* glen, const u64* qoff, const u32* qlen, size_t nrec,
                        u8** out, size_t* out_len, u32* n_byte_out) {
    g_failed = 0; g_err[0] = 0;
    rice_w w[3]; memset(w, 0, sizeof w);
    for (int i = 0; i < 3; i++) { w[i].A = 256; w[i].N = 1; }
    u64 genofs = 0, last[3] = {0, 0, 0};
    u32 n_byte = 0;
    for (size_t r = 0; r < nrec && !g_failed; r++) {
        for (u32 i = 0; i < glen[r]; i++) {
            u8 g = base[goff[r] + i];
            const u8 q = i < qlen[r] ? base[qoff[r] + i] : 40;                /* gens.cpp:153 */
            if (g == 'a' || g == 'c' || g == 'g' || g == 't' || g == 'n') {
                rice_put(&w[2], genofs + 1 - last[2]); last[2] = genofs + 1;
                if (g == 'n') g = 'N';
            }
            const int n = gencode_of(g);
            const int bad_q = q == '!', bad_n = n > 3;
            if (n > 4) { fail("illegal base character %d", g); break; }
            genofs++;
            if (bad_n) {
                if (!n_byte) n_byte = g;
                if (g != n_byte) { fail("switched N_byte: %c", g); break; }
                if (!bad_q) { rice_put(&w[0], genofs - last[0]); last[0] = genofs; }
            } else if (bad_q) { rice_put(&w[1], genofs - last[1]); last[1] = genofs; }
        }
    }
    for (int i = 0; i < 3; i++) { rice_finish(&w[i]); out[i] = w[i].p; out_len[i] = w[i].n; }
    if (n_byte_out) *n_byte_out = n_byte;
    return g_failed ? -1 : 0;
}
/* the way back: the positions (counted from 1) a list holds -> pos[cap]; returns their number, or -1 for a stream that does
   not end properly */
long long sfqo_exc_rice_decode(const u8* p, size_t n, u64* pos, size_t cap) {
    if (!n) return 0;
    u64 A = 256; u32 N = 1; size_t bit = 0, cnt = 0; u64 at = 0;
    #define RBIT() ((bit >> 3) < n ? (p[bit >> 3] >> (bit & 7)) & 1u : 0u)
    for (;;) {
        if ((bit >> 3) >= n + 16) return -1;
        const u32 k = rice_k(A, N);
        u32 q = 0; while (q < 32 && RBIT()) { q++; bit++; }
        u64 v = 0;
        if (q < 32) { bit++; for (u32 j = 0; j < k; j++, bit++) v |= (u64)RBIT() << j; v |= (u64)q << k; }
        else for (u32 j = 0; j < 40; j++, bit++) v |= (u64)RBIT() << j;
        if (!v) break;
        at += v;
        if (cnt < cap) pos[cnt] = at;
        cnt++;
        A += v; N++;
        if (N >= 32) { A >>= 1; N >>= 1; }
    }
    #undef RBIT
    return (long long)cnt;
}

/* ---- headers: frozen PowerRanger rows ---- */
#define REC_ROWS (66 * 16)
static void hcount_hook(void* arg, int row, u8 sym) { u32** a = arg; if (a[1]) a[0][(size_t)row * 256 + sym]++; }
/* the counting pass: the header model over runs of `run` records starting every `stride` records; the first record of a
   run is its "first header", the second warms the field types up, the rest are counted.  counts[REC_ROWS][256] */
int sfqo_rec_count(const u8* base, const u64* off, const u32* len, size_t nrec, size_t stride, size_t run, size_t nruns, u32* counts) {
    g_failed = 0; g_err[0] = 0;
    for (size_t i = 0; i < nruns; i++) {
        const size_t r0 = i * stride;
        if (r0 >= nrec) break;
        const size_t n = nrec - r0 < run ? nrec - r0 : run;
        recm r; rec_alloc(&r);
        void* arg[2] = { counts, NULL };
        r.hook = hcount_hook; r.hook_arg = arg; r.no_x = 1; r.lossless = 1;
        const u8* prev = NULL;
        for (size_t k = 0; k < n && !g_failed; k++) {
            arg[1] = k >= 2 ? (void*)counts : NULL;
            rec_save(&r, (u64)k + 1, base + off[r0 + k], base + off[r0 + k] + len[r0 + k], prev, NULL, NULL);
            prev = base + off[r0 + k];
        }
        free(r.ranger);
    }
    return g_failed ? -1 : 0;
}
/* counts -> scaled frequencies f = (14 * count) >> s, the row's largest <= 32000 (what "rec.pri" carries) */
int sfqo_rec_prior_freqs(const u32* counts, u32* f) {
    for (int r = 0; r < REC_ROWS; r++) {
        u64 mx = 0;
        for (int s2 = 0; s2 < 256; s2++) if (counts[r * 256 + s2] > mx) mx = counts[r * 256 + s2];
        int sh = 0;
        while (((mx * 14) >> sh) > 32000) sh++;
        for (int s2 = 0; s2 < 256; s2++) f[r * 256 + s2] = (u32)(((u64)counts[r * 256 + s2] * 14) >> sh);
    }
    return 0;
}
int sfqo_rec_frozen_rows(const u32* f, u32* rows) {
    for (int r = 0; r < REC_ROWS; r++) {
        u32 x[256];
        for (int s2 = 0; s2 < 256; s2++) x[s2] = f[r * 256 + s2] + 1;           /* power_ranger.hpp:100 */
        frozen_row(x, 256, rows + (size_t)r * 256);
    }
    return 0;
}
typedef struct { chenc* c; const u32* rows; } hfz;
static void hfz_hook(void* arg, int row, u8 sym) {
    hfz* h = arg;
    const u32 e = h->rows[(size_t)row * 256 + sym];
    ch_encode(h->c, e & 0xffff, e >> 16, 65536);
}
/* "rec" streams through frozen rows: a block's headers are cut into chains of chain_reads records; every chain starts
   from the block's first header (the base, never coded) with cold field types.  out = the chains' streams back to back,
   sizes[c] / hdr_bytes[c] per chain.  Returns the number of chains. */
long long sfqo_rec_encode_chains_frozen(const u8* base, const u64* off, const u32* len, size_t nrec, size_t block_reads, size_t chain_reads,
                                        const u32* frozen_rows, u8** out, size_t* out_len, u32* sizes, u32* hdr_bytes) {
    g_failed = 0; g_err[0] = 0;
    obuf o = { 0, 0, 0 };
    size_t nc = 0;
    for (size_t b0 = 0; b0 < nrec; b0 += block_reads) {
        const size_t b1 = b0 + block_reads < nrec ? b0 + block_reads : nrec;
        for (size_t r0 = b0; r0 < b1; r0 += chain_reads, nc++) {
            const size_t r1 = r0 + chain_reads < b1 ? r0 + chain_reads : b1;
            chenc c; ch_init(&c);
            hfz h = { &c, frozen_rows };
            recm r; rec_alloc(&r);
            r.hook = hfz_hook; r.hook_arg = &h; r.lossless = 1;
            rec_save(&r, 1, base + off[b0], base + off[b0] + len[b0], NULL, NULL, NULL);      /* the base */
            const u8* prev = base + off[b0];
            u32 hb = 0;
            for (size_t i = r0; i < r1 && !g_failed; i++) {
                hb += len[i];
                if (i == b0) continue;
                rec_save(&r, (u64)(i - b0) + 1, base + off[i], base + off[i] + len[i], prev, NULL, NULL);
                prev = base + off[i];
            }
            free(r.ranger);
            const size_t n = ch_finish(&c);
            ob_write(&o, c.out, n);
            if (sizes) sizes[nc] = (u32)n;
            if (hdr_bytes) hdr_bytes[nc] = hb;
            free(c.out);
        }
    }
    *out = o.p ? o.p : xmalloc(1); *out_len = o.n;
    return g_failed ? -1 : (long long)nc;
}


/* ================================================================================================
 * A "rec" stream in the PRE-VERSION-5 layout, for testing RecLoad::load_pre5 (recs.cpp:463-510).  The reference no longer
 * has the encoder of that layout; this one is derived from the decoder: per changed field the type ST_DGT / ST_DLT
 * (both the previous and the current text are numbers by is_number, recs.cpp:265-275, the gap between them) or ST_STR
 * (length and characters); values come from the previous header's TEXT, nothing is cached.  Headers must keep their
 * shape (no rec.x) and no numeric field may become 0 (sprintf("%lld") of 0 is "0", as the text was).
 * ============================================================================================== */
int sfqo_rec_encode_pre5(const u8* base, const u64* off, const u32* len, size_t nrec, u8** out, size_t* out_len) {
    g_failed = 0; g_err[0] = 0;
    recm r; rec_alloc(&r);
    wr* w = wr_new_plain(); rc_init_save(&r.rc, w);
    const u8* prev = NULL;
    for (size_t k = 0; k < nrec && !g_failed; k++) {
        const u8* buf = base + off[k];
        if (k == 0) { map_space(&r, buf, 0); prev = buf; continue; }
        space_map mp = r.smap[0];
        map_space(&r, buf, 0);                                   /* the decoder maps the PREVIOUS header into smap[0]; we need both */
        space_map mi = r.smap[0];
        if (mi.len != mp.len || memcmp(mi.str, mp.str, (size_t)mi.len)) { fail("pre-5 test stream: the header shape changed"); break; }
        u64 map = 0;
        for (int i = 0; i < mi.len; i++)
            if (mi.wln[i] != mp.wln[i] || memcmp(buf + mi.off[i], prev + mp.off[i], (size_t)mi.wln[i])) map |= 1ULL << i;
        pwu_put(&r.ranger[0].num, &r.rc, map);
        for (int i = 0; i < mi.len; i++) {
            if (!(map & (1ULL << i))) continue;
            long long pv, cv;
            if (is_number(prev + mp.off[i], mp.wln[i], &pv) && is_number(buf + mi.off[i], mi.wln[i], &cv) && mp.wln[i] > 0 && mi.wln[i] > 0 && mi.wln[i] <= 18 && mp.wln[i] <= 18) {
                if (cv >= pv) { pw_put(&r.ranger[i + 1].type, &r.rc, ST_DGT); pwu_put(&r.ranger[i + 1].num, &r.rc, (u64)(cv - pv)); }
                else          { pw_put(&r.ranger[i + 1].type, &r.rc, ST_DLT); pwu_put(&r.ranger[i + 1].num, &r.rc, (u64)(pv - cv)); }
            } else {
                pw_put(&r.ranger[i + 1].type, &r.rc, ST_STR);
                pwu_put(&r.ranger[i + 1].num, &r.rc, (u64)mi.wln[i]);
                for (int j = 0; j < mi.wln[i]; j++) pw_put(&r.ranger[i + 1].str, &r.rc, buf[mi.off[i] + j]);
            }
        }
        prev = buf;
    }
    rc_done(&r.rc);
    free(r.ranger);
    take(w, out, out_len);
    return g_failed ? -1 : 0;
}
