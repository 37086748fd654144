/* problem.c -- SDPA reader, flat problem image, pre-solve, rank rule, start point (host, plain C).
 *
 * Own implementation.  Behaviour follows the reference so that the same .dat-s file yields the same
 * problem, the same branch decisions and the same start point:
 *   reader conventions   src_semi/io/lorads_file_io.c:21-293
 *   matrix / cone typing  src_semi/data/lorads_sdp_data.c:811-828, io/lorads_user_data.c:58-71
 *   union pattern         src_semi/data/lorads_sdp_conic.c:868-1076
 *   norms                 src_semi/data/lorads_solver.c:149-185,1054-1073; lorads_sdp_data.c:148-183
 *   rank rule             src_semi/data/lorads_solver.c:290-319
 *   start point           src_semi/data/lorads_solver.c:361-371,415,652-653
 */
#include "lorads_host.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int mat, blk, row, col;
    double val;
    int64_t seq;
} ent_t;

static int ent_cmp(const void *pa, const void *pb) {
    const ent_t *a = (const ent_t *)pa, *b = (const ent_t *)pb;
    if (a->blk != b->blk) return a->blk < b->blk ? -1 : 1;
    if (a->mat != b->mat) return a->mat < b->mat ? -1 : 1;
    if (a->col != b->col) return a->col < b->col ? -1 : 1;
    if (a->row != b->row) return a->row < b->row ? -1 : 1;
    return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

typedef struct {
    int row, col;
} pos_t;

/* stable LSD radix sort of n (key, payload) pairs on the low `bits` bits of the keys, 11 bits per pass (the qsort calls this replaces
 * were the larger half of the reader's time on files of 10^5+ entries).  Returns 0, or 1 when the scratch could not be allocated */
static int radix_pairs(uint64_t *key, uint32_t *val, size_t n, int bits) {
    if (n < 2) return 0;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * n);
    size_t *cnt = (size_t *)malloc(sizeof(size_t) * 2048);
    if (!k2 || !v2 || !cnt) { free(k2); free(v2); free(cnt); return 1; }
    uint64_t *ka = key, *kb = k2;
    uint32_t *va = val, *vb = v2;
    for (int shift = 0; shift < bits; shift += 11) {
        memset(cnt, 0, sizeof(size_t) * 2048);
        for (size_t i = 0; i < n; ++i) cnt[(ka[i] >> shift) & 2047u]++;
        if (cnt[(ka[0] >> shift) & 2047u] == n) continue; /* (all keys share this digit) */
        size_t run = 0;
        for (int d = 0; d < 2048; ++d) { const size_t c = cnt[d]; cnt[d] = run; run += c; }
        for (size_t i = 0; i < n; ++i) {
            const size_t dst = cnt[(ka[i] >> shift) & 2047u]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != key) { memcpy(key, ka, sizeof(uint64_t) * n); memcpy(val, va, sizeof(uint32_t) * n); }
    free(k2); free(v2); free(cnt);
    return 0;
}
static int bits_of(uint64_t maxval) { int b = 1; while (b < 64 && (maxval >> b) != 0) ++b; return b; }

static void block_free(lrd_block *b) {
    free(b->row_idx); free(b->a_ptr); free(b->a_row); free(b->a_col); free(b->a_val);
    free(b->c_row); free(b->c_col); free(b->c_val);
    free(b->p_row); free(b->p_col); free(b->a_pidx); free(b->c_pidx);
    memset(b, 0, sizeof *b);
}

void lrd_problem_free(lrd_problem *p) {
    if (!p) return;
    for (int k = 0; k < p->nblk; ++k) block_free(&p->blk[k]);
    free(p->blk);
    free(p->b);
    free(p->con_global);
    free(p);
}

/* pre-solve of one block: cone type, dense/sparse scratch decision, union pattern + index maps */
static int block_presolve(lrd_block *b, int m) {
    const int n = b->n;
    const int64_t npack = (int64_t)n * (n + 1) / 2;
    const int na = b->a_ptr[b->nrow];
    b->cone_sparse = !((double)b->nrow > 0.3 * (double)m);
    int dense = 0;
    if (n < 20 && !b->is_lp) dense = 1;
    if (!dense && !b->is_lp) {
        if ((double)b->c_nnz > 0.1 * (double)npack) dense = 1;
        for (int i = 0; i < b->nrow && !dense; ++i)
            if ((double)(b->a_ptr[i + 1] - b->a_ptr[i]) > 0.1 * (double)npack) dense = 1;
    }
    pos_t *u = NULL;
    int np = 0;
    uint32_t *who = NULL; /* sorted position -> stored entry (C's first, then the A_i's) */
    int *upos = NULL;     /* sorted position -> its place in the union pattern */
    if (!dense) {
        const int tot = b->c_nnz + na;
        const size_t cap = (size_t)(tot > 0 ? tot : 1);
        u = (pos_t *)malloc(sizeof(pos_t) * cap);
        uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * cap);
        who = (uint32_t *)malloc(sizeof(uint32_t) * cap);
        upos = (int *)malloc(sizeof(int) * cap);
        if (!u || !key || !who || !upos) { free(u); free(key); free(who); free(upos); return 1; }
        const uint64_t nn = (uint64_t)(n > 0 ? n : 1);
        for (int k = 0; k < b->c_nnz; ++k) key[k] = (uint64_t)b->c_col[k] * nn + (uint64_t)b->c_row[k];
        for (int k = 0; k < na; ++k) key[b->c_nnz + k] = (uint64_t)b->a_col[k] * nn + (uint64_t)b->a_row[k];
        for (int k = 0; k < tot; ++k) who[k] = (uint32_t)k;
        if (radix_pairs(key, who, (size_t)tot, bits_of(nn * nn))) { free(u); free(key); free(who); free(upos); return 1; } /* by column, then row */
        for (int k = 0; k < tot; ++k) {
            if (k == 0 || key[k] != key[k - 1]) { u[np].row = (int)(key[k] % nn); u[np].col = (int)(key[k] / nn); ++np; }
            upos[k] = np - 1;
        }
        free(key);
        if ((double)np / (double)npack >= 0.1 && !b->is_lp) { dense = 1; free(u); u = NULL; free(who); who = NULL; free(upos); upos = NULL; }
    }
    b->dense_mode = dense;
    free(b->p_row); free(b->p_col); free(b->a_pidx); free(b->c_pidx);
    b->a_pidx = (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
    b->c_pidx = (int *)malloc(sizeof(int) * (size_t)(b->c_nnz > 0 ? b->c_nnz : 1));
    if (dense) {
        if (npack > 0x7fffffff) return 1;
        b->np = (int)npack;
        b->p_row = (int *)malloc(sizeof(int) * (size_t)npack);
        b->p_col = (int *)malloc(sizeof(int) * (size_t)npack);
        int t = 0;
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n; ++i) { b->p_row[t] = i; b->p_col[t] = j; ++t; }
        /* packed column-major lower index, src_semi/lorads_utils.h:45-47 */
        for (int k = 0; k < na; ++k)
            b->a_pidx[k] = (int)(((int64_t)(2 * n - b->a_col[k] - 1) * b->a_col[k]) / 2 + b->a_row[k]);
        for (int k = 0; k < b->c_nnz; ++k)
            b->c_pidx[k] = (int)(((int64_t)(2 * n - b->c_col[k] - 1) * b->c_col[k]) / 2 + b->c_row[k]);
    } else {
        b->np = np;
        b->p_row = (int *)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
        b->p_col = (int *)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
        for (int k = 0; k < np; ++k) { b->p_row[k] = u[k].row; b->p_col[k] = u[k].col; }
        for (int k = 0; k < b->c_nnz + na; ++k) { /* every stored entry's place in the pattern, from the sort's payload */
            const int w = (int)who[k];
            if (w < b->c_nnz) b->c_pidx[w] = upos[k];
            else b->a_pidx[w - b->c_nnz] = upos[k];
        }
        free(u); free(who); free(upos);
    }
    return 0;
}

static void problem_norms(lrd_problem *p) {
    double n1 = 0, n2 = 0, ninf = 0;
    for (int k = 0; k < p->nblk; ++k) {
        const lrd_block *b = &p->blk[k];
        double lp1 = 0;
        for (int t = 0; t < b->c_nnz; ++t) {
            double a = fabs(b->c_val[t]);
            if (b->is_lp) { n1 += a; lp1 += a; }
            else if (b->c_row[t] == b->c_col[t]) { n1 += a; n2 += a * a; }
            else { n1 += 2 * a; n2 += 2 * a * a; }
            if (a > ninf) ninf = a;
        }
        /* the LP cone reports (||c||_1)^2 as its "nrm2 square" (data/lorads_lp_conic.c:128-133) -- kept */
        if (b->is_lp) n2 += lp1 * lp1;
    }
    p->cObjNrm1 = n1; p->cObjNrm2 = sqrt(n2); p->cObjNrmInf = ninf;
    double b1 = 0, b2 = 0, binf = 0;
    for (int i = 0; i < p->m; ++i) {
        double a = fabs(p->b[i]);
        b1 += a; b2 += a * a;
        if (a > binf) binf = a;
    }
    p->bNrm1 = b1; p->bNrm2 = sqrt(b2); p->bNrmInf = binf;
}

/* entries into (cone, matrix, column, row, file order): two radix sorts on packed keys -- (column, row) first, (cone, matrix) second,
 * both stable, starting from file order -- and one gather; qsort with ent_cmp when the keys do not pack (or LORADS_READER=sscanf: tests) */
static int sort_entries(ent_t *e, int64_t ne, int nblk, int m, const lrd_problem *p) {
    uint64_t nmax = 1;
    for (int k = 0; k < nblk; ++k)
        if ((uint64_t)p->blk[k].n > nmax) nmax = (uint64_t)p->blk[k].n;
    const int plain = getenv("LORADS_READER") && !strcmp(getenv("LORADS_READER"), "sscanf");
    if (plain || ne < 2 || ne > 0xffffffffll || nmax > 0x7fffffffull || (uint64_t)nblk * ((uint64_t)m + 1) > ((uint64_t)1 << 62)) {
        qsort(e, (size_t)ne, sizeof(ent_t), ent_cmp);
        return 0;
    }
    uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)ne);
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)ne);
    ent_t *tmp = (ent_t *)malloc(sizeof(ent_t) * (size_t)ne);
    int rc = !key || !idx || !tmp;
    if (!rc) {
        for (int64_t t = 0; t < ne; ++t) { key[t] = (uint64_t)e[t].col * nmax + (uint64_t)e[t].row; idx[t] = (uint32_t)t; }
        rc = radix_pairs(key, idx, (size_t)ne, bits_of(nmax * nmax));
    }
    if (!rc) {
        for (int64_t t = 0; t < ne; ++t) key[t] = (uint64_t)e[idx[t]].blk * ((uint64_t)m + 1) + (uint64_t)e[idx[t]].mat;
        rc = radix_pairs(key, idx, (size_t)ne, bits_of((uint64_t)nblk * ((uint64_t)m + 1)));
    }
    if (!rc) {
        for (int64_t t = 0; t < ne; ++t) tmp[t] = e[idx[t]];
        memcpy(e, tmp, sizeof(ent_t) * (size_t)ne);
    }
    free(key); free(idx); free(tmp);
    return rc;
}

/* shared back end of the reader and of lrd_problem_from_triplets: entries are 0-based, mat 0 = F0 */
static int build_problem(int m, const double *bvec, int nblk, const int *dims, ent_t *e, int64_t ne,
                         lrd_problem **out) {
    lrd_problem *p = (lrd_problem *)calloc(1, sizeof *p);
    p->m = m;
    p->b = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    memcpy(p->b, bvec, sizeof(double) * (size_t)m);
    p->nblk = p->nblk_global = nblk;
    p->blk = (lrd_block *)calloc((size_t)nblk, sizeof(lrd_block));
    p->sum_dims_global = 0;
    for (int k = 0; k < nblk; ++k) {
        p->blk[k].is_lp = dims[k] < 0; /* negative dimension = the diagonal (LP) block */
        p->blk[k].n = dims[k] < 0 ? -dims[k] : dims[k];
        p->blk[k].global_id = k;
        if (!p->blk[k].is_lp) { p->sum_dims_global += p->blk[k].n; p->nsdp_global += 1; }
    }
    /* normalise: drop tiny, lower triangle, negate F0 */
    int64_t w = 0;
    for (int64_t t = 0; t < ne; ++t) {
        ent_t x = e[t];
        if (fabs(x.val) < 1e-12) continue;
        if (x.blk < 0 || x.blk >= nblk || x.mat < 0 || x.mat > m) { lrd_problem_free(p); return 2; }
        if (p->blk[x.blk].is_lp) x.col = x.row; /* the reference keys LP entries by their row index only (:264) */
        if (x.row < x.col) { int tmp = x.row; x.row = x.col; x.col = tmp; }
        if (x.col < 0 || x.row >= p->blk[x.blk].n) { lrd_problem_free(p); return 2; }
        if (x.mat == 0) x.val = -x.val;
        x.seq = t;
        e[w++] = x;
    }
    ne = w;
    if (sort_entries(e, ne, nblk, m, p)) { lrd_problem_free(p); return 3; }
    int64_t pos = 0;
    for (int k = 0; k < nblk; ++k) {
        lrd_block *b = &p->blk[k];
        int64_t beg = pos;
        while (pos < ne && e[pos].blk == k) ++pos;
        int64_t cnt = pos - beg, cn = 0;
        while (cn < cnt && e[beg + cn].mat == 0) ++cn;
        b->c_nnz = (int)cn;
        b->c_row = (int *)malloc(sizeof(int) * (size_t)(cn > 0 ? cn : 1));
        b->c_col = (int *)malloc(sizeof(int) * (size_t)(cn > 0 ? cn : 1));
        b->c_val = (double *)malloc(sizeof(double) * (size_t)(cn > 0 ? cn : 1));
        for (int64_t t = 0; t < cn; ++t) {
            b->c_row[t] = e[beg + t].row; b->c_col[t] = e[beg + t].col; b->c_val[t] = e[beg + t].val;
        }
        int64_t na = cnt - cn;
        int nrow = 0;
        for (int64_t t = beg + cn; t < pos; ++t)
            if (t == beg + cn || e[t].mat != e[t - 1].mat) ++nrow;
        b->nrow = nrow;
        b->row_idx = (int *)malloc(sizeof(int) * (size_t)(nrow > 0 ? nrow : 1));
        b->a_ptr = (int *)malloc(sizeof(int) * (size_t)(nrow + 1));
        b->a_row = (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
        b->a_col = (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
        b->a_val = (double *)malloc(sizeof(double) * (size_t)(na > 0 ? na : 1));
        int r = -1;
        for (int64_t t = 0; t < na; ++t) {
            const ent_t *x = &e[beg + cn + t];
            if (t == 0 || x->mat != x[-1].mat) { ++r; b->row_idx[r] = x->mat - 1; b->a_ptr[r] = (int)t; }
            b->a_row[t] = x->row; b->a_col[t] = x->col; b->a_val[t] = x->val;
        }
        b->a_ptr[nrow] = (int)na;
        if (block_presolve(b, m)) { lrd_problem_free(p); return 3; }
    }
    problem_norms(p);
    *out = p;
    return 0;
}

int lrd_problem_from_triplets(int m, const double *b, int nblk, const int *dims, int64_t nent, const int *e_mat,
                              const int *e_blk, const int *e_row, const int *e_col, const double *e_val,
                              lrd_problem **out) {
    ent_t *e = (ent_t *)malloc(sizeof(ent_t) * (size_t)(nent > 0 ? nent : 1));
    for (int64_t t = 0; t < nent; ++t) {
        e[t].mat = e_mat[t]; e[t].blk = e_blk[t]; e[t].row = e_row[t]; e[t].col = e_col[t]; e[t].val = e_val[t];
        e[t].seq = t;
    }
    int rc = build_problem(m, b, nblk, dims, e, nent, out);
    free(e);
    return rc;
}

/* ---- SDPA sparse format ----
 * The file is read in one piece and scanned in place.  Entry lines ("mat blk i j value", the bulk of any file) go through a parser of
 * their own instead of sscanf: four decimal integers and a value whose conversion is EXACTLY strtod's -- a decimal token of at most
 * 19 significant digits whose mantissa fits 2^53 and whose power of ten is within 10^+-22 is one exact integer times or over one exact
 * power of ten, i.e. one correctly rounded IEEE operation (Clinger's fast path); everything else (longer mantissas, big exponents,
 * inf / nan / hex) is handed to strtod itself.  Same values bit for bit, ~4x less time per line (io/lorads_file_io.c:21-293 uses fscanf). */
typedef struct {
    char *p, *end; /* cursor and one past the last byte of the file (the byte at `end` is a NUL we own) */
} scan_t;

/* next line, NUL-terminated in place (the '\n' is overwritten); NULL at the end of the file */
static char *next_line(scan_t *sc) {
    if (sc->p >= sc->end) return NULL;
    char *ln = sc->p;
    char *nl = (char *)memchr(ln, '\n', (size_t)(sc->end - ln));
    if (nl) { *nl = 0; sc->p = nl + 1; }
    else sc->p = sc->end;
    return ln;
}

/* numbers separated by anything that is not part of a number ({ } ( ) , ' and blanks) */
static int scan_double(const char **sp, double *out);
static int next_number(char **s, double *v) {
    char *c = *s;
    while (*c && !(isdigit((unsigned char)*c) || *c == '-' || *c == '+' || *c == '.')) ++c;
    if (!*c) { *s = c; return 0; }
    const char *end = c;
    if (!scan_double(&end, v)) { *s = c + 1; return next_number(s, v); }
    *s = (char *)end;
    return 1;
}

static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20,
                               1e21, 1e22};

/* "%d": optional blanks, optional sign, digits.  Returns 0 when there is no integer here (as a failed sscanf field) */
static int scan_int(const char **sp, int *out) {
    const char *s = *sp;
    while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\v' || *s == '\f') ++s;
    int neg = 0;
    if (*s == '-') { neg = 1; ++s; }
    else if (*s == '+') ++s;
    if (!isdigit((unsigned char)*s)) return 0;
    long long v = 0;
    while (isdigit((unsigned char)*s)) { v = v * 10 + (*s - '0'); if (v > 4000000000ll) return 0; ++s; }
    *out = (int)(neg ? -v : v);
    *sp = s;
    return 1;
}

/* "%lg": the value of the token at *sp exactly as strtod converts it.  Returns 0 when there is no number */
static int scan_double(const char **sp, double *out) {
    const char *s = *sp;
    while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\v' || *s == '\f') ++s;
    const char *t = s;
    int neg = 0;
    if (*t == '-') { neg = 1; ++t; }
    else if (*t == '+') ++t;
    unsigned long long mant = 0;
    int nsig = 0, e10 = 0, ndig = 0, fast = 1;
    while (isdigit((unsigned char)*t)) {
        if (nsig < 19) { mant = mant * 10 + (unsigned)(*t - '0'); if (mant) ++nsig; }
        else fast = 0; /* (a 20th significant digit: strtod decides the rounding) */
        ++t; ++ndig;
    }
    if (*t == '.') {
        ++t;
        while (isdigit((unsigned char)*t)) {
            if (nsig < 19) { mant = mant * 10 + (unsigned)(*t - '0'); if (mant) ++nsig; --e10; }
            else fast = 0;
            ++t; ++ndig;
        }
    }
    if (ndig > 0 && (*t == 'e' || *t == 'E')) {
        const char *u = t + 1;
        int eneg = 0, ev = 0, ed = 0;
        if (*u == '-') { eneg = 1; ++u; }
        else if (*u == '+') ++u;
        while (isdigit((unsigned char)*u)) { if (ev < 100000) ev = ev * 10 + (*u - '0'); ++u; ++ed; }
        if (ed) { e10 += eneg ? -ev : ev; t = u; }
    }
    /* the short way only for a plain decimal token that ends where a number ends */
    if (fast && ndig > 0 && !isalnum((unsigned char)*t) && *t != '.' && mant <= (1ull << 53) && e10 >= -22 && e10 <= 22) {
        double v = (double)mant;
        if (e10 < 0) v /= P10[-e10];
        else v *= P10[e10];
        *out = neg ? -v : v;
        *sp = t;
        return 1;
    }
    char *end;
    const double v = strtod(s, &end);
    if (end == s) return 0;
    *out = v;
    *sp = end;
    return 1;
}

/* one entry line: 5 = all fields there (what sscanf("%d %d %d %d %lg") returns for it), less otherwise.  Exported for the tests */
int lrd_parse_entry_line(const char *line, int ij[4], double *val) {
    const char *s = line;
    for (int k = 0; k < 4; ++k)
        if (!scan_int(&s, &ij[k])) return k;
    return scan_double(&s, val) ? 5 : 4;
}

int lrd_read_sdpa(const char *fname, lrd_problem **out) {
    FILE *f = fopen(fname, "rb");
    if (!f) return 1;
    char *ln, *text = NULL;
    int rc = 2, m = -1, nblk = -1, *dims = NULL;
    double *b = NULL, v;
    ent_t *e = NULL;
    int64_t ne = 0, ecap = 0;
    size_t len = 0, cap = (size_t)1 << 20;
    text = (char *)malloc(cap + 1);
    for (;;) { /* (grows by doubling: works for pipes too) */
        const size_t got = text ? fread(text + len, 1, cap - len, f) : 0;
        len += got;
        if (!text || got == 0) break;
        if (len == cap) { cap *= 2; text = (char *)realloc(text, cap + 1); }
    }
    fclose(f);
    if (!text) return 2;
    text[len] = 0;
    for (size_t i = 0; i < len; ++i) /* a stray NUL would end a line early: the old line reader stopped at it too; make it a blank */
        if (!text[i]) text[i] = ' ';
    scan_t sc = {text, text + len};
    const int timing = getenv("LORADS_HOST_TIMING") != NULL; /* stage times of the reader on stderr */
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int use_sscanf = getenv("LORADS_READER") && !strcmp(getenv("LORADS_READER"), "sscanf"); /* (the former per-line conversion: tests) */
    /* header: skip comment lines (first char '*' or '"') */
    do { ln = next_line(&sc); } while (ln && (ln[0] == '*' || ln[0] == '"'));
    if (!ln) goto done;
    { char *s = ln; if (!next_number(&s, &v)) goto done; m = (int)v; }
    ln = next_line(&sc);
    if (!ln) goto done;
    { char *s = ln; if (!next_number(&s, &v)) goto done; nblk = (int)v; }
    if (m < 0 || nblk <= 0) goto done;
    dims = (int *)malloc(sizeof(int) * (size_t)nblk);
    { int got = 0;
      while (got < nblk) {
          ln = next_line(&sc);
          if (!ln) goto done;
          char *s = ln;
          while (got < nblk && next_number(&s, &v)) dims[got++] = (int)v;
      } }
    for (int k = 0; k < nblk; ++k) /* one diagonal (LP) block, and only at the end, as the reference (:120-124) */
        if (dims[k] == 0 || (dims[k] < 0 && k != nblk - 1)) { rc = 4; goto done; }
    b = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    { int got = 0;
      while (got < m) {
          ln = next_line(&sc);
          if (!ln) goto done;
          char *s = ln;
          while (got < m && next_number(&s, &v)) b[got++] = v;
      } }
    while ((ln = next_line(&sc)) != NULL) {
        int ij[4];
        double val;
        const int nf = use_sscanf ? sscanf(ln, "%d %d %d %d %lg", &ij[0], &ij[1], &ij[2], &ij[3], &val) : lrd_parse_entry_line(ln, ij, &val);
        if (nf != 5) {
            char *s = ln;
            while (*s && isspace((unsigned char)*s)) ++s;
            if (!*s) continue;
            break; /* trailing comment section */
        }
        if (ne == ecap) { ecap = ecap ? 2 * ecap : 1 << 16; e = (ent_t *)realloc(e, sizeof(ent_t) * (size_t)ecap); }
        e[ne].mat = ij[0]; e[ne].blk = ij[1] - 1; e[ne].row = ij[2] - 1; e[ne].col = ij[3] - 1; e[ne].val = val; e[ne].seq = ne;
        ++ne;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    rc = build_problem(m, b, nblk, dims, e, ne, out);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (timing)
        fprintf(stderr, "lorads_host: reader: %lld entry lines scanned in %.1f ms, problem image (sort, cones, pre-solve) %.1f ms\n", (long long)ne,
                1e3 * (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_nsec - t0.tv_nsec),
                1e3 * (double)(t2.tv_sec - t1.tv_sec) + 1e-6 * (double)(t2.tv_nsec - t1.tv_nsec));
done:
    free(text); free(dims); free(b); free(e);
    return rc;
}

/* FNV-1a over everything the image holds (dimensions, every array, the norms): two images with the same digest are the same
 * problem (used by the tests to compare the reader's two conversion paths) */
static uint64_t fnv(uint64_t h, const void *data, size_t bytes) {
    const unsigned char *c = (const unsigned char *)data;
    for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
uint64_t lrd_problem_digest(const lrd_problem *p) {
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, &p->m, sizeof p->m);
    h = fnv(h, &p->nblk, sizeof p->nblk);
    h = fnv(h, p->b, sizeof(double) * (size_t)p->m);
    const double nr[6] = {p->cObjNrm1, p->cObjNrm2, p->cObjNrmInf, p->bNrm1, p->bNrm2, p->bNrmInf};
    h = fnv(h, nr, sizeof nr);
    for (int k = 0; k < p->nblk; ++k) {
        const lrd_block *b = &p->blk[k];
        const int hd[8] = {b->n, b->nrow, b->c_nnz, b->cone_sparse, b->dense_mode, b->np, b->is_lp, b->global_id};
        h = fnv(h, hd, sizeof hd);
        const size_t na = (size_t)b->a_ptr[b->nrow];
        h = fnv(h, b->row_idx, sizeof(int) * (size_t)b->nrow);
        h = fnv(h, b->a_ptr, sizeof(int) * ((size_t)b->nrow + 1));
        h = fnv(h, b->a_row, sizeof(int) * na);
        h = fnv(h, b->a_col, sizeof(int) * na);
        h = fnv(h, b->a_val, sizeof(double) * na);
        h = fnv(h, b->c_row, sizeof(int) * (size_t)b->c_nnz);
        h = fnv(h, b->c_col, sizeof(int) * (size_t)b->c_nnz);
        h = fnv(h, b->c_val, sizeof(double) * (size_t)b->c_nnz);
        h = fnv(h, b->p_row, sizeof(int) * (size_t)b->np);
        h = fnv(h, b->p_col, sizeof(int) * (size_t)b->np);
        h = fnv(h, b->a_pidx, sizeof(int) * na);
        h = fnv(h, b->c_pidx, sizeof(int) * (size_t)b->c_nnz);
    }
    return h;
}

void lrd_problem_select(lrd_problem *p, const int *keep) {
    int w = 0;
    for (int k = 0; k < p->nblk; ++k) {
        if (keep[k]) { if (w != k) { p->blk[w] = p->blk[k]; memset(&p->blk[k], 0, sizeof(lrd_block)); } ++w; }
        else block_free(&p->blk[k]);
    }
    p->nblk = w;
}

/* Sharded cones whose constraints are block-separable over the ranks (no constraint touches cones of two ranks; cone k lives on
 * rank k % world): every m-vector of the method -- constrValSum, lambda, q1, q2 -- then splits into per-rank pieces nobody else
 * reads, and the ranks only share SCALARS.  keep[] = the cones of this rank.  Returns 1 and rewrites the image into the rank's own
 * sub-problem (its constraints renumbered 0..m_local-1 in ascending global order, b cut to them, row indices remapped; a constraint
 * no cone touches goes to rank 0; the norms stay those of the whole problem) when the deal is separable; returns 0 and leaves
 * the image alone when some constraint is shared.  Call BEFORE lrd_problem_select (it looks at every cone of the file). */
int lrd_problem_localize(lrd_problem *p, int world, int rank_id) {
    const int m = p->m;
    int *owner = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) owner[i] = -1;
    int shared = 0;
    for (int k = 0; k < p->nblk && !shared; ++k) {
        const lrd_block *b = &p->blk[k];
        const int rk = k % world;
        for (int t = 0; t < b->nrow; ++t) {
            const int i = b->row_idx[t];
            if (owner[i] >= 0 && owner[i] != rk) { shared = 1; break; }
            owner[i] = rk;
        }
    }
    if (shared) { free(owner); return 0; }
    int *loc = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int ml = 0;
    for (int i = 0; i < m; ++i) {
        const int o = owner[i] < 0 ? 0 : owner[i];
        loc[i] = o == rank_id ? ml++ : -1;
    }
    int *glob = (int *)malloc(sizeof(int) * (size_t)(ml > 0 ? ml : 1));
    double *bl = (double *)malloc(sizeof(double) * (size_t)(ml > 0 ? ml : 1));
    for (int i = 0; i < m; ++i)
        if (loc[i] >= 0) { glob[loc[i]] = i; bl[loc[i]] = p->b[i]; }
    for (int k = 0; k < p->nblk; ++k) {
        if (k % world != rank_id) continue;
        lrd_block *b = &p->blk[k];
        for (int t = 0; t < b->nrow; ++t) b->row_idx[t] = loc[b->row_idx[t]]; /* (ascending order is kept) */
    }
    free(p->b);
    p->b = bl;
    p->m_global = m;
    p->m = ml;
    p->con_global = glob;
    p->separable = 1;
    free(owner); free(loc);
    return 1;
}

void lrd_determine_rank(lrd_problem *p, double times) {
    for (int k = 0; k < p->nblk; ++k) {
        lrd_block *b = &p->blk[k];
        if (b->is_lp) { b->rank = b->rank_max = 1; continue; }
        int nnz_rows = b->nrow;
        int cap = (int)sqrt(2.0 * nnz_rows) + 1;
        if (cap > b->n) cap = b->n;
        int r;
        if (times <= 1e-6) r = cap;
        else if (nnz_rows / b->n >= 20 && b->n <= 400 && p->nsdp_global <= 3) r = cap;
        else {
            double lr = ceil(times * log((double)b->n));
            r = (int)(lr < (double)cap ? lr : (double)cap);
        }
        if (r < 1) r = 1;
        b->rank = r;
        b->rank_max = cap;
    }
}

/* The start point is the reference's: srand(925), then two rand() draws per element, R of every cone first, then U and V cone by
 * cone (data/lorads_solver.c:361-371,415,652-653).  glibc's rand() is the TYPE_3 additive-feedback generator behind a lock
 * (r[i] = r[i-3] + r[i-31] mod 2^32, output r[i] >> 1, state seeded by the Lehmer generator 16807 mod 2^31 - 1 and run 310 steps): the
 * same recurrence inline draws the 2 x 3 x n x r numbers of cfg5 in 0.03 s instead of 0.19.  It is used only when its first 4096
 * outputs equal this C library's rand() -- on a platform whose rand() is another generator the library's own is called, as the
 * reference would there. */
typedef struct {
    uint32_t r[34];
    int i; /* next write position in the ring of 34 (holds r[k-31..k-1] and r[k-3]) */
    int use_libc;
} rng_t;
static uint32_t rng_next_raw(rng_t *g) { /* one step of r[k] = r[k-31] + r[k-3] on a ring of 31 */
    const int k = g->i;
    const uint32_t v = g->r[k] + g->r[(k + 28) % 31]; /* r[k-31] sits in slot k, r[k-3] in slot k-3 = k+28 mod 31 */
    g->r[k] = v;
    g->i = (k + 1) % 31;
    return v;
}
static void rng_seed_glibc(rng_t *g, unsigned seed) {
    int32_t st[34];
    st[0] = (int32_t)(seed ? seed : 1);
    for (int i = 1; i < 31; ++i) {
        const long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        st[i] = (int32_t)w;
    }
    /* glibc keeps r[0..30] in a ring with the front pointer at 3 and the rear at 0, and discards 310 outputs */
    for (int i = 0; i < 31; ++i) g->r[i] = (uint32_t)st[i];
    /* in sequence terms: r[31 + j] = r[j] for j = 0..2 (front starts three ahead), then the recurrence.  Rotate so that slot 0 holds the
     * oldest value the recurrence needs: after the three copies the window is r[3..33] = st[3..30], st[0..2] */
    uint32_t win[31];
    for (int i = 0; i < 28; ++i) win[i] = (uint32_t)st[i + 3];
    for (int i = 0; i < 3; ++i) win[28 + i] = (uint32_t)st[i];
    memcpy(g->r, win, sizeof win);
    g->i = 0;
    for (int i = 0; i < 310; ++i) (void)rng_next_raw(g);
}
static void rng_init(rng_t *g, unsigned seed) {
    g->use_libc = 0;
    rng_seed_glibc(g, seed);
    rng_t probe = *g;
    srand(seed);
    for (int i = 0; i < 4096 && !g->use_libc; ++i)
        if ((int)(rng_next_raw(&probe) >> 1) != rand()) g->use_libc = 1;
    if (getenv("LORADS_LIBC_RAND")) g->use_libc = 1; /* (tests) */
    srand(seed); /* (the library's generator starts over for the path that uses it) */
}
static inline int rng_rand(rng_t *g) { return g->use_libc ? rand() : (int)(rng_next_raw(g) >> 1); }

/* 1: the inline generator reproduces this C library's rand() (and is the one lrd_init_point draws from); 0: rand() itself is used */
int lrd_start_generator_is_inline(void) {
    rng_t g;
    rng_init(&g, 925);
    return !g.use_libc;
}

int lrd_init_point(const lrd_problem *p, double ***Rp, double ***Up, double ***Vp) {
    int nb = p->nblk;
    double **R = (double **)calloc((size_t)nb, sizeof(double *));
    double **U = (double **)calloc((size_t)nb, sizeof(double *));
    double **V = (double **)calloc((size_t)nb, sizeof(double *));
    rng_t g;
    rng_init(&g, 925);
    for (int k = 0; k < nb; ++k) {
        size_t cnt = (size_t)p->blk[k].n * p->blk[k].rank;
        R[k] = (double *)malloc(sizeof(double) * cnt);
        for (size_t i = 0; i < cnt; ++i) {
            double x = (double)rng_rand(&g) / RAND_MAX;
            x -= (double)rng_rand(&g) / RAND_MAX;
            R[k][i] = x;
        }
    }
    for (int k = 0; k < nb; ++k) {
        size_t cnt = (size_t)p->blk[k].n * p->blk[k].rank;
        U[k] = (double *)malloc(sizeof(double) * cnt);
        V[k] = (double *)malloc(sizeof(double) * cnt);
        for (size_t i = 0; i < cnt; ++i) { double x = (double)rng_rand(&g) / RAND_MAX; x -= (double)rng_rand(&g) / RAND_MAX; U[k][i] = x; }
        for (size_t i = 0; i < cnt; ++i) { double x = (double)rng_rand(&g) / RAND_MAX; x -= (double)rng_rand(&g) / RAND_MAX; V[k][i] = x; }
    }
    *Rp = R; *Up = U; *Vp = V;
    return 0;
}

void lrd_free_point(int nblk, double **R, double **U, double **V) {
    for (int k = 0; k < nblk; ++k) {
        if (R) free(R[k]);
        if (U) free(U[k]);
        if (V) free(V[k]);
    }
    free(R); free(U); free(V);
}
