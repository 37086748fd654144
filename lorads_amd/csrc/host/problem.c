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

static int pos_cmp(const void *pa, const void *pb) {
    const pos_t *a = (const pos_t *)pa, *b = (const pos_t *)pb;
    if (a->col != b->col) return a->col < b->col ? -1 : 1;
    return (a->row > b->row) - (a->row < b->row);
}

static int pos_find(const pos_t *u, int np, int row, int col) {
    int lo = 0, hi = np - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (u[mid].col < col || (u[mid].col == col && u[mid].row < row)) lo = mid + 1;
        else if (u[mid].col == col && u[mid].row == row) return mid;
        else hi = mid - 1;
    }
    return -1;
}

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
    if (!dense) {
        int tot = b->c_nnz + na;
        u = (pos_t *)malloc(sizeof(pos_t) * (size_t)(tot > 0 ? tot : 1));
        for (int k = 0; k < b->c_nnz; ++k) { u[k].row = b->c_row[k]; u[k].col = b->c_col[k]; }
        for (int k = 0; k < na; ++k) { u[b->c_nnz + k].row = b->a_row[k]; u[b->c_nnz + k].col = b->a_col[k]; }
        qsort(u, (size_t)tot, sizeof(pos_t), pos_cmp);
        for (int k = 0; k < tot; ++k)
            if (k == 0 || u[k].row != u[k - 1].row || u[k].col != u[k - 1].col) u[np++] = u[k];
        if ((double)np / (double)npack >= 0.1 && !b->is_lp) { dense = 1; free(u); u = NULL; }
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
        for (int k = 0; k < na; ++k) b->a_pidx[k] = pos_find(u, np, b->a_row[k], b->a_col[k]);
        for (int k = 0; k < b->c_nnz; ++k) b->c_pidx[k] = pos_find(u, np, b->c_row[k], b->c_col[k]);
        free(u);
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
    qsort(e, (size_t)ne, sizeof(ent_t), ent_cmp);
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

/* ---- SDPA sparse format ---- */
static char *read_line_dyn(FILE *f, char **buf, size_t *cap) {
    size_t len = 0;
    int ch;
    if (!*buf) { *cap = 4096; *buf = (char *)malloc(*cap); }
    while ((ch = fgetc(f)) != EOF) {
        if (len + 2 > *cap) { *cap *= 2; *buf = (char *)realloc(*buf, *cap); }
        (*buf)[len++] = (char)ch;
        if (ch == '\n') break;
    }
    if (len == 0 && ch == EOF) return NULL;
    (*buf)[len] = 0;
    return *buf;
}

/* numbers separated by anything that is not part of a number ({ } ( ) , ' and blanks) */
static int next_number(char **s, double *v) {
    char *c = *s;
    while (*c && !(isdigit((unsigned char)*c) || *c == '-' || *c == '+' || *c == '.')) ++c;
    if (!*c) { *s = c; return 0; }
    char *end;
    *v = strtod(c, &end);
    if (end == c) { *s = c + 1; return next_number(s, v); }
    *s = end;
    return 1;
}

int lrd_read_sdpa(const char *fname, lrd_problem **out) {
    FILE *f = fopen(fname, "r");
    if (!f) return 1;
    char *buf = NULL;
    size_t cap = 0;
    char *ln;
    int rc = 2, m = -1, nblk = -1, *dims = NULL;
    double *b = NULL, v;
    ent_t *e = NULL;
    int64_t ne = 0, ecap = 0;
    /* header: skip comment lines (first char '*' or '"') */
    do { ln = read_line_dyn(f, &buf, &cap); } while (ln && (ln[0] == '*' || ln[0] == '"'));
    if (!ln) goto done;
    { char *s = ln; if (!next_number(&s, &v)) goto done; m = (int)v; }
    ln = read_line_dyn(f, &buf, &cap);
    if (!ln) goto done;
    { char *s = ln; if (!next_number(&s, &v)) goto done; nblk = (int)v; }
    if (m < 0 || nblk <= 0) goto done;
    dims = (int *)malloc(sizeof(int) * (size_t)nblk);
    { int got = 0;
      while (got < nblk) {
          ln = read_line_dyn(f, &buf, &cap);
          if (!ln) goto done;
          char *s = ln;
          while (got < nblk && next_number(&s, &v)) dims[got++] = (int)v;
      } }
    for (int k = 0; k < nblk; ++k) /* one diagonal (LP) block, and only at the end, as the reference (:120-124) */
        if (dims[k] == 0 || (dims[k] < 0 && k != nblk - 1)) { rc = 4; goto done; }
    b = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    { int got = 0;
      while (got < m) {
          ln = read_line_dyn(f, &buf, &cap);
          if (!ln) goto done;
          char *s = ln;
          while (got < m && next_number(&s, &v)) b[got++] = v;
      } }
    while ((ln = read_line_dyn(f, &buf, &cap)) != NULL) {
        int mat, blk, i, j;
        double val;
        if (sscanf(ln, "%d %d %d %d %lg", &mat, &blk, &i, &j, &val) != 5) {
            char *s = ln;
            while (*s && isspace((unsigned char)*s)) ++s;
            if (!*s) continue;
            break; /* trailing comment section */
        }
        if (ne == ecap) { ecap = ecap ? 2 * ecap : 1 << 16; e = (ent_t *)realloc(e, sizeof(ent_t) * (size_t)ecap); }
        e[ne].mat = mat; e[ne].blk = blk - 1; e[ne].row = i - 1; e[ne].col = j - 1; e[ne].val = val; e[ne].seq = ne;
        ++ne;
    }
    rc = build_problem(m, b, nblk, dims, e, ne, out);
done:
    fclose(f);
    free(buf); free(dims); free(b); free(e);
    return rc;
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

int lrd_init_point(const lrd_problem *p, double ***Rp, double ***Up, double ***Vp) {
    int nb = p->nblk;
    double **R = (double **)calloc((size_t)nb, sizeof(double *));
    double **U = (double **)calloc((size_t)nb, sizeof(double *));
    double **V = (double **)calloc((size_t)nb, sizeof(double *));
    srand(925);
    for (int k = 0; k < nb; ++k) {
        size_t cnt = (size_t)p->blk[k].n * p->blk[k].rank;
        R[k] = (double *)malloc(sizeof(double) * cnt);
        for (size_t i = 0; i < cnt; ++i) {
            double x = (double)rand() / RAND_MAX;
            x -= (double)rand() / RAND_MAX;
            R[k][i] = x;
        }
    }
    for (int k = 0; k < nb; ++k) {
        size_t cnt = (size_t)p->blk[k].n * p->blk[k].rank;
        U[k] = (double *)malloc(sizeof(double) * cnt);
        V[k] = (double *)malloc(sizeof(double) * cnt);
        for (size_t i = 0; i < cnt; ++i) { double x = (double)rand() / RAND_MAX; x -= (double)rand() / RAND_MAX; U[k][i] = x; }
        for (size_t i = 0; i < cnt; ++i) { double x = (double)rand() / RAND_MAX; x -= (double)rand() / RAND_MAX; V[k][i] = x; }
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
