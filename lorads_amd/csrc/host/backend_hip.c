/* backend_hip.c -- wires the host's operator table (lrd_backend) to the HIP C-ABI library
 * (include/lorads_hip.h).  The library is loaded with dlopen so that the plain-C host has no link-time
 * dependency on ROCm; if the library, a symbol or a GPU is missing this FAILS -- the product has no
 * CPU path behind this table. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lorads_hip.h"
#include "lorads_host.h"

typedef struct {
    void *dl;
    lorads_hip_ctx *ctx;
    lrd_allreduce_fn ar;
    void *ar_user;
    /* entry points */
    int (*create)(const lorads_hip_problem *, lorads_hip_ctx **);
    void (*destroy)(lorads_hip_ctx *);
    const char *(*last_error)(void);
    int (*init_constr)(lorads_hip_ctx *, int32_t);
    int (*alm_cal_grad)(lorads_hip_ctx *, double, double *);
    int (*lbfgs_direction)(lorads_hip_ctx *, int32_t);
    int (*alm_q12p12)(lorads_hip_ctx *, double *);
    int (*alm_linesearch_coeffs)(lorads_hip_ctx *, double, double, double, double *);
    int (*set_y_as_neg_grad)(lorads_hip_ctx *);
    int (*alm_update_var)(lorads_hip_ctx *, double);
    int (*set_lbfgs_his_two)(lorads_hip_ctx *, double);
    int (*update_dimacs)(lorads_hip_ctx *, int32_t, double *);
    int (*cal_obj)(lorads_hip_ctx *, int32_t, double *);
    int (*admm_update_var)(lorads_hip_ctx *, double, double, int32_t, int32_t *);
    int (*update_dual_var)(lorads_hip_ctx *, double);
    int (*cal_dual_obj)(lorads_hip_ctx *, double *);
    int (*alm_to_admm)(lorads_hip_ctx *);
    int (*average_uv_to_v)(lorads_hip_ctx *);
    int (*scale_obj)(lorads_hip_ctx *, double);
    int (*resize_rank)(lorads_hip_ctx *, const int32_t *);
    int (*set_mat)(lorads_hip_ctx *, int32_t, int32_t, const double *);
    int (*get_mat)(lorads_hip_ctx *, int32_t, int32_t, double *);
    int (*set_vec)(lorads_hip_ctx *, int32_t, const double *);
    int (*get_vec)(lorads_hip_ctx *, int32_t, double *);
    int (*set_allreduce)(lorads_hip_ctx *, lorads_hip_allreduce_fn, void *);
    int (*admm_step)(lorads_hip_ctx *, double, double, int32_t, double *);
    int (*dual_infeasibility)(lorads_hip_ctx *, double, int32_t, int32_t, double *, double *, int32_t *);
    int (*alm_front)(lorads_hip_ctx *, double, int32_t, double *);
    int (*alm_step)(lorads_hip_ctx *, double, double, int32_t, double *);
} hipbe;

#define H ((hipbe *)cx)
static int report(hipbe *h, int rc, const char *what) {
    if (rc) fprintf(stderr, "lorads_hip: %s failed: %s\n", what, h->last_error ? h->last_error() : "?");
    return rc;
}
static int b_init_constr(void *cx, int pair) { return report(H, H->init_constr(H->ctx, pair), "init_constr"); }
static int b_alm_cal_grad(void *cx, double rho, double *lag) { return report(H, H->alm_cal_grad(H->ctx, rho, lag), "alm_cal_grad"); }
static int b_lbfgs_direction(void *cx, int it) { return report(H, H->lbfgs_direction(H->ctx, it), "lbfgs_direction"); }
static int b_alm_q12p12(void *cx, double p[2]) { return report(H, H->alm_q12p12(H->ctx, p), "alm_q12p12"); }
static int b_ls(void *cx, double rho, double p1, double p2, double k[4]) {
    return report(H, H->alm_linesearch_coeffs(H->ctx, rho, p1, p2, k), "alm_linesearch_coeffs");
}
static int b_set_y(void *cx) { return report(H, H->set_y_as_neg_grad(H->ctx), "set_y_as_neg_grad"); }
static int b_upd(void *cx, double tau) { return report(H, H->alm_update_var(H->ctx, tau), "alm_update_var"); }
static int b_his(void *cx, double tau) { return report(H, H->set_lbfgs_his_two(H->ctx, tau), "set_lbfgs_his_two"); }
static int b_dimacs(void *cx, int pair, double *e) { return report(H, H->update_dimacs(H->ctx, pair, e), "update_dimacs"); }
static int b_obj(void *cx, int pair, double *v) { return report(H, H->cal_obj(H->ctx, pair, v), "cal_obj"); }
static int b_admm(void *cx, double rho, double tol, int mx, int *it) {
    int32_t n = 0;
    int rc = report(H, H->admm_update_var(H->ctx, rho, tol, mx, &n), "admm_update_var");
    *it = n;
    return rc;
}
static int b_step(void *cx, double rho, double tol, int mx, double o[4]) { return report(H, H->admm_step(H->ctx, rho, tol, mx, o), "admm_step"); }
/* ARPACK parameters of dual_infeasible (data/lorads_sdp_conic.c:1288-1319): tol 1e-2, ncv 40, 600 restarts */
static int b_afront(void *cx, double rho, int inner, double o[6]) { return report(H, H->alm_front(H->ctx, rho, inner, o), "alm_front"); }
static int b_astep(void *cx, double rho, double tau, int nx, double o[8]) { return report(H, H->alm_step(H->ctx, rho, tau, nx, o), "alm_step"); }
static int b_dinf(void *cx, double *v) { return report(H, H->dual_infeasibility(H->ctx, 1e-2, 40, 600, v, NULL, NULL), "dual_infeasibility"); }
static int b_dual(void *cx, double rho) { return report(H, H->update_dual_var(H->ctx, rho), "update_dual_var"); }
static int b_dobj(void *cx, double *v) { return report(H, H->cal_dual_obj(H->ctx, v), "cal_dual_obj"); }
static int b_a2a(void *cx) { return report(H, H->alm_to_admm(H->ctx), "alm_to_admm"); }
static int b_avg(void *cx) { return report(H, H->average_uv_to_v(H->ctx), "average_uv_to_v"); }
static int b_scale(void *cx, double s) { return report(H, H->scale_obj(H->ctx, s), "scale_obj"); }
static int b_resize(void *cx, const int *nr) { return report(H, H->resize_rank(H->ctx, (const int32_t *)nr), "resize_rank"); }
static int b_set_mat(void *cx, int w, int k, const double *a) { return report(H, H->set_mat(H->ctx, w, k, a), "set_mat"); }
static int b_get_mat(void *cx, int w, int k, double *a) { return report(H, H->get_mat(H->ctx, w, k, a), "get_mat"); }
static int b_set_vec(void *cx, int w, const double *a) { return report(H, H->set_vec(H->ctx, w, a), "set_vec"); }
static int b_get_vec(void *cx, int w, double *a) { return report(H, H->get_vec(H->ctx, w, a), "get_vec"); }
static int ar_tramp(void *user, double *buf, int32_t count, int32_t on_device) {
    hipbe *h = (hipbe *)user;
    return h->ar ? h->ar(h->ar_user, buf, count, on_device) : 0;
}
static int b_set_ar(void *cx, lrd_allreduce_fn fn, void *user) {
    H->ar = fn;
    H->ar_user = user;
    return H->set_allreduce(H->ctx, fn ? ar_tramp : NULL, H);
}
static void b_destroy(void *cx) {
    if (H->ctx) H->destroy(H->ctx);
    /* the library stays loaded for the life of the process (HIP runtime teardown order) */
    free(cx);
}
#undef H

/* raw context for callers that need the measurement hooks (bench.py) */
void *lrd_hip_backend_raw_ctx(const lrd_backend *be) { return be && be->ctx ? ((hipbe *)be->ctx)->ctx : NULL; }

int lrd_hip_backend_create(const lrd_problem *p, int lbfgs_len, const char *libpath, lrd_backend *out) {
    hipbe *h = (hipbe *)calloc(1, sizeof *h);
    h->dl = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!h->dl) {
        fprintf(stderr, "lorads: cannot load the HIP backend %s: %s\n", libpath, dlerror());
        free(h);
        return 10;
    }
#define SYM(field, name)                                                             \
    do {                                                                             \
        *(void **)(&h->field) = dlsym(h->dl, "lorads_hip_" name);                    \
        if (!h->field) {                                                             \
            fprintf(stderr, "lorads: %s lacks symbol lorads_hip_%s\n", libpath, name); \
            free(h);                                                                 \
            return 11;                                                               \
        }                                                                            \
    } while (0)
    SYM(create, "create"); SYM(destroy, "destroy"); SYM(last_error, "last_error"); SYM(init_constr, "init_constr");
    SYM(alm_cal_grad, "alm_cal_grad"); SYM(lbfgs_direction, "lbfgs_direction"); SYM(alm_q12p12, "alm_q12p12");
    SYM(alm_linesearch_coeffs, "alm_linesearch_coeffs"); SYM(set_y_as_neg_grad, "set_y_as_neg_grad");
    SYM(alm_update_var, "alm_update_var"); SYM(set_lbfgs_his_two, "set_lbfgs_his_two"); SYM(update_dimacs, "update_dimacs");
    SYM(cal_obj, "cal_obj"); SYM(admm_update_var, "admm_update_var"); SYM(update_dual_var, "update_dual_var");
    SYM(cal_dual_obj, "cal_dual_obj"); SYM(alm_to_admm, "alm_to_admm"); SYM(average_uv_to_v, "average_uv_to_v");
    SYM(scale_obj, "scale_obj"); SYM(resize_rank, "resize_rank"); SYM(set_mat, "set_mat"); SYM(get_mat, "get_mat");
    SYM(set_vec, "set_vec"); SYM(get_vec, "get_vec"); SYM(set_allreduce, "set_allreduce"); SYM(admm_step, "admm_step");
    SYM(dual_infeasibility, "dual_infeasibility"); SYM(alm_front, "alm_front"); SYM(alm_step, "alm_step");
#undef SYM
    lorads_hip_block *hb = (lorads_hip_block *)calloc((size_t)(p->nblk > 0 ? p->nblk : 1), sizeof *hb);
    for (int k = 0; k < p->nblk; ++k) {
        const lrd_block *b = &p->blk[k];
        hb[k].n = b->n; hb[k].rank = b->rank; hb[k].nrow = b->nrow; hb[k].row_idx = b->row_idx; hb[k].a_ptr = b->a_ptr;
        hb[k].a_row = b->a_row; hb[k].a_col = b->a_col; hb[k].a_val = b->a_val; hb[k].c_nnz = b->c_nnz;
        hb[k].c_row = b->c_row; hb[k].c_col = b->c_col; hb[k].c_val = b->c_val;
        hb[k].is_lp = b->is_lp;
    }
    lorads_hip_problem hp;
    memset(&hp, 0, sizeof hp);
    hp.m = p->m; hp.b = p->b; hp.b_nrm1 = p->bNrm1; hp.nblocks = p->nblk; hp.blocks = hb; hp.lbfgs_len = lbfgs_len;
    hp.device = -1;
    int rc = h->create(&hp, &h->ctx);
    free(hb);
    if (rc) {
        fprintf(stderr, "lorads: lorads_hip_create failed: %s\n", h->last_error());
        free(h);
        return 12;
    }
    if (p->separable) { /* this image is one rank's sub-problem of a block-separable deal: only scalars cross the ranks */
        int (*set_sep)(lorads_hip_ctx *, int32_t);
        *(void **)(&set_sep) = dlsym(h->dl, "lorads_hip_set_separable");
        if (!set_sep || set_sep(h->ctx, 1)) {
            fprintf(stderr, "lorads: %s cannot take a separable shard (lorads_hip_set_separable)\n", libpath);
            h->destroy(h->ctx);
            free(h);
            return 13;
        }
    }
    memset(out, 0, sizeof *out);
    out->ctx = h;
    out->name = "hip-gfx950";
    out->init_constr = b_init_constr; out->alm_cal_grad = b_alm_cal_grad; out->lbfgs_direction = b_lbfgs_direction;
    out->alm_q12p12 = b_alm_q12p12; out->alm_linesearch_coeffs = b_ls; out->set_y_as_neg_grad = b_set_y;
    out->alm_update_var = b_upd; out->set_lbfgs_his_two = b_his; out->update_dimacs = b_dimacs; out->cal_obj = b_obj;
    out->admm_update_var = b_admm; out->update_dual_var = b_dual; out->cal_dual_obj = b_dobj; out->alm_to_admm = b_a2a;
    out->average_uv_to_v = b_avg; out->scale_obj = b_scale; out->resize_rank = b_resize; out->set_mat = b_set_mat;
    out->get_mat = b_get_mat; out->set_vec = b_set_vec; out->get_vec = b_get_vec; out->set_allreduce = b_set_ar;
    out->destroy = b_destroy;
    out->admm_step = b_step;
    out->dual_infeasibility = b_dinf;
    out->alm_front = b_afront;
    out->alm_step = b_astep;
    return 0;
}
