/* session.c -- small C API around problem + solver used by the CLI, by bench.py and by the tests
 * (ctypes).  A session owns the problem image, the parameter block, the start point and, once a
 * backend table has been attached, the solver state. */
#include "lorads_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct lrd_session {
    lrd_params par;
    lrd_problem *prob;
    lrd_backend be;
    int have_be;
    lrd_solver sol;
    int have_sol;
    int nblk_all;          /* blocks in the file (before sharding) */
    double **R0, **U0, **V0; /* start point of the blocks this process holds */
    char fname[4096];
} lrd_session;

size_t lrd_backend_sizeof(void) { return sizeof(lrd_backend); }

lrd_session *lrd_session_open(const char *fname) {
    lrd_session *s = (lrd_session *)calloc(1, sizeof *s);
    lrd_params_default(&s->par);
    snprintf(s->fname, sizeof s->fname, "%s", fname);
    s->par.fname = s->fname;
    int rc = lrd_read_sdpa(fname, &s->prob);
    if (rc) {
        fprintf(stderr, "lorads: cannot read %s (code %d%s)\n", fname, rc, rc == 4 ? ": LP block is out of scope" : "");
        free(s);
        return NULL;
    }
    s->nblk_all = s->prob->nblk;
    return s;
}

lrd_session *lrd_session_from_triplets(int m, const double *b, int nblk, const int *dims, int64_t nent,
                                       const int *e_mat, const int *e_blk, const int *e_row, const int *e_col,
                                       const double *e_val) {
    lrd_session *s = (lrd_session *)calloc(1, sizeof *s);
    lrd_params_default(&s->par);
    snprintf(s->fname, sizeof s->fname, "<memory>");
    s->par.fname = s->fname;
    if (lrd_problem_from_triplets(m, b, nblk, dims, nent, e_mat, e_blk, e_row, e_col, e_val, &s->prob)) {
        free(s);
        return NULL;
    }
    s->nblk_all = s->prob->nblk;
    return s;
}

int lrd_session_set_param(lrd_session *s, const char *key, const char *val) { return lrd_params_set(&s->par, key, val); }
lrd_params *lrd_session_params(lrd_session *s) { return &s->par; }
lrd_problem *lrd_session_problem(lrd_session *s) { return s->prob; }
lrd_solver *lrd_session_solver(lrd_session *s) { return s->have_sol ? &s->sol : NULL; }
lrd_backend *lrd_session_backend(lrd_session *s) { return s->have_be ? &s->be : NULL; }

/* rank rule + start point + (optional) round-robin sharding of blocks over `world` processes.
 * allow_separable: the backend can work on a rank's own sub-problem and share only scalars (the HIP library:
 * lorads_hip_set_separable) -- then a block-separable deal is cut down to this rank's constraints (lrd_problem_localize) */
int lrd_session_prepare_sharded(lrd_session *s, int world, int rank_id, int allow_separable) {
    lrd_determine_rank(s->prob, s->par.timesLogRank);
    double **R, **U, **V;
    lrd_init_point(s->prob, &R, &U, &V);
    int nb = s->prob->nblk;
    if (allow_separable) lrd_problem_localize(s->prob, world > 1 ? world : 1, world > 1 ? rank_id : 0); /* (world 1: the whole problem is "its own") */
    if (world > 1) {
        int *keep = (int *)calloc((size_t)nb, sizeof(int));
        int w = 0;
        for (int k = 0; k < nb; ++k) {
            keep[k] = (k % world) == rank_id;
            if (keep[k]) { R[w] = R[k]; U[w] = U[k]; V[w] = V[k]; ++w; }
            else { free(R[k]); free(U[k]); free(V[k]); }
        }
        lrd_problem_select(s->prob, keep);
        free(keep);
    }
    s->R0 = R; s->U0 = U; s->V0 = V;
    return 0;
}

/* override the rank chosen by the rule (bench: r = 40 at n = 20000 is --timesLogRank 4.0) */
int lrd_session_block_info(lrd_session *s, int k, int *n, int *rank, int *nrow, int *na, int *nc, int *np,
                           int *dense_mode, int *cone_sparse) {
    if (k < 0 || k >= s->prob->nblk) return 1;
    const lrd_block *b = &s->prob->blk[k];
    *n = b->n; *rank = b->rank; *nrow = b->nrow; *na = b->a_ptr[b->nrow]; *nc = b->c_nnz; *np = b->np;
    *dense_mode = b->dense_mode; *cone_sparse = b->cone_sparse;
    return 0;
}
int lrd_session_prepare(lrd_session *s, int world, int rank_id) { return lrd_session_prepare_sharded(s, world, rank_id, 0); }
/* separable deal (see lrd_session_prepare_sharded)?  m_global = constraints of the file; map[i] (m entries, may be NULL) = index of
 * local constraint i in the file */
int lrd_session_separable(lrd_session *s, int *m_global, int *map) {
    const lrd_problem *p = s->prob;
    if (!p->separable) { if (m_global) *m_global = p->m; return 0; }
    if (m_global) *m_global = p->m_global;
    if (map) memcpy(map, p->con_global, sizeof(int) * (size_t)p->m);
    return 1;
}
int lrd_session_dims(lrd_session *s, int *m, int *nblk, int *nblk_global) {
    *m = s->prob->m; *nblk = s->prob->nblk; *nblk_global = s->prob->nblk_global;
    return 0;
}
const double *lrd_session_start(lrd_session *s, int which, int k) {
    double **a = which == LRD_MAT_R ? s->R0 : which == LRD_MAT_U ? s->U0 : s->V0;
    return a ? a[k] : NULL;
}

/* attach an operator table (copied), upload the start point, initialise the solver state */
int lrd_session_attach(lrd_session *s, const lrd_backend *be) {
    s->be = *be;
    s->have_be = 1;
    for (int k = 0; k < s->prob->nblk; ++k) {
        if (s->be.set_mat(s->be.ctx, LRD_MAT_R, k, s->R0[k])) return 1;
        if (s->be.set_mat(s->be.ctx, LRD_MAT_U, k, s->U0[k])) return 1;
        if (s->be.set_mat(s->be.ctx, LRD_MAT_V, k, s->V0[k])) return 1;
    }
    lrd_solver_init(&s->sol, s->prob, &s->be, &s->par);
    s->have_sol = 1;
    return 0;
}

int lrd_session_set_allreduce(lrd_session *s, lrd_allreduce_fn fn, void *user) {
    if (!s->have_sol) return 1;
    s->sol.allreduce = fn;
    s->sol.allreduce_user = user;
    return s->be.set_allreduce(s->be.ctx, fn, user);
}

int lrd_session_use_fused_step(lrd_session *s, int on) {
    if (!s->have_sol) return 1;
    s->sol.use_fused_step = on;
    return 0;
}

int lrd_session_solve(lrd_session *s) { return s->have_sol ? lrd_solve(&s->par, &s->sol) : 1; }

/* finer-grained drivers for tests / bench */
int lrd_session_alm(lrd_session *s) { return lrd_alm_optimize(&s->par, &s->sol, 0, 0, s->par.ALMRhoFactor, lrd_time()); }
void lrd_session_alm_to_admm(lrd_session *s) { lrd_alm_to_admm(&s->par, &s->sol); }
int lrd_session_admm(lrd_session *s, int iter_ceiling) {
    return lrd_admm_optimize(&s->par, &s->sol, 0, iter_ceiling, lrd_time());
}

/* `steps` ADMM iterations at fixed rho with the CG tolerance min(1e-2 err1, 1e-8) refreshed every
 * iteration (the measured step of bench.py; one iteration = lorads_admm.c:76-81 + :120).
 * io = {err1 (in/out), cg iterations (out), pobj (out), dobj (out)} */
int lrd_session_admm_steps(lrd_session *s, int steps, double rho, double io[4]) {
    if (!s->have_be) return 1;
    lrd_backend *be = &s->be;
    double err1 = io[0], pobj = 0, dobj = 0;
    long cg = 0;
    for (int it = 0; it < steps; ++it) {
        double tol = err1 * 1e-2 < 1e-8 ? err1 * 1e-2 : 1e-8;
        if (be->admm_step && s->sol.use_fused_step) {
            double o[4];
            if (be->admm_step(be->ctx, rho, tol, 800, o)) return 1;
            cg += (long)o[0]; pobj = o[1]; dobj = o[2]; err1 = o[3];
        } else {
            int c = 0;
            if (be->admm_update_var(be->ctx, rho, tol, 800, &c)) return 1;
            cg += c;
            if (be->cal_obj(be->ctx, LRD_PAIR_UV, &pobj) || be->cal_dual_obj(be->ctx, &dobj) ||
                be->update_dimacs(be->ctx, LRD_PAIR_UV, &err1))
                return 1;
        }
        if (be->update_dual_var(be->ctx, rho)) return 1;
    }
    io[0] = err1; io[1] = (double)cg; io[2] = pobj; io[3] = dobj;
    return 0;
}

/* DIMACS error 2 of the current multipliers (-1 when the table lacks the slot) */
int lrd_session_dual_infeasibility(lrd_session *s, double *err_dual_l1) {
    if (!s->have_sol) return 1;
    int rc = lrd_dual_infeasibility(&s->sol);
    *err_dual_l1 = s->sol.err_dual_l1;
    return rc;
}

/* results: [pObj, dObj, constrVio(1), pdGap, alm_outer, alm_inner, alm_rho, admm_iter, cg_iter, admm_rho,
 *           t_alm, t_admm, status, admm_iters_first, cg_iters_first, constrVio(Inf)];
 * lrd_session_results2 appends [dualInfeas(1), dualInfeas(Inf), t_dual_infeas, scaleObjHis] */
int lrd_session_results2(lrd_session *s, double out[4]) {
    if (!s->have_sol) return 1;
    lrd_solver *v = &s->sol;
    out[0] = v->err_dual_l1;
    out[1] = v->err_dual_l1 < 0 ? -1.0 : v->err_dual_l1 * (1 + s->prob->cObjNrm1) / (1 + s->prob->cObjNrmInf);
    out[2] = v->t_dual_infeas;
    out[3] = v->scaleObjHis;
    return 0;
}
int lrd_session_results(lrd_session *s, double out[16]) {
    if (!s->have_sol) return 1;
    lrd_solver *v = &s->sol;
    out[0] = v->pObjVal; out[1] = v->dObjVal; out[2] = v->err_constr_l1; out[3] = v->err_pdgap;
    out[4] = v->alm.outerIter; out[5] = v->alm.innerIter; out[6] = v->alm.rho;
    out[7] = v->admm.iter; out[8] = v->admm.cg_iter; out[9] = v->admm.rho;
    out[10] = v->t_alm; out[11] = v->t_admm; out[12] = v->status;
    out[13] = v->admm_iters_first; out[14] = v->cg_iters_first;
    out[15] = v->err_constr_l1 * (1 + s->prob->bNrm1) / (1 + s->prob->bNrmInf);
    return 0;
}

int lrd_session_current_rank(lrd_session *s, int k) { return s->have_sol ? s->sol.rank[k] : s->prob->blk[k].rank; }

void lrd_session_close(lrd_session *s) {
    if (!s) return;
    if (s->have_sol) lrd_solver_clear(&s->sol);
    if (s->have_be && s->be.destroy) s->be.destroy(s->be.ctx);
    if (s->prob) { lrd_free_point(s->prob->nblk, s->R0, s->U0, s->V0); lrd_problem_free(s->prob); }
    free(s);
}
