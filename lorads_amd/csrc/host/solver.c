/* solver.c -- scalar outer-loop control of both phases (host, plain C).
 *
 * the reference keeps these as host-side scalar code and so do
 * we.  Everything numerical goes through `lrd_backend` (the HIP C-ABI in the product).  The control
 * flow below restates, decision for decision, the reference loops
 *   phase 1  LORADS_ALMOptimize / LORADS_ALMOptimize_reopt   src_semi/lorads_alg/lorads_alm.c:991-1255, 745-987
 *   phase 2  LORADSADMMOptimize / LORADSADMMOptimize_reopt   src_semi/lorads_alg/lorads_admm.c:33-157, 160-307
 *   hand-off LORADS_ALMtoADMM, reopt                         src_semi/data/lorads_solver.c:968-1004, 1075-1117
 *   driver   main()                                          src_semi/main.c:321-398
 * with the two variants of each loop folded into one function and a `reopt_variant` switch (the
 * differences are listed next to each switch), so that iterate-for-iterate parity is possible.
 */
#include "lorads_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define LMIN(a, b) ((a) < (b) ? (a) : (b))
#define LMAX(a, b) ((a) > (b) ? (a) : (b))
/* every table slot returns 0 / non-zero (lorads_hip.h); the reference's kernels return void and cannot fail, ours can
 * (device error, rank refused, all-reduce hook failed).  A failure is remembered in the solver and the loops below leave
 * with LRD_RET_NUM_ERR at their next check. */
#define BE(call) do { if ((call) != 0) s->be_fail = 1; } while (0)

double lrd_time(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

void lrd_params_default(lrd_params *p) {
    memset(p, 0, sizeof *p);
    p->fname = "NULL";
    p->initRho = 0.0; p->rhoMax = 5000.0; p->rhoCellingALM = 1e8; p->rhoCellingADMM = 5000.0 * 200;
    p->maxALMIter = 200; p->maxADMMIter = 10000; p->timesLogRank = 2.0; p->rhoFreq = 5; p->rhoFactor = 1.2;
    p->ALMRhoFactor = 2.0; p->phase1Tol = 1e-3; p->phase2Tol = 1e-5; p->timeSecLimit = 3600.0;
    p->heuristicFactor = 1.0; p->lbfgsListLength = 2; p->endTauTol = 1e-16; p->endALMSubTol = 1e-10;
    p->l2Rescaling = 0; p->reoptLevel = 2; p->dyrankLevel = 2; p->highAccMode = 0; p->verbose = 1;
}

int lrd_params_set(lrd_params *p, const char *k, const char *v) {
#define D(name) if (!strcmp(k, #name)) { p->name = atof(v); goto ok; }
#define I(name) if (!strcmp(k, #name)) { p->name = atoi(v); goto ok; }
    D(initRho) D(rhoMax) D(rhoCellingALM) D(rhoCellingADMM) I(maxALMIter) I(maxADMMIter) D(timesLogRank)
    I(rhoFreq) D(rhoFactor) D(ALMRhoFactor) D(phase1Tol) D(phase2Tol) D(timeSecLimit) D(heuristicFactor)
    I(lbfgsListLength) D(endTauTol) D(endALMSubTol) I(l2Rescaling) I(reoptLevel) I(dyrankLevel) I(highAccMode)
    I(verbose)
#undef D
#undef I
    return 1;
ok:
    p->rhoCellingADMM = p->rhoMax * 200; /* main.c:236 */
    return 0;
}

/* ---- line search scalars (lorads_alm.c:102-228) ---- */
static double nthroot3(double x) { return x > 0 ? pow(x, 1.0 / 3) : -pow(-x, 1.0 / 3); }

int lrd_cubic_roots(double a, double b, double c, double d, double r[3]) {
    /* Shengjin's formulas, same case split as LORADScubic_equation (lorads_alm.c:114-154) */
    double A = b * b - 3 * a * c, B = b * c - 9 * a * d, C = c * c - 3 * b * d;
    double delta = B * B - 4 * A * C;
    r[0] = r[1] = r[2] = 0.0;
    if (A == 0 && B == 0) { r[0] = LMAX(r[0], -c / b); return 1; }
    if (delta > 0) {
        double sq = sqrt(delta);
        double y1 = A * b + 1.5 * a * (-B + sq), y2 = A * b + 1.5 * a * (-B - sq);
        r[0] = LMAX(r[0], (-b - nthroot3(y1) - nthroot3(y2)) / 3 / a);
        return 1;
    }
    if (delta == 0 && A != 0 && B != 0) { double K = B / A; r[0] = -b / a + K; r[1] = -K / 2; return 2; }
    if (delta < 0) {
        double sA = sqrt(A), T = (A * b - 1.5 * a * B) / (A * sA), th = acos(T);
        double cs = cos(th / 3), sn = sqrt(3) * sin(th / 3);
        r[0] = (-b - 2 * sA * cs) / 3 / a;
        r[1] = (-b + sA * (cs + sn)) / 3 / a;
        r[2] = (-b + sA * (cs - sn)) / 3 / a;
        return 3;
    }
    return 0;
}

static double quartic(const double k[4], double x) {
    return k[0] * pow(x, 4) + k[1] * pow(x, 3) + k[2] * pow(x, 2) + k[3] * x;
}

int lrd_linesearch_tau(const double k[4], double *tau) {
    /* argmin of a t^4 + b t^3 + c t^2 + d t over {0, 1, stationary points in (1e-20, 1]};
     * later candidates win ties within 1e-10, as lorads_alm.c:173-227 */
    double roots[3];
    int nr = lrd_cubic_roots(4 * k[0], 3 * k[1], 2 * k[2], k[3], roots);
    double f[5] = {0.0, quartic(k, 1.0), 1e30, 1e30, 1e30}, cand[5] = {0.0, 1.0, roots[0], roots[1], roots[2]};
    for (int i = 0; i < 3; ++i)
        if (nr >= i + 1 && !(i == 2 && nr != 3) && roots[i] > 1e-20 && roots[i] <= 1.0) f[2 + i] = quartic(k, roots[i]);
    double fmin = f[0];
    for (int i = 1; i < 5; ++i) fmin = LMIN(fmin, f[i]);
    for (int i = 0; i < 5; ++i)
        if (fabs(fmin - f[i]) < 1e-10) *tau = cand[i];
    return nr;
}

/* ---- solver state ---- */
int lrd_solver_init(lrd_solver *s, lrd_problem *prob, lrd_backend *be, const lrd_params *par) {
    memset(s, 0, sizeof *s);
    s->prob = prob;
    s->be = be;
    s->rank = (int *)malloc(sizeof(int) * (size_t)(prob->nblk > 0 ? prob->nblk : 1));
    for (int k = 0; k < prob->nblk; ++k) s->rank[k] = prob->blk[k].rank;
    /* initial_solver_state, data/lorads_solver.c:1148-1170 */
    double rho = par->initRho == 0 ? 1.0 / sqrt((double)prob->sum_dims_global) : par->initRho;
    lrd_alm_state *a = &s->alm;
    a->dual_objective_value = a->primal_objective_value = 1e30;
    a->l_1_dual_infeasibility = a->l_1_primal_infeasibility = 1e30;
    a->l_inf_dual_infeasibility = a->l_inf_primal_infeasibility = 1e30;
    a->rho = rho;
    lrd_admm_state *d = &s->admm;
    d->dual_objective_value = d->primal_objective_value = d->primal_dual_gap = 1e30;
    d->l_1_dual_infeasibility = d->l_1_primal_infeasibility = 1e30;
    d->l_inf_dual_infeasibility = d->l_inf_primal_infeasibility = 1e30;
    d->l_2_dual_infeasibility = d->l_2_primal_infeasibility = 1e30;
    d->rho = rho;
    d->nBlks = prob->nsdp_global > 0 ? prob->nsdp_global : 1;
    s->scaleObjHis = 1.0;
    s->max_alm_sub_iter = 5000;
    s->status = LRD_UNKNOWN;
    s->use_fused_step = 1;
    return 0;
}

void lrd_solver_clear(lrd_solver *s) {
    free(s->rank);
    s->rank = NULL;
}

/* "time is up?" -- with sharded cones every rank must take the same branch, or the ones that stay enter the next
 * collective alone and wait for ever: the clocks are the only control input that is not rank-uniform, so the verdicts
 * are summed over the ranks (anybody's time-out ends it for all).  Called at the same control points on every rank. */
/* A table slot that failed on ONE rank (be_fail) travels with the same sum: every rank then leaves at this control point with
 * LRD_RET_NUM_ERR instead of the healthy ones entering the next device collective alone.  (A rank whose slot fails inside a sweep
 * that contains a collective cannot be waited for by its peers: their hand-over reports it after LORADS_HANDOVER_TIMEOUT_S, default
 * 300 s, and the solve ends with a non-zero return code -- lorads_hip.h, "result hand-over".) */
static int time_is_up(lrd_solver *s, double t_start, double limit) {
    double v[2] = {(lrd_time() - t_start >= limit) ? 1.0 : 0.0, s->be_fail ? 1.0 : 0.0};
    if (s->allreduce) {
        if (s->allreduce(s->allreduce_user, v, 2, 0) != 0) s->be_fail = 1;
        if (v[1] > 0.0) s->be_fail = 1; /* somebody's backend failed: everybody stops */
    }
    return (s->allreduce && s->be_fail) ? 2 : (v[0] > 0.0 ? 1 : 0); /* 2: a backend failure somewhere, 1: time */
}

static double inf_from_l1(const lrd_solver *s, double l1) {
    return l1 * (1 + s->prob->bNrm1) / (1 + s->prob->bNrmInf);
}
static double l2_from_l1(const lrd_solver *s, double l1) { return l1 * (1 + s->prob->bNrm1) / (1 + s->prob->bNrm2); }

static void refresh_obj(lrd_solver *s, int pair) {
    double v = 0.0;
    BE(s->be->cal_obj(s->be->ctx, pair, &v));
    s->pObjVal = v / s->scaleObjHis;
    v = 0.0;
    BE(s->be->cal_dual_obj(s->be->ctx, &v));
    s->dObjVal = v / s->scaleObjHis;
}

static void refresh_dimacs(lrd_solver *s, int pair) {
    BE(s->be->update_dimacs(s->be->ctx, pair, &s->err_constr_l1));
    double gap = s->pObjVal - s->dObjVal;
    s->err_pdgap = fabs(gap) / (1 + fabs(s->pObjVal) + fabs(s->dObjVal));
}

static double cert(const lrd_solver *s, double lag_sq) { return sqrt(lag_sq) / (1 + s->prob->cObjNrmInf); }

static void alm_log(const lrd_params *par, const lrd_alm_state *a, double t) {
    if (!par->verbose) return;
    printf("ALM OuterIter:%d InnerIter:%d pObj:%5.5e dObj:%5.5e pInfea(1):%5.5e pInfea(Inf):%5.5e pdGap:%5.5e rho:%3.2f Time:%3.2f\n",
           a->outerIter, a->innerIter, a->primal_objective_value, a->dual_objective_value, a->l_1_primal_infeasibility,
           a->l_inf_primal_infeasibility, a->primal_dual_gap, a->rho, t);
}

static void admm_log(const lrd_params *par, const lrd_admm_state *d, double t) {
    if (!par->verbose) return;
    printf("ADMM Iter:%d pObj:%5.5e dObj:%5.5e pInfea(1):%5.5e pInfea(Inf):%5.5e pdGap:%5.5e rho:%3.2f cgIter:%d Time:%3.2f\n",
           d->iter, d->primal_objective_value, d->dual_objective_value, d->l_1_primal_infeasibility,
           d->l_inf_primal_infeasibility, d->primal_dual_gap, d->rho, (int)((double)d->cg_iter / (double)d->nBlks), t);
}

/* EMA stall detector, lorads_utils.c:404-434 */
static int ema_check(double *cur, double *old, double val, double alpha, double thr, int interval, int *counter) {
    int ok = 1;
    *cur = alpha * val + (1 - alpha) * (*cur);
    if (*counter >= interval) {
        if (*old != 0) {
            double ch = (*cur - *old) / *old;
            ok = (ch >= -thr) && (ch <= thr);
        }
        *old = *cur;
        *counter = 1;
    } else {
        (*counter)++;
    }
    return ok;
}

/* CheckAllRankMax, data/lorads_solver.c:758-774 */
static int all_rank_max(lrd_solver *s, double factor) {
    int cnt = 0;
    for (int k = 0; k < s->prob->nblk; ++k) {
        int nr = (int)LMIN(ceil(s->rank[k] * factor), (double)s->prob->blk[k].rank_max);
        if (nr >= s->prob->blk[k].rank_max) ++cnt;
    }
    if (s->allreduce) { /* blocks are sharded: every rank must take the same decision */
        double v[2] = {(double)cnt, (double)s->prob->nblk};
        BE(s->allreduce(s->allreduce_user, v, 2, 0));
        return v[0] == v[1];
    }
    return cnt == s->prob->nblk;
}

/* AUG_RANK, data/lorads_solver.c:806-906: new_r = min(ceil(1.5 r), rank_max); new columns get 1/sqrt(r_new_cols) on
 * their leading diagonal; L-BFGS history is cleared */
static int augment_rank(lrd_solver *s, double factor) {
    if (all_rank_max(s, 1.0)) return 1;
    int nb = s->prob->nblk;
    int *nr = (int *)malloc(sizeof(int) * (size_t)(nb > 0 ? nb : 1));
    for (int k = 0; k < nb; ++k) nr[k] = (int)LMIN(ceil(s->rank[k] * factor), (double)s->prob->blk[k].rank_max);
    if (s->be->resize_rank(s->be->ctx, nr) != 0) {
        /* refused (e.g. a rank the device kernels do not cover): host and device keep the OLD ranks -- treated as
         * "rank cannot grow any further", and flagged so that the caller stops */
        s->be_fail = 1;
        free(nr);
        return 1;
    }
    for (int k = 0; k < nb; ++k) s->rank[k] = nr[k];
    free(nr);
    return all_rank_max(s, factor);
}

/* ------------------------------------------------------------------ phase 1 ------------------- */
int lrd_alm_optimize(lrd_params *par, lrd_solver *s, int reopt, int early_stop, double rho_update_factor,
                     double t_start) {
    lrd_backend *be = s->be;
    void *cx = be->ctx;
    lrd_alm_state *st = &s->alm;
    const double t_ori = lrd_time();
    int ret = LRD_RET_OK;
    if (!reopt) s->max_alm_sub_iter = 5000; /* lorads_alm.c:993 (the reopt variant keeps the global) */
    int is_rank_max = all_rank_max(s, 1.0);
    int last_outer_start = 1;
    double tau = 0.0, lag = 0.0, cert_val, cert_tol;
    const double cert0 = 0.1;
    int k, k0;
    char difficulty;
    int local_iter, clear_lbfgs, rank_flag, rho_factor_flag, sub_counter;
    const double rank_factor = 1.5;
    double rank_thres = 15;
    if (par->dyrankLevel == 0) rank_thres = 1e8;
    else if (par->dyrankLevel == 1) rank_thres = 150;
    else if (par->dyrankLevel == 2) rank_thres = 15;
    else if (par->dyrankLevel == 3) rank_thres = 5;

restart:
    cert_tol = cert0 / st->rho;
    BE(be->init_constr(cx, LRD_PAIR_RR));
    BE(be->alm_cal_grad(cx, st->rho, &lag));
    cert_val = cert(s, lag);
    difficulty = 'h';
    local_iter = 0; clear_lbfgs = 0; rank_flag = 0; rho_factor_flag = 0; sub_counter = 0;
    if (!reopt) rho_update_factor = par->ALMRhoFactor; /* lorads_alm.c:1020 */
    k = k0 = st->outerIter;

    for (;;) {
        /* loop condition: for(k<=maxALMIter) vs while(true)+break (lorads_alm.c:1039 vs :791-796) */
        if (!reopt) {
            if (k > par->maxALMIter) break;
        } else if (k > par->maxALMIter && st->l_inf_primal_infeasibility <= par->phase1Tol &&
                   (st->primal_dual_gap <= LMAX(par->phase1Tol, par->phase2Tol * 5) || !par->highAccMode)) {
            break;
        }
        double ema_cur = 0.0, ema_old = 0.0;
        int ema_counter = 1, cur_iter_counter = 1;
        if (sub_counter >= 2) {
            sub_counter = 0;
            s->max_alm_sub_iter = LMIN(s->max_alm_sub_iter + 10000, 25000);
        }
        int jump_update_rho = 0;
        while (difficulty != 'e') {
            local_iter = 0;
            int steady = ema_check(&ema_cur, &ema_old, cert_val, 0.1, 0.005, 5, &ema_counter);
            if (!steady && !par->highAccMode) break;
            if (cur_iter_counter >= s->max_alm_sub_iter) { sub_counter += 1; break; }
            if (rank_flag >= rank_thres && !is_rank_max && (k - last_outer_start >= 3)) break;
            if (cert_val <= cert_tol) break;
            /* fused path: the table offers the inner iteration as two calls with one host round trip each and
             * pre-computes the next direction while the host looks at this one's results */
            /* (with sharded cones too: every collective inside is entered by all ranks alike) */
            const int fused = be->alm_step && be->alm_front && s->use_fused_step;
            int have_front = 0;
            double front[6] = {0, 0, 0, 0, 0, 0};
            while (cert_val - cert_tol > par->endALMSubTol) {
                /* L-BFGS memory reset every 300 steps: `localIter % 300` vs `(localIter-1) % 300` */
                if ((!reopt && local_iter % 300 == 0) || (reopt && (local_iter - 1) % 300 == 0)) clear_lbfgs = 0;
                double p12[2], coef[4];
                if (!fused) {
                    BE(be->lbfgs_direction(cx, clear_lbfgs));
                    BE(be->alm_q12p12(cx, p12));
                    BE(be->alm_linesearch_coeffs(cx, st->rho, p12[0], p12[1], coef));
                } else {
                    if (!have_front && be->alm_front(cx, st->rho, clear_lbfgs, front)) { ret = LRD_RET_NUM_ERR; goto end_alm; }
                    memcpy(coef, front + 2, sizeof coef);
                    have_front = 0;
                }
                int nroot = lrd_linesearch_tau(coef, &tau);
                if (nroot == 0) { ret = LRD_RET_NUM_ERR; goto end_alm; }
                if (fabs(tau) < par->endTauTol) {
                    if (par->verbose) printf("update rho:%5.8e since tau is too small.\n", tau);
                    st->innerIter++; local_iter++; cur_iter_counter++; clear_lbfgs++;
                    jump_update_rho = 1;
                    break;
                }
                if (!fused) {
                    BE(be->set_y_as_neg_grad(cx));
                    BE(be->alm_update_var(cx, tau));
                    BE(be->alm_cal_grad(cx, st->rho, &lag));
                    BE(be->set_lbfgs_his_two(cx, tau));
                    BE(be->update_dimacs(cx, LRD_PAIR_RR, &s->err_constr_l1));
                } else {
                    /* the counter the next pass of this loop would hand to lbfgs_direction */
                    int next_clear = clear_lbfgs + 1;
                    if ((!reopt && (local_iter + 1) % 300 == 0) || (reopt && local_iter % 300 == 0)) next_clear = 0;
                    double o[8];
                    if (be->alm_step(cx, st->rho, tau, next_clear, o)) { ret = LRD_RET_NUM_ERR; goto end_alm; }
                    lag = o[0];
                    s->err_constr_l1 = o[1];
                    memcpy(front, o + 2, sizeof front);
                    have_front = 1;
                }
                if (s->be_fail) { ret = LRD_RET_NUM_ERR; goto end_alm; }
                { double gap = s->pObjVal - s->dObjVal;
                  s->err_pdgap = fabs(gap) / (1 + fabs(s->pObjVal) + fabs(s->dObjVal)); }
                st->l_1_primal_infeasibility = s->err_constr_l1;
                st->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
                if (!reopt && st->l_inf_primal_infeasibility <= par->phase1Tol &&
                    (st->primal_dual_gap <= par->phase1Tol || !par->highAccMode)) {
                    st->outerIter = k; /* warm start is good enough: lorads_alm.c:1132-1140 */
                    st->innerIter++; local_iter++; cur_iter_counter++; clear_lbfgs++;
                    goto end_alm;
                }
                cert_val = cert(s, lag);
                st->innerIter++; local_iter++; cur_iter_counter++; clear_lbfgs++;
                if (local_iter > 800) break;
            }
            if (jump_update_rho) break;
            BE(be->update_dual_var(cx, st->rho));
            BE(be->alm_cal_grad(cx, st->rho, &lag));
            cert_val = cert(s, lag);
            if (local_iter <= 20) difficulty = 'e';
            else if (local_iter <= 100) { difficulty = 'm'; rank_flag += 2; }
            else if (reopt || local_iter < 400) { difficulty = 'h'; rank_flag += 3; } /* :896 vs :1161-1168 */
            else { difficulty = 's'; rank_flag += 4; }
            if (difficulty == 'e') rank_flag = 0;
        }
        /* UpdateRho */
        do {
            st->rho *= rho_update_factor;
            BE(be->alm_cal_grad(cx, st->rho, &lag));
            cert_val = cert(s, lag);
            cert_tol = cert0 / st->rho;
        } while (cert_tol >= cert_val && !s->be_fail);
        if (s->be_fail) { ret = LRD_RET_NUM_ERR; goto end_alm; }
        if (st->rho >= 5e4 && rho_factor_flag < 4) { rho_update_factor = sqrt(sqrt(rho_update_factor)); rho_factor_flag = 4; }
        else if (st->rho >= 5e6 && rho_factor_flag < 6) { rho_update_factor = sqrt(sqrt(rho_update_factor)); rho_factor_flag = 6; }
        else if (st->rho >= 5e8 && rho_factor_flag < 8) { rho_update_factor = sqrt(sqrt(rho_update_factor)); rho_factor_flag = 8; }
        difficulty = 'h';
        clear_lbfgs = 0;
        if (reopt) k += 1;
        st->outerIter = k;
        if (!reopt && st->l_inf_primal_infeasibility <= par->phase1Tol &&
            (st->primal_dual_gap <= par->phase1Tol || !par->highAccMode))
            goto end_alm;
        refresh_obj(s, LRD_PAIR_RR);
        refresh_dimacs(s, LRD_PAIR_RR);
        st->primal_dual_gap = s->err_pdgap;
        st->primal_objective_value = s->pObjVal;
        st->dual_objective_value = s->dObjVal;
        st->l_1_primal_infeasibility = s->err_constr_l1;
        st->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
        st->l_1_dual_infeasibility = st->l_inf_dual_infeasibility = 99;
        if (!reopt) {
            if (st->primal_dual_gap <= par->phase1Tol * 1e-3 && st->l_1_primal_infeasibility <= par->phase1Tol * 1e-3)
                goto print_and_exit;
        } else if (early_stop) {
            if (st->l_1_primal_infeasibility <= par->phase1Tol &&
                st->primal_dual_gap <= LMAX(par->phase1Tol, par->phase2Tol * 5) && (k - k0) > 1)
                goto print_and_exit;
        } else if (st->primal_dual_gap <= par->phase2Tol && st->l_1_primal_infeasibility <= par->phase2Tol && (k - k0) > 1) {
            goto print_and_exit;
        }
        alm_log(par, st, lrd_time() - t_ori);
        { const int tu = time_is_up(s, t_start, par->timeSecLimit); if (tu == 2) { ret = LRD_RET_NUM_ERR; goto end_alm; } if (tu) goto print_and_exit; }
        if (rank_flag >= rank_thres && !is_rank_max && (!reopt || s->prob->nsdp_global <= 10)) {
            rank_flag = 0;
            if (k - last_outer_start >= 2) {
                if (par->verbose) printf("increase the rank, factor:%f.\n", rank_factor);
                is_rank_max = augment_rank(s, rank_factor);
                if (s->be_fail) { ret = LRD_RET_NUM_ERR; goto end_alm; }
                st->outerIter = k;
                last_outer_start = st->outerIter;
                goto restart;
            }
        }
        if (!reopt) k += 1;
    }
end_alm:
    refresh_obj(s, LRD_PAIR_RR);
    refresh_dimacs(s, LRD_PAIR_RR);
    st->primal_dual_gap = s->err_pdgap;
    if (!reopt) {
        st->primal_objective_value = s->pObjVal;
        st->dual_objective_value = s->dObjVal;
        st->l_1_primal_infeasibility = s->err_constr_l1;
        st->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
    } else { /* lorads_alm.c:978 derives l1 back from the stored linf */
        st->l_1_primal_infeasibility = st->l_inf_primal_infeasibility * (1 + s->prob->bNrmInf) / (1 + s->prob->bNrm1);
    }
    st->l_1_dual_infeasibility = st->l_inf_dual_infeasibility = 99;
print_and_exit:
    if (s->be_fail) ret = LRD_RET_NUM_ERR;
    if (par->verbose) {
        printf("-----------------------------------------------------------------------\nExit ALM:\n");
        alm_log(par, st, lrd_time() - t_ori);
        printf("-----------------------------------------------------------------------\n");
    }
    return ret;
}

/* ------------------------------------------------------------------ phase 2 ------------------- */
static void admm_pull_state(lrd_solver *s, int with_l1) {
    lrd_admm_state *d = &s->admm;
    d->primal_objective_value = s->pObjVal;
    d->dual_objective_value = s->dObjVal;
    d->primal_dual_gap = s->err_pdgap;
    if (with_l1) d->l_1_primal_infeasibility = s->err_constr_l1;
}

int lrd_admm_optimize(lrd_params *par, lrd_solver *s, int reopt, int iter_ceiling, double t_start) {
    lrd_backend *be = s->be;
    void *cx = be->ctx;
    lrd_admm_state *d = &s->admm;
    if (d->primal_dual_gap <= par->phase2Tol && d->l_1_primal_infeasibility <= par->phase2Tol) return LRD_RET_OK;
    const int cg_max = 800;
    d->rho = LMIN(d->rho, par->rhoMax);
    s->cgIter = 0;
    /* prologue, lorads_admm.c:47-58 */
    BE(be->init_constr(cx, LRD_PAIR_UV));
    refresh_obj(s, LRD_PAIR_UV);
    refresh_dimacs(s, LRD_PAIR_UV);
    admm_pull_state(s, 1);
    d->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
    d->l_2_primal_infeasibility = l2_from_l1(s, s->err_constr_l1);
    if (reopt && par->verbose) { printf("enter admm reopt \n"); admm_log(par, d, 0); }
    double cur_rho_max = par->rhoMax, old_mean = 1e30, buf[10] = {0};
    int bad_pd = 0;
    const int count = 0; /* the reference never advances it */
    const double t_ori = lrd_time();
    const int bad_limit = reopt ? 200 : 800;
    while (d->iter <= par->maxADMMIter || d->primal_dual_gap >= par->phase2Tol ||
           d->l_1_primal_infeasibility >= par->phase2Tol) {
        if (d->iter >= iter_ceiling) {
            if (reopt) admm_log(par, d, 0);
            break;
        }
        /* CG tolerance from the (possibly stale, quirk Q5) l1 infeasibility: 1e-2 vs 1e-4 */
        double cg_tol = LMIN(d->l_1_primal_infeasibility * (reopt ? 1e-4 : 1e-2), 1e-8);
        int cg_its = 0;
        if (be->admm_step && s->use_fused_step) {
            double o[4] = {0.0, 0.0, 0.0, 0.0};
            if (be->admm_step(cx, d->rho, cg_tol, cg_max, o) != 0) { s->be_fail = 1; return LRD_RET_NUM_ERR; }
            cg_its = (int)o[0];
            s->pObjVal = o[1] / s->scaleObjHis;
            s->dObjVal = o[2] / s->scaleObjHis;
            s->err_constr_l1 = o[3];
            s->err_pdgap = fabs(s->pObjVal - s->dObjVal) / (1 + fabs(s->pObjVal) + fabs(s->dObjVal));
            s->cgIter += cg_its;
            d->cg_iter = s->cgIter;
        } else {
            if (be->admm_update_var(cx, d->rho, cg_tol, cg_max, &cg_its) != 0) { s->be_fail = 1; return LRD_RET_NUM_ERR; }
            s->cgIter += cg_its;
            d->cg_iter = s->cgIter;
            refresh_obj(s, LRD_PAIR_UV);
            refresh_dimacs(s, LRD_PAIR_UV);
        }
        if (s->be_fail) return LRD_RET_NUM_ERR;
        admm_pull_state(s, reopt); /* the first-pass loop does not refresh l1 here (lorads_admm.c:82-85) */
        d->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
        if (reopt) d->l_2_primal_infeasibility = l2_from_l1(s, s->err_constr_l1);
        if (d->l_inf_primal_infeasibility >= 1e10 || d->primal_dual_gap >= 1 - 1e-8) {
            if (reopt) admm_log(par, d, lrd_time() - t_ori);
            if (par->verbose) printf("Numerical Error!\n");
            return LRD_RET_NUM_ERR;
        }
        if (d->primal_dual_gap <= par->phase2Tol * 5) bad_pd = LMAX(0, bad_pd - 5);
        else if (d->primal_dual_gap <= par->phase2Tol) bad_pd = LMAX(0, bad_pd - 10);
        if (d->primal_dual_gap >= par->phase1Tol * 1e2) bad_pd += 2;
        if (bad_pd >= bad_limit) return LRD_RET_OK;
        buf[count % 10] = d->l_inf_primal_infeasibility;
        if ((!reopt && d->l_inf_primal_infeasibility <= par->phase2Tol) ||
            (reopt && d->l_1_primal_infeasibility <= par->phase2Tol)) {
            refresh_dimacs(s, LRD_PAIR_UV);
            admm_pull_state(s, 1);
            if (!reopt || d->primal_dual_gap <= par->phase2Tol) {
                admm_log(par, d, lrd_time() - t_ori);
                return LRD_RET_OK;
            }
        }
        BE(be->update_dual_var(cx, d->rho));
        /* rho schedule: tested on iter+1 in the first pass, on iter in the reopt pass */
        int it_sched = reopt ? d->iter : d->iter + 1;
        if (it_sched % par->rhoFreq == 0) {
            d->rho *= par->rhoFactor;
            if (d->rho >= cur_rho_max) {
                d->rho = cur_rho_max;
                if (it_sched % (par->rhoFreq * 100) == 0) {
                    double mean = 0.0;
                    for (int i = 0; i < 10; ++i) mean += fabs(buf[i]);
                    mean /= 10.0;
                    if (mean / old_mean >= 0.65) {
                        d->rho *= pow(par->rhoFactor, round(log(par->rhoFreq * 100) / log(par->rhoFreq)));
                        cur_rho_max = d->rho;
                    }
                    old_mean = mean;
                }
            }
            if (d->rho >= par->rhoCellingADMM) d->rho = par->rhoCellingADMM;
        }
        if (d->iter % 50 == 0) {
            refresh_dimacs(s, LRD_PAIR_UV);
            admm_pull_state(s, 1);
            admm_log(par, d, lrd_time() - t_ori);
            { const int tu = time_is_up(s, t_start, par->timeSecLimit); if (tu == 2) return LRD_RET_NUM_ERR; if (tu) return LRD_RET_TIME_OUT; }
        }
        if (d->primal_dual_gap <= par->phase2Tol * 1e-3 && d->l_1_primal_infeasibility <= par->phase2Tol * 1e-3) {
            if (par->verbose) printf("Early Stop When DIMACS Errors Are Well-Satisfied");
            return LRD_RET_OK;
        }
        d->iter++;
    }
    if (reopt) admm_log(par, d, lrd_time() - t_ori);
    return s->be_fail ? LRD_RET_NUM_ERR : LRD_RET_OK;
}

/* LORADS_ALMtoADMM, data/lorads_solver.c:968-1004 */
void lrd_alm_to_admm(lrd_params *par, lrd_solver *s) {
    BE(s->be->alm_to_admm(s->be->ctx));
    lrd_alm_state *a = &s->alm;
    lrd_admm_state *d = &s->admm;
    d->l_1_dual_infeasibility = a->l_1_dual_infeasibility;
    d->l_1_primal_infeasibility = a->l_1_primal_infeasibility;
    d->l_2_dual_infeasibility = a->l_2_dual_infeasibility;
    d->l_inf_dual_infeasibility = a->l_inf_dual_infeasibility;
    d->l_inf_primal_infeasibility = a->l_inf_primal_infeasibility;
    d->l_2_primal_infeasibility = a->l_2_primal_infeasibility;
    d->primal_dual_gap = a->primal_dual_gap;
    d->rho = a->rho * par->heuristicFactor;
    if (a->rho > par->rhoMax) {
        d->rho = LMIN(sqrt(LMAX(par->rhoMax, a->rho) / par->rhoMax) * par->rhoMax, a->rho);
        par->rhoMax = d->rho;
    }
}

/* reopt, data/lorads_solver.c:1075-1117 */
double lrd_reopt(lrd_params *par, lrd_solver *s, double reopt_param, int reopt_alm_iter, int reopt_admm_iter,
                 double t_start, int *bad_flag, int level) {
    int old_alm = par->maxALMIter, old_admm = par->maxADMMIter;
    double old_rho_max = par->rhoMax;
    par->maxALMIter = reopt_alm_iter - 1 + s->alm.outerIter;
    par->maxADMMIter = reopt_admm_iter;
    s->scaleObjHis *= reopt_param; /* objScale_dualvar */
    BE(s->be->scale_obj(s->be->ctx, reopt_param));
    if (s->admm.rho <= par->rhoMax) s->alm.rho = LMAX(s->admm.rho, s->alm.rho);
    double t0 = lrd_time();
    lrd_alm_optimize(par, s, 1, 1, sqrt(par->ALMRhoFactor), t_start);
    par->rhoMax = LMAX(sqrt(LMAX(s->admm.rho, s->alm.rho) / s->admm.rho) * s->admm.rho, par->rhoMax);
    lrd_alm_to_admm(par, s);
    if (*bad_flag == 0 || level < 2) {
        int rc = lrd_admm_optimize(par, s, 1, LMIN(s->admm.iter * 4, s->admm.iter + old_admm), t_start);
        *bad_flag = (rc == LRD_RET_BAD_ITER);
    }
    double t1 = lrd_time();
    par->maxALMIter = old_alm;
    par->maxADMMIter = old_admm;
    par->rhoMax = old_rho_max;
    return t1 - t0;
}

/* calculate_dual_infeasibility_solver (data/lorads_solver.c:1007-1037): the table's optional slot returns
 * sum_k |min(lambda_min(C_k - A_k^*(lambda)), 0)| over the cones it holds; ranks add theirs; then the two
 * divisions of :1034-1035.  Copies the value into the ADMM state like main.c:404-409. */
int lrd_dual_infeasibility(lrd_solver *s) {
    if (!s->be->dual_infeasibility) { s->err_dual_l1 = -1.0; return 1; }
    double t = lrd_time(), v = 0.0;
    if (s->be->dual_infeasibility(s->be->ctx, &v)) { s->err_dual_l1 = -1.0; return 1; }
    if (s->allreduce) BE(s->allreduce(s->allreduce_user, &v, 1, 0));
    s->err_dual_l1 = v / s->scaleObjHis / (s->prob->cObjNrm1 + 1);
    lrd_admm_state *d = &s->admm;
    d->l_1_dual_infeasibility = s->err_dual_l1;
    d->l_inf_dual_infeasibility = s->err_dual_l1 * (1 + s->prob->cObjNrm1) / (1 + s->prob->cObjNrmInf);
    d->l_2_dual_infeasibility = s->err_dual_l1 * (1 + s->prob->cObjNrm1) / (1 + s->prob->cObjNrm2);
    d->primal_dual_gap = s->err_pdgap;
    d->l_1_primal_infeasibility = s->err_constr_l1;
    d->l_inf_primal_infeasibility = inf_from_l1(s, s->err_constr_l1);
    d->l_2_primal_infeasibility = l2_from_l1(s, s->err_constr_l1);
    s->t_dual_infeas += lrd_time() - t;
    return 0;
}

/* main.c:321-476 + status classification :478-487.  A table without the dual_infeasibility slot stops after
 * the level-1 round and classifies without the dual term (err_dual_l1 stays -1; that is also what
 * oracle/ref_driver.c does with the reference, whose ARPACK dependency cannot be linked here). */
int lrd_solve(lrd_params *par, lrd_solver *s) {
    double t0 = lrd_time();
    int bad = 0;
    s->status = LRD_UNKNOWN;
    s->err_dual_l1 = -1.0;
    s->t_dual_infeas = 0.0;
    double ta = lrd_time();
    s->be_fail = 0;
    lrd_alm_optimize(par, s, 0, 0, par->ALMRhoFactor, t0);
    s->t_alm = lrd_time() - ta;
    if (s->be_fail) return LRD_RET_NUM_ERR; /* a table slot failed: nothing below would be computed from real numbers */
    { const int tu = time_is_up(s, t0, par->timeSecLimit); if (tu == 2) return LRD_RET_NUM_ERR; if (tu) { s->status = LRD_TIME_LIMIT; return 0; } }
    lrd_alm_to_admm(par, s);
    ta = lrd_time();
    if (lrd_admm_optimize(par, s, 0, par->maxADMMIter, t0) == LRD_RET_BAD_ITER) bad = 1;
    s->t_admm = lrd_time() - ta;
    if (s->be_fail) return LRD_RET_NUM_ERR;
    s->admm_iters_first = s->admm.iter;
    s->cg_iters_first = s->cgIter;
    const int admm_reopt_min_iter = par->highAccMode ? 1000 : 50;
    int cnt = 0;
    if (par->reoptLevel >= 1) {
        while ((s->alm.primal_dual_gap > par->phase2Tol || s->alm.l_1_primal_infeasibility > par->phase2Tol) &&
               (s->admm.primal_dual_gap > par->phase2Tol || s->admm.l_1_primal_infeasibility > par->phase2Tol)) {
            if (cnt >= 1) break;
            if (par->verbose) printf("******  reopt parameter:%.3f\n", 5.0);
            lrd_reopt(par, s, 5.0, 3, admm_reopt_min_iter, t0, &bad, 1);
            cnt += 1;
            if (s->be_fail) return LRD_RET_NUM_ERR;
            { const int tu = time_is_up(s, t0, par->timeSecLimit); if (tu == 2) return LRD_RET_NUM_ERR; if (tu) { s->status = LRD_TIME_LIMIT; return 0; } }
        }
    }
    const int have_dual = lrd_dual_infeasibility(s) == 0; /* main.c:400-413, evaluated at every reoptLevel */
    if (have_dual) {
        lrd_admm_state *d = &s->admm;
        if (par->verbose)
            printf("Dual infeasibility: l_1 = %f, l_inf = %f, l_2 = %f\n", d->l_1_dual_infeasibility, d->l_inf_dual_infeasibility,
                   d->l_2_dual_infeasibility);
        int dual_cnt = 0; /* main.c:414-476 */
        while (par->reoptLevel >= 2 && (d->l_1_dual_infeasibility > par->phase2Tol || d->primal_dual_gap > par->phase2Tol ||
                                        d->l_1_primal_infeasibility > par->phase2Tol)) {
            if (dual_cnt >= 2) break;
            if (!par->highAccMode && d->l_1_dual_infeasibility <= 5 * par->phase2Tol && d->primal_dual_gap <= 5 * par->phase2Tol &&
                d->l_1_primal_infeasibility <= par->phase2Tol)
                break;
            if (par->verbose) printf("******  reopt parameter:%.3f\n", 5.0);
            lrd_reopt(par, s, 5.0, 3, 50, t0, &bad, 2);
            BE(s->be->average_uv_to_v(s->be->ctx)); /* averageUV + copyRtoV, main.c:438-448 */
            if (s->be_fail) return LRD_RET_NUM_ERR;
            if (lrd_dual_infeasibility(s)) break;
            if (par->verbose)
                printf("reopt %d:Dual infeasibility: l_1 = %f, l_inf = %f, l_2 = %f\n", dual_cnt, d->l_1_dual_infeasibility,
                       d->l_inf_dual_infeasibility, d->l_2_dual_infeasibility);
            dual_cnt += 1;
            { const int tu = time_is_up(s, t0, par->timeSecLimit); if (tu == 2) return LRD_RET_NUM_ERR; if (tu) { s->status = LRD_TIME_LIMIT; return 0; } }
        }
        if (d->l_1_dual_infeasibility <= 5 * par->phase2Tol && d->primal_dual_gap <= 5 * par->phase2Tol &&
            d->l_1_primal_infeasibility <= par->phase2Tol)
            s->status = LRD_PRIMAL_DUAL_OPTIMAL;
        else if (d->primal_dual_gap <= 5 * par->phase2Tol && d->l_1_primal_infeasibility <= par->phase2Tol)
            s->status = LRD_PRIMAL_OPTIMAL;
        else
            s->status = LRD_MAXITER;
        return 0;
    }
    if (s->err_pdgap <= 5 * par->phase2Tol && s->err_constr_l1 <= par->phase2Tol) s->status = LRD_PRIMAL_OPTIMAL;
    else s->status = LRD_MAXITER;
    return 0;
}
