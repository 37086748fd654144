// lorads_hip.hip -- MI355X (gfx950 / CDNA4) backend of the LoRADS per-iteration path.
//
// Hand-written HIP, FP64 throughout, wave64.  Implements the C ABI of include/lorads_hip.h; every
// entry point cites there the reference function it replaces.  Nothing here falls back to a CPU
// path: without a GPU lorads_hip_create fails.
//
// Data layout in HBM (per cone k, n x r factors):
//   * factors R,U,V,Grad and the CG vectors are ROW-major n x r (ld = r): one factor row is one
//     contiguous 8r-byte segment, so the row gathers of the sparse contractions are coalesced
//     (the reference's column-major layout makes them stride-n level-1 BLAS calls).
//     All cones are concatenated in one flat allocation per factor so that the L-BFGS vector
//     operations run over sum_k n_k r_k in one launch.  The ABI movers transpose to/from the
//     reference's column-major layout.
//   * A_i: unique lower-triangular positions of all A_i of the cone ("A-pattern", PA entries) +
//     CSR constraint -> (entry, value) + its transpose entry -> (constraint, value) + the Gram
//     matrix G = A A^T over pattern entries (S = G T in one pass, when it is small) + a full
//     symmetric adjacency row -> (neighbour row, entry) for gather-form S*X (no atomics,
//     bitwise reproducible).
//   * C u A "union pattern" (PU) with transpose + adjacency, used by the RHS / gradient weighted-sum
//     products (C + sum_i w_i A_i) X.
//
// Source layout: this file = context + the C ABI; kernels.inc = every kernel; build.inc = a cone's device image
// (patterns, adjacency, Gram, operator variants, merged block-diagonal cone, LP block); sweep.inc = launch helpers,
// the CG operator, speculative solves, the cone-by-cone and lockstep sweeps, the evaluation, result hand-over;
// lanczos.inc = dual infeasibility.
//
// Kernels (HBM/L2-bound integer+FP64 gather work; bytes per unit in DESIGN.md):
//   k_pairdots   T_e = X_p.Y_q + X_q.Y_p on a pattern            (reference LORADSUVt)
//   k_cv         w_i = sum_k a_k T_e(k) (+ running-sum update)     (mul_inner_rk_double / coneAUV)
//   k_cw         w_i straight from the factors, one wavefront per constraint (LORADSUVt + coneAUV)
//   k_sval       S_e = [C_e] + sum_(i,a) weight_i a               (sdpDataWSum / addObjCoeff)
//   k_sgram      S = G T                                          (coneAUV + sdpDataWSum fused)
//   k_spmm2      Y_p = epilogue(sum_(q,e) S_e X_q)  + fused dots  (mul_rk + axpy + dot/nrm), neighbour list read 8 slots
//                per trip; <FRONT>: right-hand side AND initial residual of a CG solve from one gather of the rows
//   k_spmm_ell   the operator's second half, slot coefficient a w_i, fixed-width slot list (k_spmm<CW>: CSR fallback)
//   k_refresh_w  constraint bookkeeping after a solve when A(sym(U V^T)) has followed the CG updates (no gather pass)
//   k_op_diag    fused operator when every A_i = a e_p e_p^T (Max-Cut): one pass
//   k_op_entry   fused operator when every A_i holds one entry (matrix completion)
//   k_dense_cx   W = C X for a dense objective on the FP64 matrix cores (v_mfma_f64_16x16x4_f64)
//   k_cg_*       CG vector updates with device-resident scalars   (CGSolve); *_seg: many cones in lockstep
//   k_lbfgs_stage, k_pairdots_rd, k_cv_rd, k_obj_rd, k_alm_update, k_alm_tail, k_linesearch   phase 1
//   k_lp_*       LP block: level-scheduled closed-form column sweep, column values, dual infeasibility
//   k_eval_*, k_publish[_final]   closing sums of an evaluation, result hand-over to the host
//
// Launch structure: one ADMM iteration (2 CG solves per cone, constraint refreshes, objective,
// DIMACS) is enqueued on one stream WITHOUT host round trips, using the iteration counts of the
// previous ADMM iteration as a speculation and device-side gates (struct Guard); the host
// synchronises once per ADMM iteration and resumes a solve that needed more iterations.
// The scalar steps of CGSolve (start of a solve, convergence test, restart bookkeeping) are not launches
// of their own: they ride on the neighbouring kernel (structs InitArgs / Deferred / Carry below).
//
// Reductions: every reducing kernel writes one partial per workgroup; the consumer re-sums the
// partials (<= 4096) in a fixed order, so results do not depend on scheduling.  Wave-level sums use
// 64-lane __shfl_xor butterflies.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#include <tuple>
#include <type_traits>
#include <vector>

#include "lorads_hip.h"
#include "lorads_hip_dev.h"

namespace {

constexpr int TPB = 256;
constexpr int MINPART = 4096; // least capacity of one partial-sum slot; a context sizes its slots to its largest cone (lorads_hip_ctx::maxpart)
constexpr int SX_WORD = 56; // control-block words [56, 60): a rank's local sums of an evaluation on their way to its host (lorads_hip_set_scalar_exchange)
constexpr int NSLOT = 20; // (18, 19: start-of-solve and restart residual partials of the ADMM sweep, see solve_front)

thread_local std::string g_err;
int fail(const char *what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int fail_msg(const std::string &m) {
    g_err = m;
    return 1;
}
#define HC(call)                                        \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) return fail(#call, e__); \
    } while (0)

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int LG>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = LG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over the 256-thread workgroup, result in every thread; sh = 4 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return t;
}
// N sums over the workgroup with one barrier pair; sh = 4*N doubles of LDS, results in every thread.  Each value is
// reduced exactly as block_sum reduces it.
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *sh) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        v[k] = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) sh[k * 4 + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = (sh[k * 4] + sh[k * 4 + 1]) + (sh[k * 4 + 2] + sh[k * 4 + 3]);
    __syncthreads();
}
// this thread's share of n partial sums, part[tid], part[tid + TPB], ... added in that order.  Four loads are issued
// per trip, unconditionally (clamped index, masked value): the plain loop `v += part[i]` compiles to one load + full
// wait per trip, i.e. n / TPB memory round trips in a row at the head of every kernel that consumes partials
__device__ __forceinline__ double private_partials(const double *__restrict__ part, int n) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 4 * TPB) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = i + k * TPB;
            t[k] = part[j < n ? j : i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v += (i + k * TPB < n) ? t[k] : 0.0;
    }
    return v;
}
__device__ __forceinline__ double sum_partials(const double *part, int n, double *sh) {
    return block_sum(private_partials(part, n), sh);
}
// the same for N arrays at once (all 4 N loads of a trip in flight together); every pointer must be readable at [0]
template <int N>
__device__ __forceinline__ void private_partials_n(const double *const (&part)[N], const int (&n)[N], double (&v)[N]) {
    int nmax = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) { v[a] = 0.0; nmax = n[a] > nmax ? n[a] : nmax; }
    for (int i = threadIdx.x; i < nmax; i += 4 * TPB) {
        double t[N][4];
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = i + k * TPB;
                t[a][k] = part[a][j < n[a] ? j : 0];
            }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[a] += (i + k * TPB < n[a]) ? t[a][k] : 0.0;
    }
}

struct CGState {
    double rr, bnorm, beta;
    int done; // 0 running, 1 converged, 2 converged at the initial residual, 3 max iterations
    int iter;
    int nan;
    int pad;
};
// Launch gating.  A whole ADMM iteration is enqueued without host round trips: kernels of a CG solve
// exit at once when that solve is finished (`skip` -> its done word != 0), and everything that comes
// after a solve (the next solve, the constraint refresh, the evaluation) exits at once when that solve
// is NOT finished yet (`need` -> its done word == 0); the host then resumes from exactly that point.
struct Guard {
    const int *skip;
    const int *need;
};
// Split in two so that the gate words travel while the kernel issues its own loads: gate_load() at the top requests both
// words at once, without a branch between them (a null pointer borrows the other word's address), GATE_OPEN looks at them
// where the first predicated store needs the answer.  (`(skip && *skip) || (need && *need == 0)` at the top of a kernel
// compiles to load - wait - branch - load - wait: two memory round trips before the kernel's first own load.)
struct Gate { int vs, vn; };
__device__ int g_gate_words[2] = {0, 1}; // what an absent gate reads: "do not skip", "satisfied"
__device__ __forceinline__ Gate gate_load(const Guard &g) {
    // two UNCONDITIONAL loads (an absent gate reads the constant words): a load inside an `if` makes the compiler wait
    // for it -- and for every load issued before it -- where the branches join
    const int *ps = g.skip ? g.skip : &g_gate_words[0], *pn = g.need ? g.need : &g_gate_words[1];
    Gate t;
    t.vs = *ps;
    t.vn = *pn;
    return t;
}
__device__ __forceinline__ bool gate_blocked(const Guard &g, const Gate &t) { return t.vs != 0 || t.vn == 0; }
#define GATE_OPEN (!gate_blocked(g, gt))
__device__ __forceinline__ bool blocked(const Guard &g) { return gate_blocked(g, gate_load(g)); }
// The gate words were written by the previous kernel, so reading them costs a cache-missing scalar load
// (~1.5 us when it heads the kernel).  Kernels therefore evaluate `live` first but only USE it to predicate
// their stores: the gate loads fly together with the kernel's own first loads.  A blocked kernel computes on
// whatever is there and writes nothing.

// Scalar steps of CGSolve that ride on a neighbouring kernel instead of being one-workgroup launches of their own
// (each such launch costs ~4.5 us + a boundary for a handful of flops):
//  * InitArgs -- the start of a solve (k_cg_init: ||rhs||_1, ||r_0||, converged at once?) evaluated by the FIRST kernel of
//    iteration 0 (k_cw): every workgroup sums the same partials in the same order and gates itself on the result,
//    workgroup 0 writes the state for the kernels that follow.  Nothing of the old state is read, so there is no race.
//  * Deferred -- the convergence test after an update (k_cg_check, CHK_ITER) evaluated by the kernel that FOLLOWS the
//    update (k_cg_dir, k_refresh_w, k_average).  The test reads the state before it; that state is taken from `shadow`,
//    a copy k_cg_update (and the solve's init) leave behind, so that workgroup 0 can write the new state into `st`
//    while the other workgroups are still reading.  Re-running a test whose update was skipped (solve already finished)
//    reproduces the same state from the same shadow and partials.
// A scalar of the step that is either passed by value or read from the device-resident step parameters (lorads_hip_ctx::par).
// The launch chain of an ADMM iteration that is replayed as a captured hipGraph must not carry rho, the CG tolerance or the
// iteration limit in its kernel ARGUMENTS (they are frozen at capture): such a chain is enqueued with p != null, and the host
// changes the values with one tiny launch (k_set_par) only when they change (rho every rhoFreq iterations, lorads_admm.c:121-138;
// the tolerance follows the primal infeasibility, :76).  Same bits either way.
struct DS {
    const double *p;
    double v;
    __host__ __device__ DS(double x = 0.0) : p(nullptr), v(x) {}
    __host__ __device__ DS(const double *q, double x) : p(q), v(x) {}
};
__device__ __forceinline__ double dsv(const DS &s) { return s.p ? *s.p : s.v; }
struct InitArgs {
    CGState *st;              // nullptr: nothing to do
    CGState *shadow;
    const double *part_rr, *part_b;
    int nrr, nb;
    DS tol;
};
struct Deferred {
    CGState *st;              // nullptr: nothing pending
    const CGState *shadow;
    const double *part;       // partials of ||r||^2 left by the update
    int np;
    DS maxit;                 // (an integer kept as a double)
    DS tol;
    const int *need;          // the solve's own gate (previous stage finished)
};
struct DefOut { int done; double beta; };
// in two halves, like the carried start: the loads first (before the carrier's own), the decision where it is needed
struct DefPriv { double rr_old, bnorm, beta0, a, tol; int it0, done0, needw, maxit; };
__device__ __forceinline__ DefPriv deferred_private(const Deferred &d) {
    DefPriv v;
    v.rr_old = d.shadow->rr; v.bnorm = d.shadow->bnorm; v.beta0 = d.shadow->beta;
    v.it0 = d.shadow->iter; v.done0 = d.shadow->done;
    v.needw = *(d.need ? d.need : &d.shadow->done); // (no branch around the load)
    v.tol = dsv(d.tol); v.maxit = (int)dsv(d.maxit);
    v.a = private_partials(d.part, d.np);
    return v;
}
__device__ __forceinline__ DefOut finish_deferred(const Deferred &d, const DefPriv &v, double *sh) {
    const bool open = !(d.need && v.needw == 0);
    const double a = block_sum(v.a, sh);
    DefOut o;
    if (!open) { o.done = 0; o.beta = 0.0; return o; }                // the whole stage is still gated: its words read 0
    if (v.done0 != 0) { o.done = v.done0; o.beta = v.beta0; return o; } // finished before this test (e.g. at the initial residual)
    const int it = v.it0 + 1;                                           // lorads_cgs.c:189-194, :217-224 (as k_cg_check)
    o.beta = a / v.rr_old;
    o.done = sqrt(a) / v.bnorm < v.tol ? 1 : (it >= v.maxit ? 3 : 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.st->iter = it;
        if (a != a) d.st->nan = 1;
        d.st->beta = o.beta;
        d.st->rr = a;
        if (o.done) d.st->done = o.done;
    }
    return o;
}
__device__ __forceinline__ DefOut run_deferred(const Deferred &d, double *sh) { return finish_deferred(d, deferred_private(d), sh); }
// gate of a kernel that carries a Deferred test: the word of the solve under test is the test's result, not memory
__device__ __forceinline__ bool blocked_after(const Guard &g, const Deferred &d, int done_now) {
    const int sk = g.skip ? (g.skip == &d.st->done ? done_now : *g.skip) : 0;
    const int nd = g.need ? (g.need == &d.st->done ? done_now : *g.need) : 1;
    return sk != 0 || nd == 0;
}

// what a kernel may carry: the start of a solve or a convergence test (never both)
struct Carry {
    InitArgs ia;
    Deferred d;
};

// k_cg_init in two halves for a carrier that wants its own loads in between: per-thread sums of the partials ...
__device__ __forceinline__ void init_private(const InitArgs &ia, double (&iv)[2]) {
    const double *const pp[2] = {ia.part_rr, ia.part_b};
    const int nn[2] = {ia.nrr, ia.nb};
    private_partials_n<2>(pp, nn, iv);
}
// ... and the workgroup reduction + decision (same sums as sum_partials); returns the solve's `done` word or -1 (gated)
__device__ __forceinline__ int finish_init(const InitArgs &ia, bool open, double (&iv)[2], double *sh8) {
    block_sum_n<2>(iv, sh8);
    const bool conv = sqrt(iv[0]) / iv[1] < dsv(ia.tol);
    if (!open) return -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CGState s;
        s.rr = iv[0]; s.bnorm = iv[1]; s.beta = 0.0; s.iter = 0; s.nan = 0; s.pad = 0;
        s.done = conv ? 2 : 0;
        *ia.st = s;
        *ia.shadow = s;
    }
    return conv ? 2 : 0;
}
// k_cg_init carried by the first kernel of iteration 0 (see InitArgs): returns the solve's `done` word, or -1 while the
// stage is gated; every workgroup computes it, workgroup 0 writes the state
__device__ __forceinline__ int run_init(const InitArgs &ia, const Guard &g, double *sh) {
    const bool open = !(g.need && *g.need == 0);
    const double a = sum_partials(ia.part_rr, ia.nrr, sh);
    const double b = sum_partials(ia.part_b, ia.nb, sh);
    const bool conv = sqrt(a) / b < dsv(ia.tol);
    if (!open) return -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CGState s;
        s.rr = a; s.bnorm = b; s.beta = 0.0; s.iter = 0; s.nan = 0; s.pad = 0;
        s.done = conv ? 2 : 0;
        *ia.st = s;
        *ia.shadow = s;
    }
    return conv ? 2 : 0;
}

enum { W_COMPACT = 0, W_ADMM = 1, W_ALM = 2, W_DUAL = 3, W_ADMM_V = 4, W_ADMM_VS = 5 }; // (VS: as V, the row dot formed by the front itself, see k_spmm2)
enum { OP_CG = 0, OP_RES = 1, OP_RHS = 2, OP_GRAD = 3 };
enum { CHK_ITER = 1, CHK_RESTART = 2 };
enum { DIR_BETA = 1, DIR_RESTART = 2 };
enum { CV_SET = 0, CV_ADD = 1, CV_DELTA = 2 };

#include "kernels.inc"
#ifdef LORADS_DEV_BUILD
#include "dev_kernels.inc"
#endif

// ------------------------------------------------------------------ host side
template <typename T>
int dalloc(T **p, size_t n) {
    *p = nullptr;
    HC(hipMalloc((void **)p, sizeof(T) * (n > 0 ? n : 1)));
    return 0;
}
template <typename T>
int upload(T **p, const std::vector<T> &v) {
    if (dalloc(p, v.size())) return 1;
    if (!v.empty()) HC(hipMemcpy(*p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return 0;
}

struct Pattern {              // one symmetric sparsity pattern with everything the kernels need
    int ne = 0;               // unique lower-tri entries
    int nslot = 0;            // adjacency slots
    int n_owned = 0;          // constraints that have an owner pair in e_con_own
    int *erow = nullptr, *ecol = nullptr;
    int *e_ptr = nullptr, *e_con = nullptr; // entry -> (local constraint, value)
    int *e_con_own = nullptr; // e_con with the sign bit set on ONE pair per constraint (its "owner": k_sval with a dual update on board); null if some constraint has no entry
    double *e_val = nullptr;
    int *adj_ptr = nullptr, *adj_col = nullptr, *adj_e = nullptr; // row -> (neighbour, entry)
    int *adj_dyn = nullptr;    // per slot: the entry if some constraint touches it (its coefficient changes: gathered from S), else -1
    double *adj_sval = nullptr; // per slot: the coefficient of an entry no constraint touches (C on the pattern; streamed with the list)
    double *S = nullptr;      // values on the pattern
    double *S2 = nullptr;     // 2 ne doubles: {S_e, second image sum_i w_i A_i} side by side for the fused front of a CG solve (one 16-byte gather per slot); union pattern of k_cw cones only
    double *cbase = nullptr;  // C on the pattern (union pattern only)
    void release() {
        hipFree(erow); hipFree(ecol); hipFree(e_ptr); hipFree(e_con); hipFree(e_con_own); hipFree(e_val);
        hipFree(adj_ptr); hipFree(adj_col); hipFree(adj_e); hipFree(adj_dyn); hipFree(adj_sval); hipFree(S); hipFree(S2); hipFree(cbase);
    }
};

struct Block {
    int n = 0, r = 0, nrow = 0, na = 0, nc = 0;
    // r is the rank the kernels see: the cone's rank rl rounded up to even (dev_rank), the extra column zero in every factor.  A zero
    // column stays zero through every step of both phases (each column of S V, of the operator's result, of a gradient or direction is
    // built from the same column of its input), and every even rank takes the 16-byte row accesses and the kernels that need them
    // (k_front_cw, the quad form of k_cw): r = 41 ran at 0.175 ms per headline iteration, r = 42 at 0.131.
    int rl = 0;
    size_t off = 0;           // offset of this cone in the flat factor arrays
    int *row_idx = nullptr;
    bool row_idx_identity = false; // row_idx[i] == i for every local constraint (a cone that sees all constraints in order)
    // Every entry of the union pattern that a constraint touches is touched by exactly ONE (sv_direct): the coefficient of such an entry
    // is cbase + weight a of that one constraint, so whoever holds the constraint's weight can write it -- phase 1's k_alm_update does,
    // and k_sval is not launched (constraint -> its entries of the union pattern, CSR: sv_ptr, sv_e, sv_a)
    bool sv_direct = false;
    int *sv_ptr = nullptr, *sv_e = nullptr;
    double *sv_a = nullptr;
    int *a_ptr = nullptr, *a_e = nullptr;  // constraint CSR over the A-pattern
    double *a_val = nullptr;
    Pattern pa, pu;
    bool has_gram = false;    // G = A A^T over A-pattern entries
    int *g_ptr = nullptr, *g_col = nullptr;
    double *g_val = nullptr;
    double *T = nullptr;      // pair dots on the A-pattern
    bool cv_borrowed = false; // cv points into the merged cone's array
    double *cv = nullptr;     // constrVal[k], compact
    double *wtmp = nullptr;   // compact weights inside the CG operator
    int *c_row = nullptr, *c_col = nullptr;
    double *c_val = nullptr;
    bool dense_c = false;     // C stored dense (reference rule nnz > 0.1 n(n+1)/2): C X runs on MFMA
    int npad = 0;
    double *Cfull = nullptr;  // npad x npad row-major, symmetric, zero padded
    double *Wd = nullptr;     // n x r result of C X
    double *Wpart = nullptr;  // split-K slabs of it
    int ksplit_b = 0;         // K split over workgroups of the second dense form (k_dense_cx_b; 0: not applicable)
    // DENSE constraint matrices (the reference's sdp_coeff_dense rule, data/lorads_sdp_data.c:820: nnz > 0.1 n(n+1)/2): kept
    // out of the sparse patterns (their rows of the constraint CSR are empty there) and stored as full symmetric npad x npad
    // matrices; their part of every A / A^* runs through the dense GEMM (k_dense_cx_b) -- see dense_constr_fix / dense_part
    bool dense_a = false;
    int nd = 0;               // how many
    int *d_con = nullptr;     // [nd] local constraint index
    std::vector<int> d_con_h; // (host copy)
    double *Adense = nullptr; // [nd][npad * npad]
    double *Sfull = nullptr;  // npad * npad scratch: [C +] sum_j mu_j A_j
    double *d_mu = nullptr;   // [nd] the weights of the current combination
    double *Wj = nullptr;     // [nd][n * r]: W_j = A_j Y of the dense constraint matrices for the factor Y = wj_for (dense_cache_fill) ...
    double *Wc = nullptr;     // ... and C Y of a dense objective beside them
    const double *wj_for = nullptr; // the factor array the kept products belong to (null: none)
    bool t_uv_valid = false;  // B.T currently holds the pair dots of (U,V) (symmetric in the pair)
    double *T2 = nullptr;     // second pair-dot buffer (evaluation on R) so that T(U,V) survives it
    bool diag_only = false;   // every A_i is a single diagonal entry (Max-Cut)
    double *gdiag = nullptr;
    int *diag_row = nullptr;  // Max-Cut-type cone: row of constraint i ...
    int rc_w = 0;             // Max-Cut-type cone: constraints per row, fixed width (k_eval_diag; 0: a row holds too many)
    int *rc_con = nullptr;    // [n * rc_w] row -> constraint (-1 = none)
    double *diag_a = nullptr; // ... and its coefficient (the cone then keeps w_uv = row dots U_p . V_p and w_op = p_p . V_p, n each)
    bool use_cw = false;      // operator = k_cw (constraint values from the factors) + k_spmm<CW> (slot coefficient a w_i)
    int *cadj_ptr = nullptr, *cadj_col = nullptr, *cadj_con = nullptr; // row -> (neighbour, compact constraint, a)
    int *ca_row = nullptr, *ca_col = nullptr; // (row, col) of every constraint entry, in constraint-CSR order
    double *ca_val = nullptr;                 // ... and its coefficient; all three padded to ca_ell per constraint when ca_ell > 0
    int ca_ell = 0;
    double *cadj_a = nullptr;
    int cadj_ptr_host_n = 0;                  // number of slots in the CSR list (cadj_col / cadj_con / cadj_a)
    int cell_w = 0;                           // > 0: the slot list also in fixed width (8 or 16 per row) for k_spmm_ell
    int *cell_col = nullptr, *cell_con = nullptr;
    double *cell_a = nullptr;
    double *w_uv = nullptr, *w_op = nullptr; // A(sym(U V^T)) kept for re-use (valid <=> t_uv_valid); operator scratch
    // k_front_cw (the whole front of a solve without a coefficient pass): the objective's own adjacency row -> (neighbour, c),
    // constraint -> positions of its slots in the CSR slot list (k_wsum), one contribution per slot
    bool front_cw = false;
    int *fc_ptr = nullptr, *fc_col = nullptr;
    double *fc_val = nullptr;
    int fc_nslot = 0;
    int cs_w = 0;             // slots per constraint in the contribution array (fixed width, padding stays 0)
    int *cell_dst = nullptr, *cadj_dst = nullptr; // slot (fixed-width / CSR numbering) -> its place in the contribution array
    double *w_contrib = nullptr;
    bool w0_ready = false;    // w_contrib holds the contributions of A(sym(r0 V^T)) for the residual the front has just left in cr
    bool is_lp = false;       // the LP block: generic diagonal cone everywhere except the ADMM update (k_lp_sweep)
    int lp_nlev = 0;
    int *lp_lvl_ptr = nullptr, *lp_lvl_cols = nullptr, *lp_ptr = nullptr, *lp_grow = nullptr;
    double *lp_a = nullptr, *lp_nrm2sq = nullptr, *lp_cobj = nullptr, *lp_cv = nullptr;
    bool entry_only = false;  // every A_i is a single (off-)diagonal entry (matrix completion): k_op_entry
    double *gentry = nullptr; // sum of a_i^2 per A-pattern entry
    int *bip_rows[2] = {nullptr, nullptr}; // single-entry cone whose entry graph is bipartite: the rows of either colour (k_op_entry_bip)
    int bip_n[2] = {0, 0};
    double *bip_we = nullptr;  // per entry: ge (x_p.V_q + x_q.V_p) of the running operator application, colour 0 -> colour 1
    int cg_iter_last = 0;     // lorads_cg_linsys.iter survives an immediate exit (lorads_cgs.c:157-160,173)
    int spec[2] = {1, 1};     // speculated CG iterations of the U- and V-solve: the largest count of the last few sweeps
    int spec_hist[2][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1}};
    int spec_pos = 0;
    double bytes_mv = 0, bytes_cg = 0;
};

struct Ring {
    double *s = nullptr, *y = nullptr;
};
struct GraphCache; // captured launch chains (graph.inc)
struct PersistPlan; // the one-launch ADMM iteration of Max-Cut-type cones (persist.inc)
struct LTeamPlan;   // the one-launch L-BFGS history update + direction of phase 1 (lbfgs_team.inc)

} // namespace

constexpr size_t LZ_PINNED = 256; // doubles of pinned read-back per Lanczos worker (covers ncv <= 126)
struct LzWorker {
    hipStream_t stream = nullptr; // nullptr: the context's own stream (the first worker of a one-thread run)
    double *pinned = nullptr;
};

struct lorads_hip_ctx {
    int m = 0, nb = 0, L = 2;
    double b_nrm1 = 0;
    hipStream_t stream = nullptr;
    std::vector<Block> blk;
    std::vector<LzWorker> lz_workers; // dual-infeasibility eigen-solves (lanczos.inc)
    Block merged;             // all cones as ONE block-diagonal cone (see build_merged); valid when has_merged
    bool has_merged = false;
    std::vector<int> seg_row0_h;              // padded first row of every cone in the merged cone (+ end)
    int *seg_row0 = nullptr, *seg_vt0 = nullptr, *seg_vt_seg = nullptr;
    long long *seg_vt_e0 = nullptr;
    int seg_nvt = 0;
    int *phase_done = nullptr;                // [2]
    int spec_b[2] = {1, 1};                   // speculated lockstep iterations of the U and V phase
    int spec_b_hist[2][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1}};
    int spec_b_pos = 0;
    int spec_window = 4;                      // sweeps looked back at (LORADS_SPEC_WINDOW, 1 = the previous sweep's count alone)
    bool merged_ok = false;   // structure allows it (separable constraints, no dense C); ranks decide has_merged
    size_t all_elem = 0;
    double *R = nullptr, *U = nullptr, *V = nullptr, *G = nullptr;   // flat factors
    double *cr = nullptr, *cp = nullptr, *cQ = nullptr, *rhs = nullptr; // flat CG vectors
    double *Dtmp = nullptr;
    double *cstage = nullptr; // m+2 (+ objective partials): [local constrValSum | objective part | miss flag | partials] on its way through the all-reduce
    bool stage_clean = false; // cstage[0..m) is known to be zero (left so by k_commit_eval)
    int ar_fast_all = -1;     // every rank holds one cone that can send its objective partials through the all-reduce (-1: not yet agreed)
    bool opt_ar_fast = true;
    double *b = nullptr, *lambda = nullptr, *csum = nullptr, *q12 = nullptr; // csum: m+2, q12: 2m+2
    double *part = nullptr;   // NSLOT x maxpart partial sums
    int maxpart = MINPART;    // capacity of one slot: one partial per workgroup of the widest row kernel of the largest cone (4 rows per
                              // workgroup at 64 lanes per row), so no cone dimension and no rank is refused for want of partial-sum room
    int ls_np = 0;            // line-search partials (slots 10..16) currently valid for q1, q2: how many per sum
    char *ctrl = nullptr, *h_ctrl = nullptr; // [64 scalars | CG states] device + pinned mirror: ONE readback copy
    char *h_ctrl_dev = nullptr;              // device address of the pinned mirror (k_publish writes it directly)
    unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr, pub_seq = 0; // published sequence number (the host's count)
    unsigned long long *seq_dev = nullptr;   // ... and the device's: every hand-over kernel bumps it (see k_publish)
    // Step parameters in device memory {rho, CG tolerance, rho of the pending dual update, CG iteration limit} (see DS): the launch
    // chain of an ADMM iteration refers to them by address while par_mode is on, so that a captured chain can be replayed
    double *par = nullptr;
    double par_h[4] = {0, 0, 0, 0};          // what the device holds
    int par_valid = 0;                       // bit k: par_h[k] is on the device
    bool par_mode = false;
    const char *launch_err = nullptr;        // a step that could not be enqueued (reported by the next hand-over / read-back)
    bool par_changed_last = false;           // the previous step's parameters differed from the step's before it (see graph_this_step)
    double par_req[3] = {0, 0, 0};           // {rho, tolerance, iteration limit} the previous step asked for
    bool par_req_valid = false;
    bool opt_graph = true;                   // LORADS_GRAPH=0: every iteration enqueued launch by launch
    bool opt_graph_batched = false;          // LORADS_GRAPH=2: the lockstep sweep of a merged cone is replayed too
    bool opt_graph_forced = false;           // LORADS_GRAPH=1 or 2: replay whatever the size (default: small cones only, see graph_ok)
    GraphCache *graphs = nullptr;            // captured launch chains by shape (graph.inc)
    PersistPlan *persist = nullptr;          // teams of resident workgroups, one launch per ADMM iteration (persist.inc; LORADS_PERSIST=0: off)
    bool opt_persist = true;                 // (read at creation)
    int shared_gpu_fd = -1;                  // LORADS_SHARED_GPU=1: lock file of the device (several processes on one card take turns, see run_sweep_persist)
    bool opt_persist_carry = true;           // the evaluation leaves (C V) for the next iteration's U front (LORADS_PERSIST_CARRY=0: every front gathers)
    bool opt_persist_l2 = true;              // granules / rows of workgroups verified to share an XCD go through its L2 (LORADS_PERSIST_L2=0: always written through)
    bool persist_stamps = false;             // team 0's leader leaves its phase times (lorads_hip_persist_stamps)
    long long n_persist = 0;                 // ADMM iterations run that way
    long long n_launch = 0;                  // kernels enqueued through LAUNCH / the one-launch forms (lorads_hip_launch_count: bench.py's launches per step)
    LTeamPlan *lteam = nullptr;              // phase 1: setlbfgsHisTwo + LBFGSDirection as one launch of resident workgroups (lbfgs_team.inc)
    bool opt_lbfgs_team = true;              // (LORADS_LBFGS_TEAM=0: launch by launch)
    bool opt_alm_sval_direct = true;         // ... and k_alm_update writes the entries' coefficients where Block::sv_direct holds (LORADS_ALM_SVAL_DIRECT=0: k_sval)
    bool opt_alm_fold_cv = true;             // ... where every constraint has one entry the pattern pass does the constraint pass's work (LORADS_ALM_FOLD_CV=0)
    bool opt_alm_fused_tail = true;          // ... and, behind it, shared passes for A(R R^T), q1, q2 and one closing workgroup (LORADS_ALM_FUSED_TAIL=0)
    bool use_publish = true;
    // LORADSUpdateDualVar waiting for the first kernel of the next sweep (k_sval of the U-solve's front forms the weights
    // from the updated multipliers and stores them to lambda_alt, then the two vectors swap); sent off as k_dual_update
    // by flush_pending if anything else comes first
    bool pend_dual = false;
    double pend_dual_rho = 0.0;
    double *lambda_alt = nullptr;
    bool virt_refresh = false; // the V-solve's front forms its weights as if the refresh after the U-solve had been stored (see enqueue_sweep)
    bool opt_entry_bip = true; // single-entry cones with a bipartite entry graph: k_op_entry_bip x 2 (LORADS_ENTRY_BIP=0: k_op_entry)
    bool opt_fuse_dir = true; // Max-Cut-type cones: the direction update inside the operator kernel (LORADS_FUSE_DIR=0: k_cg_dir)
    bool opt_cw_quad = true;  // k_cw with 4 lanes per entry where it applies (LORADS_CW_QUAD=0: 8 lanes)
    bool opt_fuse_eval = true; // single cone on the k_cw path: constraint values and objective partials in one launch (LORADS_FUSE_EVAL=0)
    bool opt_dense_rem = true; // dense GEMM: 1..4 columns beyond the full tiles on plain FMAs instead of a tile of their own (LORADS_DENSE_REM=0)
    bool opt_pad_rank = true; // odd ranks run as the next even rank with a zero column (LORADS_PAD_ODD_RANK=0: as they are)
    bool opt_dev_presolve = true;   // pattern work of the pre-solve on the device (presolve.inc; LORADS_DEV_PRESOLVE=0: host)
    bool opt_presolve_check = false; // LORADS_PRESOLVE_CHECK=1: build every device pattern on the host too and compare
    size_t dev_presolve_min = (size_t)1 << 15; // stored entries below which a pattern is built on the host (LORADS_DEV_PRESOLVE_MIN)
    long long n_dev_patterns = 0, n_checked_patterns = 0;
    bool opt_dense_cache = true; // dense constraint matrices: A_j V kept for the length of a CG solve (LORADS_DENSE_CACHE=0: nd + 1 GEMMs per application)
    bool opt_front_diag = true; // Max-Cut-type cones: the front forms its diagonal coefficients itself, no k_sval (LORADS_FRONT_DIAG=0)
    bool opt_eval_diag = true; // Max-Cut-type cones: k_eval_diag instead of k_average + k_pairdots + k_cv_res (LORADS_EVAL_DIAG=0)
    bool opt_fold_avg = true; // the sweep's last k_cg_update also forms R = (U + V) / 2 (LORADS_FOLD_AVG=0: k_average)
    bool opt_tile_update = true; // Max-Cut-type cones: k_cg_update on the row tiles of k_op_diag (LORADS_TILE_UPDATE=0: grid-stride over the flat vector)
    bool avg_folded = false;  // ... and has done so for the evaluation that is enqueued next
    bool opt_front_lds = true; // k_front_cw parks the first four slot rows in LDS for its second visit (LORADS_FRONT_LDS=0: gathered again)
    bool opt_front_cw = true; // k_front_cw + k_wsum instead of k_sval + k_spmm2<FRONT> + iteration 0's k_cw (LORADS_FRONT_CW=0: the latter)
    bool pend_dual_virtual = false; // the pending dual update has already been USED (formed on the fly by k_front_cw) but not stored
    bool opt_exact_refresh = false, opt_split_front = false; // test knobs (read at creation): see constr_by_recurrence, fused_front
    bool final_pending = false;              // an evaluation's closing sums wait for the next hand-over (k_publish_final)
    EvalFinalArgs final_args;
    double *scal = nullptr;   // 64 device scalars
    CGState *st = nullptr;    // one per (cone, half)
    CGState *st_shadow = nullptr; // state before a pending convergence test (see Deferred)
    InitArgs pend_init{};     // a solve's start waiting for the kernel that will carry it (flushed as k_cg_init otherwise)
    Guard pend_init_g{};
    Deferred pend_chk{};      // a convergence test waiting likewise (flushed as k_cg_check otherwise)
    Guard pend_chk_g{};
    bool opt_lazy_scalars = true; // LORADS_LAZY_SCALARS=0: every scalar step as its own launch
    // a direction update (k_cg_dir) waiting for the row-local operator that follows it (k_op_diag forms the rows of p
    // itself); sent off as its own launch if anything else comes first
    struct PendDir { int kind = 0; CGState *st = nullptr; double *r = nullptr, *p = nullptr; size_t len = 0; int gv = 0;
                     const double *rs_part = nullptr; int rs_np = 0; Guard g{}; bool seg = false; int half = 0;
                     const double *chk_part = nullptr; double chk_tol = 0; int chk_maxit = 0, chk_k = 0; // chk_*: lockstep test riding with the direction
                     const double *rs_seg = nullptr; } pend_dir; // rs_seg: lockstep restart's scalars riding with it (chk_tol / chk_maxit / chk_k set too)
    struct PendSegChk { bool on = false; int half = 0, k = 0; const double *part = nullptr; double tol = 0; int maxit = 0;
                        bool ride_res = false; } pend_segchk; // ride_res: the restart's residual pass that follows may take it (see op_diag)
    struct PendSegInit { bool on = false; int half = 0; double tol = 0; Guard front{}; const double *part_rr = nullptr, *part_b = nullptr; } pend_seginit; // lockstep sweep: the solves' start waiting for iteration 0's operator kernel
    bool opt_seg_carry_init = true; // (LORADS_SEG_CARRY_INIT=0: k_cg_init_seg as a launch of its own)
    // separable shards: the evaluation's four scalars summed by the ranks' hosts after the hand-over (lorads_hip_set_scalar_exchange)
    lorads_hip_scalar_exchange_fn sx = nullptr;
    void *sx_user = nullptr;
    bool sx_pending = false; // the local sums are on their way to the mirror (scal[SX_WORD..+4)); the next wait_publish exchanges them
    int sx_with_obj = 0;
    long long n_sx = 0;
    bool opt_seg_virt = true; // lockstep sweep, Max-Cut-type merged cone, evaluation at the end: the refresh after the U-solves is not stored, the V front forms its weights from U_p.V_p itself (LORADS_SEG_VIRT=0: k_pairdots + k_cv)
    bool opt_seg_carry_dual = true; // lockstep sweep: a waiting dual update rides on the U front of the merged cone (LORADS_SEG_CARRY_DUAL=0: k_dual_update)
    bool opt_seg_carry_restart = true; // lockstep sweep: the k % 20 == 0 restart's test and scalars ride on its two operator kernels (LORADS_SEG_CARRY_RESTART=0)
    double *seg_rr_alt = nullptr; // second slot of every stage's r.r (see SegArgs)
    int *seg_tile_info = nullptr; // int4 per row tile of the merged cone (see DirArgs.seg_info)
    bool opt_seg_carry = true;    // LORADS_SEG_CARRY=0: k_cg_check_seg after every update of the lockstep sweep
    int *seg_tile_cone = nullptr;             // row tile (workgroup of a row kernel) -> cone, merged cone
    CGState *h_st = nullptr;  // pinned mirror
    double *h_scal = nullptr; // pinned mirror of scalars
    std::vector<Ring> ring;
    double *ring_ab = nullptr; // [2L] alpha,beta per node
    int head = 0;
    lorads_hip_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    bool ar_stream_ordered = false; // the hook enqueues on our stream (RCCL): no host sync around it
    bool sep = false; // lorads_hip_set_separable: this context holds one rank's own constraints; only scalars cross the ranks
    double *sepbuf = nullptr; // 32: scalars on their way through the all-reduce in that mode
    SepExtra sep_extra{};     // scalars the next evaluation's all-reduce also carries (set by the caller, taken by enqueue_eval)
    bool sep_pending = false; // the reduced scalars of a separable evaluation wait for the hand-over kernel to commit them (k_publish_sep)
    int sep_with_obj = 0;
    SepExtra sep_ex{};
    double *gram = nullptr;   // 128: [0..66) products V_a.V_b of the Gram-form L-BFGS direction, [80..91) the coefficients of D
    bool opt_gram = true;     // LORADS_LBFGS_GRAM=0: sharded direction by the sequential recursion, one collective per dot
    bool opt_gram_single = false; // LORADS_LBFGS_GRAM=2: the Gram-form direction on a single GPU as well
    // profiling
    int prof = 0, prof_every = 8;
    int prof_target = 0;      // what a profiling window times: 0 = CG operator applications, 1 = solve fronts (lorads_hip_profile_target)
    long n_front = 0;
    long n_sweeps = 0;        // ADMM sweeps run so far (cadence of the exact constraint refresh, see constr_by_recurrence)
    long n_matvec = 0, n_cg_it = 0, n_solves = 0, n_samp = 0, n_samp_spmm = 0, n_resume = 0;
    double ms_samp = 0, ms_samp_spmm = 0;
    std::vector<float> samp_ms; // every timed operator application of the current profiling window
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend_mv, pend_sp;
    std::vector<hipEvent_t> ev_pool; // pre-created (hipEventCreate is too slow for the timed region)
    size_t ev_next = 0;
};

namespace {

inline DS ds_rho(const lorads_hip_ctx *c, double v) { return c->par_mode ? DS(c->par + 0, v) : DS(v); }
inline DS ds_tol(const lorads_hip_ctx *c, double v) { return c->par_mode ? DS(c->par + 1, v) : DS(v); }
inline DS ds_rho_dual(const lorads_hip_ctx *c, double v) { return c->par_mode ? DS(c->par + 2, v) : DS(v); }
inline DS ds_maxit(const lorads_hip_ctx *c, int v) { return c->par_mode ? DS(c->par + 3, (double)v) : DS((double)v); }
inline int nblocks_for(size_t items, int per_block) { return (int)((items + per_block - 1) / per_block); }
// grid of the L-BFGS stage kernels: every workgroup re-sums the previous stage's partials, so keep them few
inline int grid_lbfgs(size_t len);
inline int grid1d(size_t len) {
    size_t g = (len + TPB - 1) / TPB;
    return (int)std::min<size_t>(std::max<size_t>(g, 1), 2048);
}
inline int grid_lbfgs(size_t len) {
    return std::min(grid1d(len), 512);
}
// lanes per row/entry: 16-byte loads (2 columns per lane) when r is even and fits 8 lanes x 8 steps x 2
inline bool use_v2(int r) { return (r % 2) == 0 && r <= 128; }
inline int lg_for(int r) { return use_v2(r) ? 8 : (r <= 64 ? 8 : (r <= 256 ? 32 : 64)); }
const Guard NOGUARD{nullptr, nullptr};

double *part_slot(lorads_hip_ctx *c, int k) { return c->part + (size_t)k * c->maxpart; }
// sharded cones that share constraints: constrValSum, q1, q2 are summed over the ranks as m-vectors.  Separable shards
// (lorads_hip_set_separable) hold their own constraints: only scalars are summed (c->ar && c->sep).
inline bool shard_vec(const lorads_hip_ctx *c) { return c->ar && !c->sep; }
// the rank the kernels run at (see Block::rl): odd ranks take a zero column along; the LP block's "rank" 1 is not a factor width
inline int dev_rank(const lorads_hip_ctx *c, int r, bool is_lp) { return (c->opt_pad_rank && !is_lp && (r & 1) && r < 512) ? r + 1 : r; }

inline void persist_touch(lorads_hip_ctx *c); // (persist.inc: what the one-launch iteration keeps between launches belongs to the V in memory)
#include "build.inc"
#include "sweep.inc"
#include "persist.inc"
#include "lbfgs_team.inc"

} // namespace

// ================================================================== C ABI
extern "C" {

const char *lorads_hip_last_error(void) { return g_err.c_str(); }

int lorads_hip_create(const lorads_hip_problem *prob, lorads_hip_ctx **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail_msg("no HIP device: the MI355X backend has no CPU fallback");
    if (prob->device >= 0) HC(hipSetDevice(prob->device));
    lorads_hip_ctx *c = new lorads_hip_ctx();
    c->m = prob->m; c->nb = prob->nblocks; c->L = std::max(prob->lbfgs_len, 1); c->b_nrm1 = prob->b_nrm1;
    HC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    { // partial-sum slots: room for every row kernel's grid at any rank (the merged cone of all blocks included)
        size_t rows_max = 0, rows_all = 0;
        for (int k = 0; k < c->nb; ++k) {
            const size_t nk = (size_t)std::max(prob->blocks[k].n, 0);
            rows_max = std::max(rows_max, nk);
            rows_all += (nk + 31) & ~(size_t)31;
        }
        const size_t need = (std::max(rows_max, rows_all) + 3) / 4 + 1;
        if (need > ((size_t)1 << 27)) { delete c; return fail_msg("cone dimension beyond 2^29 rows"); }
        c->maxpart = (int)std::max<size_t>((size_t)MINPART, (need + 255) & ~(size_t)255);
    }
    c->opt_pad_rank = !(getenv("LORADS_PAD_ODD_RANK") && getenv("LORADS_PAD_ODD_RANK")[0] == '0');
    c->opt_dev_presolve = !(getenv("LORADS_DEV_PRESOLVE") && getenv("LORADS_DEV_PRESOLVE")[0] == '0');
    c->opt_presolve_check = getenv("LORADS_PRESOLVE_CHECK") && getenv("LORADS_PRESOLVE_CHECK")[0] == '1';
    if (getenv("LORADS_DEV_PRESOLVE_MIN")) c->dev_presolve_min = (size_t)std::max(0ll, atoll(getenv("LORADS_DEV_PRESOLVE_MIN")));
    c->blk.resize(c->nb);
    for (int k = 0; k < c->nb; ++k)
        if (build_block(c, c->blk[k], prob->blocks[k])) { lorads_hip_destroy(c); return 1; }
    if (build_merged(c, prob)) { lorads_hip_destroy(c); return 1; }
    common_rank(c);
    refresh_merged(c);
    if (getenv("LORADS_HIP_VERBOSE"))
        fprintf(stderr, "lorads_hip: %d cone(s), merged view %s\n", c->nb, c->has_merged ? "on" : (c->merged_ok ? "off (ranks differ)" : "not applicable"));
    if (alloc_factors(c)) { lorads_hip_destroy(c); return 1; }
    std::vector<double> hb(prob->b, prob->b + c->m);
    if (upload(&c->b, hb) || dalloc(&c->lambda, (size_t)c->m) || dalloc(&c->lambda_alt, (size_t)c->m) || dalloc(&c->csum, (size_t)c->m + 2) ||
        dalloc(&c->cstage, (size_t)c->m + 2 + MINPART) || dalloc(&c->gram, 128) ||
        dalloc(&c->q12, (size_t)2 * c->m + 2) || dalloc(&c->part, (size_t)NSLOT * c->maxpart) ||
        dalloc(&c->ctrl, 64 * sizeof(double) + sizeof(CGState) * (size_t)std::max(2 * c->nb, 1)) ||
        dalloc(&c->ring_ab, (size_t)2 * c->L) || dalloc(&c->par, 8) || dalloc(&c->seq_dev, 2)) {
        lorads_hip_destroy(c);
        return 1;
    }
    HC(hipMemset(c->par, 0, sizeof(double) * 8));
    HC(hipMemset(c->seq_dev, 0, sizeof(unsigned long long) * 2));
    c->opt_graph = !(getenv("LORADS_GRAPH") && getenv("LORADS_GRAPH")[0] == '0');
    c->opt_graph_batched = getenv("LORADS_GRAPH") && getenv("LORADS_GRAPH")[0] == '2';
    c->opt_graph_forced = getenv("LORADS_GRAPH") && (getenv("LORADS_GRAPH")[0] == '1' || getenv("LORADS_GRAPH")[0] == '2');
    HC(hipHostMalloc((void **)&c->h_ctrl, 64 * sizeof(double) + sizeof(CGState) * (size_t)std::max(2 * c->nb, 1), hipHostMallocMapped));
    HC(hipHostMalloc((void **)&c->h_flag, 64, hipHostMallocMapped));
    *c->h_flag = 0;
    HC(hipHostGetDevicePointer((void **)&c->h_ctrl_dev, c->h_ctrl, 0));
    HC(hipHostGetDevicePointer((void **)&c->h_flag_dev, c->h_flag, 0));
    c->use_publish = !getenv("LORADS_NO_PUBLISH");
    c->opt_lazy_scalars = !(getenv("LORADS_LAZY_SCALARS") && getenv("LORADS_LAZY_SCALARS")[0] == '0');
    c->opt_ar_fast = !(getenv("LORADS_AR_PLAIN") && getenv("LORADS_AR_PLAIN")[0] == '1');
    c->opt_gram = !(getenv("LORADS_LBFGS_GRAM") && getenv("LORADS_LBFGS_GRAM")[0] == '0');
    c->opt_gram_single = getenv("LORADS_LBFGS_GRAM") && getenv("LORADS_LBFGS_GRAM")[0] == '2';
    c->opt_fuse_dir = !(getenv("LORADS_FUSE_DIR") && getenv("LORADS_FUSE_DIR")[0] == '0');
    c->opt_entry_bip = !(getenv("LORADS_ENTRY_BIP") && getenv("LORADS_ENTRY_BIP")[0] == '0');
    c->opt_seg_carry = !(getenv("LORADS_SEG_CARRY") && getenv("LORADS_SEG_CARRY")[0] == '0');
    c->opt_seg_carry_init = !(getenv("LORADS_SEG_CARRY_INIT") && getenv("LORADS_SEG_CARRY_INIT")[0] == '0');
    c->opt_seg_virt = !(getenv("LORADS_SEG_VIRT") && getenv("LORADS_SEG_VIRT")[0] == '0');
    c->opt_seg_carry_dual = !(getenv("LORADS_SEG_CARRY_DUAL") && getenv("LORADS_SEG_CARRY_DUAL")[0] == '0');
    c->opt_seg_carry_restart = !(getenv("LORADS_SEG_CARRY_RESTART") && getenv("LORADS_SEG_CARRY_RESTART")[0] == '0');
    c->opt_cw_quad = !(getenv("LORADS_CW_QUAD") && getenv("LORADS_CW_QUAD")[0] == '0');
    c->opt_front_cw = !(getenv("LORADS_FRONT_CW") && getenv("LORADS_FRONT_CW")[0] == '0');
    c->opt_front_lds = !(getenv("LORADS_FRONT_LDS") && getenv("LORADS_FRONT_LDS")[0] == '0');
    if (getenv("LORADS_SPEC_WINDOW")) c->spec_window = std::max(1, std::min(8, atoi(getenv("LORADS_SPEC_WINDOW"))));
    c->opt_fold_avg = !(getenv("LORADS_FOLD_AVG") && getenv("LORADS_FOLD_AVG")[0] == '0');
    c->opt_tile_update = !(getenv("LORADS_TILE_UPDATE") && getenv("LORADS_TILE_UPDATE")[0] == '0');
    c->opt_eval_diag = !(getenv("LORADS_EVAL_DIAG") && getenv("LORADS_EVAL_DIAG")[0] == '0');
    c->opt_front_diag = !(getenv("LORADS_FRONT_DIAG") && getenv("LORADS_FRONT_DIAG")[0] == '0');
    c->opt_dense_cache = !(getenv("LORADS_DENSE_CACHE") && getenv("LORADS_DENSE_CACHE")[0] == '0');
    c->opt_fuse_eval = !(getenv("LORADS_FUSE_EVAL") && getenv("LORADS_FUSE_EVAL")[0] == '0');
    c->opt_dense_rem = !(getenv("LORADS_DENSE_REM") && getenv("LORADS_DENSE_REM")[0] == '0');
    c->opt_exact_refresh = getenv("LORADS_EXACT_REFRESH") && getenv("LORADS_EXACT_REFRESH")[0] == '1';
    c->opt_split_front = getenv("LORADS_SPLIT_FRONT") && getenv("LORADS_SPLIT_FRONT")[0] == '1';
    c->scal = (double *)c->ctrl;
    c->st = (CGState *)(c->ctrl + 64 * sizeof(double));
    c->h_scal = (double *)c->h_ctrl;
    c->h_st = (CGState *)(c->h_ctrl + 64 * sizeof(double));
    HC(hipMemset(c->lambda, 0, sizeof(double) * (size_t)std::max(c->m, 1)));
    HC(hipMemset(c->csum, 0, sizeof(double) * (size_t)(c->m + 2)));
    HC(hipMemset(c->q12, 0, sizeof(double) * (size_t)(2 * c->m + 2)));
    HC(hipMemset(c->scal, 0, sizeof(double) * 64));
    HC(hipMemset(c->ring_ab, 0, sizeof(double) * (size_t)2 * c->L));
    HC(hipMemset(c->st, 0, sizeof(CGState) * (size_t)std::max(2 * c->nb, 1)));
    if (dalloc(&c->st_shadow, (size_t)std::max(2 * c->nb, 1))) return 1;
    HC(hipMemset(c->st_shadow, 0, sizeof(CGState) * (size_t)std::max(2 * c->nb, 1)));
    c->persist = new PersistPlan();
    c->opt_persist = !(getenv("LORADS_PERSIST") && getenv("LORADS_PERSIST")[0] == '0');
    c->opt_persist_l2 = !(getenv("LORADS_PERSIST_L2") && getenv("LORADS_PERSIST_L2")[0] == '0');
    c->opt_persist_carry = !(getenv("LORADS_PERSIST_CARRY") && getenv("LORADS_PERSIST_CARRY")[0] == '0');
    c->lteam = new LTeamPlan();
    c->opt_lbfgs_team = !(getenv("LORADS_LBFGS_TEAM") && getenv("LORADS_LBFGS_TEAM")[0] == '0');
    c->opt_alm_fused_tail = !(getenv("LORADS_ALM_FUSED_TAIL") && getenv("LORADS_ALM_FUSED_TAIL")[0] == '0');
    c->opt_alm_fold_cv = !(getenv("LORADS_ALM_FOLD_CV") && getenv("LORADS_ALM_FOLD_CV")[0] == '0');
    c->opt_alm_sval_direct = !(getenv("LORADS_ALM_SVAL_DIRECT") && getenv("LORADS_ALM_SVAL_DIRECT")[0] == '0');
    if (getenv("LORADS_SHARED_GPU") && getenv("LORADS_SHARED_GPU")[0] == '1') {
        int dev = 0;
        char bus[64] = "0";
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
        for (char *q = bus; *q; ++q) if (*q == ':' || *q == '.' || *q == '/') *q = '_';
        const std::string path = std::string("/tmp/lorads_gpu_") + bus + ".lock";
        c->shared_gpu_fd = open(path.c_str(), O_CREAT | O_RDWR, 0666);
    }
    HC(hipDeviceSynchronize());
    *out = c;
    return 0;
}

void lorads_hip_destroy(lorads_hip_ctx *c) {
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);
    drain_events(c);
    for (auto &e : c->ev_pool) hipEventDestroy(e);
    std::vector<Block *> all;
    for (auto &B : c->blk) all.push_back(&B);
    if (c->merged_ok) all.push_back(&c->merged);
    for (Block *bp : all) {
        Block &B = *bp;
        B.pa.release(); B.pu.release();
        if (!B.cv_borrowed) hipFree(B.cv);
        hipFree(B.sv_ptr); hipFree(B.sv_e); hipFree(B.sv_a);
        hipFree(B.row_idx); hipFree(B.a_ptr); hipFree(B.a_e); hipFree(B.a_val); hipFree(B.Cfull); hipFree(B.T); hipFree(B.T2); hipFree(B.wtmp);
        hipFree(B.c_row); hipFree(B.c_col); hipFree(B.c_val); hipFree(B.gdiag); hipFree(B.diag_row); hipFree(B.diag_a); hipFree(B.rc_con); hipFree(B.gentry); hipFree(B.bip_rows[0]); hipFree(B.bip_rows[1]); hipFree(B.bip_we); hipFree(B.ca_row); hipFree(B.ca_col); hipFree(B.ca_val); hipFree(B.cadj_ptr); hipFree(B.cadj_col); hipFree(B.cadj_con); hipFree(B.cadj_a); hipFree(B.cell_col); hipFree(B.cell_con); hipFree(B.cell_a);
        hipFree(B.d_con); hipFree(B.Adense); hipFree(B.Sfull); hipFree(B.d_mu); hipFree(B.fc_ptr); hipFree(B.fc_col); hipFree(B.fc_val); hipFree(B.cell_dst); hipFree(B.cadj_dst); hipFree(B.w_contrib);
        hipFree(B.w_uv); hipFree(B.w_op); hipFree(B.lp_lvl_ptr); hipFree(B.lp_lvl_cols); hipFree(B.lp_ptr); hipFree(B.lp_grow);
        hipFree(B.lp_a); hipFree(B.lp_nrm2sq); hipFree(B.lp_cobj); hipFree(B.lp_cv); hipFree(B.g_ptr); hipFree(B.g_col);
        hipFree(B.g_val);
    }
    free_factors(c);
    hipFree(c->cstage); hipFree(c->sepbuf); hipFree(c->gram);
    hipFree(c->b); hipFree(c->lambda); hipFree(c->lambda_alt); hipFree(c->csum); hipFree(c->q12); hipFree(c->part); hipFree(c->ctrl); hipFree(c->st_shadow); hipFree(c->seg_tile_cone); hipFree(c->seg_rr_alt); hipFree(c->seg_tile_info);
    hipFree(c->ring_ab); hipFree(c->par); hipFree(c->seq_dev);
    graph_cache_free(c);
    if (c->persist) { c->persist->release(); delete c->persist; c->persist = nullptr; }
    if (c->lteam) { c->lteam->release(); delete c->lteam; c->lteam = nullptr; }
    if (c->shared_gpu_fd >= 0) close(c->shared_gpu_fd);
    hipFree(c->seg_row0); hipFree(c->seg_vt0); hipFree(c->seg_vt_seg); hipFree(c->seg_vt_e0); hipFree(c->phase_done);
    if (c->h_ctrl) hipHostFree(c->h_ctrl);
    if (c->h_flag) hipHostFree(c->h_flag);
    for (auto &w : c->lz_workers) { if (w.stream) hipStreamDestroy(w.stream); if (w.pinned) hipHostFree(w.pinned); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int lorads_hip_sync(lorads_hip_ctx *c) {
    flush_pending(c);
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_set_allreduce(lorads_hip_ctx *c, lorads_hip_allreduce_fn fn, void *user) {
    c->ar = fn;
    c->ar_user = user;
    c->ar_fast_all = -1;
    return 0;
}

int lorads_hip_set_scalar_exchange(lorads_hip_ctx *c, lorads_hip_scalar_exchange_fn fn, void *user) {
    if (!c) return fail_msg("set_scalar_exchange: no context");
    if (c->sx_pending || c->sep_pending) { // (an evaluation's sums are still on their way: finish that hand-over under the old rule first)
        if (read_states(c)) return 1;
    }
    c->sx = fn; c->sx_user = user;
    return 0;
}
int lorads_hip_set_separable(lorads_hip_ctx *c, int32_t on) {
    flush_pending(c);
    c->sep = on != 0;
    if (c->sep && !c->sepbuf && dalloc(&c->sepbuf, 32)) return 1;
    return 0;
}

int lorads_hip_set_allreduce_stream_ordered(lorads_hip_ctx *c, int32_t on) {
    c->ar_stream_ordered = on != 0;
    return 0;
}

void *lorads_hip_stream(lorads_hip_ctx *c) { return (void *)c->stream; }

/* all-reduce constrValSum through the registered hook (self-check of a hook / of the stream-ordered mode) */
int lorads_hip_selfcheck_allreduce(lorads_hip_ctx *c) {
    LAUNCH(k_scale, grid1d((size_t)c->m), (size_t)c->m, 1.0, c->csum); // some work on the stream before the collective
    if (c->ar && c->sep) {
        // separable shards hold different numbers of constraints: the collective is one double (constrValSum[0]), spread over the
        // vector afterwards -- the caller's check (every entry = the sum over the ranks) reads the same
        if (c->m == 0) LAUNCH(k_zero, 1, (size_t)1, c->sepbuf + 16, NOGUARD);
        else HC(hipMemcpyAsync(c->sepbuf + 16, c->csum, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (allreduce_dev(c, c->sepbuf + 16, 1)) return 1;
        if (c->m) LAUNCH(k_fill_from, grid1d((size_t)c->m), (size_t)c->m, (const double *)(c->sepbuf + 16), c->csum);
        HC(hipStreamSynchronize(c->stream));
        return 0;
    }
    if (allreduce_dev(c, c->csum, c->m)) return 1;
    LAUNCH(k_scale, grid1d((size_t)c->m), (size_t)c->m, 1.0, c->csum); // and after it
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_init_constr(lorads_hip_ctx *c, int32_t pair) {
    c->ls_np = 0;
    const double *X = pair == LORADS_HIP_PAIR_RR ? c->R : c->U, *Y = pair == LORADS_HIP_PAIR_RR ? c->R : c->V;
    LAUNCH(k_zero, grid1d((size_t)c->m + 2), (size_t)c->m + 2, c->csum, NOGUARD);
    for (auto &B : c->blk) {
        constr_val(c, B, X + B.off, Y + B.off, 1.0, B.cv, CV_ADD, c->csum, NOGUARD);
        lp_col_values(c, B, X, Y, NOGUARD);
    }
    return shard_vec(c) ? allreduce_dev(c, c->csum, c->m) : 0; // (separable shards: these are the rank's own constraints)
}

static int enqueue_alm_grad(lorads_hip_ctx *c, double rho, bool reduce = true) {
    LAUNCH(k_zero, 1, (size_t)1, c->scal + 8, NOGUARD);
    std::vector<Block *> cones;
    Block *S1 = solo(c);
    if (S1 && !S1->dense_c) cones.push_back(S1);
    else for (auto &B0 : c->blk) cones.push_back(&B0);
    for (Block *bp : cones) {
        Block &B = *bp;
        WArgs wa{};
        wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.row_idx = B.row_idx_identity ? nullptr : B.row_idx; wa.rho = rho;
        sval(c, B.pu, true, W_ALM, wa, NOGUARD);
        if (B.dense_c || B.dense_a) dense_part(c, B, c->R + B.off, W_ALM, wa, 1.0, true, NOGUARD);
        int g = spmm(c, B, B.pu, c->R + B.off, OP_GRAD, nullptr, nullptr, rho, c->G + B.off, part_slot(c, 0), NOGUARD,
                     (B.dense_c || B.dense_a) ? B.Wd : nullptr);
        LAUNCH(k_finalize, 1, part_slot(c, 0), g, 1.0, 1, c->scal + 8, NOGUARD);
    }
    return reduce ? allreduce_dev(c, c->scal + 8, 1) : 0;
}
int lorads_hip_alm_cal_grad(lorads_hip_ctx *c, double rho, double *lag) {
    if (enqueue_alm_grad(c, rho)) return 1;
    return read_scalars(c, 8, 1, lag);
}

int lorads_hip_lbfgs_direction(lorads_hip_ctx *c, int32_t inner) {
    c->merged.t_uv_valid = false;
    for (auto &B : c->blk) B.t_uv_valid = false; // U is overwritten by the direction D
    const size_t n = c->all_elem;
    const int gv = c->ar ? grid1d(n) : grid_lbfgs(n);
    double *D = c->U;
    // (LORADS_LBFGS_GRAM=2: the Gram form on one GPU too -- two passes over the vectors instead of five.  Its dots are combinations of
    // the 15 products and round differently from the recursion's (1e-16 of scale): on hard instances whole solves then take other
    // iteration counts than the reference's (theta30 at phase1Tol 1e-3: 3608 inner iterations against 2991), so the default is the recursion)
    if (!c->ar && !(c->opt_gram_single && c->L <= 5)) { // single rank: one kernel per stage of the recursion (5 + 1 launches for history 2)
        double *pp[2] = {part_slot(c, 3), part_slot(c, 5)};
        int cur = 0;
        if (inner == 0) {
            LAUNCH(k_lbfgs_stage, gv, n, (int)ST_FIRST, (const double *)nullptr, 0, (double *)nullptr, (const double *)nullptr,
                   (double *)nullptr, c->G, c->G, pp[cur], D);
        } else {
            double *q = c->Dtmp;
            const int nn = inner <= c->L - 1 ? inner : c->L;
            int node = (c->head - 1 + c->L) % c->L;
            LAUNCH(k_lbfgs_stage, gv, n, (int)ST_FIRST, (const double *)nullptr, 0, (double *)nullptr, (const double *)nullptr, q,
                   c->G, c->ring[node].s, pp[cur], (double *)nullptr);
            for (int t = 0; t < nn; ++t) { // first loop, newest -> oldest: alpha_i = beta_i s_i.q, q -= alpha_i y_i
                const int nxt = (node - 1 + c->L) % c->L;
                const double *dv = t + 1 < nn ? c->ring[nxt].s : c->ring[node].y; // next dot: s of the older node, or y of the oldest
                LAUNCH(k_lbfgs_stage, gv, n, (int)ST_ALPHA, pp[cur], gv, c->ring_ab + 2 * node, c->ring[node].y, q,
                       (const double *)nullptr, dv, pp[cur ^ 1], (double *)nullptr);
                cur ^= 1;
                if (t + 1 < nn) node = nxt;
            }
            for (int t = 0; t < nn; ++t) { // second loop, oldest -> newest: q += (alpha_i - beta_i y_i.q) s_i
                const int nxt = (node + 1) % c->L;
                const bool last = t + 1 == nn;
                LAUNCH(k_lbfgs_stage, gv, n, (int)ST_W, pp[cur], gv, c->ring_ab + 2 * node, c->ring[node].s, q,
                       (const double *)nullptr, last ? c->G : c->ring[nxt].y, pp[cur ^ 1], last ? D : (double *)nullptr);
                cur ^= 1;
                node = nxt;
            }
        }
        LAUNCH(k_use_grad_p, gv, n, pp[cur], gv, c->G, D);
        return 0;
    }
    if (c->opt_gram && c->L <= 5) { // sharded cones: ONE collective for the whole direction (see k_gram)
        if (inner == 0) { // D = -Grad; LBFGSDirUseGrad could only replace it by itself
            LAUNCH(k_scale_copy, gv, n, -1.0, c->G, D);
            return 0;
        }
        const int nn = inner <= c->L - 1 ? inner : c->L, nv = 2 * nn + 1, npair = nv * (nv + 1) / 2;
        GramVecs V{};
        GramPlan pl{};
        pl.nn = nn;
        V.v[0] = c->G;
        for (int t = 0, node = (c->head - 1 + c->L) % c->L; t < nn; ++t, node = (node - 1 + c->L) % c->L) {
            pl.node[t] = node;
            V.v[1 + t] = c->ring[node].y;
            V.v[1 + nn + t] = c->ring[node].s;
        }
        const int g = std::max(1, std::min(std::min(gv, 1024), 5 * c->maxpart / npair)); // (partials: slots 3..7, consumed at once)
        double *part = part_slot(c, 3);
#define GRAM_NV(NV_) case NV_: LAUNCH((k_gram<NV_>), g, n, V, part); break
        switch (nv) { GRAM_NV(3); GRAM_NV(5); GRAM_NV(7); GRAM_NV(9); default: LAUNCH((k_gram<11>), g, n, V, part); break; }
#undef GRAM_NV
        if (!c->ar) { // one rank: the sums and the recursion on coefficients in one launch
            LAUNCH(k_gram_final, 1, (const double *)part, npair, g, c->gram, pl, c->ring_ab, c->gram + 80);
        } else {
            LAUNCH(k_gram_final, 1, (const double *)part, npair, g, c->gram, pl, (double *)nullptr, (double *)nullptr);
            if (allreduce_dev(c, c->gram, npair)) return 1;
            hipLaunchKernelGGL(k_lbfgs_coef, dim3(1), dim3(1), 0, c->stream, pl, (const double *)c->gram, c->ring_ab, c->gram + 80);
        }
#define LINC_NV(NV_) case NV_: LAUNCH((k_lincomb<NV_>), gv, n, V, (const double *)(c->gram + 80), D); break
        switch (nv) { LINC_NV(3); LINC_NV(5); LINC_NV(7); LINC_NV(9); default: LAUNCH((k_lincomb<11>), gv, n, V, (const double *)(c->gram + 80), D); break; }
#undef LINC_NV
        return 0;
    }
    if (inner == 0) {
        LAUNCH(k_scale_copy, gv, n, -1.0, c->G, D);
    } else {
        double *q = c->Dtmp;
        HC(hipMemcpyAsync(q, c->G, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
        const int nn = inner <= c->L - 1 ? inner : c->L;
        int node = (c->head - 1 + c->L) % c->L;
        double *dot = c->scal + 9, *coef = c->scal + 10;
        for (int t = 0; t < nn; ++t) {
            if (dot_to_slot(c, c->ring[node].s, q, 9)) return 1;
            hipLaunchKernelGGL(k_scalar_op, dim3(1), dim3(1), 0, c->stream, (int)SOP_ALPHA, dot, c->ring_ab + 2 * node,
                               c->ring_ab + 2 * node + 1, coef);
            LAUNCH(k_axpy_dev, gv, n, coef, c->ring[node].y, q);
            node = (node - 1 + c->L) % c->L;
        }
        node = (node + 1) % c->L;
        for (int t = 0; t < nn; ++t) {
            if (dot_to_slot(c, c->ring[node].y, q, 9)) return 1;
            hipLaunchKernelGGL(k_scalar_op, dim3(1), dim3(1), 0, c->stream, (int)SOP_W, dot, c->ring_ab + 2 * node,
                               c->ring_ab + 2 * node + 1, coef);
            LAUNCH(k_axpy_dev, gv, n, coef, c->ring[node].s, q);
            node = (node + 1) % c->L;
        }
        LAUNCH(k_scale_copy, gv, n, -1.0, q, D);
    }
    if (dot_to_slot(c, D, c->G, 11)) return 1;
    LAUNCH(k_use_grad, gv, n, c->scal + 11, c->G, D);
    return 0;
}

static int enqueue_q12p12(lorads_hip_ctx *c, int *defer_p12 = nullptr) {
    const int m = c->m;
    Block *S1 = solo(c);
    if (S1 && S1->nrow == m && m > 0 && !S1->dense_c && !S1->dense_a && S1->nc > 0) {
        // one cone that sees every constraint: (R,D) and (D,D) share each row visit -- 4 launches
        Block &B = *S1;
        const Shape sh = shape_for(B.r);
        const double *R = c->R + B.off, *D = c->U + B.off;
        SHAPE_DISPATCH(sh, LAUNCH((k_pairdots_rd<LG_, V2_, NS_>), nblocks_for((size_t)B.pa.ne, TPB / sh.lg), B.pa.ne, B.pa.erow,
                                  B.pa.ecol, R, D, B.r, B.T2, B.T));
        B.t_uv_valid = false;
        c->ls_np = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 1024);
        LAUNCH(k_cv_rd, c->ls_np, B.nrow, B.a_ptr, B.a_e, B.a_val, B.T2, B.T, B.cv, B.row_idx, c->q12, c->q12 + m, c->b, c->csum,
               c->lambda, part_slot(c, 10), c->maxpart);
        const int go = std::min(nblocks_for((size_t)B.nc, TPB / sh.lg), 2048);
        SHAPE_DISPATCH(sh, LAUNCH((k_obj_rd<LG_, V2_, NS_>), go, B.nc, B.c_row, B.c_col, B.c_val, R, D, B.r, part_slot(c, 4),
                                  part_slot(c, 6)));
        if (defer_p12) *defer_p12 = go; // the caller's line-search kernel sums the partials (same order, same values)
        else LAUNCH(k_finalize2, 2, part_slot(c, 4), part_slot(c, 6), go, 2.0, 1.0, c->q12 + 2 * m);
        return 0;
    }
    c->ls_np = 0;
    LAUNCH(k_zero, grid1d((size_t)2 * m + 2), (size_t)2 * m + 2, c->q12, NOGUARD);
    for (int pass = 0; pass < 2; ++pass) {
        const double *X = pass == 0 ? c->R : c->U; // D lives in U
        const double scale = pass == 0 ? 2.0 : 1.0;
        for (auto &B : c->blk) {
            constr_val(c, B, X + B.off, c->U + B.off, scale, B.cv, CV_ADD, c->q12 + (size_t)pass * m, NOGUARD);
            int g = obj_partials(c, B, X + B.off, c->U + B.off, part_slot(c, 4), NOGUARD);
            if (g) LAUNCH(k_finalize, 1, part_slot(c, 4), g, scale, 1, c->q12 + 2 * m + pass, NOGUARD);
        }
    }
    // (separable shards: q1, q2 are the rank's own; p1, p2 are summed where they are read)
    return shard_vec(c) ? allreduce_dev(c, c->q12, 2 * m + 2) : 0;
}
int lorads_hip_alm_q12p12(lorads_hip_ctx *c, double p12[2]) {
    if (enqueue_q12p12(c)) return 1;
    if (c->ar && c->sep) {
        // the slot-by-slot line search that follows takes p1, p2 from the host: the summed pair goes to a buffer of its own
        // and q12[2m..] keeps the local parts (launch_linesearch copies them next to the five local sums, unused there)
        HC(hipMemcpyAsync(c->sepbuf + 8, c->q12 + 2 * c->m, sizeof(double) * 2, hipMemcpyDeviceToDevice, c->stream));
        if (allreduce_dev(c, c->sepbuf + 8, 2)) return 1;
        return read_scalars_at(c, c->sepbuf + 8, 2, p12);
    }
    return read_scalars_at(c, c->q12 + 2 * c->m, 2, p12);
}

// quartic coefficients of the line search from the five m-vector sums (lorads_alm.c:164-172)
static void quartic_coeffs(double rho, double p1, double p2, const double s[5], double k[4]) {
    k[0] = rho * s[0] / 2;              // rho ||q2||^2 / 2
    k[1] = rho * s[1];                  // rho q1.q2
    k[2] = p2 - rho * s[3] + rho * s[2] / 2;
    k[3] = p1 - rho * s[4];
}
// the five dots (+ p1, p2) into scal[16..22]; np_obj > 0: the objective partials of q12p12 are still to be summed
static void launch_linesearch(lorads_hip_ctx *c, double rho, int np_obj) {
    if (c->ls_np == 0) { // q1, q2 came from the general path: stream the m-vectors once, many workgroups
        c->ls_np = std::min(grid1d((size_t)c->m), 1024);
        LAUNCH(k_linesearch_part, c->ls_np, c->m, c->b, c->csum, c->lambda, c->q12, c->q12 + c->m, part_slot(c, 10), c->maxpart);
    }
    LAUNCH(k_linesearch, 1, c->m, 1.0 / rho, part_slot(c, 10), c->ls_np, c->q12, c->scal + 16,
           np_obj ? part_slot(c, 4) : (const double *)nullptr, np_obj ? part_slot(c, 6) : (const double *)nullptr, np_obj, c->maxpart);
}
int lorads_hip_alm_linesearch_coeffs(lorads_hip_ctx *c, double rho, double p1, double p2, double k[4]) {
    launch_linesearch(c, rho, 0);
    if (c->ar && c->sep && allreduce_dev(c, c->scal + 16, 5)) return 1; // the five sums over every rank's constraints
    double s[5];
    if (read_scalars(c, 16, 5, s)) return 1;
    quartic_coeffs(rho, p1, p2, s, k);
    return 0;
}

int lorads_hip_set_y_as_neg_grad(lorads_hip_ctx *c) {
    LAUNCH(k_scale_copy, grid1d(c->all_elem), c->all_elem, -1.0, c->G, c->ring[c->head].y);
    return 0;
}

int lorads_hip_alm_update_var(lorads_hip_ctx *c, double tau) {
    c->ls_np = 0;
    LAUNCH(k_axpy, grid1d(c->all_elem), c->all_elem, tau, c->U, c->R);
    LAUNCH(k_csum_step, nblocks_for((size_t)c->m, TPB), c->m, tau, c->q12, c->q12 + c->m, c->csum);
    return 0;
}

int lorads_hip_set_lbfgs_his_two(lorads_hip_ctx *c, double tau) {
    Ring &h = c->ring[c->head];
    if (!c->ar) {
        const int g = grid_lbfgs(c->all_elem);
        LAUNCH(k_his_two_dot, g, c->all_elem, tau, c->U, c->G, h.s, h.y, part_slot(c, 3));
        LAUNCH(k_finalize_beta, 1, part_slot(c, 3), g, c->ring_ab + 2 * c->head + 1);
        c->head = (c->head + 1) % c->L;
        return 0;
    }
    LAUNCH(k_his_two, grid1d(c->all_elem), c->all_elem, tau, c->U, c->G, h.s, h.y);
    if (dot_to_slot(c, h.y, h.s, 9)) return 1;
    hipLaunchKernelGGL(k_scalar_op, dim3(1), dim3(1), 0, c->stream, (int)SOP_BETA, c->scal + 9, c->ring_ab + 2 * c->head,
                       c->ring_ab + 2 * c->head + 1, c->scal + 10);
    c->head = (c->head + 1) % c->L;
    return 0;
}

// Fused phase-1 inner iteration (a16-a19).  The reference's inner loop body is
//   direction, q12p12, line search | setAsNegGrad, ALMupdateVar(tau), ALMCalGrad, setlbfgsHisTwo, updateDimacsALM
// (lorads_alm.c:1066-1131) with the scalar line search in the middle.  alm_front enqueues the first half and reads
// {p1, p2, a, b, c, d}; alm_step takes tau, enqueues the second half AND -- speculatively -- the first half of the
// next iteration, and reads everything with ONE host synchronisation: out = {lagNormSq, err1, p1, p2, a, b, c, d}.
// Same kernels in the same order as the slot-by-slot calls, so the results are identical; if the host leaves the
// loop, the speculated direction is simply never used (it only touched D, q1/q2 and scratch).
static int enqueue_alm_front(lorads_hip_ctx *c, double rho, int32_t inner, bool have_direction = false) {
    int np = 0;
    if ((!have_direction && lorads_hip_lbfgs_direction(c, inner)) || enqueue_q12p12(c, &np)) return 1;
    launch_linesearch(c, rho, np);
    // separable shards: the five m-vector sums and p1, p2 are this rank's parts -- one all-reduce of seven doubles
    return (c->ar && c->sep) ? allreduce_dev(c, c->scal + 16, 7) : 0;
}
int lorads_hip_alm_front(lorads_hip_ctx *c, double rho, int32_t inner, double out[6]) {
    if (enqueue_alm_front(c, rho, inner)) return 1;
    double s[7];
    if (read_scalars(c, 16, 7, s)) return 1;
    out[0] = s[5]; out[1] = s[6];
    quartic_coeffs(rho, s[5], s[6], s, out + 2);
    return 0;
}
int lorads_hip_alm_step(lorads_hip_ctx *c, double rho, double tau, int32_t next_inner, double out[8]) {
    Block *S1 = solo(c);
    bool team = false;
    if (S1 && !c->ar && S1->nrow == c->m && c->m > 0 && !S1->dense_c && !S1->dense_a) {
        // one cone that sees every constraint (one rank: k_alm_tail forms 1/(y.s) from local sums): 7 launches for the whole second half
        Block &B = *S1;
        Ring &h = c->ring[c->head];
        const int gv = grid_lbfgs(c->all_elem);
        c->ls_np = 0;
        // (the history update and the NEXT direction as one launch of resident workgroups, where that applies: lbfgs_team.inc)
        team = lteam_ready(c, next_inner);
        SvDirect sd{};
        if (c->opt_alm_sval_direct && B.sv_direct && B.pu.cbase && B.nrow == c->m) {
            sd.ptr = B.sv_ptr; sd.e = B.sv_e; sd.a = B.sv_a; sd.cbase = B.pu.cbase; sd.S = B.pu.S; sd.row_idx = B.row_idx; sd.b = c->b; sd.lambda = c->lambda;
            sd.rho = rho; sd.nrow = B.nrow;
        }
        LAUNCH(k_alm_update, gv, c->all_elem, tau, c->G, c->U, h.y, c->R, c->m, c->q12, c->q12 + c->m, c->csum, sd);
        if (!sd.ptr) {
            WArgs wa{};
            wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.row_idx = B.row_idx_identity ? nullptr : B.row_idx; wa.rho = rho;
            sval(c, B.pu, true, W_ALM, wa, NOGUARD);
        }
        const int glag = spmm(c, B, B.pu, c->R, OP_GRAD, nullptr, nullptr, rho, c->G, part_slot(c, 0), NOGUARD);
        if (team) { if (launch_lbfgs_team(c, tau, next_inner)) return 1; }
        else LAUNCH(k_his_two_dot, gv, c->all_elem, tau, c->U, c->G, h.s, h.y, part_slot(c, 3));
        if (team && c->use_publish && c->opt_alm_fused_tail && B.nc > 0 && B.pa.ne > 0 && lteam_scratch(c, (size_t)B.pa.ne) == 0) {
            // the next direction exists already: the step's A(R R^T) and the next line search's q1, q2 share their passes, one
            // workgroup closes the iteration and hands it over (lbfgs_team.inc) -- 3 launches where the forms below take 8
            const Shape sh = shape_for(B.r);
            const double *R = c->R + B.off, *D = c->U + B.off;
            const int g1 = nblocks_for((size_t)B.pa.ne, TPB / sh.lg), go = std::min(nblocks_for((size_t)B.nc, TPB / sh.lg), 2048);
            // cones whose constraints have ONE entry each (Max-Cut type, single-entry): the pattern pass does the constraints' bookkeeping too
            // (up to 2048 workgroups' worth of entries: beyond that the nine block sums per workgroup and their partials cost more than
            // the launch they save -- cfg5, 6250 workgroups: 191.7 -> 195.9 us per inner iteration)
            const bool fold = c->opt_alm_fold_cv && (B.diag_only || B.entry_only) && B.na == B.nrow && B.pa.e_ptr && g1 <= std::min(c->maxpart, 2048);
            CvFold cf{};
            if (fold) {
                cf.e_ptr = B.pa.e_ptr; cf.e_con = B.pa.e_con; cf.e_val = B.pa.e_val; cf.row_idx = B.row_idx; cf.cv = B.cv; cf.csum = c->csum;
                cf.q1 = c->q12; cf.q2 = c->q12 + c->m; cf.b = c->b; cf.lambda = c->lambda; cf.part_v = part_slot(c, 8); cf.part_d = part_slot(c, 9);
                cf.part_ls = part_slot(c, 10); cf.pstride = c->maxpart;
            }
            SHAPE_DISPATCH(sh, LAUNCH((k_pairdots_rrd<LG_, V2_, NS_>), g1 + go, B.pa.ne, B.pa.erow, B.pa.ecol, R, D, B.r, c->lteam->t0, B.T2, B.T, g1,
                                      B.nc, B.c_row, B.c_col, B.c_val, part_slot(c, 4), part_slot(c, 6), cf));
            B.t_uv_valid = false;
            int nls = g1;
            if (!fold) {
                nls = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 1024);
                LAUNCH(k_cv_res_rd, nls, B.nrow, B.a_ptr, B.a_e, B.a_val, (const double *)c->lteam->t0, (const double *)B.T2, (const double *)B.T,
                       B.cv, B.row_idx, c->csum, c->q12, c->q12 + c->m, c->b, c->lambda, part_slot(c, 8), part_slot(c, 9), part_slot(c, 10), c->maxpart);
            }
            c->ls_np = nls;
            c->head = (c->head + 1) % c->L;
            AlmCloseArgs ca{};
            ca.lag_part = part_slot(c, 0); ca.nlag = glag;
            ca.part_v = part_slot(c, 8); ca.part_d = part_slot(c, 9); ca.nres = nls;
            ca.part_ls = part_slot(c, 10); ca.nls = nls; ca.pstride = c->maxpart;
            ca.obj1 = part_slot(c, 4); ca.obj2 = part_slot(c, 6); ca.nobj = go;
            ca.rinv = 1.0 / rho; ca.scal = c->scal; ca.q12_tail = c->q12 + 2 * (size_t)c->m;
            flush_pending(c);
            flush_final(c);
            ++c->pub_seq;
            LAUNCH(k_alm_close, 1, ca, (const unsigned long long *)c->ctrl, 23, (unsigned long long *)c->h_ctrl_dev, c->h_flag_dev, c->seq_dev);
            if (wait_publish(c)) return 1;
            const double *s = c->h_scal;
            if (s[7] != 0.0) {
                c->lteam->failed = true; // (not again in this context)
                return fail_msg("one-launch L-BFGS direction: the team of workgroups did not complete (are other processes holding the GPU's compute units?); "
                                "LORADS_LBFGS_TEAM=0 selects the launch-by-launch form");
            }
            out[0] = s[8];
            out[1] = std::sqrt(s[0]) / (1 + c->b_nrm1);
            out[2] = s[21]; out[3] = s[22];
            quartic_coeffs(rho, s[21], s[22], s + 16, out + 4);
            return 0;
        }
        pairdots(c, B.pa, c->R, c->R, B.r, B.T2, NOGUARD);
        const int nres = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 2048);
        LAUNCH(k_cv_res, nres, B.nrow, B.a_ptr, B.a_e, B.a_val, B.T2, B.cv, B.row_idx, c->csum, c->b, c->lambda, part_slot(c, 8),
               part_slot(c, 9), NOGUARD);
        LAUNCH(k_alm_tail, 1, part_slot(c, 0), glag, c->scal + 8, part_slot(c, 3), team ? 0 : gv, c->ring_ab + 2 * c->head + 1, part_slot(c, 8),
               part_slot(c, 9), nres, c->scal);
        c->head = (c->head + 1) % c->L;
    } else if (c->ar && c->sep && c->opt_gram) {
        // separable shards: ||Grad||^2 and y.s of the step are local sums like the evaluation's -- they ride on ITS all-reduce
        // (six doubles, one collective for the whole second half), and 1/(y.s) is formed from the summed value afterwards
        Ring &h = c->ring[c->head];
        const int hd = c->head;
        if (lorads_hip_set_y_as_neg_grad(c) || lorads_hip_alm_update_var(c, tau) || enqueue_alm_grad(c, rho, false)) return 1;
        LAUNCH(k_his_two, grid1d(c->all_elem), c->all_elem, tau, c->U, c->G, h.s, h.y);
        if (dot_to_slot(c, h.y, h.s, 9, false)) return 1;
        c->head = (c->head + 1) % c->L;
        c->sep_extra.n = 2;
        c->sep_extra.p[0] = c->scal + 8;
        c->sep_extra.p[1] = c->scal + 9;
        if (enqueue_eval(c, LORADS_HIP_PAIR_RR, nullptr, false)) return 1;
        flush_pending(c); // (the reduced y.s must be committed before it is read)
        hipLaunchKernelGGL(k_scalar_op, dim3(1), dim3(1), 0, c->stream, (int)SOP_BETA, c->scal + 9, c->ring_ab + 2 * hd, c->ring_ab + 2 * hd + 1,
                           c->scal + 10);
    } else if (lorads_hip_set_y_as_neg_grad(c) || lorads_hip_alm_update_var(c, tau) || enqueue_alm_grad(c, rho) ||
               lorads_hip_set_lbfgs_his_two(c, tau) || enqueue_eval(c, LORADS_HIP_PAIR_RR, nullptr, false)) {
        return 1;
    }
    if (next_inner >= 0 && enqueue_alm_front(c, rho, next_inner, team)) return 1;
    double s[23];
    if (read_scalars(c, 0, 23, s)) return 1;
    if (team && s[7] != 0.0) {
        c->lteam->failed = true; // (not again in this context)
        return fail_msg("one-launch L-BFGS direction: the team of workgroups did not complete (are other processes holding the GPU's compute units?); "
                        "LORADS_LBFGS_TEAM=0 selects the launch-by-launch form");
    }
    out[0] = s[8];
    out[1] = std::sqrt(s[0]) / (1 + c->b_nrm1);
    out[2] = s[21]; out[3] = s[22];
    if (next_inner >= 0) quartic_coeffs(rho, s[21], s[22], s + 16, out + 4);
    return 0;
}

int lorads_hip_update_dimacs(lorads_hip_ctx *c, int32_t pair, double *err1) {
    if (enqueue_eval(c, pair, nullptr, false)) return 1;
    double s[3];
    if (read_scalars(c, 0, 3, s)) return 1;
    *err1 = std::sqrt(s[0]) / (1 + c->b_nrm1);
    return 0;
}

int lorads_hip_cal_obj(lorads_hip_ctx *c, int32_t pair, double *pobj) {
    if (pair == LORADS_HIP_PAIR_UV) LAUNCH(k_average, grid1d(c->all_elem), c->all_elem, c->U, c->V, c->R, NOGUARD);
    LAUNCH(k_zero, 1, (size_t)1, c->scal + 3, NOGUARD);
    std::vector<Block *> cones;
    Block *S1 = solo(c);
    if (S1 && !S1->dense_c) cones.push_back(S1);
    else for (auto &B0 : c->blk) cones.push_back(&B0);
    for (Block *bp : cones) {
        Block &B = *bp;
        int g = obj_partials(c, B, c->R + B.off, c->R + B.off, part_slot(c, 4), NOGUARD);
        if (g) LAUNCH(k_finalize, 1, part_slot(c, 4), g, 1.0, 1, c->scal + 3, NOGUARD);
    }
    if (allreduce_dev(c, c->scal + 3, 1)) return 1;
    return read_scalars(c, 3, 1, pobj);
}

int lorads_hip_admm_update_var(lorads_hip_ctx *c, double rho, double tol, int32_t maxit, int32_t *iters) {
    if (run_sweep(c, rho, tol, maxit, false)) return 1;
    int its = 0;
    finish_sweep(c, &its);
    *iters = its;
    return 0;
}

// fused ADMM step: admmUpdateVar + calObj_admm + LORADSCalDualObj + updateDimacsADMM with ONE host
// synchronisation (lorads_admm.c:76-81); out = {cg iterations, <C,RR^T>, b.lambda, err1}
int lorads_hip_admm_step(lorads_hip_ctx *c, double rho, double tol, int32_t maxit, double out[4]) {
    if (run_sweep(c, rho, tol, maxit, true)) return 1;
    int its = 0;
    finish_sweep(c, &its);
    out[0] = its;
    out[1] = c->h_scal[2];
    out[2] = c->h_scal[1];
    out[3] = std::sqrt(c->h_scal[0]) / (1 + c->b_nrm1);
    return 0;
}

int lorads_hip_update_dual_var(lorads_hip_ctx *c, double rho) {
    c->ls_np = 0;
    // (see pend_dual; with sharded cones only the one-kernel front can take it: it needs no owner pairs, k_wsum stores all of lambda)
    // (lockstep sweep of a Max-Cut-type merged cone that sees every constraint: its U front takes it, see enqueue_batched)
    const bool merged_front = c->has_merged && c->opt_seg_carry_dual && c->merged.diag_only && c->merged.rc_w > 0 && c->merged.w_uv &&
                              c->merged.nrow == c->m && c->opt_front_diag && !c->opt_split_front && !shard_vec(c) && !getenv("LORADS_NO_BATCH");
    if (c->opt_lazy_scalars && c->lambda_alt && ((c->nb == 1 && (!shard_vec(c) || front_cw_ok(c, c->blk[0]))) || merged_front ||
                                                 (c->persist && !c->persist->failed && persist_eligible(c)))) {
        flush_pending(c);
        c->pend_dual = true;
        c->pend_dual_rho = rho;
        return 0;
    }
    LAUNCH(k_dual_update, nblocks_for((size_t)c->m, TPB), c->m, rho, c->b, c->csum, c->lambda);
    return 0;
}

int lorads_hip_cal_dual_obj(lorads_hip_ctx *c, double *dobj) {
    const int g = std::min(grid1d((size_t)c->m), 256);
    LAUNCH(k_dot, g, (size_t)c->m, c->b, c->lambda, part_slot(c, 7), NOGUARD);
    LAUNCH(k_finalize, 1, part_slot(c, 7), g, 1.0, 0, c->scal + 4, NOGUARD);
    if (c->ar && c->sep && allreduce_dev(c, c->scal + 4, 1)) return 1; // (b and lambda are the rank's own pieces)
    return read_scalars(c, 4, 1, dobj);
}

static void invalidate_t(lorads_hip_ctx *c) {
    persist_touch(c);
    c->merged.t_uv_valid = false;
    for (auto &B : c->blk) { B.t_uv_valid = false; B.wj_for = nullptr; }
}

int lorads_hip_alm_to_admm(lorads_hip_ctx *c) {
    flush_pending(c);
    invalidate_t(c);
    HC(hipMemcpyAsync(c->V, c->R, sizeof(double) * c->all_elem, hipMemcpyDeviceToDevice, c->stream));
    HC(hipMemcpyAsync(c->U, c->V, sizeof(double) * c->all_elem, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int lorads_hip_average_uv_to_v(lorads_hip_ctx *c) {
    invalidate_t(c);
    LAUNCH(k_average, grid1d(c->all_elem), c->all_elem, c->U, c->V, c->R, NOGUARD);
    HC(hipMemcpyAsync(c->V, c->R, sizeof(double) * c->all_elem, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int lorads_hip_scale_obj(lorads_hip_ctx *c, double s) {
    c->ls_np = 0;
    persist_touch(c);
    for (auto &B0 : c->blk) B0.wj_for = nullptr; // (kept products C Y of a dense objective are those of the old C)
    std::vector<Block *> all;
    for (auto &B0 : c->blk) all.push_back(&B0);
    if (c->merged_ok) all.push_back(&c->merged); // the merged cone carries its own copy of the objective
    for (Block *bp : all) {
        Block &B = *bp;
        if (B.nc) LAUNCH(k_scale, grid1d((size_t)B.nc), (size_t)B.nc, s, B.c_val);
        if (B.pu.ne) LAUNCH(k_scale, grid1d((size_t)B.pu.ne), (size_t)B.pu.ne, s, B.pu.cbase);
        if (B.pu.nslot) LAUNCH(k_scale, grid1d((size_t)B.pu.nslot), (size_t)B.pu.nslot, s, B.pu.adj_sval); // (the list's own copy of C)
        if (B.fc_nslot) LAUNCH(k_scale, grid1d((size_t)B.fc_nslot), (size_t)B.fc_nslot, s, B.fc_val); // (k_front_cw's copy)
        if (B.dense_c) LAUNCH(k_scale, grid1d((size_t)B.npad * B.npad), (size_t)B.npad * B.npad, s, B.Cfull);
        if (B.is_lp && B.n) LAUNCH(k_scale, grid1d((size_t)B.n), (size_t)B.n, s, B.lp_cobj);
    }
    if (c->m) LAUNCH(k_scale, grid1d((size_t)c->m), (size_t)c->m, s, c->lambda);
    return 0;
}

int lorads_hip_set_mat(lorads_hip_ctx *c, int32_t which, int32_t k, const double *cm) {
    flush_pending(c);
    double *base = mat_base(c, which);
    if (!base || k < 0 || k >= c->nb) return fail_msg("set_mat: bad argument");
    persist_touch(c);
    Block &B = c->blk[k];
    B.t_uv_valid = false;
    B.wj_for = nullptr;
    c->merged.t_uv_valid = false; // (its pair values of (U, V) cover this cone too)
    std::vector<double> rm((size_t)B.n * B.r); // (zero-initialised: the padding column of an odd rank)
    for (int j = 0; j < B.rl; ++j)
        for (int i = 0; i < B.n; ++i) rm[(size_t)i * B.r + j] = cm[(size_t)j * B.n + i];
    HC(hipMemcpyAsync(base + B.off, rm.data(), sizeof(double) * rm.size(), hipMemcpyHostToDevice, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_get_mat(lorads_hip_ctx *c, int32_t which, int32_t k, double *cm) {
    flush_pending(c);
    double *base = mat_base(c, which);
    if (!base || k < 0 || k >= c->nb) return fail_msg("get_mat: bad argument");
    Block &B = c->blk[k];
    std::vector<double> rm((size_t)B.n * B.r);
    HC(hipMemcpyAsync(rm.data(), base + B.off, sizeof(double) * rm.size(), hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    for (int j = 0; j < B.rl; ++j)
        for (int i = 0; i < B.n; ++i) cm[(size_t)j * B.n + i] = rm[(size_t)i * B.r + j];
    return 0;
}

int lorads_hip_set_vec(lorads_hip_ctx *c, int32_t which, const double *v) {
    flush_pending(c);
    c->ls_np = 0;
    double *d = vec_base(c, which);
    if (!d) return fail_msg("set_vec: bad argument");
    HC(hipMemcpyAsync(d, v, sizeof(double) * (size_t)c->m, hipMemcpyHostToDevice, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_get_vec(lorads_hip_ctx *c, int32_t which, double *v) {
    flush_pending(c);
    double *d = vec_base(c, which);
    if (!d) return fail_msg("get_vec: bad argument");
    HC(hipMemcpyAsync(v, d, sizeof(double) * (size_t)c->m, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_resize_rank(lorads_hip_ctx *c, const int32_t *nr) {
    flush_pending(c);
    persist_touch(c);
    // AUG_RANK (data/lorads_solver.c:806-906): keep the old columns, new columns = 1/sqrt(k) on their
    // leading diagonal (lpRandomDiag :776-786), clear L-BFGS history and CG workspaces.  Everything stays in HBM: the new
    // arrays are allocated, one kernel per factor and cone copies row i's old entries and writes the new columns, the old
    // arrays are released -- no copy through the host, no transposition (the device layout is row-major on both sides).
    for (int k = 0; k < c->nb; ++k) { // (refuse before anything is touched: a refused call leaves host and device at the old ranks)
        const Block &B = c->blk[k];
        if (nr[k] < B.rl || nr[k] > 512) return fail_msg("resize_rank: bad rank");
        const int nd = dev_rank(c, nr[k], B.is_lp);
        if (B.dense_c && nd > 128 && B.ksplit_b == 0)
            return fail_msg("resize_rank: this cone's dense objective kernel supports rank <= 128");
        if (B.dense_a && nd > 128 && B.ksplit_b == 0)
            return fail_msg("resize_rank: this cone's dense constraint kernel supports rank <= 128");
        if (nblocks_for((size_t)B.n, TPB / lg_for(nd)) > c->maxpart)
            return fail_msg("resize_rank: cone dimension too large for the partial-sum slots at this rank");
    }
    double *old[4] = {c->R, c->U, c->V, c->G};
    std::vector<size_t> off_old(c->nb);
    std::vector<int> r_old(c->nb), rl_old(c->nb);
    for (int k = 0; k < c->nb; ++k) { off_old[k] = c->blk[k].off; r_old[k] = c->blk[k].r; rl_old[k] = c->blk[k].rl; }
    c->R = c->U = c->V = c->G = nullptr; // (kept alive across free_factors)
    free_factors(c);
    invalidate_t(c);
    for (int k = 0; k < c->nb; ++k) { c->blk[k].rl = nr[k]; c->blk[k].r = dev_rank(c, nr[k], c->blk[k].is_lp); }
    common_rank(c);
    refresh_merged(c);
    if (alloc_factors(c)) { for (auto p : old) hipFree(p); return 1; }
    double *now[4] = {c->R, c->U, c->V, c->G};
    for (int a = 0; a < 4; ++a)
        for (int k = 0; k < c->nb; ++k) {
            const Block &B = c->blk[k];
            const int aug = B.rl - rl_old[k], rr = std::min(B.n, aug);
            const size_t len = (size_t)B.n * B.r;
            LAUNCH(k_grow_rank, grid1d(len), len, rl_old[k], r_old[k], B.rl, B.r, (const double *)(old[a] + off_old[k]), now[a] + B.off, rr,
                   rr > 0 ? 1 / std::sqrt((double)rr) : 0.0);
        }
    HC(hipMemsetAsync(c->ring_ab, 0, sizeof(double) * (size_t)2 * c->L, c->stream));
    HC(hipStreamSynchronize(c->stream)); // (the old arrays are read by the kernels above)
    for (auto p : old) hipFree(p);
    return 0;
}

int lorads_hip_profile(lorads_hip_ctx *c, int32_t enable, int32_t every) {
    flush_pending(c);
    HC(hipStreamSynchronize(c->stream));
    drain_events(c);
    c->prof = enable;
    c->prof_every = std::max(1, every);
    if (enable && c->ev_pool.empty()) {
        c->ev_pool.resize(3 * 1024);
        for (auto &e : c->ev_pool) HC(hipEventCreate(&e));
    }
    c->n_matvec = c->n_cg_it = c->n_solves = c->n_samp = c->n_samp_spmm = c->n_resume = 0;
    c->n_front = 0;
    c->ms_samp = c->ms_samp_spmm = 0;
    c->samp_ms.clear();
    return 0;
}

int lorads_hip_profile_target(lorads_hip_ctx *c, int32_t target) {
    if (target != 0 && target != 1) return fail_msg("profile_target: 0 (operator applications) or 1 (solve fronts)");
    c->prof_target = target;
    return 0;
}

int lorads_hip_profile_samples(lorads_hip_ctx *c, double *out, int32_t cap, int32_t *n) {
    flush_pending(c);
    HC(hipStreamSynchronize(c->stream));
    drain_events(c);
    *n = (int32_t)c->samp_ms.size();
    for (int32_t i = 0; i < cap && i < *n; ++i) out[i] = c->samp_ms[(size_t)i];
    return 0;
}

int lorads_hip_profile_read(lorads_hip_ctx *c, double s[8]) {
    flush_pending(c);
    HC(hipStreamSynchronize(c->stream));
    drain_events(c);
    s[0] = (double)c->n_matvec;
    s[1] = (double)c->n_resume;
    s[2] = (double)c->n_cg_it;
    s[3] = (double)c->n_solves;
    s[4] = (double)c->n_samp;
    s[5] = c->ms_samp;
    s[6] = (double)c->n_samp_spmm;
    s[7] = c->ms_samp_spmm;
    return 0;
}

// `reps` applications of the live CG operator of cone 0, back to back between ONE event pair, nothing riding along (milliseconds for
// all of them): bench.py's operator_alone_back_to_back
int lorads_hip_time_operator(lorads_hip_ctx *c, int32_t reps, double *ms) {
    if (c->nb < 1) return fail_msg("time_operator: no cone");
    Block &B = c->blk[0];
    flush_pending(c);
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int prof0 = c->prof;
    c->prof = 0;
    HC(hipMemsetAsync(c->st, 0, sizeof(CGState) * 2, c->stream));
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(TPB), 0, c->stream, (size_t)NSLOT * c->maxpart, 1.0, c->part);
    const bool tv = B.t_uv_valid;
    for (int it = -3; it < reps; ++it) {
        if (it == 0) HC(hipEventRecord(e0, c->stream));
        apply_operator(c, B, c->V + B.off, c->cp + B.off, OP_CG, nullptr, c->cQ + B.off, part_slot(c, 0), NOGUARD);
    }
    HC(hipEventRecord(e1, c->stream));
    HC(hipEventSynchronize(e1));
    float f = 0;
    HC(hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    hipEventDestroy(e0); hipEventDestroy(e1);
    c->prof = prof0;
    c->n_matvec -= reps + 3;
    B.t_uv_valid = (B.use_cw || B.diag_only || B.entry_only) ? tv : false; // (those write w_op only; the others overwrite the pair dots)
    return 0;
}

#ifdef LORADS_DEV_BUILD
#include "dev.inc"
#endif

int lorads_hip_operator_kind(lorads_hip_ctx *c, int32_t k, int32_t *kind) {
    if (k < 0 || k >= c->nb) return fail_msg("bad block");
    const Block &B = c->blk[k];
    *kind = B.diag_only ? 2 : B.entry_only ? 3 : B.use_cw ? 4 : B.has_gram ? 0 : 1;
    if (B.dense_a) *kind += 16; // + dense constraint matrices through the dense GEMM
    if (B.entry_only && B.bip_rows[0] && c->opt_entry_bip) *kind += 32; // single-entry operator in its two-colour form (k_op_entry_bip x 2)
    return 0;
}

int lorads_hip_block_image(lorads_hip_ctx *c, int32_t k, int64_t im[16]) {
    if (k < 0 || k >= c->nb) return fail_msg("bad block");
    const Block &B = c->blk[k];
    const int64_t v[16] = {B.n, B.rl, B.nrow, B.na, B.nc, B.pa.ne, B.pu.ne, B.dense_c, B.dense_a ? B.nd : 0, B.diag_only, B.entry_only,
                           B.use_cw, B.has_gram, B.front_cw, B.cell_w, B.bip_n[0]};
    for (int i = 0; i < 16; ++i) im[i] = v[i];
    return 0;
}

int lorads_hip_scalar_exchange_count(lorads_hip_ctx *c, int64_t *n) {
    *n = c->n_sx;
    return 0;
}

int lorads_hip_persist_stats(lorads_hip_ctx *c, int64_t stats[6]) {
    const bool ok = c->persist && persist_ready(c, 800);
    stats[0] = c->n_persist;
    stats[1] = ok ? 1 : 0;
    stats[2] = ok ? c->persist->grid : 0;
    stats[3] = ok ? c->persist->rows : 0;
    stats[4] = ok ? c->persist->ns : 0;
    stats[5] = ok ? (int64_t)c->persist->lds : 0;
    return 0;
}

int lorads_hip_launch_count(lorads_hip_ctx *c, int64_t *n) {
    *n = c->n_launch;
    return 0;
}

int lorads_hip_lbfgs_team_stats(lorads_hip_ctx *c, int64_t stats[4]) {
    const bool ok = c->lteam && c->lteam->valid;
    stats[0] = c->lteam ? c->lteam->launches : 0;
    stats[1] = ok ? 1 : 0;
    stats[2] = ok ? c->lteam->grid : 0;
    stats[3] = ok ? c->lteam->np : 0;
    return 0;
}

int lorads_hip_persist_stamps(lorads_hip_ctx *c, int32_t enable, uint64_t ticks[16]) {
    flush_pending(c);
    HC(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; ++i) ticks[i] = 0;
    if (c->persist && c->persist->valid && c->persist->stamps && c->persist_stamps) {
        HC(hipMemcpy(c->persist->h_stamps, c->persist->stamps, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) ticks[i] = c->persist->h_stamps[i];
        HC(hipMemset(c->persist->stamps, 0, sizeof(unsigned long long) * 16));
    }
    c->persist_stamps = enable != 0;
    return 0;
}

int lorads_hip_presolve_stats(lorads_hip_ctx *c, int64_t stats[2]) {
    stats[0] = c->n_dev_patterns;
    stats[1] = c->n_checked_patterns;
    return 0;
}

int lorads_hip_graph_stats(lorads_hip_ctx *c, int64_t stats[4]) {
    stats[0] = c->graphs ? c->graphs->n_capture : 0;
    stats[1] = c->graphs ? c->graphs->n_replay : 0;
    stats[2] = c->graphs ? (int64_t)c->graphs->map.size() : 0;
    stats[3] = graph_ok(c) ? 1 : 0;
    return 0;
}

int lorads_hip_algorithmic_bytes(lorads_hip_ctx *c, int32_t k, double *mv, double *cg) {
    if (k < 0 || k >= c->nb) return fail_msg("bad block");
    const Block &B = c->blk[k]; // (from the CURRENT rank: phase 1 may have grown it since the cone was built)
    const double F = 8.0 * (double)B.n * (double)B.rl; // (the problem's rank, not the padded one)
    *mv = 4 * F + 32.0 * B.na + 16.0 * B.nrow;
    *cg = *mv + 9 * F;
    return 0;
}

} // extern "C"

#include "lanczos.inc"
