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
// Kernels (HBM/L2-bound integer+FP64 gather work; bytes per unit in DESIGN.md):
//   k_pairdots   T_e = X_p.Y_q + X_q.Y_p on a pattern            (reference LORADSUVt)
//   k_cv         w_i = sum_k a_k T_e(k) (+ running-sum update)     (mul_inner_rk_double / coneAUV)
//   k_sval       S_e = [C_e] + sum_(i,a) weight_i a               (sdpDataWSum / addObjCoeff)
//   k_sgram      S = G T                                          (coneAUV + sdpDataWSum fused)
//   k_spmm       Y_p = epilogue(sum_(q,e) S_e X_q)  + fused dots  (mul_rk + axpy + dot/nrm)
//   k_op_diag    fused operator when every A_i = a e_p e_p^T (Max-Cut): one pass
//   k_cg_*       CG vector updates with device-resident scalars   (CGSolve)
//   k_eval_final ||b - sum||^2, b.lambda in one workgroup
//   misc         averaging, dual update, line-search dots, L-BFGS axpy/dot
//
// Launch structure: one ADMM iteration (2 CG solves per cone, constraint refreshes, objective,
// DIMACS) is enqueued on one stream WITHOUT host round trips, using the iteration counts of the
// previous ADMM iteration as a speculation and device-side gates (struct Guard); the host
// synchronises once per ADMM iteration and resumes a solve that needed more iterations.
//
// Reductions: every reducing kernel writes one partial per workgroup; the consumer re-sums the
// partials (<= 4096) in a fixed order, so results do not depend on scheduling.  Wave-level sums use
// 64-lane __shfl_xor butterflies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "lorads_hip.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXPART = 4096; // capacity of one partial-sum slot
constexpr int NSLOT = 18;

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
__device__ __forceinline__ double sum_partials(const double *part, int n, double *sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) v += part[i];
    return block_sum(v, sh);
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
__device__ __forceinline__ bool blocked(const Guard &g) { return (g.skip && *g.skip != 0) || (g.need && *g.need == 0); }
// The gate words were written by the previous kernel, so reading them costs a cache-missing scalar load
// (~1.5 us when it heads the kernel).  Kernels therefore evaluate `live` first but only USE it to predicate
// their stores: the gate loads fly together with the kernel's own first loads.  A blocked kernel computes on
// whatever is there and writes nothing.

enum { W_COMPACT = 0, W_ADMM = 1, W_ALM = 2, W_DUAL = 3 };
enum { OP_CG = 0, OP_RES = 1, OP_RHS = 2, OP_GRAD = 3 };
enum { CHK_ITER = 1, CHK_RESTART = 2 };
enum { DIR_BETA = 1, DIR_RESTART = 2 };
enum { CV_SET = 0, CV_ADD = 1, CV_DELTA = 2 };

// ------------------------------------------------------------------ kernels
// Column slices: lane l of an LG-lane group owns columns (l + c*LG)*W .. +W-1 for c = 0..NS-1 (W = 2: one
// 16-byte load per step).  NS is a compile-time constant and every load is UNCONDITIONAL (the address is
// clamped into the row, the value masked afterwards): a per-element `if (j < r) load` makes hipcc branch
// around each load and wait vmcnt(0) per element, which serialises the gather (cdna_hip_programming.md 5, trap c).
template <int LG, bool V2, int NS>
struct Slice {
    static constexpr int W = V2 ? 2 : 1;
    __device__ static __forceinline__ void load(const double *__restrict__ row, int r, int lane, double (&v)[NS][W]) {
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const int j = (lane + c * LG) * W;
            const bool ok = j < r;
            const int jc = ok ? j : 0;
            if (V2) {
                const double2 t = *(const double2 *)(row + jc);
                v[c][0] = ok ? t.x : 0.0;
                v[c][W - 1] = ok ? t.y : 0.0;
            } else {
                const double t = row[jc];
                v[c][0] = ok ? t : 0.0;
            }
        }
    }
};

// pair dot of one pattern entry, LG lanes: x_p.y_q + x_q.y_p (p != q) or x_p.y_p; all row loads issued first
template <int LG, bool V2, int NS>
__device__ __forceinline__ double pair_dot(const double *__restrict__ X, const double *__restrict__ Y, int p, int q, int r,
                                           int lane) {
    constexpr int W = V2 ? 2 : 1;
    double a[NS][W], b[NS][W], cc[NS][W], d[NS][W];
    Slice<LG, V2, NS>::load(X + (size_t)p * r, r, lane, a);
    Slice<LG, V2, NS>::load(Y + (size_t)q * r, r, lane, b);
    Slice<LG, V2, NS>::load(X + (size_t)q * r, r, lane, cc);
    Slice<LG, V2, NS>::load(Y + (size_t)p * r, r, lane, d);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int w = 0; w < W; ++w) { s1 += a[c][w] * b[c][w]; s2 += cc[c][w] * d[c][w]; }
    return p == q ? s1 : s1 + s2;
}

// T_e = X_p.Y_q + X_q.Y_p  (p != q)   |   X_p.Y_p  (p == q); LG lanes share one pattern entry
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_pairdots(int ne, const int *__restrict__ erow, const int *__restrict__ ecol,
                                                  const double *__restrict__ X, const double *__restrict__ Y, int r,
                                                  double *__restrict__ T, Guard g) {
    const bool live = !blocked(g);
    const int e = (blockIdx.x * TPB + threadIdx.x) / LG, lane = threadIdx.x % LG;
    const bool act = e < ne;
    const int p = act ? erow[e] : 0, q = act ? ecol[e] : 0;
    double s = pair_dot<LG, V2, NS>(X, Y, p, q, r, lane);
    s = group_sum<LG>(s);
    if (live && act && lane == 0) T[e] = s;
}

// Constraint values straight from the factors: w_i = sum_{entries e of A_i} a_e pairdot_e(X, Y), ONE WAVEFRONT PER
// CONSTRAINT (64 / LG entries in flight, LG lanes each), then the same bookkeeping as k_cv -- the pair-dot array T is
// never written.  w_out (may be null) additionally keeps the plain values (the operator's constraint weights).
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_cw(int nrow, const int *__restrict__ a_ptr, const int *__restrict__ a_p,
                                            const double *__restrict__ a_val, const int *__restrict__ a_q,
                                            const double *__restrict__ X,
                                            const double *__restrict__ Y, int r, double scale, double *__restrict__ w_out,
                                            double *__restrict__ cv, int mode, const int *__restrict__ row_idx,
                                            double *__restrict__ vec, Guard g, int ell_w) {
    const bool live = !blocked(g);
    constexpr int GR = 64 / LG; // entry groups per wavefront
    const int i = (blockIdx.x * TPB + threadIdx.x) >> 6, lane64 = threadIdx.x & 63;
    const int grp = lane64 / LG, lane = lane64 % LG;
    const bool act = i < nrow;
    const int ic = act ? i : 0;
    // ell_w > 0: the entry arrays are padded to ell_w per constraint (a = 0 in the padding) -- no row-pointer load
    // heads the dependent-load chain
    const int t0 = ell_w ? ic * ell_w : a_ptr[ic], t1 = !act ? t0 : ell_w ? t0 + ell_w : a_ptr[ic + 1];
    // two entries per group and trip: the (row, col, a) of both are fetched before the first row gather (rows and
    // columns are stored per constraint entry -- no detour through the pattern entry)
    double acc = 0.0;
    for (int t = t0 + grp; t < t1; t += 2 * GR) {
        const int tb = t + GR < t1 ? t + GR : t;
        const int p0 = a_p[t], q0 = a_q[t], p1 = a_p[tb], q1 = a_q[tb];
        const double c0 = a_val[t], c1 = t + GR < t1 ? a_val[tb] : 0.0;
        const double d0 = pair_dot<LG, V2, NS>(X, Y, p0, q0, r, lane);
        const double d1 = pair_dot<LG, V2, NS>(X, Y, p1, q1, r, lane);
        acc += c0 * d0;
        acc += c1 * d1;
    }
    const double s = wave_sum(acc);
    if (live && act && lane64 == 0) {
        if (w_out) w_out[i] = s;
        if (vec) {
            const int gi = row_idx[i];
            if (mode == CV_SET) vec[gi] = s * scale;
            else if (mode == CV_ADD) vec[gi] += s * scale;
            else vec[gi] += s - cv[i];
        }
        if (cv) cv[i] = s;
    }
}

// Both pair dots of the line search from one visit of the four rows (ALMCalq12p12, lorads_alm.c:540-560):
// t1 = sym-pair(R, D), t2 = sym-pair(D, D), each summed exactly as pair_dot does.
template <int LG, bool V2, int NS>
__device__ __forceinline__ void pair_dot_rd(const double *__restrict__ R, const double *__restrict__ D, int p, int q, int r,
                                            int lane, double &t1, double &t2) {
    constexpr int W = V2 ? 2 : 1;
    double rp[NS][W], rq[NS][W], dp[NS][W], dq[NS][W];
    Slice<LG, V2, NS>::load(R + (size_t)p * r, r, lane, rp);
    Slice<LG, V2, NS>::load(D + (size_t)q * r, r, lane, dq);
    Slice<LG, V2, NS>::load(R + (size_t)q * r, r, lane, rq);
    Slice<LG, V2, NS>::load(D + (size_t)p * r, r, lane, dp);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int w = 0; w < W; ++w) {
            a1 += rp[c][w] * dq[c][w]; a2 += rq[c][w] * dp[c][w];
            b1 += dp[c][w] * dq[c][w]; b2 += dq[c][w] * dp[c][w];
        }
    t1 = p == q ? a1 : a1 + a2;
    t2 = p == q ? b1 : b1 + b2;
}
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_pairdots_rd(int ne, const int *__restrict__ erow, const int *__restrict__ ecol,
                                                     const double *__restrict__ R, const double *__restrict__ D, int r,
                                                     double *__restrict__ T1, double *__restrict__ T2) {
    const int e = (blockIdx.x * TPB + threadIdx.x) / LG, lane = threadIdx.x % LG;
    const bool act = e < ne;
    const int p = act ? erow[e] : 0, q = act ? ecol[e] : 0;
    double t1, t2;
    pair_dot_rd<LG, V2, NS>(R, D, p, q, r, lane, t1, t2);
    t1 = group_sum<LG>(t1);
    t2 = group_sum<LG>(t2);
    if (act && lane == 0) { T1[e] = t1; T2[e] = t2; }
}
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_obj_rd(int ne, const int *__restrict__ erow, const int *__restrict__ ecol,
                                                const double *__restrict__ cval, const double *__restrict__ R,
                                                const double *__restrict__ D, int r, double *__restrict__ part1,
                                                double *__restrict__ part2) {
    __shared__ double sh[4];
    const int lane = threadIdx.x % LG, per = TPB / LG;
    double s1 = 0.0, s2 = 0.0;
    for (int e = blockIdx.x * per + threadIdx.x / LG; e < ne; e += gridDim.x * per) {
        double t1, t2;
        pair_dot_rd<LG, V2, NS>(R, D, erow[e], ecol[e], r, lane, t1, t2);
        s1 += t1 * cval[e];
        s2 += t2 * cval[e];
    }
    const double u1 = block_sum(s1, sh), u2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { part1[blockIdx.x] = u1; part2[blockIdx.x] = u2; }
}
// q1 = 2 A(T1), q2 = A(T2) for a cone that sees every constraint (vec1/vec2 are SET); cv keeps the second one,
// as the two successive k_cv passes leave it.  The rows a workgroup owns also give its share of the seven
// line-search sums (see k_linesearch), so no kernel has to stream the m-vectors again.
__global__ __launch_bounds__(TPB) void k_cv_rd(int nrow, const int *__restrict__ a_ptr, const int *__restrict__ a_e,
                                               const double *__restrict__ a_val, const double *__restrict__ T1,
                                               const double *__restrict__ T2, double *__restrict__ cv,
                                               const int *__restrict__ row_idx, double *__restrict__ vec1,
                                               double *__restrict__ vec2, const double *__restrict__ b,
                                               const double *__restrict__ csum, const double *__restrict__ lambda,
                                               double *__restrict__ part);
// out[b] = scale_b * sum(part_b), b = blockIdx.x in {0, 1}
__global__ __launch_bounds__(TPB) void k_finalize2(const double *__restrict__ part0, const double *__restrict__ part1, int n,
                                                   double scale0, double scale1, double *out) {
    __shared__ double sh[4];
    const double t = sum_partials(blockIdx.x ? part1 : part0, n, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = (blockIdx.x ? scale1 : scale0) * t;
}

// partial of sum_e c_e * pairdot_e  (objective <C, sym(X Y^T)>); grid-stride so that the grid stays <= MAXPART
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_obj(int ne, const int *__restrict__ erow, const int *__restrict__ ecol,
                                             const double *__restrict__ cval, const double *__restrict__ X,
                                             const double *__restrict__ Y, int r, double *__restrict__ part, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const int lane = threadIdx.x % LG, per = TPB / LG;
    double s = 0.0; // every lane keeps its own slice; the block sum adds the slices
    for (int e = blockIdx.x * per + threadIdx.x / LG; e < ne; e += gridDim.x * per)
        s += pair_dot<LG, V2, NS>(X, Y, erow[e], ecol[e], r, lane) * cval[e];
    const double t = block_sum(s, sh);
    if (live && threadIdx.x == 0) part[blockIdx.x] = t;
}

// w_i = sum_k a_k T[e_k]; 8 lanes per constraint.  `mode` says what happens to the running
// m-vector `vec` (constrValSum / q1 / q2): CV_SET vec[g] = scale w, CV_ADD vec[g] += scale w,
// CV_DELTA vec[g] += w - old (the subtract/recompute/add bookkeeping of LORADSUpdateSDPVar,
// lorads_alg_common.c:199-203); cv (compact constrVal[k]) gets w
__global__ __launch_bounds__(TPB) void k_cv(int nrow, const int *__restrict__ a_ptr, const int *__restrict__ a_e,
                                            const double *__restrict__ a_val, const double *__restrict__ T, double scale,
                                            double *__restrict__ cv, int mode, const int *__restrict__ row_idx,
                                            double *__restrict__ vec, Guard g) {
    const bool live = !blocked(g);
    const int i = (blockIdx.x * TPB + threadIdx.x) / 8, lane = threadIdx.x & 7;
    const bool act = i < nrow;
    double s = 0.0;
    if (act)
        for (int t = a_ptr[i] + lane; t < a_ptr[i + 1]; t += 8) s += a_val[t] * T[a_e[t]];
    s = group_sum<8>(s);
    if (live && act && lane == 0) {
        if (vec) {
            const int gi = row_idx[i];
            if (mode == CV_SET) vec[gi] = s * scale;
            else if (mode == CV_ADD) vec[gi] += s * scale;
            else vec[gi] += s - cv[i];
        }
        if (cv) cv[i] = s;
    }
}

struct WArgs {
    const double *w;      // W_COMPACT: compact weights [nrow]
    const double *csum;   // global m-vectors
    const double *b;
    const double *lambda;
    const double *cv;     // compact [nrow]
    const int *row_idx;
    double rho;
};
__device__ __forceinline__ double weight_of(int mode, const WArgs &a, int i) {
    if (mode == W_COMPACT) return a.w[i];
    const int g = a.row_idx[i];
    if (mode == W_ADMM) return ((a.csum[g] - a.b[g]) - a.cv[i]) * a.rho - a.lambda[g]; // lorads_admm.c:432-445
    if (mode == W_DUAL) return -a.lambda[g];                                            // data/lorads_solver.c:1011
    return (-a.lambda[g] - a.rho * a.b[g]) + a.rho * a.csum[g];                         // lorads_alm.c:22-26
}
// S_e = [cbase_e] + sum over the constraints touching e of weight_i * a
__global__ __launch_bounds__(TPB) void k_sval(int ne, const int *__restrict__ e_ptr, const int *__restrict__ e_con,
                                              const double *__restrict__ e_val, const double *__restrict__ cbase, int mode,
                                              WArgs wa, double *__restrict__ S, Guard g, CGState *reset, int nreset) {
    // first kernel of a sweep: mark every later stage "not finished" (stage 0 itself is reset by its k_cg_init,
    // nothing before that reads it)
    if (reset && blockIdx.x == 0 && threadIdx.x < nreset) reset[threadIdx.x].done = 0;
    const bool live = !blocked(g);
    const int e = blockIdx.x * TPB + threadIdx.x;
    if (e >= ne) return;
    double s = cbase ? cbase[e] : 0.0;
    for (int t = e_ptr[e]; t < e_ptr[e + 1]; ++t) s += weight_of(mode, wa, e_con[t]) * e_val[t];
    if (live) S[e] = s;
}
// S = G T with G = A A^T over pattern entries (w = A T and S = A^T w in one pass); 8 lanes per entry so
// that the dependent (index -> T) loads of one Gram row are in flight together
__global__ __launch_bounds__(TPB) void k_sgram(int ne, const int *__restrict__ g_ptr, const int *__restrict__ g_col,
                                               const double *__restrict__ g_val, const double *__restrict__ T,
                                               double *__restrict__ S, Guard g) {
    const bool live = !blocked(g);
    const int e = (blockIdx.x * TPB + threadIdx.x) / 8, lane = threadIdx.x & 7;
    const bool act = e < ne;
    double s = 0.0;
    if (act)
        for (int t = g_ptr[e] + lane; t < g_ptr[e + 1]; t += 8) s += g_val[t] * T[g_col[t]];
    s = group_sum<8>(s);
    if (live && act && lane == 0) S[e] = s;
}

// Y_p = epilogue( sum over the neighbours (q,e) of row p of S_e X_q ), LG lanes per row, + fused reduction.
// CW = true: the slot list is per (neighbour, constraint): adj_e holds the compact constraint index, S the constraint
// weights w and adj_a the coefficient a, so that the slot coefficient a * w[con] is formed without an S array.
template <int LG, bool V2, int NS, bool CW = false>
__global__ __launch_bounds__(TPB) void k_spmm(int n, const int *__restrict__ adj_ptr, const int *__restrict__ adj_col,
                                              const int *__restrict__ adj_e, const double *__restrict__ S,
                                              const double *__restrict__ X, int r, int mode, const double *__restrict__ xin,
                                              const double *__restrict__ rhs, double rho, double *__restrict__ out,
                                              double *__restrict__ part, Guard g, const double *__restrict__ dense_add,
                                              const double *__restrict__ adj_a = nullptr) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    constexpr int W = V2 ? 2 : 1;
    const int row = (blockIdx.x * TPB + threadIdx.x) / LG, lane = threadIdx.x % LG;
    const bool act = row < n;
    const int rowc = act ? row : 0;
    double acc[NS][W];
    // dense part of (C + sum_i w_i A_i) X, computed by k_dense_cx, when C is stored dense
    if (dense_add) Slice<LG, V2, NS>::load(dense_add + (size_t)rowc * r, r, lane, acc);
    else {
#pragma unroll
        for (int c = 0; c < NS; ++c)
#pragma unroll
            for (int w = 0; w < W; ++w) acc[c][w] = 0.0;
    }
    const int t0 = adj_ptr[rowc], t1 = act ? adj_ptr[rowc + 1] : t0;
    // epilogue operands are fetched up front: the kernel is bound by its chain of dependent loads
    // (row pointer -> neighbour index -> coefficient / neighbour row), not by bandwidth
    const size_t base = (size_t)rowc * r;
    double xi[NS][W], rh[NS][W];
    if (mode == OP_CG || mode == OP_RES) Slice<LG, V2, NS>::load(xin + base, r, lane, xi);
    if (mode == OP_RES) Slice<LG, V2, NS>::load(rhs + base, r, lane, rh);
    if (mode == OP_RHS) Slice<LG, V2, NS>::load(X + base, r, lane, xi);
    // 4 neighbours per trip: the 8 index loads, then the 4 coefficient and 4 row gathers are all issued
    // before the first use (out-of-range slots repeat the last neighbour with coefficient 0)
    for (int t = t0; t < t1; t += 4) {
        int q[4], e[4];
        double aa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = t + u < t1 ? t + u : t1 - 1;
            q[u] = adj_col[tt];
            e[u] = adj_e[tt];
            aa[u] = CW ? adj_a[tt] : 1.0;
        }
        double sc[4], v[4][NS][W];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sc[u] = CW ? aa[u] * S[e[u]] : S[e[u]];
            Slice<LG, V2, NS>::load(X + (size_t)q[u] * r, r, lane, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double su = t + u < t1 ? sc[u] : 0.0;
#pragma unroll
            for (int c = 0; c < NS; ++c)
#pragma unroll
                for (int w = 0; w < W; ++w) acc[c][w] += su * v[u][c][w];
        }
    }
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int j0 = (lane + c * LG) * W;
        double v[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const double a = acc[c][w];
            if (mode == OP_CG) { v[w] = xi[c][w] + a; local += xi[c][w] * v[w]; }
            else if (mode == OP_RES) { v[w] = rh[c][w] - (xi[c][w] + a); local += v[w] * v[w]; }
            else if (mode == OP_RHS) { v[w] = xi[c][w] - a / rho; local += fabs(v[w]); }
            else { v[w] = 2.0 * a; local += v[w] * v[w]; }
        }
        if (live && act && j0 < r) {
            if (V2) *(double2 *)(out + base + j0) = make_double2(v[0], v[W - 1]);
            else out[base + j0] = v[0];
        }
    }
    const double t = block_sum(act ? local : 0.0, sh);
    if (live && threadIdx.x == 0) part[blockIdx.x] = t;
}

// Max-Cut-type cones (every A_i = a_i e_p e_p^T): the whole operator is row-local,
//   out_p = x_p + g_p (x_p . V_p) V_p,  g_p = sum_i a_i^2  -> one pass over x and V
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_op_diag(int n, const double *__restrict__ gd, const double *__restrict__ V, int r,
                                                 int mode, const double *__restrict__ xin, const double *__restrict__ rhs,
                                                 double *__restrict__ out, double *__restrict__ part, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    constexpr int W = V2 ? 2 : 1;
    const int row = (blockIdx.x * TPB + threadIdx.x) / LG, lane = threadIdx.x % LG;
    const bool act = row < n;
    const int rowc = act ? row : 0;
    const size_t base = (size_t)rowc * r;
    double xv[NS][W], vv[NS][W], rh[NS][W];
    Slice<LG, V2, NS>::load(xin + base, r, lane, xv);
    Slice<LG, V2, NS>::load(V + base, r, lane, vv);
    if (mode == OP_RES) Slice<LG, V2, NS>::load(rhs + base, r, lane, rh);
    const double gr = gd[rowc];
    double d = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int w = 0; w < W; ++w) d += xv[c][w] * vv[c][w];
    d = group_sum<LG>(d) * gr;
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int j0 = (lane + c * LG) * W;
        double v[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            v[w] = xv[c][w] + d * vv[c][w];
            if (mode == OP_CG) local += xv[c][w] * v[w];
            else { v[w] = rh[c][w] - v[w]; local += v[w] * v[w]; }
        }
        if (live && act && j0 < r) {
            if (V2) *(double2 *)(out + base + j0) = make_double2(v[0], v[W - 1]);
            else out[base + j0] = v[0];
        }
    }
    const double t = block_sum(act ? local : 0.0, sh);
    if (live && threadIdx.x == 0) part[blockIdx.x] = t;
}

// Cones whose constraints each hold ONE pattern entry (matrix completion: A_k = a (e_i e_j^T + e_j e_i^T)/2 ...):
// w_i = a_i T_e and S_e = (sum_i a_i^2) T_e = ge_e T_e stay entry-local, so the whole operator is one row-centric pass
//   out_p = x_p + sum_{(q,e) adjacent to p} ge_e (x_p.V_q + x_q.V_p) V_q        (q = p: ge_e (x_p.V_p) V_p)
// -- no T, no S, no second and third kernel.  Every entry is visited from both of its rows (the pair dot is formed
// twice); what it saves is two launches and the 4-rows-per-entry gather of the pair-dot kernel.
template <int LG, bool V2, int NS>
__global__ __launch_bounds__(TPB) void k_op_entry(int n, const int *__restrict__ adj_ptr, const int *__restrict__ adj_col,
                                                  const int *__restrict__ adj_e, const double *__restrict__ ge,
                                                  const double *__restrict__ V, int r, int mode, const double *__restrict__ xin,
                                                  const double *__restrict__ rhs, double *__restrict__ out,
                                                  double *__restrict__ part, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    constexpr int W = V2 ? 2 : 1;
    const int row = (blockIdx.x * TPB + threadIdx.x) / LG, lane = threadIdx.x % LG;
    const bool act = row < n;
    const int rowc = act ? row : 0;
    const size_t base = (size_t)rowc * r;
    const int t0 = adj_ptr[rowc], t1 = act ? adj_ptr[rowc + 1] : t0;
    double xp[NS][W], vp[NS][W], rh[NS][W], acc[NS][W];
    Slice<LG, V2, NS>::load(xin + base, r, lane, xp);
    Slice<LG, V2, NS>::load(V + base, r, lane, vp);
    if (mode == OP_RES) Slice<LG, V2, NS>::load(rhs + base, r, lane, rh);
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int w = 0; w < W; ++w) acc[c][w] = 0.0;
    for (int t = t0; t < t1; t += 2) { // two neighbours per trip: 4 row gathers in flight
        int q[2], e[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tt = t + u < t1 ? t + u : t1 - 1;
            q[u] = adj_col[tt];
            e[u] = adj_e[tt];
        }
        double gc[2], xq[2][NS][W], vq[2][NS][W];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            gc[u] = ge[e[u]];
            Slice<LG, V2, NS>::load(xin + (size_t)q[u] * r, r, lane, xq[u]);
            Slice<LG, V2, NS>::load(V + (size_t)q[u] * r, r, lane, vq[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int c = 0; c < NS; ++c)
#pragma unroll
                for (int w = 0; w < W; ++w) { s1 += xp[c][w] * vq[u][c][w]; s2 += xq[u][c][w] * vp[c][w]; }
            double td = q[u] == rowc ? s1 : s1 + s2;
            td = group_sum<LG>(td);
            const double su = t + u < t1 ? gc[u] * td : 0.0;
#pragma unroll
            for (int c = 0; c < NS; ++c)
#pragma unroll
                for (int w = 0; w < W; ++w) acc[c][w] += su * vq[u][c][w];
        }
    }
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int j0 = (lane + c * LG) * W;
        double v[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            v[w] = xp[c][w] + acc[c][w];
            if (mode == OP_CG) local += xp[c][w] * v[w];
            else { v[w] = rh[c][w] - v[w]; local += v[w] * v[w]; }
        }
        if (live && act && j0 < r) {
            if (V2) *(double2 *)(out + base + j0) = make_double2(v[0], v[W - 1]);
            else out[base + j0] = v[0];
        }
    }
    const double t = block_sum(act ? local : 0.0, sh);
    if (live && threadIdx.x == 0) part[blockIdx.x] = t;
}

// Dense objective matrix: W = C X with C dense symmetric (n_pad x n_pad row-major, zero padded) and X n x r
// row-major -- the reference's dense branch (unpack + dsymm, lorads_sdp_data.c:646-671) -- on the FP64 matrix
// cores: v_mfma_f64_16x16x4_f64.  Workgroup (bx, by): 64 rows of C (16 per wave) x the by-th K range
// (split-K over workgroups so that the grid fills 256 CUs; the KS partial results are summed in a fixed
// order by k_sum_slabs).  The 4 waves share each 32-row slab of X, staged through LDS as a plain linear copy
// (unpadded [32][r]; the column padding to 16 NT is done when the B fragment is read).  Each C element is read
// once from HBM: 2 n^2 r flop over 8 n^2 bytes = r/4 flop/B.  Lane l = (i = l & 15, g = l >> 4) loads
// C[row0+i][k+4g .. +3] (32 B) and feeds MFMA t (t = 0..3) with a = C[row0+i][k+4g+t], b = X[k+4g+t][col];
// the k index a lane group stands for only has to agree between A and B.  D layout of the f64 form:
// col = l & 15, row = (l >> 4) + 4 reg.
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NT> // column tiles of 16 (r <= 16 NT)
__global__ __launch_bounds__(TPB) void k_dense_cx(int n, int npad, int krange, const double *__restrict__ Cf,
                                                  const double *__restrict__ X, int r, double *__restrict__ Wpart, Guard g) {
    extern __shared__ __attribute__((aligned(16))) double xs[]; // [32][r]
    const bool live = !blocked(g);
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 15, gk = l >> 4;
    const int row0 = blockIdx.x * 64 + wave * 16;
    const int kbeg = blockIdx.y * krange;
    v4f64 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    // column of the B fragment per tile, clamped into the row; padded columns are masked to 0
    int colc[NT];
    double colm[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int col = 16 * ct + i;
        colc[ct] = col < r ? col : 0;
        colm[ct] = col < r ? 1.0 : 0.0;
    }
    const double *crow = Cf + (size_t)(row0 + i) * npad + 4 * gk;
    const size_t xlen = (size_t)n * r;
    for (int k0 = kbeg; k0 < kbeg + krange; k0 += 32) {
        const double2 c0 = *(const double2 *)(crow + k0), c1 = *(const double2 *)(crow + k0 + 2);
        const double2 c2 = *(const double2 *)(crow + k0 + 16), c3 = *(const double2 *)(crow + k0 + 18);
        __syncthreads();
        const size_t xoff = (size_t)k0 * r;
        for (int idx = threadIdx.x; idx < 32 * r; idx += TPB) // rows beyond n are zero (C is zero padded too)
            xs[idx] = xoff + idx < xlen ? X[xoff + idx] : 0.0;
        __syncthreads();
        const double cv8[8] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, c3.x, c3.y};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double *xr = xs + (16 * (t >> 2) + 4 * gk + (t & 3)) * r;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv8[t], xr[colc[ct]] * colm[ct], acc[ct], 0, 0, 0);
        }
    }
    if (!live) return;
    double *Wo = Wpart + (size_t)blockIdx.y * n * r;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = row0 + gk + 4 * q, col = 16 * ct + i;
            if (row < n && col < r) Wo[(size_t)row * r + col] = acc[ct][q];
        }
}
// W = sum of the split-K slabs, fixed order
__global__ void k_sum_slabs(size_t len, int ks, const double *__restrict__ part, double *__restrict__ W, Guard g) {
    const bool live = !blocked(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        double v = part[i];
        for (int k = 1; k < ks; ++k) v += part[(size_t)k * len + i];
        if (live) W[i] = v;
    }
}

// start of CGSolve (lorads_cgs.c:115,149-172): ||b||_1 and the initial residual norm from partials; every
// workgroup recomputes the two sums (same order -> same value), workgroup 0 publishes the state; p = r
// (one workgroup; the direction of iteration 0 is r itself -- its tail is always the k = 0 restart, which
// sets p = 2 r_true without reading the old p -- so no p = r copy is made)
__global__ __launch_bounds__(TPB) void k_cg_init(CGState *st, const double *__restrict__ part_rr, int nrr,
                                                 const double *__restrict__ part_b, int nb, double tol, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const double a = sum_partials(part_rr, nrr, sh);
    const double b = sum_partials(part_b, nb, sh);
    const bool conv = sqrt(a) / b < tol;
    if (!live || threadIdx.x != 0) return;
    st->rr = a; st->bnorm = b; st->beta = 0.0; st->iter = 0; st->nan = 0; st->pad = 0;
    st->done = conv ? 2 : 0;
}

// x += alpha p, r -= alpha Q, partial ||r||^2; alpha = rr / (p.Q) from device scalars (lorads_cgs.c:181-189)
__global__ __launch_bounds__(TPB) void k_cg_update(size_t len, const CGState *st, const double *__restrict__ part_pq, int npq,
                                                   double *__restrict__ x, double *r, const double *p,
                                                   const double *__restrict__ Q, double *__restrict__ part_rr, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const double rr = st->rr; // issued before the partial sums so that the loads overlap
    const double pq = sum_partials(part_pq, npq, sh);
    const double alpha = rr / pq;
    if (!live) return;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < len; i += (size_t)gridDim.x * TPB) {
        const double pv = p[i]; // p may alias r (iteration 0): read both before writing r
        x[i] += alpha * pv;
        const double rv = r[i] - alpha * Q[i];
        r[i] = rv;
        local += rv * rv;
    }
    const double t = block_sum(local, sh);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
}

// scalar bookkeeping of CGSolve, one workgroup
__global__ __launch_bounds__(TPB) void k_cg_check(CGState *st, int kind, const double *__restrict__ part_a, int na, double tol,
                                                  int maxiter, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const double rr_old = st->rr, bnorm = st->bnorm; // issued before the partial sums: the loads overlap
    const int it = st->iter;
    const double a = sum_partials(part_a, na, sh);
    if (!live || threadIdx.x != 0) return;
    if (kind == CHK_ITER) { // lorads_cgs.c:189-194, :217-224
        st->iter = it + 1;
        if (a != a) st->nan = 1;
        st->beta = a / rr_old;
        st->rr = a;
        if (sqrt(a) / bnorm < tol) st->done = 1;
        else if (it + 1 >= maxiter) st->done = 3;
    } else { // restart: true residual, then beta = qTrNew/qTr = 1 (:195-221)
        st->rr = a;
        st->beta = 1.0;
    }
}

__global__ void k_cg_dir(size_t len, const CGState *st, int kind, const double *__restrict__ r, double *__restrict__ p, Guard g) {
    if (blocked(g)) return;
    const double beta = st->beta;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        const double rv = r[i];
        p[i] = kind == DIR_RESTART ? rv + rv : rv + beta * p[i];
    }
}

// ---- CG of MANY cones in lockstep on the merged block-diagonal cone (block-separable constraints, equal rank).
// The gather kernels run once for all cones; what stays per cone are the scalars of CGSolve (alpha, beta, norms,
// stopping): SegArgs maps workgroups to cones.  row0[k] = first (padded) row of cone k, a multiple of 32, so the
// per-workgroup partials of the row kernels never straddle two cones: cone k owns tiles [row0[k]/rpw, row0[k+1]/rpw).
// The vector kernels run over chunks of SEG_CH elements, chunk j of the launch belongs to cone vt_seg[j] and starts
// at element vt_e0[j]; cone k owns chunk partials [vt0[k], vt0[k+1]).  State of (cone k, half) is st[2k + half].
// phase_done[half] != 0 <=> every cone's solve of that half has finished (uniform gate for the gather kernels).
constexpr int SEG_CH = 2048;
struct SegArgs {
    const int *row0, *vt0, *vt_seg;
    const long long *vt_e0;
    CGState *st;
    int *phase_done;
    int half, ncones, rpw, r;
};
__device__ __forceinline__ double wave_partials(const double *part, int n) { // whole wavefront, result in every lane
    double v = 0.0;
    for (int i = threadIdx.x & 63; i < n; i += 64) v += part[i];
    return wave_sum(v);
}
// all cones finished? (after a barrier that follows the state writes)
__device__ __forceinline__ void publish_phase_done(const SegArgs &sa, int *flag_sh) {
    if (threadIdx.x == 0) *flag_sh = 1;
    __syncthreads();
    for (int k = threadIdx.x; k < sa.ncones; k += TPB)
        if (sa.st[2 * k + sa.half].done == 0) *flag_sh = 0;
    __syncthreads();
    if (threadIdx.x == 0) sa.phase_done[sa.half] = *flag_sh;
}
// k_cg_init for every cone (one workgroup, one wavefront per cone in turn); the U half also re-arms the V half
__global__ __launch_bounds__(TPB) void k_cg_init_seg(SegArgs sa, const double *__restrict__ part_rr,
                                                     const double *__restrict__ part_b, double tol, Guard g) {
    __shared__ int flag;
    if (blocked(g)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < sa.ncones; k += TPB / 64) {
        const int tb = sa.row0[k] / sa.rpw, te = sa.row0[k + 1] / sa.rpw;
        const double a = wave_partials(part_rr + tb, te - tb), b = wave_partials(part_b + tb, te - tb);
        if (lane == 0) {
            CGState *st = sa.st + 2 * k + sa.half;
            st->rr = a; st->bnorm = b; st->beta = 0.0; st->iter = 0; st->nan = 0; st->pad = 1; // pad = "started"
            st->done = sqrt(a) / b < tol ? 2 : 0;
            if (sa.half == 0) { sa.st[2 * k + 1].done = 0; sa.st[2 * k + 1].pad = 0; }
        }
    }
    __threadfence_block();
    if (sa.half == 0 && threadIdx.x == 0) sa.phase_done[1] = 0;
    publish_phase_done(sa, &flag);
}
// k_cg_update per chunk with the alpha of the chunk's cone
__global__ __launch_bounds__(TPB) void k_cg_update_seg(SegArgs sa, const double *__restrict__ part_pq, double *__restrict__ x,
                                                       double *r, const double *p, const double *__restrict__ Q,
                                                       double *__restrict__ part_rr, Guard g) {
    __shared__ double sh[4];
    if (blocked(g)) return;
    const int k = sa.vt_seg[blockIdx.x];
    const CGState *st = sa.st + 2 * k + sa.half;
    if (st->done != 0) return;
    const int tb = sa.row0[k] / sa.rpw, te = sa.row0[k + 1] / sa.rpw;
    const double rr = st->rr;
    const double pq = sum_partials(part_pq + tb, te - tb, sh);
    const double alpha = rr / pq;
    const long long e0 = sa.vt_e0[blockIdx.x], eend = (long long)sa.row0[k + 1] * sa.r;
    const long long e1 = e0 + SEG_CH < eend ? e0 + SEG_CH : eend;
    double local = 0.0;
    for (long long i = e0 + threadIdx.x; i < e1; i += TPB) {
        const double pv = p[i]; // p may alias r (iteration 0)
        x[i] += alpha * pv;
        const double rv = r[i] - alpha * Q[i];
        r[i] = rv;
        local += rv * rv;
    }
    const double t = block_sum(local, sh);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
}
// k_cg_check for every cone (one workgroup); part: chunk partials (CHK_ITER) or row-tile partials (CHK_RESTART)
__global__ __launch_bounds__(TPB) void k_cg_check_seg(SegArgs sa, int kind, const double *__restrict__ part, double tol,
                                                      int maxiter, Guard g) {
    __shared__ int flag;
    if (blocked(g)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < sa.ncones; k += TPB / 64) {
        CGState *st = sa.st + 2 * k + sa.half;
        if (st->done != 0) continue; // uniform per wavefront
        int lo, hi;
        if (kind == CHK_ITER) { lo = sa.vt0[k]; hi = sa.vt0[k + 1]; }
        else { lo = sa.row0[k] / sa.rpw; hi = sa.row0[k + 1] / sa.rpw; }
        const double a = wave_partials(part + lo, hi - lo);
        if (lane == 0) {
            if (kind == CHK_ITER) {
                const int it = st->iter;
                const double rr_old = st->rr;
                st->iter = it + 1;
                if (a != a) st->nan = 1;
                st->beta = a / rr_old;
                st->rr = a;
                if (sqrt(a) / st->bnorm < tol) st->done = 1;
                else if (it + 1 >= maxiter) st->done = 3;
            } else {
                st->rr = a;
                st->beta = 1.0;
            }
        }
    }
    __threadfence_block();
    publish_phase_done(sa, &flag);
}
__global__ __launch_bounds__(TPB) void k_cg_dir_seg(SegArgs sa, int kind, const double *__restrict__ r, double *__restrict__ p,
                                                    Guard g) {
    if (blocked(g)) return;
    const int k = sa.vt_seg[blockIdx.x];
    const CGState *st = sa.st + 2 * k + sa.half;
    if (st->done != 0) return;
    const double beta = st->beta;
    const long long e0 = sa.vt_e0[blockIdx.x], eend = (long long)sa.row0[k + 1] * sa.r;
    const long long e1 = e0 + SEG_CH < eend ? e0 + SEG_CH : eend;
    for (long long i = e0 + threadIdx.x; i < e1; i += TPB) {
        const double rv = r[i];
        p[i] = kind == DIR_RESTART ? rv + rv : rv + beta * p[i];
    }
}

// 1.0 when the gate word says "not finished" (this rank missed its speculation), else 0.0 -- rides on the all-reduce
__global__ void k_miss_flag(const int *need, double *out) { *out = (need && *need == 0) ? 1.0 : 0.0; }
// constrValSum <- the all-reduced staging vector, unless some rank reported a miss (stage[m+1] = number of such ranks):
// then every rank keeps its constrValSum -- the unfinished sweeps still need it -- and the evaluation is repeated
__global__ void k_commit_csum(int m, const double *__restrict__ stage, double *__restrict__ csum) {
    if (stage[m + 1] != 0.0) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) csum[i] = stage[i];
}

// ---- LP block (one diagonal cone, rank 1, lorads_hip_block.is_lp).  In phase 1 and in every evaluation it runs
// through the generic cone kernels.  The ADMM update is the reference's closed form, column by column in file order
// (LORADSUpdateLPVarOne, lorads_admm.c:595-629; bookkeeping lorads_alg_common.c:236-246):
//   m1_k = rho (csum[g_k] - cvLP_k - b[g_k]) - lambda[g_k],  w = c + sum_k m1_k a_k,
//   u <- (-(w v - rho v) / rho) / (1 + ||a||^2 v^2),  cvLP_k <- a_k u v (csum follows),  then the same for v.
// Columns that share no constraint row do not see each other, so the sweep is LEVEL-SCHEDULED: level(j) =
// 1 + max level of the earlier columns sharing a row with j; columns of one level are updated in parallel, levels in
// order -- exactly the sequential result.  Slack-type blocks (every column alone in its row) are one level.
struct LpArgs {
    int ncols, nlev;
    const int *lvl_ptr, *lvl_cols, *ptr, *grow; // level -> columns; column -> entries; entry -> global constraint
    const double *a, *nrm2sq, *cobj;
    double *cv, *U, *V, *csum;
    const double *b, *lambda;
    double rho;
};
__device__ __forceinline__ double lp_new_value(const LpArgs &A, int col, double fixed) {
    double w = A.cobj[col];
    for (int t = A.ptr[col]; t < A.ptr[col + 1]; ++t) {
        const int g = A.grow[t];
        double m1 = A.b[g];
        m1 *= -1.0;
        m1 += A.csum[g];
        m1 += -1.0 * A.cv[t];
        m1 *= A.rho;
        m1 += -1.0 * A.lambda[g];
        w += m1 * A.a[t];
    }
    double M2 = w * fixed;
    M2 = M2 - A.rho * fixed;
    const double blin = -1.0 * M2 / A.rho;
    return blin / (1 + A.nrm2sq[col] * fixed * fixed);
}
__device__ __forceinline__ void lp_refresh(const LpArgs &A, int col, double uv) {
    for (int t = A.ptr[col]; t < A.ptr[col + 1]; ++t) {
        const int g = A.grow[t];
        double cs = A.csum[g] + -1.0 * A.cv[t];
        const double nv = A.a[t] * uv;
        A.cv[t] = nv;
        A.csum[g] = cs + nv;
    }
}
__device__ __forceinline__ void lp_column(const LpArgs &A, int col) {
    const double u = lp_new_value(A, col, A.V[col]);
    A.U[col] = u;
    lp_refresh(A, col, u * A.V[col]);
    const double v = lp_new_value(A, col, u);
    A.V[col] = v;
    lp_refresh(A, col, u * v);
}
// one workgroup walks the levels; done words of the block's two stages are set at the end
__global__ __launch_bounds__(TPB) void k_lp_sweep(LpArgs A, CGState *st, Guard g) {
    if (blocked(g)) return;
    for (int lev = 0; lev < A.nlev; ++lev) {
        for (int i = A.lvl_ptr[lev] + threadIdx.x; i < A.lvl_ptr[lev + 1]; i += TPB) lp_column(A, A.lvl_cols[i]);
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st[0].iter = 0; st[0].nan = 0; st[0].pad = 1; st[0].done = 1;
        st[1].iter = 0; st[1].nan = 0; st[1].pad = 1; st[1].done = 1;
    }
}
// single level: every column independent, many workgroups; the done words are set by k_lp_done afterwards
__global__ __launch_bounds__(TPB) void k_lp_sweep_flat(LpArgs A, Guard g) {
    if (blocked(g)) return;
    const int col = blockIdx.x * TPB + threadIdx.x;
    if (col < A.ncols) lp_column(A, col);
}
__global__ void k_lp_done(CGState *st, Guard g) {
    if (blocked(g)) return;
    st[0].iter = 0; st[0].nan = 0; st[0].pad = 1; st[0].done = 1;
    st[1].iter = 0; st[1].nan = 0; st[1].pad = 1; st[1].done = 1;
}
// constrValLP: cv_t = a_t x_col y_col for every stored entry (lp_cone_AUV, data/lorads_lp_conic.c:172-175)
__global__ void k_lp_cv(int ncols, const int *__restrict__ ptr, const double *__restrict__ a, const double *__restrict__ X,
                        const double *__restrict__ Y, double *__restrict__ cv, Guard g) {
    if (blocked(g)) return;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const double uv = X[col] * Y[col];
    for (int t = ptr[col]; t < ptr[col + 1]; ++t) cv[t] = a[t] * uv;
}
// sum over the columns of |min(c - sum_k lambda a, 0)| (data/lorads_solver.c:1015-1023), one workgroup
__global__ __launch_bounds__(TPB) void k_lp_dual(int ncols, const int *__restrict__ ptr, const int *__restrict__ grow,
                                                 const double *__restrict__ a, const double *__restrict__ cobj,
                                                 const double *__restrict__ lambda, double *out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int col = threadIdx.x; col < ncols; col += TPB) {
        double w = cobj[col];
        for (int t = ptr[col]; t < ptr[col + 1]; ++t) w += -lambda[grow[t]] * a[t];
        acc += fabs(w < 0.0 ? w : 0.0);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) *out = acc;
}

// Result hand-over without hipStreamSynchronize: one workgroup copies `nwords` 8-byte words of the control block into
// host-mapped pinned memory, fences at system scope and then publishes a sequence number the host spins on.  Cuts the
// per-iteration wake-up latency of an interrupt-driven stream synchronisation (and the separate copy kernel).
__global__ __launch_bounds__(TPB) void k_publish(const unsigned long long *__restrict__ src, int nwords, unsigned long long *dst,
                                                 unsigned long long *flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < nwords; i += TPB) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
        __threadfence_system();
    }
}

// ---- small vector kernels
__global__ void k_average(size_t len, const double *__restrict__ u, const double *__restrict__ v, double *__restrict__ out, Guard g) {
    if (blocked(g)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (u[i] + v[i]) / 2;
}
__global__ void k_zero(size_t len, double *__restrict__ y, Guard g) {
    if (blocked(g)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) y[i] = 0.0;
}
__global__ void k_axpy(size_t len, double a, const double *__restrict__ x, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
__global__ void k_axpy_dev(size_t len, const double *coef, const double *__restrict__ x, double *__restrict__ y) {
    const double a = *coef;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
__global__ void k_scale_copy(size_t len, double a, const double *__restrict__ x, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * x[i];
}
__global__ void k_scale(size_t len, double a, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) y[i] *= a;
}
// s = tau*D, y += G  (setlbfgsHisTwo)
__global__ void k_his_two(size_t len, double tau, const double *__restrict__ D, const double *__restrict__ G,
                          double *__restrict__ s, double *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        s[i] = tau * D[i];
        y[i] += G[i];
    }
}
// D = -G when <D,G> >= 0 (LBFGSDirectionUseGrad)
__global__ void k_use_grad(size_t len, const double *ip, const double *__restrict__ G, double *__restrict__ D) {
    if (!(*ip >= 0)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
        D[i] = -G[i];
}
__global__ __launch_bounds__(TPB) void k_dot(size_t len, const double *__restrict__ x, const double *__restrict__ y,
                                             double *__restrict__ part, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < len; i += (size_t)gridDim.x * TPB) local += x[i] * y[i];
    const double t = block_sum(local, sh);
    if (live && threadIdx.x == 0) part[blockIdx.x] = t;
}
// out (= or +=) scale * sum(part)
__global__ __launch_bounds__(TPB) void k_finalize(const double *__restrict__ part, int n, double scale, int accumulate,
                                                  double *out, Guard g) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const double t = sum_partials(part, n, sh);
    if (live && threadIdx.x == 0) *out = accumulate ? *out + scale * t : scale * t;
}
// One stage of the L-BFGS two-loop recursion (lorads_alm.c:230-391) per launch:
//   [coefficient from the previous stage's dot partials]  ->  q += coef * ax  ->  partials of <dv, q>.
// Every workgroup re-sums the previous partials in the order k_finalize uses, so all of them see the same
// coefficient and the arithmetic is that of the k_dot / k_finalize / k_scalar_op / k_axpy_dev chain it replaces.
// ab = {alpha, beta} of the history node the coefficient belongs to.  dout != nullptr: last stage, D = -q and
// the partials are those of <D, dv> (dv = Grad).
enum { ST_FIRST = 0, ST_ALPHA = 1, ST_W = 2 };
__global__ __launch_bounds__(TPB) void k_lbfgs_stage(size_t len, int op, const double *__restrict__ prev_part, int nprev,
                                                     double *ab, const double *__restrict__ ax, double *q,
                                                     const double *__restrict__ qsrc, const double *__restrict__ dv,
                                                     double *__restrict__ part_out, double *__restrict__ dout) {
    __shared__ double sh[4];
    double coef = 0.0;
    if (op != ST_FIRST) {
        const double dot = sum_partials(prev_part, nprev, sh);
        if (op == ST_ALPHA) {
            const double alpha = ab[1] * dot;
            coef = -1 * alpha;
            if (blockIdx.x == 0 && threadIdx.x == 0) ab[0] = alpha;
        } else {
            coef = ab[0] - ab[1] * dot;
        }
    }
    double local = 0.0;
    auto one = [&](double qi, double axi, double dvi, double &store) { // one element of the stage
        const double v = op == ST_FIRST ? qi : qi + coef * axi;
        if (dout) { store = -1.0 * v; local += store * dvi; }
        else { store = v; local += dvi * v; }
    };
    // 16-byte accesses: these stages only stream (3 vectors in, 1 out)
    const size_t gid = (size_t)blockIdx.x * TPB + threadIdx.x, stride = (size_t)gridDim.x * TPB, n2 = len / 2;
    const double2 *q2 = (const double2 *)(op == ST_FIRST ? qsrc : q), *ax2 = (const double2 *)ax, *dv2 = (const double2 *)dv;
    double2 *o2 = (double2 *)(dout ? dout : q);
    for (size_t i = gid; i < n2; i += stride) {
        const double2 qv = q2[i], dvv = dv2[i];
        const double2 axv = op == ST_FIRST ? qv : ax2[i];
        double2 st;
        one(qv.x, axv.x, dvv.x, st.x);
        one(qv.y, axv.y, dvv.y, st.y);
        o2[i] = st;
    }
    if ((len & 1) && gid == 0) {
        const size_t i = len - 1;
        double st;
        one(op == ST_FIRST ? qsrc[i] : q[i], op == ST_FIRST ? 0.0 : ax[i], dv[i], st);
        (dout ? dout : q)[i] = st;
    }
    const double t = block_sum(local, sh);
    if (threadIdx.x == 0) part_out[blockIdx.x] = t;
}
// D = -G when <D,G> >= 0 (LBFGSDirectionUseGrad), <D,G> given as partials
__global__ __launch_bounds__(TPB) void k_use_grad_p(size_t len, const double *__restrict__ part, int npart,
                                                    const double *__restrict__ G, double *__restrict__ D) {
    __shared__ double sh[4];
    const double ip = sum_partials(part, npart, sh);
    if (!(ip >= 0)) return;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < len; i += (size_t)gridDim.x * TPB) D[i] = -G[i];
}
// s = tau*D, y += G (setlbfgsHisTwo) and the partials of <y, s>
__global__ __launch_bounds__(TPB) void k_his_two_dot(size_t len, double tau, const double *__restrict__ D,
                                                     const double *__restrict__ G, double *__restrict__ s,
                                                     double *__restrict__ y, double *__restrict__ part) {
    __shared__ double sh[4];
    double local = 0.0;
    const size_t gid = (size_t)blockIdx.x * TPB + threadIdx.x, stride = (size_t)gridDim.x * TPB, n2 = len / 2;
    for (size_t i = gid; i < n2; i += stride) {
        const double2 d = ((const double2 *)D)[i], gg = ((const double2 *)G)[i], yo = ((const double2 *)y)[i];
        double2 si, yi;
        si.x = tau * d.x; yi.x = yo.x + gg.x;
        si.y = tau * d.y; yi.y = yo.y + gg.y;
        ((double2 *)s)[i] = si;
        ((double2 *)y)[i] = yi;
        local += yi.x * si.x;
        local += yi.y * si.y;
    }
    if ((len & 1) && gid == 0) {
        const size_t i = len - 1;
        const double si = tau * D[i], yi = y[i] + G[i];
        s[i] = si;
        y[i] = yi;
        local += yi * si;
    }
    const double t = block_sum(local, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// beta = 1 / sum(part)
__global__ __launch_bounds__(TPB) void k_finalize_beta(const double *__restrict__ part, int n, double *beta) {
    __shared__ double sh[4];
    const double t = sum_partials(part, n, sh);
    if (threadIdx.x == 0) *beta = 1.0 / t;
}
enum { SOP_ALPHA = 0, SOP_W = 1, SOP_BETA = 2 };
// scalar algebra of the two-loop recursion
__global__ void k_scalar_op(int op, const double *dot, double *alpha, double *beta, double *coef) {
    if (op == SOP_ALPHA) { *alpha = *beta * *dot; *coef = -1 * *alpha; }
    else if (op == SOP_W) { *coef = *alpha - *beta * *dot; }
    else { *beta = 1.0 / *dot; }
}
// partials of sum (b-csum)^2 and b.lambda over the m constraints (primalInfeasibility, LORADSCalDualObj) when no
// constraint-value kernel could produce them on the way (cones that share constraints)
__global__ __launch_bounds__(TPB) void k_eval_part(int m, const double *__restrict__ b, const double *__restrict__ csum,
                                                   const double *__restrict__ lambda, double *__restrict__ part_v,
                                                   double *__restrict__ part_d, Guard g) {
    __shared__ double sh[8];
    const bool live = !blocked(g);
    double vd[2] = {0.0, 0.0};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < m; i += gridDim.x * TPB) {
        const double t = b[i] - csum[i];
        vd[0] += t * t;
        vd[1] += b[i] * lambda[i];
    }
    block_sum_n<2>(vd, sh);
    if (live && threadIdx.x == 0) { part_v[blockIdx.x] = vd[0]; part_d[blockIdx.x] = vd[1]; }
}
// cv_i = w_i = sum_k a_k T[e_k] and vec[g_i] = w_i for a cone that sees every constraint (k_cv with CV_SET, scale 1),
// plus the partials of sum (b - w)^2 and b.lambda over the rows this workgroup owns
__global__ __launch_bounds__(TPB) void k_cv_res(int nrow, const int *__restrict__ a_ptr, const int *__restrict__ a_e,
                                                const double *__restrict__ a_val, const double *__restrict__ T,
                                                double *__restrict__ cv, const int *__restrict__ row_idx, double *__restrict__ vec,
                                                const double *__restrict__ b, const double *__restrict__ lambda,
                                                double *__restrict__ part_v, double *__restrict__ part_d, Guard g) {
    __shared__ double sh[8];
    const bool live = !blocked(g);
    const int lane = threadIdx.x & 7, per = TPB / 8;
    double vd[2] = {0.0, 0.0};
    for (int i0 = blockIdx.x * per; i0 < nrow; i0 += gridDim.x * per) {
        const int i = i0 + threadIdx.x / 8;
        const bool act = i < nrow;
        double s = 0.0;
        if (act)
            for (int t = a_ptr[i] + lane; t < a_ptr[i + 1]; t += 8) s += a_val[t] * T[a_e[t]];
        s = group_sum<8>(s);
        if (act && lane == 0) {
            const int gi = row_idx[i];
            const double w = s * 1.0, bi = b[gi], t = bi - w;
            if (live) { vec[gi] = w; cv[i] = s; }
            vd[0] += t * t;
            vd[1] += bi * lambda[gi];
        }
    }
    block_sum_n<2>(vd, sh);
    if (live && threadIdx.x == 0) { part_v[blockIdx.x] = vd[0]; part_d[blockIdx.x] = vd[1]; }
}
// one workgroup: out[0] = sum (b-csum)^2, out[1] = b.lambda from their partials, out[2] = <C, R R^T> (one cone, one rank)
__global__ __launch_bounds__(TPB) void k_eval_final(const double *__restrict__ part_v, const double *__restrict__ part_d, int np,
                                                    double *out, Guard g, const double *__restrict__ obj_part, int nobj) {
    __shared__ double sh[4];
    const bool live = !blocked(g);
    const double o = obj_part ? sum_partials(obj_part, nobj, sh) : 0.0;
    const double v = sum_partials(part_v, np, sh), d = sum_partials(part_d, np, sh);
    if (live && threadIdx.x == 0) { out[0] = v; out[1] = d; if (obj_part) out[2] = o; }
}
// phase-1 step with the line-search result: y_head = -Grad (setAsNegGrad), R += tau D (ALMupdateVar) and
// constrValSum += tau q1 + tau^2 q2 (lorads_alm.c:583-598,619-648,1122-1124) in one pass
__global__ __launch_bounds__(TPB) void k_alm_update(size_t len, double tau, const double *__restrict__ G,
                                                    const double *__restrict__ D, double *__restrict__ y, double *__restrict__ R,
                                                    int m, const double *__restrict__ q1, const double *__restrict__ q2,
                                                    double *__restrict__ csum) {
    const size_t gid = (size_t)blockIdx.x * TPB + threadIdx.x, stride = (size_t)gridDim.x * TPB;
    const size_t n2 = len / 2;
    for (size_t i = gid; i < n2; i += stride) {
        const double2 gg = ((const double2 *)G)[i], d = ((const double2 *)D)[i];
        double2 rr = ((double2 *)R)[i], yy;
        yy.x = -1.0 * gg.x; yy.y = -1.0 * gg.y;
        rr.x += tau * d.x; rr.y += tau * d.y;
        ((double2 *)y)[i] = yy;
        ((double2 *)R)[i] = rr;
    }
    if ((len & 1) && gid == 0) {
        y[len - 1] = -1.0 * G[len - 1];
        R[len - 1] += tau * D[len - 1];
    }
    for (size_t i = gid; i < (size_t)m; i += stride) {
        const double cs = csum[i] + tau * q1[i];
        csum[i] = cs + (tau * tau) * q2[i];
    }
}
// one workgroup closes the inner iteration: lagNormSq, beta of the new history pair, primal residual and b.lambda,
// all from partials
__global__ __launch_bounds__(TPB) void k_alm_tail(const double *__restrict__ lag_part, int nlag, double *lag_out,
                                                  const double *__restrict__ ys_part, int nys, double *beta_out,
                                                  const double *__restrict__ part_v, const double *__restrict__ part_d, int np,
                                                  double *out) {
    __shared__ double sh[4];
    const double lag = sum_partials(lag_part, nlag, sh);
    const double ys = sum_partials(ys_part, nys, sh);
    const double v = sum_partials(part_v, np, sh), d = sum_partials(part_d, np, sh);
    if (threadIdx.x == 0) { *lag_out = 1.0 * lag; *beta_out = 1.0 / ys; out[0] = v; out[1] = d; }
}
// lambda += rho b - rho csum
__global__ void k_dual_update(int m, double rho, const double *__restrict__ b, const double *__restrict__ csum,
                              double *__restrict__ lambda) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { double l = lambda[i] + rho * b[i]; lambda[i] = l + (-rho) * csum[i]; }
}
// csum += tau q1 + tau^2 q2
__global__ void k_csum_step(int m, double tau, const double *__restrict__ q1, const double *__restrict__ q2,
                            double *__restrict__ csum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { double c = csum[i] + tau * q1[i]; csum[i] = c + (tau * tau) * q2[i]; }
}
// Line search sums (lorads_alm.c:164-172).  With q0 = (b - csum) + lambda / rho the five dots are
//   ||q2||^2, q1.q2, ||q1||^2, q0.q2, q0.q1;  the rho-free pieces are accumulated as SEVEN partial sums per workgroup
//   {q2.q2, q1.q2, q1.q1, (b-csum).q2, lambda.q2, (b-csum).q1, lambda.q1}  (slot k at part + k * MAXPART)
// either by the kernel that produces q1, q2 (k_cv_rd) or by k_linesearch_part; k_linesearch adds them up.
constexpr int LS_NSUM = 7;
__device__ __forceinline__ void ls_accumulate(double (&s)[LS_NSUM], double a1, double a2, double bc, double lam) {
    s[0] += a2 * a2; s[1] += a1 * a2; s[2] += a1 * a1; s[3] += bc * a2; s[4] += lam * a2; s[5] += bc * a1; s[6] += lam * a1;
}
__global__ __launch_bounds__(TPB) void k_cv_rd(int nrow, const int *__restrict__ a_ptr, const int *__restrict__ a_e,
                                               const double *__restrict__ a_val, const double *__restrict__ T1,
                                               const double *__restrict__ T2, double *__restrict__ cv,
                                               const int *__restrict__ row_idx, double *__restrict__ vec1,
                                               double *__restrict__ vec2, const double *__restrict__ b,
                                               const double *__restrict__ csum, const double *__restrict__ lambda,
                                               double *__restrict__ part) {
    __shared__ double sh[4 * LS_NSUM];
    const int lane = threadIdx.x & 7, per = TPB / 8;
    double acc[LS_NSUM] = {0, 0, 0, 0, 0, 0, 0};
    for (int i0 = blockIdx.x * per; i0 < nrow; i0 += gridDim.x * per) {
        const int i = i0 + threadIdx.x / 8;
        const bool act = i < nrow;
        double s1 = 0.0, s2 = 0.0;
        if (act)
            for (int t = a_ptr[i] + lane; t < a_ptr[i + 1]; t += 8) {
                const int e = a_e[t];
                const double a = a_val[t];
                s1 += a * T1[e];
                s2 += a * T2[e];
            }
        s1 = group_sum<8>(s1);
        s2 = group_sum<8>(s2);
        if (act && lane == 0) {
            const int gi = row_idx[i];
            const double a1 = s1 * 2.0, a2 = s2 * 1.0;
            vec1[gi] = a1;
            vec2[gi] = a2;
            cv[i] = s2;
            ls_accumulate(acc, a1, a2, b[gi] - csum[gi], lambda[gi]);
        }
    }
    block_sum_n<LS_NSUM>(acc, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < LS_NSUM; ++k) part[(size_t)k * MAXPART + blockIdx.x] = acc[k];
}
__global__ __launch_bounds__(TPB) void k_linesearch_part(int m, const double *__restrict__ b, const double *__restrict__ csum,
                                                         const double *__restrict__ lambda, const double *__restrict__ q1,
                                                         const double *__restrict__ q2, double *__restrict__ part) {
    __shared__ double sh[4 * LS_NSUM];
    double s[LS_NSUM] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < m; i += gridDim.x * TPB) ls_accumulate(s, q1[i], q2[i], b[i] - csum[i], lambda[i]);
    block_sum_n<LS_NSUM>(s, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < LS_NSUM; ++k) part[(size_t)k * MAXPART + blockIdx.x] = s[k];
}
// one workgroup: out[0..4] = the five dots, out[5], out[6] = p1, p2 -- taken from the q12 tail or, when the objective
// partials are handed over (fused step), summed here and stored in both places
__global__ __launch_bounds__(TPB) void k_linesearch(int m, double rinv, const double *__restrict__ part, int np, double *q12,
                                                    double *__restrict__ out, const double *__restrict__ part1,
                                                    const double *__restrict__ part2, int npart) {
    __shared__ double sh[4];
    if (part1) {
        const double t1 = sum_partials(part1, npart, sh), t2 = sum_partials(part2, npart, sh);
        if (threadIdx.x == 0) {
            out[5] = q12[2 * (size_t)m] = 2.0 * t1;
            out[6] = q12[2 * (size_t)m + 1] = 1.0 * t2;
        }
    } else if (threadIdx.x == 0) {
        out[5] = q12[2 * (size_t)m];
        out[6] = q12[2 * (size_t)m + 1];
    }
    double s[LS_NSUM];
    for (int k = 0; k < LS_NSUM; ++k) s[k] = sum_partials(part + (size_t)k * MAXPART, np, sh);
    if (threadIdx.x == 0) {
        out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
        out[3] = s[3] + rinv * s[4];
        out[4] = s[5] + rinv * s[6];
    }
}

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
    int *erow = nullptr, *ecol = nullptr;
    int *e_ptr = nullptr, *e_con = nullptr; // entry -> (local constraint, value)
    double *e_val = nullptr;
    int *adj_ptr = nullptr, *adj_col = nullptr, *adj_e = nullptr; // row -> (neighbour, entry)
    double *S = nullptr;      // values on the pattern
    double *cbase = nullptr;  // C on the pattern (union pattern only)
    void release() {
        hipFree(erow); hipFree(ecol); hipFree(e_ptr); hipFree(e_con); hipFree(e_val);
        hipFree(adj_ptr); hipFree(adj_col); hipFree(adj_e); hipFree(S); hipFree(cbase);
    }
};

struct Block {
    int n = 0, r = 0, nrow = 0, na = 0, nc = 0;
    size_t off = 0;           // offset of this cone in the flat factor arrays
    int *row_idx = nullptr;
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
    int ksplit = 1;
    bool t_uv_valid = false;  // B.T currently holds the pair dots of (U,V) (symmetric in the pair)
    double *T2 = nullptr;     // second pair-dot buffer (evaluation on R) so that T(U,V) survives it
    bool diag_only = false;   // every A_i is a single diagonal entry (Max-Cut)
    double *gdiag = nullptr;
    bool use_cw = false;      // operator = k_cw (constraint values from the factors) + k_spmm<CW> (slot coefficient a w_i)
    int *cadj_ptr = nullptr, *cadj_col = nullptr, *cadj_con = nullptr; // row -> (neighbour, compact constraint, a)
    int *ca_row = nullptr, *ca_col = nullptr; // (row, col) of every constraint entry, in constraint-CSR order
    double *ca_val = nullptr;                 // ... and its coefficient; all three padded to ca_ell per constraint when ca_ell > 0
    int ca_ell = 0;
    double *cadj_a = nullptr;
    double *w_uv = nullptr, *w_op = nullptr; // A(sym(U V^T)) kept for re-use (valid <=> t_uv_valid); operator scratch
    bool is_lp = false;       // the LP block: generic diagonal cone everywhere except the ADMM update (k_lp_sweep)
    int lp_nlev = 0;
    int *lp_lvl_ptr = nullptr, *lp_lvl_cols = nullptr, *lp_ptr = nullptr, *lp_grow = nullptr;
    double *lp_a = nullptr, *lp_nrm2sq = nullptr, *lp_cobj = nullptr, *lp_cv = nullptr;
    bool entry_only = false;  // every A_i is a single (off-)diagonal entry (matrix completion): k_op_entry
    double *gentry = nullptr; // sum of a_i^2 per A-pattern entry
    int cg_iter_last = 0;     // lorads_cg_linsys.iter survives an immediate exit (lorads_cgs.c:157-160,173)
    int spec[2] = {1, 1};     // speculated CG iterations of the U- and V-solve
    double bytes_mv = 0, bytes_cg = 0;
};

struct Ring {
    double *s = nullptr, *y = nullptr;
};

} // namespace

struct lorads_hip_ctx {
    int m = 0, nb = 0, L = 2;
    double b_nrm1 = 0;
    hipStream_t stream = nullptr;
    std::vector<Block> blk;
    Block merged;             // all cones as ONE block-diagonal cone (see build_merged); valid when has_merged
    bool has_merged = false;
    std::vector<int> seg_row0_h;              // padded first row of every cone in the merged cone (+ end)
    int *seg_row0 = nullptr, *seg_vt0 = nullptr, *seg_vt_seg = nullptr;
    long long *seg_vt_e0 = nullptr;
    int seg_nvt = 0;
    int *phase_done = nullptr;                // [2]
    int spec_b[2] = {1, 1};                   // speculated lockstep iterations of the U and V phase
    bool merged_ok = false;   // structure allows it (separable constraints, no dense C); ranks decide has_merged
    size_t all_elem = 0;
    double *R = nullptr, *U = nullptr, *V = nullptr, *G = nullptr;   // flat factors
    double *cr = nullptr, *cp = nullptr, *cQ = nullptr, *rhs = nullptr; // flat CG vectors
    double *Dtmp = nullptr;
    double *cstage = nullptr; // m+2: [local constrValSum | objective part | miss flag] on its way through the all-reduce
    double *b = nullptr, *lambda = nullptr, *csum = nullptr, *q12 = nullptr; // csum: m+2, q12: 2m+2
    double *part = nullptr;   // NSLOT x MAXPART partial sums
    int ls_np = 0;            // line-search partials (slots 10..16) currently valid for q1, q2: how many per sum
    char *ctrl = nullptr, *h_ctrl = nullptr; // [64 scalars | CG states] device + pinned mirror: ONE readback copy
    char *h_ctrl_dev = nullptr;              // device address of the pinned mirror (k_publish writes it directly)
    unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr, pub_seq = 0; // published sequence number
    bool use_publish = true;
    double *scal = nullptr;   // 64 device scalars
    CGState *st = nullptr;    // one per (cone, half)
    CGState *h_st = nullptr;  // pinned mirror
    double *h_scal = nullptr; // pinned mirror of scalars
    std::vector<Ring> ring;
    double *ring_ab = nullptr; // [2L] alpha,beta per node
    int head = 0;
    lorads_hip_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    bool ar_stream_ordered = false; // the hook enqueues on our stream (RCCL): no host sync around it
    // profiling
    int prof = 0, prof_every = 8;
    long n_matvec = 0, n_cg_it = 0, n_solves = 0, n_samp = 0, n_samp_spmm = 0, n_resume = 0;
    double ms_samp = 0, ms_samp_spmm = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend_mv, pend_sp;
    std::vector<hipEvent_t> ev_pool; // pre-created (hipEventCreate is too slow for the timed region)
    size_t ev_next = 0;
};

namespace {

inline int nblocks_for(size_t items, int per_block) { return (int)((items + per_block - 1) / per_block); }
// grid of the L-BFGS stage kernels: every workgroup re-sums the previous stage's partials, so keep them few
inline int grid_lbfgs(size_t len);
inline int grid1d(size_t len) {
    size_t g = (len + TPB - 1) / TPB;
    return (int)std::min<size_t>(std::max<size_t>(g, 1), 2048);
}
inline int grid_lbfgs(size_t len) {
    static const int cap = getenv("LORADS_LBFGS_GRID") ? atoi(getenv("LORADS_LBFGS_GRID")) : 512;
    return std::min(grid1d(len), cap);
}
// lanes per row/entry: 16-byte loads (2 columns per lane) when r is even and fits 8 lanes x 8 steps x 2
inline bool use_v2(int r) { return (r % 2) == 0 && r <= 128; }
inline int lg_for(int r) { return use_v2(r) ? 8 : (r <= 64 ? 8 : (r <= 256 ? 32 : 64)); }
const Guard NOGUARD{nullptr, nullptr};

double *part_slot(lorads_hip_ctx *c, int k) { return c->part + (size_t)k * MAXPART; }

// ---- pattern construction (host, once per context)
struct HostPattern {
    std::vector<int> erow, ecol;
    std::vector<int> e_ptr, e_con;
    std::vector<double> e_val;
    std::vector<int> adj_ptr, adj_col, adj_e;
};

void unique_positions(const std::vector<std::pair<int, int>> &pos, std::vector<std::pair<int, int>> &uniq,
                      std::vector<int> &index) {
    std::vector<int> order(pos.size());
    for (size_t i = 0; i < pos.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (pos[a].second != pos[b].second) return pos[a].second < pos[b].second; // by column, then row
        if (pos[a].first != pos[b].first) return pos[a].first < pos[b].first;
        return a < b;
    });
    uniq.clear();
    index.assign(pos.size(), -1);
    for (size_t k = 0; k < order.size(); ++k) {
        int i = order[k];
        if (uniq.empty() || uniq.back() != pos[i]) uniq.push_back(pos[i]);
        index[i] = (int)uniq.size() - 1;
    }
}

void build_adjacency(int n, const std::vector<std::pair<int, int>> &uniq, HostPattern &hp) {
    hp.erow.resize(uniq.size());
    hp.ecol.resize(uniq.size());
    std::vector<int> deg(n + 1, 0);
    for (size_t e = 0; e < uniq.size(); ++e) {
        hp.erow[e] = uniq[e].first;
        hp.ecol[e] = uniq[e].second;
        deg[uniq[e].first + 1]++;
        if (uniq[e].first != uniq[e].second) deg[uniq[e].second + 1]++;
    }
    hp.adj_ptr.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) hp.adj_ptr[i + 1] = hp.adj_ptr[i] + deg[i + 1];
    hp.adj_col.resize(hp.adj_ptr[n]);
    hp.adj_e.resize(hp.adj_ptr[n]);
    std::vector<int> fill(hp.adj_ptr.begin(), hp.adj_ptr.end() - 1);
    for (size_t e = 0; e < uniq.size(); ++e) {
        int p = uniq[e].first, q = uniq[e].second;
        hp.adj_col[fill[p]] = q; hp.adj_e[fill[p]] = (int)e; fill[p]++;
        if (p != q) { hp.adj_col[fill[q]] = p; hp.adj_e[fill[q]] = (int)e; fill[q]++; }
    }
    // neighbours in ascending row order: fixed summation order and better locality
    for (int i = 0; i < n; ++i) {
        int s = hp.adj_ptr[i], t = hp.adj_ptr[i + 1];
        std::vector<std::pair<int, int>> tmp(t - s);
        for (int k = s; k < t; ++k) tmp[k - s] = {hp.adj_col[k], hp.adj_e[k]};
        std::sort(tmp.begin(), tmp.end());
        for (int k = s; k < t; ++k) { hp.adj_col[k] = tmp[k - s].first; hp.adj_e[k] = tmp[k - s].second; }
    }
}

// transpose of the constraint CSR over pattern entries
void build_transpose(int ne, int nrow, const int *a_ptr, const std::vector<int> &a_e, const double *a_val, HostPattern &hp) {
    hp.e_ptr.assign(ne + 1, 0);
    int na = a_ptr[nrow];
    for (int t = 0; t < na; ++t) hp.e_ptr[a_e[t] + 1]++;
    for (int e = 0; e < ne; ++e) hp.e_ptr[e + 1] += hp.e_ptr[e];
    hp.e_con.resize(na);
    hp.e_val.resize(na);
    std::vector<int> fill(hp.e_ptr.begin(), hp.e_ptr.end() - 1);
    for (int i = 0; i < nrow; ++i)
        for (int t = a_ptr[i]; t < a_ptr[i + 1]; ++t) {
            int e = a_e[t];
            hp.e_con[fill[e]] = i;
            hp.e_val[fill[e]] = a_val[t];
            fill[e]++;
        }
}

int upload_pattern(Pattern &P, const HostPattern &hp, const std::vector<double> *cbase) {
    P.ne = (int)hp.erow.size();
    if (upload(&P.erow, hp.erow) || upload(&P.ecol, hp.ecol) || upload(&P.e_ptr, hp.e_ptr) || upload(&P.e_con, hp.e_con) ||
        upload(&P.e_val, hp.e_val) || upload(&P.adj_ptr, hp.adj_ptr) || upload(&P.adj_col, hp.adj_col) ||
        upload(&P.adj_e, hp.adj_e))
        return 1;
    if (dalloc(&P.S, (size_t)P.ne)) return 1;
    if (cbase && upload(&P.cbase, *cbase)) return 1;
    return 0;
}

// G = A A^T over pattern entries, when sum_i nnz_i^2 stays small
int build_gram(Block &B, const lorads_hip_block &hb, const std::vector<int> &a_e, int ne) {
    double tot = 0;
    for (int i = 0; i < hb.nrow; ++i) { double k = hb.a_ptr[i + 1] - hb.a_ptr[i]; tot += k * k; }
    if (tot > std::max(64.0 * B.na, 1.0e6) || tot > 2.0e8) return 0;
    std::vector<std::pair<uint64_t, double>> tri;
    tri.reserve((size_t)tot);
    for (int i = 0; i < hb.nrow; ++i)
        for (int t1 = hb.a_ptr[i]; t1 < hb.a_ptr[i + 1]; ++t1)
            for (int t2 = hb.a_ptr[i]; t2 < hb.a_ptr[i + 1]; ++t2)
                tri.push_back({((uint64_t)a_e[t1] << 32) | (uint32_t)a_e[t2], hb.a_val[t1] * hb.a_val[t2]});
    std::stable_sort(tri.begin(), tri.end(), [](const std::pair<uint64_t, double> &x, const std::pair<uint64_t, double> &y) {
        return x.first < y.first;
    });
    std::vector<int> g_ptr(ne + 1, 0), g_col;
    std::vector<double> g_val;
    for (size_t k = 0; k < tri.size(); ++k) {
        if (k > 0 && tri[k].first == tri[k - 1].first) { g_val.back() += tri[k].second; continue; }
        g_col.push_back((int)(tri[k].first & 0xffffffffu));
        g_val.push_back(tri[k].second);
        g_ptr[(int)(tri[k].first >> 32) + 1]++;
    }
    for (int e = 0; e < ne; ++e) g_ptr[e + 1] += g_ptr[e];
    if (upload(&B.g_ptr, g_ptr) || upload(&B.g_col, g_col) || upload(&B.g_val, g_val)) return 1;
    B.has_gram = true;
    return 0;
}

int build_block(lorads_hip_ctx *c, Block &B, const lorads_hip_block &hb) {
    B.n = hb.n; B.r = hb.rank; B.nrow = hb.nrow; B.na = hb.a_ptr[hb.nrow]; B.nc = hb.c_nnz;
    if (B.r > 512) return fail_msg("rank > 512 is not supported by the row kernels");
    if (nblocks_for((size_t)B.n, TPB / lg_for(B.r)) > MAXPART)
        return fail_msg("cone dimension too large for the partial-sum slots of this build");
    std::vector<std::pair<int, int>> posA(B.na), posU;
    for (int t = 0; t < B.na; ++t) {
        if (hb.a_row[t] < hb.a_col[t] || hb.a_col[t] < 0 || hb.a_row[t] >= hb.n) return fail_msg("A entry out of range / not lower-triangular");
        posA[t] = {hb.a_row[t], hb.a_col[t]};
    }
    for (int t = 0; t < B.nc; ++t)
        if (hb.c_row[t] < hb.c_col[t] || hb.c_col[t] < 0 || hb.c_row[t] >= hb.n) return fail_msg("C entry out of range / not lower-triangular");
    for (int i = 0; i < hb.nrow; ++i)
        if (hb.row_idx[i] < 0 || hb.row_idx[i] >= c->m) return fail_msg("constraint index out of range");
    // A-pattern
    std::vector<std::pair<int, int>> uniqA;
    std::vector<int> a_e;
    unique_positions(posA, uniqA, a_e);
    HostPattern hpA;
    build_adjacency(B.n, uniqA, hpA);
    build_transpose((int)uniqA.size(), hb.nrow, hb.a_ptr, a_e, hb.a_val, hpA);
    if (upload_pattern(B.pa, hpA, nullptr)) return 1;
    // dense objective (the reference's rule for a dense coefficient, lorads_sdp_data.c:818-821)
    B.dense_c = !hb.is_lp && (double)B.nc > 0.1 * (double)((int64_t)B.n * (B.n + 1) / 2) && B.r <= 128;
    if (B.dense_c) {
        B.npad = (B.n + 63) / 64 * 64;
        // split K over workgroups until the grid has >= 512 of them (K range a multiple of 32)
        B.ksplit = 1;
        while (B.ksplit < 16 && (B.npad / 64) * B.ksplit < 512 && (B.npad / (2 * B.ksplit)) % 32 == 0) B.ksplit *= 2;
        std::vector<double> cf((size_t)B.npad * B.npad, 0.0);
        for (int t = 0; t < B.nc; ++t) {
            cf[(size_t)hb.c_row[t] * B.npad + hb.c_col[t]] += hb.c_val[t];
            if (hb.c_row[t] != hb.c_col[t]) cf[(size_t)hb.c_col[t] * B.npad + hb.c_row[t]] += hb.c_val[t];
        }
        if (upload(&B.Cfull, cf)) return 1;
    }
    // union pattern C u A (A only when C is dense: the dense part is added by k_dense_cx)
    posU = posA;
    for (int t = 0; t < B.nc && !B.dense_c; ++t) posU.push_back({hb.c_row[t], hb.c_col[t]});
    std::vector<std::pair<int, int>> uniqU;
    std::vector<int> u_idx;
    unique_positions(posU, uniqU, u_idx);
    HostPattern hpU;
    build_adjacency(B.n, uniqU, hpU);
    std::vector<int> a_eU(u_idx.begin(), u_idx.begin() + B.na);
    build_transpose((int)uniqU.size(), hb.nrow, hb.a_ptr, a_eU, hb.a_val, hpU);
    std::vector<double> cbase(uniqU.size(), 0.0);
    for (int t = 0; t < B.nc && !B.dense_c; ++t) cbase[u_idx[B.na + t]] += hb.c_val[t];
    if (upload_pattern(B.pu, hpU, &cbase)) return 1;
    // constraint CSR + misc
    std::vector<int> v_rowidx(hb.row_idx, hb.row_idx + hb.nrow), v_aptr(hb.a_ptr, hb.a_ptr + hb.nrow + 1);
    std::vector<double> v_aval(hb.a_val, hb.a_val + B.na), v_cval(hb.c_val, hb.c_val + B.nc);
    std::vector<int> v_crow(hb.c_row, hb.c_row + B.nc), v_ccol(hb.c_col, hb.c_col + B.nc);
    if (upload(&B.row_idx, v_rowidx) || upload(&B.a_ptr, v_aptr) || upload(&B.a_e, a_e) || upload(&B.a_val, v_aval) ||
        upload(&B.c_row, v_crow) || upload(&B.c_col, v_ccol) || upload(&B.c_val, v_cval))
        return 1;
    if (dalloc(&B.T, (size_t)B.pa.ne) || dalloc(&B.T2, (size_t)B.pa.ne) || dalloc(&B.cv, (size_t)B.nrow) || dalloc(&B.wtmp, (size_t)B.nrow)) return 1;
    HC(hipMemset(B.cv, 0, sizeof(double) * (size_t)std::max(B.nrow, 1)));
    // Max-Cut fast path
    bool diag = B.nrow > 0;
    std::vector<double> gd(B.n, 0.0);
    for (int i = 0; i < hb.nrow && diag; ++i) {
        if (hb.a_ptr[i + 1] - hb.a_ptr[i] != 1) { diag = false; break; }
        int t = hb.a_ptr[i];
        if (hb.a_row[t] != hb.a_col[t]) { diag = false; break; }
        gd[hb.a_row[t]] += hb.a_val[t] * hb.a_val[t];
    }
    B.diag_only = diag;
    if (diag && upload(&B.gdiag, gd)) return 1;
    bool single_entry = !diag && B.nrow > 0 && !getenv("LORADS_NO_OP_ENTRY");
    for (int i = 0; i < hb.nrow && single_entry; ++i) single_entry = hb.a_ptr[i + 1] - hb.a_ptr[i] == 1;
    if (single_entry) {
        std::vector<double> gev((size_t)B.pa.ne, 0.0);
        for (int i = 0; i < hb.nrow; ++i) gev[a_e[hb.a_ptr[i]]] += hb.a_val[hb.a_ptr[i]] * hb.a_val[hb.a_ptr[i]];
        if (upload(&B.gentry, gev)) return 1;
        B.entry_only = true;
    }
    if (!diag && build_gram(B, hb, a_e, B.pa.ne)) return 1;
    // Constraint-wise operator (k_cw + k_spmm<CW>): worthwhile when there are enough constraints to fill the device
    // with one wavefront each and none of them is so large that a single wavefront would crawl through it.
    {
        int maxsz = 0;
        for (int i = 0; i < hb.nrow; ++i) maxsz = std::max(maxsz, hb.a_ptr[i + 1] - hb.a_ptr[i]);
        const char *force = getenv("LORADS_OP_CW"); // "1" force on, "0" force off (tests / comparisons)
        bool want = !diag && !B.entry_only && hb.nrow >= 256 && maxsz <= 512;
        if (force && !diag && !B.entry_only && hb.nrow > 0) want = force[0] == '1';
        if (want) {
            std::vector<int> deg(B.n + 1, 0);
            for (int t = 0; t < B.na; ++t) {
                deg[hb.a_row[t] + 1]++;
                if (hb.a_row[t] != hb.a_col[t]) deg[hb.a_col[t] + 1]++;
            }
            std::vector<int> ptr(B.n + 1, 0);
            for (int i = 0; i < B.n; ++i) ptr[i + 1] = ptr[i] + deg[i + 1];
            std::vector<int> col(ptr[B.n]), con(ptr[B.n]), fill(ptr.begin(), ptr.end() - 1);
            std::vector<double> av(ptr[B.n]);
            for (int i = 0; i < hb.nrow; ++i)
                for (int t = hb.a_ptr[i]; t < hb.a_ptr[i + 1]; ++t) {
                    const int p = hb.a_row[t], q = hb.a_col[t];
                    col[fill[p]] = q; con[fill[p]] = i; av[fill[p]] = hb.a_val[t]; fill[p]++;
                    if (p != q) { col[fill[q]] = p; con[fill[q]] = i; av[fill[q]] = hb.a_val[t]; fill[q]++; }
                }
            for (int i = 0; i < B.n; ++i) { // neighbours in ascending (row, constraint) order: fixed summation order
                const int s0 = ptr[i], s1 = ptr[i + 1];
                std::vector<std::tuple<int, int, double>> tmp(s1 - s0);
                for (int k = s0; k < s1; ++k) tmp[k - s0] = std::make_tuple(col[k], con[k], av[k]);
                std::sort(tmp.begin(), tmp.end());
                for (int k = s0; k < s1; ++k) { col[k] = std::get<0>(tmp[k - s0]); con[k] = std::get<1>(tmp[k - s0]); av[k] = std::get<2>(tmp[k - s0]); }
            }
            std::vector<int> car(hb.a_row, hb.a_row + B.na), cac(hb.a_col, hb.a_col + B.na);
            std::vector<double> cav(hb.a_val, hb.a_val + B.na);
            if (maxsz <= 32 && (double)hb.nrow * maxsz <= 1.5 * (double)B.na && !getenv("LORADS_NO_ELL")) {
                // constraints of (nearly) equal size: fixed-width layout, padding = (row 0, row 0, 0.0)
                B.ca_ell = maxsz;
                car.assign((size_t)hb.nrow * maxsz, 0); cac.assign((size_t)hb.nrow * maxsz, 0); cav.assign((size_t)hb.nrow * maxsz, 0.0);
                for (int i = 0; i < hb.nrow; ++i)
                    for (int t = hb.a_ptr[i]; t < hb.a_ptr[i + 1]; ++t) {
                        const size_t w = (size_t)i * maxsz + (size_t)(t - hb.a_ptr[i]);
                        car[w] = hb.a_row[t]; cac[w] = hb.a_col[t]; cav[w] = hb.a_val[t];
                    }
            }
            if (upload(&B.ca_row, car) || upload(&B.ca_col, cac) || upload(&B.ca_val, cav) ||
                upload(&B.cadj_ptr, ptr) || upload(&B.cadj_col, col) || upload(&B.cadj_con, con) || upload(&B.cadj_a, av) ||
                dalloc(&B.w_uv, (size_t)B.nrow) || dalloc(&B.w_op, (size_t)B.nrow))
                return 1;
            B.use_cw = true;
        }
    }
    if (hb.is_lp) { // column-wise image + level schedule of the LP block
        for (int t = 0; t < B.na; ++t)
            if (hb.a_row[t] != hb.a_col[t]) return fail_msg("LP block: off-diagonal entry");
        for (int t = 0; t < B.nc; ++t)
            if (hb.c_row[t] != hb.c_col[t]) return fail_msg("LP block: off-diagonal objective entry");
        const int n = B.n;
        std::vector<int> ptr(n + 1, 0), grow(B.na), fill(n, 0);
        std::vector<double> av(B.na), nrm(n, 0.0), cobj(n, 0.0);
        for (int t = 0; t < B.na; ++t) ptr[hb.a_row[t] + 1]++;
        for (int j = 0; j < n; ++j) ptr[j + 1] += ptr[j];
        for (int i = 0; i < hb.nrow; ++i)
            for (int t = hb.a_ptr[i]; t < hb.a_ptr[i + 1]; ++t) {
                const int col = hb.a_row[t], w = ptr[col] + fill[col]++;
                grow[w] = hb.row_idx[i];
                av[w] = hb.a_val[t];
            }
        for (int j = 0; j < n; ++j) {
            double nn = 0.0;
            for (int t = ptr[j]; t < ptr[j + 1]; ++t) nn += av[t] * av[t];
            const double nr = std::sqrt(nn); // nrm2, then squared (data/lorads_lp_conic.c:112-113)
            nrm[j] = nr * nr;
        }
        for (int t = 0; t < B.nc; ++t) cobj[hb.c_row[t]] += hb.c_val[t];
        // level(j) = 1 + max level of the earlier columns that share a constraint row with j
        std::vector<int> row_lvl((size_t)std::max(c->m, 1), 0), lvl(n, 0);
        int nlev = 0;
        for (int j = 0; j < n; ++j) {
            int l = 0;
            for (int t = ptr[j]; t < ptr[j + 1]; ++t) l = std::max(l, row_lvl[grow[t]]);
            lvl[j] = l; // 0-based level
            for (int t = ptr[j]; t < ptr[j + 1]; ++t) row_lvl[grow[t]] = l + 1;
            nlev = std::max(nlev, l + 1);
        }
        std::vector<int> lptr(nlev + 1, 0), lcols(n);
        for (int j = 0; j < n; ++j) lptr[lvl[j] + 1]++;
        for (int l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];
        std::vector<int> lf(lptr.begin(), lptr.end() - 1);
        for (int j = 0; j < n; ++j) lcols[lf[lvl[j]]++] = j;
        if (upload(&B.lp_ptr, ptr) || upload(&B.lp_grow, grow) || upload(&B.lp_a, av) || upload(&B.lp_nrm2sq, nrm) ||
            upload(&B.lp_cobj, cobj) || upload(&B.lp_lvl_ptr, lptr) || upload(&B.lp_lvl_cols, lcols) ||
            dalloc(&B.lp_cv, (size_t)B.na))
            return 1;
        HC(hipMemset(B.lp_cv, 0, sizeof(double) * (size_t)std::max(B.na, 1)));
        B.is_lp = true;
        B.lp_nlev = nlev;
    }
    return 0;
}

void block_bytes(Block &B) {
    double F = 8.0 * B.n * B.r;
    B.bytes_mv = 4 * F + 32.0 * B.na + 16.0 * B.nrow;
    B.bytes_cg = B.bytes_mv + 9 * F;
}

// rows of a cone inside the merged cone start at multiples of 32 (= the most rows one workgroup of a row kernel owns)
inline int pad_rows(int n) { return (n + 31) & ~31; }

int alloc_factors(lorads_hip_ctx *c) {
    c->all_elem = 0;
    for (auto &B : c->blk) {
        B.off = c->all_elem;
        c->all_elem += (size_t)(c->merged_ok ? pad_rows(B.n) : B.n) * B.r; // pad rows stay zero for ever
        block_bytes(B);
    }
    size_t n = c->all_elem;
    double **arrs[] = {&c->R, &c->U, &c->V, &c->G, &c->cr, &c->cp, &c->cQ, &c->rhs, &c->Dtmp};
    for (auto a : arrs) {
        if (dalloc(a, n)) return 1;
        HC(hipMemset(*a, 0, sizeof(double) * std::max<size_t>(n, 1)));
    }
    for (auto &B : c->blk)
        if (B.dense_c && (dalloc(&B.Wd, (size_t)B.n * B.r) || dalloc(&B.Wpart, (size_t)B.ksplit * B.n * B.r))) return 1;
    c->ring.resize(c->L);
    for (auto &nd : c->ring) {
        if (dalloc(&nd.s, n) || dalloc(&nd.y, n)) return 1;
        HC(hipMemset(nd.s, 0, sizeof(double) * std::max<size_t>(n, 1)));
        HC(hipMemset(nd.y, 0, sizeof(double) * std::max<size_t>(n, 1)));
    }
    HC(hipDeviceSynchronize());
    return 0;
}
// The cone the single-cone fast paths may run on: the only cone, or the block-diagonal union of all cones.
Block *solo(lorads_hip_ctx *c) {
    if (c->ar) return nullptr;
    if (c->nb == 1) return &c->blk[0];
    return c->has_merged ? &c->merged : nullptr;
}

// Several cones with block-separable constraints (every constraint lives in exactly one cone) and equal rank are,
// for every kernel that has no per-cone scalar, ONE cone with a block-diagonal pattern: the flat factor arrays are
// already the concatenation of the cones' n_k x r row-major factors.  Phase 1 (no per-cone scalars at all: the
// L-BFGS runs over the concatenation, lorads_alm.c:230-391) and the evaluation part of phase 2 then cost the
// launches of one cone instead of nb.  The CG solves keep their per-cone launches (per-cone alpha, beta, stopping).
int build_merged(lorads_hip_ctx *c, const lorads_hip_problem *prob) {
    c->merged_ok = c->has_merged = false;
    if (c->nb < 2 || getenv("LORADS_NO_MERGE")) return 0;
    for (int k = 0; k < c->nb; ++k)
        if (prob->blocks[k].is_lp) return 0; // the LP block has its own update; no merged view with one present
    std::vector<char> seen((size_t)std::max(c->m, 1), 0);
    size_t ntot = 0, nrow = 0, na = 0, nc = 0;
    for (int k = 0; k < c->nb; ++k) {
        const lorads_hip_block &hb = prob->blocks[k];
        if (c->blk[k].dense_c) return 0;
        for (int i = 0; i < hb.nrow; ++i) {
            if (seen[hb.row_idx[i]]) return 0; // a constraint couples two cones: the sweep order matters
            seen[hb.row_idx[i]] = 1;
        }
        ntot += pad_rows(hb.n); nrow += hb.nrow; na += hb.a_ptr[hb.nrow]; nc += hb.c_nnz;
    }
    // (a sharded context sees only its own cones' constraints: nrow < m is fine, the single-cone shortcuts that need
    // every constraint check nrow == m themselves)
    if (nblocks_for(ntot, TPB / 8) > MAXPART) return 0; // partial-sum slots (refresh_merged re-checks per rank)
    std::vector<int> row_idx, a_ptr(1, 0), a_row, a_col, c_row, c_col;
    std::vector<double> a_val, c_val;
    row_idx.reserve(nrow); a_row.reserve(na); a_col.reserve(na); a_val.reserve(na);
    c_row.reserve(nc); c_col.reserve(nc); c_val.reserve(nc);
    int roff = 0;
    for (int k = 0; k < c->nb; ++k) {
        const lorads_hip_block &hb = prob->blocks[k];
        for (int i = 0; i < hb.nrow; ++i) {
            row_idx.push_back(hb.row_idx[i]);
            for (int t = hb.a_ptr[i]; t < hb.a_ptr[i + 1]; ++t) {
                a_row.push_back(hb.a_row[t] + roff); a_col.push_back(hb.a_col[t] + roff); a_val.push_back(hb.a_val[t]);
            }
            a_ptr.push_back((int)a_row.size());
        }
        for (int t = 0; t < hb.c_nnz; ++t) {
            c_row.push_back(hb.c_row[t] + roff); c_col.push_back(hb.c_col[t] + roff); c_val.push_back(hb.c_val[t]);
        }
        c->seg_row0_h.push_back(roff);
        roff += pad_rows(hb.n);
    }
    c->seg_row0_h.push_back(roff);
    lorads_hip_block mb{};
    mb.n = (int)ntot; mb.rank = prob->blocks[0].rank; mb.nrow = (int)nrow; mb.row_idx = row_idx.data(); mb.a_ptr = a_ptr.data();
    mb.a_row = a_row.data(); mb.a_col = a_col.data(); mb.a_val = a_val.data(); mb.c_nnz = (int)nc; mb.c_row = c_row.data();
    mb.c_col = c_col.data(); mb.c_val = c_val.data();
    if (build_block(c, c->merged, mb)) return 1;
    if (c->merged.dense_c) return fail_msg("internal: merged cone classified dense");
    c->merged.off = 0;
    // one constrVal array: the cones' compact vectors are consecutive pieces of the merged one
    size_t o = 0;
    for (auto &B : c->blk) {
        hipFree(B.cv);
        B.cv = c->merged.cv + o;
        B.cv_borrowed = true;
        o += (size_t)B.nrow;
    }
    if (upload(&c->seg_row0, c->seg_row0_h) || dalloc(&c->phase_done, (size_t)2)) return 1;
    HC(hipMemset(c->phase_done, 0, 2 * sizeof(int)));
    c->merged_ok = true;
    return 0;
}
// equal ranks -> the merged view is usable
void refresh_merged(lorads_hip_ctx *c) {
    c->has_merged = false;
    if (!c->merged_ok) return;
    for (auto &B : c->blk)
        if (B.r != c->blk[0].r) return;
    if (nblocks_for((size_t)c->merged.n, TPB / lg_for(c->blk[0].r)) > MAXPART) return;
    c->merged.r = c->blk[0].r;
    c->merged.t_uv_valid = false;
    block_bytes(c->merged);
    // chunk tables of the per-cone vector kernels (depend on the rank)
    std::vector<int> vt0(1, 0), vt_seg;
    std::vector<long long> vt_e0;
    for (int k = 0; k < c->nb; ++k) {
        const long long e0 = (long long)c->seg_row0_h[k] * c->merged.r, e1 = (long long)c->seg_row0_h[k + 1] * c->merged.r;
        for (long long e = e0; e < e1; e += SEG_CH) { vt_seg.push_back(k); vt_e0.push_back(e); }
        vt0.push_back((int)vt_seg.size());
    }
    if (vt_seg.size() > (size_t)MAXPART) return;
    hipFree(c->seg_vt0); hipFree(c->seg_vt_seg); hipFree(c->seg_vt_e0);
    c->seg_vt0 = c->seg_vt_seg = nullptr; c->seg_vt_e0 = nullptr;
    if (upload(&c->seg_vt0, vt0) || upload(&c->seg_vt_seg, vt_seg) || upload(&c->seg_vt_e0, vt_e0)) return;
    c->seg_nvt = (int)vt_seg.size();
    c->spec_b[0] = c->spec_b[1] = 1;
    c->has_merged = true;
}
void free_factors(lorads_hip_ctx *c) {
    double *arrs[] = {c->R, c->U, c->V, c->G, c->cr, c->cp, c->cQ, c->rhs, c->Dtmp};
    for (auto a : arrs) hipFree(a);
    for (auto &B : c->blk) { hipFree(B.Wd); hipFree(B.Wpart); B.Wd = B.Wpart = nullptr; }
    for (auto &nd : c->ring) { hipFree(nd.s); hipFree(nd.y); }
    c->ring.clear();
}

double *mat_base(lorads_hip_ctx *c, int which) {
    switch (which) {
    case LORADS_HIP_MAT_R: return c->R;
    case LORADS_HIP_MAT_U: return c->U;
    case LORADS_HIP_MAT_V: return c->V;
    case LORADS_HIP_MAT_GRAD: return c->G;
    }
    return nullptr;
}
double *vec_base(lorads_hip_ctx *c, int which) {
    switch (which) {
    case LORADS_HIP_VEC_LAMBDA: return c->lambda;
    case LORADS_HIP_VEC_CONSTR_SUM: return c->csum;
    case LORADS_HIP_VEC_Q1: return c->q12;
    case LORADS_HIP_VEC_Q2: return c->q12 + c->m;
    }
    return nullptr;
}

int allreduce_dev(lorads_hip_ctx *c, double *buf, int count) {
    if (!c->ar) return 0;
    if (!c->ar_stream_ordered) HC(hipStreamSynchronize(c->stream));
    if (c->ar(c->ar_user, buf, count, 1)) return fail_msg("allreduce hook failed");
    return 0;
}

// ---- launch helpers
#define LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), 0, c->stream, __VA_ARGS__)

// (LG, V2, NS) for a rank: 8 lanes x 16-byte loads when r is even and <= 128, else 8/32/64 lanes x 8-byte loads
struct Shape {
    int lg, v2, ns;
};
inline Shape shape_for(int r) {
    Shape s;
    s.v2 = use_v2(r);
    s.lg = lg_for(r);
    const int w = s.v2 ? 2 : 1;
    s.ns = (r + s.lg * w - 1) / (s.lg * w);
    return s;
}
#define NS_SWITCH(LGV, V2V, ns, ...)                                     \
    switch (ns) {                                                        \
    case 1: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 4; __VA_ARGS__; } break; \
    case 5: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 5; __VA_ARGS__; } break; \
    case 6: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 6; __VA_ARGS__; } break; \
    case 7: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 7; __VA_ARGS__; } break; \
    default: { constexpr int LG_ = LGV; constexpr bool V2_ = V2V; constexpr int NS_ = 8; __VA_ARGS__; } break; \
    }
#define SHAPE_DISPATCH(sh, ...)                                   \
    do {                                                          \
        if ((sh).v2) { NS_SWITCH(8, true, (sh).ns, __VA_ARGS__) }        \
        else if ((sh).lg == 8) { NS_SWITCH(8, false, (sh).ns, __VA_ARGS__) } \
        else if ((sh).lg == 32) { NS_SWITCH(32, false, (sh).ns, __VA_ARGS__) } \
        else { NS_SWITCH(64, false, (sh).ns, __VA_ARGS__) }              \
    } while (0)

void pairdots(lorads_hip_ctx *c, const Pattern &P, const double *X, const double *Y, int r, double *T, Guard g) {
    if (P.ne == 0) return;
    const Shape sh = shape_for(r);
    const int grid = nblocks_for((size_t)P.ne, TPB / sh.lg);
    SHAPE_DISPATCH(sh, LAUNCH((k_pairdots<LG_, V2_, NS_>), grid, P.ne, P.erow, P.ecol, X, Y, r, T, g));
}
// returns the number of partials written
int spmm(lorads_hip_ctx *c, const Block &B, const Pattern &P, const double *X, int mode, const double *xin, const double *rhs,
         double rho, double *out, double *part, Guard g, const double *dense_add = nullptr) {
    const Shape sh = shape_for(B.r);
    const int grid = nblocks_for((size_t)B.n, TPB / sh.lg);
    SHAPE_DISPATCH(sh, LAUNCH((k_spmm<LG_, V2_, NS_>), grid, B.n, P.adj_ptr, P.adj_col, P.adj_e, P.S, X, B.r, mode, xin, rhs, rho,
                              out, part, g, dense_add));
    return grid;
}
// epilogue(x + sum over (neighbour, constraint) slots of a w_i V_q): the operator's SpMM with the coefficients formed
// from the constraint weights w
int spmm_cw(lorads_hip_ctx *c, const Block &B, const double *w, const double *X, int mode, const double *xin, const double *rhs,
            double *out, double *part, Guard g) {
    const Shape sh = shape_for(B.r);
    const int grid = nblocks_for((size_t)B.n, TPB / sh.lg);
    SHAPE_DISPATCH(sh, LAUNCH((k_spmm<LG_, V2_, NS_, true>), grid, B.n, B.cadj_ptr, B.cadj_col, B.cadj_con, w, X, B.r, mode, xin, rhs,
                              0.0, out, part, g, (const double *)nullptr, B.cadj_a));
    return grid;
}
// w_i = A_i(sym(X Y^T)) for every constraint of the cone + k_cv's bookkeeping, without the pair-dot array
void cw(lorads_hip_ctx *c, const Block &B, const double *X, const double *Y, double scale, double *w_out, double *cv, int mode,
        double *vec, Guard g) {
    if (B.nrow == 0) return;
    const Shape sh = shape_for(B.r);
    const int grid = nblocks_for((size_t)B.nrow, TPB / 64);
    SHAPE_DISPATCH(sh, LAUNCH((k_cw<LG_, V2_, NS_>), grid, B.nrow, B.a_ptr, B.ca_row, B.ca_val, B.ca_col, X, Y, B.r, scale,
                              w_out, cv, mode, B.row_idx, vec, g, B.ca_ell));
}
// W = C X on the matrix cores (dense objective only)
int dense_cx(lorads_hip_ctx *c, const Block &B, const double *X, double *W, Guard g) {
    const int nt = (B.r + 15) / 16, gx = B.npad / 64, ks = B.ksplit, krange = B.npad / ks;
    const size_t lds = sizeof(double) * 32 * (size_t)B.r;
    double *dst = ks == 1 ? W : B.Wpart;
#define DCX(NTV) hipLaunchKernelGGL(k_dense_cx<NTV>, dim3(gx, ks), dim3(TPB), lds, c->stream, B.n, B.npad, krange, B.Cfull, X, B.r, dst, g)
    switch (nt) {
    case 1: DCX(1); break;
    case 2: DCX(2); break;
    case 3: DCX(3); break;
    case 4: DCX(4); break;
    case 5: DCX(5); break;
    case 6: DCX(6); break;
    case 7: DCX(7); break;
    case 8: DCX(8); break;
    default: return 1;
    }
#undef DCX
    if (ks > 1) {
        const size_t len = (size_t)B.n * B.r;
        LAUNCH(k_sum_slabs, grid1d(len), len, ks, B.Wpart, W, g);
    }
    return 0;
}
int op_diag(lorads_hip_ctx *c, const Block &B, const double *V, int mode, const double *xin, const double *rhs, double *out,
            double *part, Guard g) {
    const Shape sh = shape_for(B.r);
    const int grid = nblocks_for((size_t)B.n, TPB / sh.lg);
    SHAPE_DISPATCH(sh, LAUNCH((k_op_diag<LG_, V2_, NS_>), grid, B.n, B.gdiag, V, B.r, mode, xin, rhs, out, part, g));
    return grid;
}
int op_entry(lorads_hip_ctx *c, const Block &B, const double *V, int mode, const double *xin, const double *rhs, double *out,
             double *part, Guard g) {
    const Shape sh = shape_for(B.r);
    const int grid = nblocks_for((size_t)B.n, TPB / sh.lg);
    SHAPE_DISPATCH(sh, LAUNCH((k_op_entry<LG_, V2_, NS_>), grid, B.n, B.pa.adj_ptr, B.pa.adj_col, B.pa.adj_e, B.gentry, V, B.r, mode, xin,
                              rhs, out, part, g));
    return grid;
}
int obj_partials(lorads_hip_ctx *c, const Block &B, const double *X, const double *Y, double *part, Guard g) {
    if (B.nc == 0) return 0;
    if (B.dense_c) { // <C, sym(X Y^T)> = sum_p X_p . (C Y)_p for symmetric C
        dense_cx(c, B, Y, B.Wd, g);
        const size_t len = (size_t)B.n * B.r;
        const int grid = grid1d(len);
        LAUNCH(k_dot, grid, len, X, B.Wd, part, g);
        return grid;
    }
    const Shape sh = shape_for(B.r);
    const int grid = std::min(nblocks_for((size_t)B.nc, TPB / sh.lg), 1024);
    SHAPE_DISPATCH(sh, LAUNCH((k_obj<LG_, V2_, NS_>), grid, B.nc, B.c_row, B.c_col, B.c_val, X, Y, B.r, part, g));
    return grid;
}

// cv = A_k(sym(X Y^T))   (LORADSInitConstrVal, lorads_alg_common.c:71-76) and, per `mode`, the running m-vector
// T(U,V) is kept in B.T (and reused by the next solve's initial residual, whose pair dots are the same
// numbers); every other pair goes through B.T2
void constr_val(lorads_hip_ctx *c, Block &B, const double *X, const double *Y, double scale, double *cv, int mode, double *vec,
                Guard g) {
    if (B.nrow == 0) return;
    const bool uv = (X == c->U + B.off && Y == c->V + B.off);
    if (B.use_cw) { // one kernel, no pair-dot array; for (U, V) the values are kept as the operator's weights
        cw(c, B, X, Y, scale, uv ? B.w_uv : (double *)nullptr, cv, mode, vec, g);
        if (uv) B.t_uv_valid = true;
        return;
    }
    double *T = uv ? B.T : B.T2;
    pairdots(c, B.pa, X, Y, B.r, T, g);
    if (uv) B.t_uv_valid = true;
    LAUNCH(k_cv, nblocks_for((size_t)B.nrow, TPB / 8), B.nrow, B.a_ptr, B.a_e, B.a_val, T, scale, cv, mode, B.row_idx, vec, g);
}
void sval(lorads_hip_ctx *c, const Pattern &P, bool with_c, int mode, const WArgs &wa, Guard g, CGState *reset = nullptr,
          int nreset = 0) {
    if (P.ne == 0 && !reset) return;
    LAUNCH(k_sval, std::max(1, nblocks_for((size_t)P.ne, TPB)), P.ne, P.e_ptr, P.e_con, P.e_val, with_c ? P.cbase : nullptr, mode,
           wa, P.S, g, reset, nreset);
}

// one application of the CG operator  out = epilogue(x + (sum_i <A_i, sym(x V^T)> A_i) V)
// (linSysProduct, lorads_admm.c:376-391); returns #partials in `part`
int apply_operator(lorads_hip_ctx *c, Block &B, const double *V, const double *x, int mode, const double *rhs, double *out,
                   double *part, Guard g) {
    const bool samp = c->prof && (c->n_matvec % c->prof_every == 0) && c->ev_next + 3 <= c->ev_pool.size();
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (samp) {
        e0 = c->ev_pool[c->ev_next++]; e1 = c->ev_pool[c->ev_next++];
        hipEventRecord(e0, c->stream);
    }
    int grid;
    if (B.diag_only) {
        grid = op_diag(c, B, V, mode, x, rhs, out, part, g);
    } else if (B.entry_only) {
        grid = op_entry(c, B, V, mode, x, rhs, out, part, g);
    } else if (B.use_cw) {
        // x and V are this cone's (U,V) in either order and B.w_uv already holds A(sym(U V^T))
        const bool is_uv = (x == c->U + B.off && V == c->V + B.off) || (x == c->V + B.off && V == c->U + B.off);
        const double *w = B.w_uv;
        if (!(is_uv && B.t_uv_valid)) {
            cw(c, B, x, V, 1.0, is_uv ? B.w_uv : B.w_op, (double *)nullptr, (int)CV_SET, (double *)nullptr, g);
            w = is_uv ? B.w_uv : B.w_op;
            B.t_uv_valid = is_uv;
        }
        if (samp) { e2 = c->ev_pool[c->ev_next++]; hipEventRecord(e2, c->stream); }
        grid = spmm_cw(c, B, w, V, mode, x, rhs, out, part, g);
    } else {
        // x and V are this cone's (U,V) in either order and B.T already holds their pair dots
        const bool is_uv = (x == c->U + B.off && V == c->V + B.off) || (x == c->V + B.off && V == c->U + B.off);
        if (!(is_uv && B.t_uv_valid)) {
            pairdots(c, B.pa, x, V, B.r, B.T, g);
            B.t_uv_valid = is_uv;
        }
        if (B.has_gram) {
            LAUNCH(k_sgram, nblocks_for((size_t)B.pa.ne, TPB / 8), B.pa.ne, B.g_ptr, B.g_col, B.g_val, B.T, B.pa.S, g);
        } else {
            LAUNCH(k_cv, nblocks_for((size_t)B.nrow, TPB / 8), B.nrow, B.a_ptr, B.a_e, B.a_val, B.T, 1.0, B.wtmp, (int)CV_SET,
                   B.row_idx, (double *)nullptr, g);
            WArgs wa{};
            wa.w = B.wtmp;
            sval(c, B.pa, false, W_COMPACT, wa, g);
        }
        if (samp) { e2 = c->ev_pool[c->ev_next++]; hipEventRecord(e2, c->stream); }
        grid = spmm(c, B, B.pa, V, mode, x, rhs, 0.0, out, part, g);
    }
    if (samp) {
        hipEventRecord(e1, c->stream);
        c->pend_mv.push_back({e0, e1});
        if (e2) c->pend_sp.push_back({e2, e1});
    }
    c->n_matvec++;
    return grid;
}

void drain_events(lorads_hip_ctx *c) {
    for (auto &pr : c->pend_mv) {
        float ms = 0;
        hipEventSynchronize(pr.second);
        hipEventElapsedTime(&ms, pr.first, pr.second);
        c->ms_samp += ms; c->n_samp++;
    }
    for (auto &pr : c->pend_sp) {
        float ms = 0;
        hipEventElapsedTime(&ms, pr.first, pr.second);
        c->ms_samp_spmm += ms; c->n_samp_spmm++;
    }
    c->pend_mv.clear();
    c->pend_sp.clear();
    c->ev_next = 0;
}

// ---- one CG solve, split so that it can be enqueued speculatively and resumed
struct Solve {
    Block *B;
    double *x;
    const double *V;
    CGState *st;
    Guard front;   // gate of everything before the iterations ({nullptr, need = previous solve done})
    size_t len;
    int gv;
};

// rhs = V - (C + sum_i M1_i A_i) V / rho, initial residual, state  (lorads_admm.c:432-463, lorads_cgs.c:115,149-172)
void solve_front(lorads_hip_ctx *c, const Solve &s, double rho, double tol, CGState *reset = nullptr, int nreset = 0) {
    Block &B = *s.B;
    double *r = c->cr + B.off, *p = c->cp + B.off, *rhs = c->rhs + B.off;
    double *pA = part_slot(c, 0), *pB = part_slot(c, 1);
    WArgs wa{};
    wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.cv = B.cv; wa.row_idx = B.row_idx; wa.rho = rho;
    sval(c, B.pu, true, W_ADMM, wa, s.front, reset, nreset);
    if (B.dense_c) dense_cx(c, B, s.V, B.Wd, s.front);
    const int nb1 = spmm(c, B, B.pu, s.V, OP_RHS, nullptr, nullptr, rho, rhs, pB, s.front, B.dense_c ? B.Wd : nullptr);
    const int na = apply_operator(c, B, s.V, s.x, OP_RES, rhs, r, pA, s.front);
    LAUNCH(k_cg_init, 1, s.st, pA, na, pB, nb1, tol, s.front);
}
// body of CG iteration k up to and including the convergence test (lorads_cgs.c:180-194)
void solve_iter_body(lorads_hip_ctx *c, const Solve &s, int k, double tol, int maxit) {
    Block &B = *s.B;
    double *r = c->cr + B.off, *p = k == 0 ? c->cr + B.off : c->cp + B.off, *Q = c->cQ + B.off; // p_0 = r_0
    double *pA = part_slot(c, 0), *pC = part_slot(c, 2);
    const Guard g{&s.st->done, s.front.need};
    const int npq = apply_operator(c, B, s.V, p, OP_CG, nullptr, Q, pA, g);
    LAUNCH(k_cg_update, s.gv, s.len, s.st, pA, npq, s.x, r, p, Q, pC, g);
    B.t_uv_valid = false; // x (= U or V) moved
    LAUNCH(k_cg_check, 1, s.st, (int)CHK_ITER, pC, s.gv, tol, maxit, g);
}
// tail of CG iteration k: restart with the true residual when k % 20 == 0 (incl. k = 0), new direction (:195-228)
void solve_iter_tail(lorads_hip_ctx *c, const Solve &s, int k, double tol, int maxit) {
    Block &B = *s.B;
    double *r = c->cr + B.off, *p = c->cp + B.off, *rhs = c->rhs + B.off;
    double *pA = part_slot(c, 0);
    const Guard g{&s.st->done, s.front.need};
    if (k % 20 == 0) {
        const int nr = apply_operator(c, B, s.V, s.x, OP_RES, rhs, r, pA, g);
        LAUNCH(k_cg_check, 1, s.st, (int)CHK_RESTART, pA, nr, tol, maxit, g);
        LAUNCH(k_cg_dir, s.gv, s.len, s.st, (int)DIR_RESTART, r, p, g);
    } else {
        LAUNCH(k_cg_dir, s.gv, s.len, s.st, (int)DIR_BETA, r, p, g);
    }
}
// iterations [k0, k1): bodies, with tails between them (the tail of the last one is left to a resume)
void solve_iters(lorads_hip_ctx *c, const Solve &s, int k0, int k1, double tol, int maxit) {
    for (int k = k0; k < k1; ++k) {
        if (k > k0) solve_iter_tail(c, s, k - 1, tol, maxit);
        solve_iter_body(c, s, k, tol, maxit);
    }
}
// constrVal[k] <- A_k(sym(U V^T)) and constrValSum += new - old (lorads_alg_common.c:199-203)
void refresh_after_solve(lorads_hip_ctx *c, Block &B, const int *need) {
    constr_val(c, B, c->U + B.off, c->V + B.off, 1.0, B.cv, CV_DELTA, c->csum, Guard{nullptr, need});
}

// copy [word0, word0 + nwords) of the control block to its pinned mirror and wait for it
int publish_and_wait(lorads_hip_ctx *c, size_t word0, size_t nwords) {
    if (!c->use_publish) {
        HC(hipMemcpyAsync(c->h_ctrl + 8 * word0, c->ctrl + 8 * word0, 8 * nwords, hipMemcpyDeviceToHost, c->stream));
        HC(hipStreamSynchronize(c->stream));
        return 0;
    }
    const unsigned long long seq = ++c->pub_seq;
    LAUNCH(k_publish, 1, (const unsigned long long *)c->ctrl + word0, (int)nwords, (unsigned long long *)c->h_ctrl_dev + word0,
           c->h_flag_dev, seq);
    volatile unsigned long long *f = c->h_flag;
    for (unsigned long spins = 0;; ++spins) {
        if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 0xfffff) == 0xfffff) { // every ~million spins: is the stream still alive?
            hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { // everything ran: the flag must be there
                if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) return 0;
                HC(hipStreamSynchronize(c->stream));
                if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) return 0;
                return fail_msg("result hand-over: sequence number not published");
            }
            if (q != hipErrorNotReady) return fail("hipStreamQuery", q);
        }
    }
}
int read_states(lorads_hip_ctx *c) {
    return publish_and_wait(c, 0, (64 * sizeof(double) + sizeof(CGState) * (size_t)(2 * c->nb)) / 8);
}

Solve make_solve(lorads_hip_ctx *c, int k, int half, const int *need) {
    Block &B = c->blk[k];
    Solve s;
    s.B = &B;
    s.x = (half == 0 ? c->U : c->V) + B.off;
    s.V = (half == 0 ? c->V : c->U) + B.off;
    s.st = c->st + 2 * k + half;
    s.front = Guard{nullptr, need};
    s.len = (size_t)B.n * B.r;
    s.gv = grid1d(s.len);
    return s;
}

// the LP block's ADMM update (see k_lp_sweep); st = the block's two stage states
void enqueue_lp_sweep(lorads_hip_ctx *c, Block &B, double rho, CGState *st, Guard g) {
    LpArgs A;
    A.ncols = B.n; A.nlev = B.lp_nlev; A.lvl_ptr = B.lp_lvl_ptr; A.lvl_cols = B.lp_lvl_cols; A.ptr = B.lp_ptr; A.grow = B.lp_grow;
    A.a = B.lp_a; A.nrm2sq = B.lp_nrm2sq; A.cobj = B.lp_cobj; A.cv = B.lp_cv; A.U = c->U + B.off; A.V = c->V + B.off;
    A.csum = c->csum; A.b = c->b; A.lambda = c->lambda; A.rho = rho;
    B.t_uv_valid = false;
    if (B.lp_nlev <= 1) {
        LAUNCH(k_lp_sweep_flat, nblocks_for((size_t)B.n, TPB), A, g);
        hipLaunchKernelGGL(k_lp_done, dim3(1), dim3(1), 0, c->stream, st, g);
    } else {
        LAUNCH(k_lp_sweep, 1, A, st, g);
    }
}
// constrValLP of the LP block from the pair (X, Y)
void lp_col_values(lorads_hip_ctx *c, Block &B, const double *X, const double *Y, Guard g) {
    if (!B.is_lp || B.n == 0) return;
    LAUNCH(k_lp_cv, nblocks_for((size_t)B.n, TPB), B.n, B.lp_ptr, B.lp_a, X + B.off, Y + B.off, B.lp_cv, g);
}

// LORADSUpdateSDPVar (lorads_alg_common.c:187-215) for all cones of this context, enqueued speculatively
// from stage `first` (stage = 2*cone + half); `resume_iter` >= 0 resumes that stage's CG after that many
// completed iterations
void enqueue_sweep(lorads_hip_ctx *c, int first, int resume_iter, double rho, double tol, int maxit, bool eval_follows) {
    for (int stg = first; stg < 2 * c->nb; ++stg) {
        const int k = stg / 2, half = stg % 2;
        const int *need = stg == 0 ? nullptr : &c->st[stg - 1].done;
        Solve s = make_solve(c, k, half, need);
        Block &B = *s.B;
        if (B.is_lp) { // both halves of every column in one go at the block's first stage; never a speculation miss
            if (half == 0) enqueue_lp_sweep(c, B, rho, c->st + stg, Guard{nullptr, need});
            continue;
        }
        if (stg == first && resume_iter >= 0) {
            const int more = std::max(2, std::min(resume_iter, 16));
            if (resume_iter > 0) solve_iter_tail(c, s, resume_iter - 1, tol, maxit);
            solve_iters(c, s, resume_iter, std::min(resume_iter + more, maxit), tol, maxit);
        } else {
            // a fresh sweep marks every later stage "not finished" in its very first kernel
            const bool fresh0 = stg == 0 && resume_iter < 0 && 2 * c->nb - 1 <= TPB;
            solve_front(c, s, rho, tol, fresh0 ? c->st + 1 : nullptr, fresh0 ? 2 * c->nb - 1 : 0);
            solve_iters(c, s, 0, std::min(B.spec[half], maxit), tol, maxit);
        }
        if (eval_follows && stg == 2 * c->nb - 1) {
            // the evaluation that follows overwrites constrVal / constrValSum with A(R R^T) (Q1): after the very last
            // solve only the pair dots are kept (the next sweep's first residual re-uses them)
            if (B.use_cw) {
                cw(c, B, c->U + B.off, c->V + B.off, 1.0, B.w_uv, (double *)nullptr, (int)CV_SET, (double *)nullptr,
                   Guard{nullptr, &s.st->done});
                B.t_uv_valid = true;
            } else if (!B.diag_only && !B.entry_only) { // (those operators never read the pair dots)
                pairdots(c, B.pa, c->U + B.off, c->V + B.off, B.r, B.T, Guard{nullptr, &s.st->done});
                B.t_uv_valid = true;
            }
        } else {
            refresh_after_solve(c, B, &s.st->done);
        }
    }
}

// after a sync: first stage >= first whose solve is not finished, or -1
int first_unfinished(lorads_hip_ctx *c, int first) {
    for (int stg = first; stg < 2 * c->nb; ++stg)
        if (c->h_st[stg].done == 0) return stg;
    return -1;
}

// objective + DIMACS refresh (calObj_admm + LORADSCalDualObj + updateDimacsADMM, lorads_admm.c:79-81):
// R = (U+V)/2 (pair UV), constrVal <- A(R R^T), constrValSum, then scal[0..2] = {||b-sum||^2, b.lambda, <C,RR^T>}
int enqueue_eval(lorads_hip_ctx *c, int pair, const int *need, bool with_obj = true) {
    const Guard g{nullptr, need};
    if (pair == LORADS_HIP_PAIR_UV) LAUNCH(k_average, grid1d(c->all_elem), c->all_elem, c->U, c->V, c->R, g);
    Block *S1 = solo(c);
    if (S1 && (S1->nrow != c->m || S1->dense_c)) S1 = c->nb == 1 ? S1 : nullptr;
    const bool single = (S1 && S1->nrow == c->m) || (c->nb == 1 && c->blk[0].nrow == c->m);
    const bool fold_res = single && !c->ar && c->m > 0; // the constraint-value kernel also delivers the residual partials
    c->ls_np = 0;
    // sharded cones: the local sums go to a staging vector, are all-reduced there and committed to constrValSum only
    // if no rank reported a speculation miss (an unfinished sweep still needs the old constrValSum)
    double *dst = c->ar ? c->cstage : c->csum;
    if (!single || c->ar) LAUNCH(k_zero, grid1d((size_t)c->m), (size_t)c->m, dst, g);
    bool first_obj = true;
    const bool fold_obj = (c->nb == 1 || S1) && !c->ar; // the final kernel sums the objective partials itself
    int nobj = 0, nres = 0;
    std::vector<Block *> cones;
    if (S1) cones.push_back(S1);
    else for (auto &B0 : c->blk) cones.push_back(&B0);
    for (Block *bp : cones) {
        Block &B = *bp;
        if (fold_res) {
            pairdots(c, B.pa, c->R + B.off, c->R + B.off, B.r, B.T2, g);
            nres = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 2048);
            LAUNCH(k_cv_res, nres, B.nrow, B.a_ptr, B.a_e, B.a_val, B.T2, B.cv, B.row_idx, c->csum, c->b, c->lambda, part_slot(c, 8),
                   part_slot(c, 9), g);
        } else {
            constr_val(c, B, c->R + B.off, c->R + B.off, 1.0, B.cv, (single && !c->ar) ? CV_SET : CV_ADD, dst, g);
        }
        lp_col_values(c, B, c->R, c->R, g); // primalInfeasibilityLP re-initialises constrValLP too (:260-262)
        if (!with_obj) continue; // DIMACS refresh alone (lorads_alg_common.c:250-290) does not touch the objective
        const int go = obj_partials(c, B, c->R + B.off, c->R + B.off, part_slot(c, 4), g);
        if (fold_obj) { nobj = go; continue; }
        if (go) { LAUNCH(k_finalize, 1, part_slot(c, 4), go, 1.0, first_obj ? 0 : 1, c->scal + 2, g); first_obj = false; }
    }
    if (with_obj && first_obj && !(fold_obj && nobj > 0)) LAUNCH(k_zero, 1, (size_t)1, c->scal + 2, g);
    if (c->ar) { // sharded cones: ONE all-reduce of [constrValSum | objective part | "I missed my speculation"]
        HC(hipMemcpyAsync(dst + c->m, c->scal + 2, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_miss_flag, dim3(1), dim3(1), 0, c->stream, need, dst + c->m + 1);
        if (allreduce_dev(c, dst, c->m + 2)) return 1;
        LAUNCH(k_commit_csum, std::max(1, std::min(grid1d((size_t)c->m), 256)), c->m, dst, c->csum);
        HC(hipMemcpyAsync(c->scal + 2, dst + c->m, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HC(hipMemcpyAsync(c->scal + 5, dst + c->m + 1, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    if (!fold_res) {
        nres = std::min(grid1d((size_t)c->m), 1024);
        LAUNCH(k_eval_part, nres, c->m, c->b, c->csum, c->lambda, part_slot(c, 8), part_slot(c, 9), g);
    }
    LAUNCH(k_eval_final, 1, part_slot(c, 8), part_slot(c, 9), nres, c->scal, g,
           (fold_obj && nobj > 0) ? part_slot(c, 4) : (const double *)nullptr, nobj);
    return 0;
}

int read_scalars_at(lorads_hip_ctx *c, const double *dptr, int count, double *out) {
    HC(hipMemcpyAsync(c->h_scal + 32, dptr, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_scal + 32, sizeof(double) * (size_t)count);
    return 0;
}
int read_scalars(lorads_hip_ctx *c, int first, int count, double *out) { // scal[first .. first+count)
    if (publish_and_wait(c, (size_t)first, (size_t)count)) return 1;
    memcpy(out, c->h_scal + first, sizeof(double) * (size_t)count);
    return 0;
}

// flat dot -> device scalar slot (+ cross-rank sum)
int dot_to_slot(lorads_hip_ctx *c, const double *x, const double *y, int slot) {
    int g = grid1d(c->all_elem);
    LAUNCH(k_dot, g, c->all_elem, x, y, part_slot(c, 3), NOGUARD);
    LAUNCH(k_finalize, 1, part_slot(c, 3), g, 1.0, 0, c->scal + slot, NOGUARD);
    return allreduce_dev(c, c->scal + slot, 1);
}

// collect the sweep results after all stages are finished
int finish_sweep(lorads_hip_ctx *c, int *iters) {
    int tot = 0;
    for (int stg = 0; stg < 2 * c->nb; ++stg) {
        Block &B = c->blk[stg / 2];
        const CGState &h = c->h_st[stg];
        if (h.nan) fprintf(stderr, "lorads_hip: NaN residual in CG (cone %d)\n", stg / 2);
        if (h.done != 2) B.cg_iter_last = h.iter; // an immediate exit leaves the stale count (reference quirk)
        tot += B.cg_iter_last;
        B.spec[stg % 2] = h.done == 2 ? 0 : h.iter;
        c->n_cg_it += (h.done == 2 ? 0 : h.iter);
        c->n_solves++;
    }
    *iters = tot;
    return 0;
}

// ---- lockstep sweep over all cones on the merged cone (see SegArgs).  Phase 0 = all U-solves, phase 1 = all V-solves;
// every kernel of a phase is gated on phase_done[] -- skip when the phase is over, wait while the previous one is not.
SegArgs seg_args(lorads_hip_ctx *c, int half) {
    SegArgs sa;
    sa.row0 = c->seg_row0; sa.vt0 = c->seg_vt0; sa.vt_seg = c->seg_vt_seg; sa.vt_e0 = c->seg_vt_e0; sa.st = c->st;
    sa.phase_done = c->phase_done; sa.half = half; sa.ncones = c->nb; sa.rpw = TPB / shape_for(c->merged.r).lg; sa.r = c->merged.r;
    return sa;
}
void batched_body(lorads_hip_ctx *c, int half, int k, double tol, int maxit) {
    Block &M = c->merged;
    const double *Vfix = half == 0 ? c->V : c->U;
    double *x = half == 0 ? c->U : c->V;
    double *r = c->cr, *p = k == 0 ? c->cr : c->cp, *Q = c->cQ; // p_0 = r_0
    double *pA = part_slot(c, 0), *pC = part_slot(c, 2);
    const Guard g{&c->phase_done[half], half ? &c->phase_done[0] : nullptr};
    const SegArgs sa = seg_args(c, half);
    apply_operator(c, M, Vfix, p, OP_CG, nullptr, Q, pA, g);
    LAUNCH(k_cg_update_seg, c->seg_nvt, sa, pA, x, r, p, Q, pC, g);
    M.t_uv_valid = false;
    LAUNCH(k_cg_check_seg, 1, sa, (int)CHK_ITER, pC, tol, maxit, g);
}
void batched_tail(lorads_hip_ctx *c, int half, int k, double tol, int maxit) {
    Block &M = c->merged;
    const double *Vfix = half == 0 ? c->V : c->U;
    double *x = half == 0 ? c->U : c->V;
    const Guard g{&c->phase_done[half], half ? &c->phase_done[0] : nullptr};
    const SegArgs sa = seg_args(c, half);
    if (k % 20 == 0) {
        apply_operator(c, M, Vfix, x, OP_RES, c->rhs, c->cr, part_slot(c, 0), g);
        LAUNCH(k_cg_check_seg, 1, sa, (int)CHK_RESTART, part_slot(c, 0), tol, maxit, g);
        LAUNCH(k_cg_dir_seg, c->seg_nvt, sa, (int)DIR_RESTART, c->cr, c->cp, g);
    } else {
        LAUNCH(k_cg_dir_seg, c->seg_nvt, sa, (int)DIR_BETA, c->cr, c->cp, g);
    }
}
void batched_iters(lorads_hip_ctx *c, int half, int k0, int k1, double tol, int maxit) {
    for (int k = k0; k < k1; ++k) {
        if (k > k0) batched_tail(c, half, k - 1, tol, maxit);
        batched_body(c, half, k, tol, maxit);
    }
}
// returns the number of lockstep iterations enqueued so far for `phase` (for a later resume)
void enqueue_batched(lorads_hip_ctx *c, int phase, int resume, double rho, double tol, int maxit, int launched[2],
                     bool eval_follows) {
    Block &M = c->merged;
    for (int half = phase; half < 2; ++half) {
        const double *Vfix = half == 0 ? c->V : c->U;
        double *x = half == 0 ? c->U : c->V;
        if (half == phase && resume >= 0) {
            const int more = std::max(2, std::min(resume, 16));
            if (resume > 0) batched_tail(c, half, resume - 1, tol, maxit);
            const int k1 = std::min(resume + more, maxit);
            batched_iters(c, half, resume, k1, tol, maxit);
            launched[half] = k1;
        } else {
            const Guard front{nullptr, half ? &c->phase_done[0] : nullptr};
            WArgs wa{};
            wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.cv = M.cv; wa.row_idx = M.row_idx; wa.rho = rho;
            sval(c, M.pu, true, W_ADMM, wa, front);
            spmm(c, M, M.pu, Vfix, OP_RHS, nullptr, nullptr, rho, c->rhs, part_slot(c, 1), front);
            apply_operator(c, M, Vfix, x, OP_RES, c->rhs, c->cr, part_slot(c, 0), front);
            LAUNCH(k_cg_init_seg, 1, seg_args(c, half), part_slot(c, 0), part_slot(c, 1), tol, front);
            const int k1 = std::min(std::max(c->spec_b[half], 0), maxit);
            batched_iters(c, half, 0, k1, tol, maxit);
            launched[half] = k1;
        }
        // constrVal <- A(sym(U V^T)), constrValSum += new - old for every cone at once
        if (eval_follows && half == 1) { // dead before the evaluation (see enqueue_sweep): keep the pair dots only
            if (M.use_cw) {
                cw(c, M, c->U, c->V, 1.0, M.w_uv, (double *)nullptr, (int)CV_SET, (double *)nullptr,
                   Guard{nullptr, &c->phase_done[half]});
                M.t_uv_valid = true;
            } else if (!M.diag_only && !M.entry_only) {
                pairdots(c, M.pa, c->U, c->V, M.r, M.T, Guard{nullptr, &c->phase_done[half]});
                M.t_uv_valid = true;
            }
        } else {
            constr_val(c, M, c->U, c->V, 1.0, M.cv, CV_DELTA, c->csum, Guard{nullptr, &c->phase_done[half]});
        }
    }
}
int run_sweep_batched(lorads_hip_ctx *c, double rho, double tol, int maxit, bool with_eval) {
    int phase = 0, resume = -1, launched[2] = {0, 0};
    bool first_pass = true, any_missed = false;
    for (;;) {
        enqueue_batched(c, phase, resume, rho, tol, maxit, launched, with_eval);
        // the evaluation rides along speculatively; with sharded cones only in the first pass (see run_sweep)
        const bool eval_now = with_eval && (!c->ar || first_pass);
        if (eval_now && enqueue_eval(c, LORADS_HIP_PAIR_UV, &c->phase_done[1])) return 1;
        if (read_states(c)) return 1;
        if (eval_now && c->ar) any_missed = c->h_scal[5] > 0.5;
        first_pass = false;
        bool u_done = true, v_done = true;
        for (int k = 0; k < c->nb; ++k) {
            u_done = u_done && c->h_st[2 * k].done != 0;
            v_done = v_done && c->h_st[2 * k + 1].pad == 1 && c->h_st[2 * k + 1].done != 0;
        }
        if (u_done && v_done) break;
        c->n_resume++;
        c->merged.t_uv_valid = false; // kernels after the miss did not run: recompute
        if (!u_done) { phase = 0; resume = launched[0]; }
        else { phase = 1; resume = c->h_st[1].pad == 1 ? launched[1] : -1; }
    }
    // speculation for the next ADMM iteration: what the slowest cone needed
    for (int half = 0; half < 2; ++half) {
        int mx = 0;
        for (int k = 0; k < c->nb; ++k) {
            const CGState &h = c->h_st[2 * k + half];
            if (h.done != 2) mx = std::max(mx, h.iter);
        }
        c->spec_b[half] = mx;
    }
    if (with_eval && c->ar && any_missed) { // some rank missed: every rank evaluates again, now with finished sweeps
        if (enqueue_eval(c, LORADS_HIP_PAIR_UV, nullptr)) return 1;
        if (read_states(c)) return 1;
    }
    return 0;
}

int run_sweep(lorads_hip_ctx *c, double rho, double tol, int maxit, bool with_eval) {
    // (no cross-rank sum inside the sweep: with sharded cones the lockstep form works unchanged)
    if (c->has_merged && !getenv("LORADS_NO_BATCH")) return run_sweep_batched(c, rho, tol, maxit, with_eval);
    int first = 0, resume = -1;
    bool first_pass = true, any_missed = false;
    // every stage starts "not finished": a stage whose predecessor misses its speculation must stay
    // blocked (and block its successors) instead of seeing last iteration's done word
    if (2 * c->nb - 1 > TPB) HC(hipMemsetAsync(c->st, 0, sizeof(CGState) * (size_t)(2 * c->nb), c->stream));
    for (;;) {
        enqueue_sweep(c, first, resume, rho, tol, maxit, with_eval);
        // With sharded cones the evaluation contains the all-reduce, a collective every rank must enter the same
        // number of times, whatever its own speculation did.  Every rank enters it once in the first pass, and the
        // reduced vector carries one extra word: how many ranks missed.  If any did, every rank finishes its sweep
        // and all enter the collective once more.  No miss anywhere (the common case): one host synchronisation.
        const bool eval_now = with_eval && (!c->ar || first_pass);
        if (eval_now && enqueue_eval(c, LORADS_HIP_PAIR_UV, c->nb ? &c->st[2 * c->nb - 1].done : nullptr)) return 1;
        if (read_states(c)) return 1;
        if (eval_now && c->ar) any_missed = c->h_scal[5] > 0.5;
        first_pass = false;
        const int stg = first_unfinished(c, first);
        if (stg < 0) break;
        first = stg;
        resume = c->h_st[stg].iter;
        c->n_resume++;
        for (auto &B : c->blk) B.t_uv_valid = false; // kernels after the miss did not run: recompute
    }
    if (with_eval && c->ar && any_missed) {
        if (enqueue_eval(c, LORADS_HIP_PAIR_UV, nullptr)) return 1;
        if (read_states(c)) return 1;
    }
    return 0;
}

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
    c->blk.resize(c->nb);
    for (int k = 0; k < c->nb; ++k)
        if (build_block(c, c->blk[k], prob->blocks[k])) { lorads_hip_destroy(c); return 1; }
    if (build_merged(c, prob)) { lorads_hip_destroy(c); return 1; }
    refresh_merged(c);
    if (getenv("LORADS_HIP_VERBOSE"))
        fprintf(stderr, "lorads_hip: %d cone(s), merged view %s\n", c->nb, c->has_merged ? "on" : (c->merged_ok ? "off (ranks differ)" : "not applicable"));
    if (alloc_factors(c)) { lorads_hip_destroy(c); return 1; }
    std::vector<double> hb(prob->b, prob->b + c->m);
    if (upload(&c->b, hb) || dalloc(&c->lambda, (size_t)c->m) || dalloc(&c->csum, (size_t)c->m + 2) ||
        dalloc(&c->cstage, (size_t)c->m + 2) ||
        dalloc(&c->q12, (size_t)2 * c->m + 2) || dalloc(&c->part, (size_t)NSLOT * MAXPART) ||
        dalloc(&c->ctrl, 64 * sizeof(double) + sizeof(CGState) * (size_t)std::max(2 * c->nb, 1)) ||
        dalloc(&c->ring_ab, (size_t)2 * c->L)) {
        lorads_hip_destroy(c);
        return 1;
    }
    HC(hipHostMalloc((void **)&c->h_ctrl, 64 * sizeof(double) + sizeof(CGState) * (size_t)std::max(2 * c->nb, 1), hipHostMallocMapped));
    HC(hipHostMalloc((void **)&c->h_flag, 64, hipHostMallocMapped));
    *c->h_flag = 0;
    HC(hipHostGetDevicePointer((void **)&c->h_ctrl_dev, c->h_ctrl, 0));
    HC(hipHostGetDevicePointer((void **)&c->h_flag_dev, c->h_flag, 0));
    c->use_publish = !getenv("LORADS_NO_PUBLISH");
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
        hipFree(B.row_idx); hipFree(B.a_ptr); hipFree(B.a_e); hipFree(B.a_val); hipFree(B.Cfull); hipFree(B.T); hipFree(B.T2); hipFree(B.wtmp);
        hipFree(B.c_row); hipFree(B.c_col); hipFree(B.c_val); hipFree(B.gdiag); hipFree(B.gentry); hipFree(B.ca_row); hipFree(B.ca_col); hipFree(B.ca_val); hipFree(B.cadj_ptr); hipFree(B.cadj_col); hipFree(B.cadj_con); hipFree(B.cadj_a);
        hipFree(B.w_uv); hipFree(B.w_op); hipFree(B.lp_lvl_ptr); hipFree(B.lp_lvl_cols); hipFree(B.lp_ptr); hipFree(B.lp_grow);
        hipFree(B.lp_a); hipFree(B.lp_nrm2sq); hipFree(B.lp_cobj); hipFree(B.lp_cv); hipFree(B.g_ptr); hipFree(B.g_col);
        hipFree(B.g_val);
    }
    free_factors(c);
    hipFree(c->cstage);
    hipFree(c->b); hipFree(c->lambda); hipFree(c->csum); hipFree(c->q12); hipFree(c->part); hipFree(c->ctrl);
    hipFree(c->ring_ab);
    hipFree(c->seg_row0); hipFree(c->seg_vt0); hipFree(c->seg_vt_seg); hipFree(c->seg_vt_e0); hipFree(c->phase_done);
    if (c->h_ctrl) hipHostFree(c->h_ctrl);
    if (c->h_flag) hipHostFree(c->h_flag);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int lorads_hip_sync(lorads_hip_ctx *c) {
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_set_allreduce(lorads_hip_ctx *c, lorads_hip_allreduce_fn fn, void *user) {
    c->ar = fn;
    c->ar_user = user;
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
    return allreduce_dev(c, c->csum, c->m);
}

static int enqueue_alm_grad(lorads_hip_ctx *c, double rho) {
    LAUNCH(k_zero, 1, (size_t)1, c->scal + 8, NOGUARD);
    std::vector<Block *> cones;
    Block *S1 = solo(c);
    if (S1 && !S1->dense_c) cones.push_back(S1);
    else for (auto &B0 : c->blk) cones.push_back(&B0);
    for (Block *bp : cones) {
        Block &B = *bp;
        WArgs wa{};
        wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.row_idx = B.row_idx; wa.rho = rho;
        sval(c, B.pu, true, W_ALM, wa, NOGUARD);
        if (B.dense_c) dense_cx(c, B, c->R + B.off, B.Wd, NOGUARD);
        int g = spmm(c, B, B.pu, c->R + B.off, OP_GRAD, nullptr, nullptr, rho, c->G + B.off, part_slot(c, 0), NOGUARD,
                     B.dense_c ? B.Wd : nullptr);
        LAUNCH(k_finalize, 1, part_slot(c, 0), g, 1.0, 1, c->scal + 8, NOGUARD);
    }
    return allreduce_dev(c, c->scal + 8, 1);
}
int lorads_hip_alm_cal_grad(lorads_hip_ctx *c, double rho, double *lag) {
    if (enqueue_alm_grad(c, rho)) return 1;
    return read_scalars(c, 8, 1, lag);
}

int lorads_hip_lbfgs_direction(lorads_hip_ctx *c, int32_t inner) {
    for (auto &B : c->blk) B.t_uv_valid = false; // U is overwritten by the direction D
    const size_t n = c->all_elem;
    const int gv = c->ar ? grid1d(n) : grid_lbfgs(n);
    double *D = c->U;
    if (!c->ar) { // single rank: one kernel per stage of the recursion (5 + 1 launches for history 2)
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
    if (S1 && S1->nrow == m && m > 0 && !S1->dense_c && S1->nc > 0) {
        // one cone that sees every constraint: (R,D) and (D,D) share each row visit -- 4 launches
        Block &B = *S1;
        const Shape sh = shape_for(B.r);
        const double *R = c->R + B.off, *D = c->U + B.off;
        SHAPE_DISPATCH(sh, LAUNCH((k_pairdots_rd<LG_, V2_, NS_>), nblocks_for((size_t)B.pa.ne, TPB / sh.lg), B.pa.ne, B.pa.erow,
                                  B.pa.ecol, R, D, B.r, B.T2, B.T));
        B.t_uv_valid = false;
        c->ls_np = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 1024);
        LAUNCH(k_cv_rd, c->ls_np, B.nrow, B.a_ptr, B.a_e, B.a_val, B.T2, B.T, B.cv, B.row_idx, c->q12, c->q12 + m, c->b, c->csum,
               c->lambda, part_slot(c, 10));
        const int go = std::min(nblocks_for((size_t)B.nc, TPB / sh.lg), 1024);
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
    return allreduce_dev(c, c->q12, 2 * m + 2);
}
int lorads_hip_alm_q12p12(lorads_hip_ctx *c, double p12[2]) {
    if (enqueue_q12p12(c)) return 1;
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
        LAUNCH(k_linesearch_part, c->ls_np, c->m, c->b, c->csum, c->lambda, c->q12, c->q12 + c->m, part_slot(c, 10));
    }
    LAUNCH(k_linesearch, 1, c->m, 1.0 / rho, part_slot(c, 10), c->ls_np, c->q12, c->scal + 16,
           np_obj ? part_slot(c, 4) : (const double *)nullptr, np_obj ? part_slot(c, 6) : (const double *)nullptr, np_obj);
}
int lorads_hip_alm_linesearch_coeffs(lorads_hip_ctx *c, double rho, double p1, double p2, double k[4]) {
    launch_linesearch(c, rho, 0);
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
static int enqueue_alm_front(lorads_hip_ctx *c, double rho, int32_t inner) {
    int np = 0;
    if (lorads_hip_lbfgs_direction(c, inner) || enqueue_q12p12(c, &np)) return 1;
    launch_linesearch(c, rho, np);
    return 0;
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
    if (S1 && S1->nrow == c->m && c->m > 0 && !S1->dense_c) {
        // one cone that sees every constraint: 7 launches for the whole second half
        Block &B = *S1;
        Ring &h = c->ring[c->head];
        const int gv = grid_lbfgs(c->all_elem);
        c->ls_np = 0;
        LAUNCH(k_alm_update, gv, c->all_elem, tau, c->G, c->U, h.y, c->R, c->m, c->q12, c->q12 + c->m, c->csum);
        WArgs wa{};
        wa.csum = c->csum; wa.b = c->b; wa.lambda = c->lambda; wa.row_idx = B.row_idx; wa.rho = rho;
        sval(c, B.pu, true, W_ALM, wa, NOGUARD);
        const int glag = spmm(c, B, B.pu, c->R, OP_GRAD, nullptr, nullptr, rho, c->G, part_slot(c, 0), NOGUARD);
        LAUNCH(k_his_two_dot, gv, c->all_elem, tau, c->U, c->G, h.s, h.y, part_slot(c, 3));
        pairdots(c, B.pa, c->R, c->R, B.r, B.T2, NOGUARD);
        const int nres = std::min(nblocks_for((size_t)B.nrow, TPB / 8), 2048);
        LAUNCH(k_cv_res, nres, B.nrow, B.a_ptr, B.a_e, B.a_val, B.T2, B.cv, B.row_idx, c->csum, c->b, c->lambda, part_slot(c, 8),
               part_slot(c, 9), NOGUARD);
        LAUNCH(k_alm_tail, 1, part_slot(c, 0), glag, c->scal + 8, part_slot(c, 3), gv, c->ring_ab + 2 * c->head + 1, part_slot(c, 8),
               part_slot(c, 9), nres, c->scal);
        c->head = (c->head + 1) % c->L;
    } else if (lorads_hip_set_y_as_neg_grad(c) || lorads_hip_alm_update_var(c, tau) || enqueue_alm_grad(c, rho) ||
               lorads_hip_set_lbfgs_his_two(c, tau) || enqueue_eval(c, LORADS_HIP_PAIR_RR, nullptr, false)) {
        return 1;
    }
    if (next_inner >= 0 && enqueue_alm_front(c, rho, next_inner)) return 1;
    double s[23];
    if (read_scalars(c, 0, 23, s)) return 1;
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
    LAUNCH(k_dual_update, nblocks_for((size_t)c->m, TPB), c->m, rho, c->b, c->csum, c->lambda);
    return 0;
}

int lorads_hip_cal_dual_obj(lorads_hip_ctx *c, double *dobj) {
    const int g = std::min(grid1d((size_t)c->m), 256);
    LAUNCH(k_dot, g, (size_t)c->m, c->b, c->lambda, part_slot(c, 7), NOGUARD);
    LAUNCH(k_finalize, 1, part_slot(c, 7), g, 1.0, 0, c->scal + 4, NOGUARD);
    return read_scalars(c, 4, 1, dobj);
}

static void invalidate_t(lorads_hip_ctx *c) {
    c->merged.t_uv_valid = false;
    for (auto &B : c->blk) B.t_uv_valid = false;
}

int lorads_hip_alm_to_admm(lorads_hip_ctx *c) {
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
    for (auto &B : c->blk) {
        if (B.nc) LAUNCH(k_scale, grid1d((size_t)B.nc), (size_t)B.nc, s, B.c_val);
        if (B.pu.ne) LAUNCH(k_scale, grid1d((size_t)B.pu.ne), (size_t)B.pu.ne, s, B.pu.cbase);
        if (B.dense_c) LAUNCH(k_scale, grid1d((size_t)B.npad * B.npad), (size_t)B.npad * B.npad, s, B.Cfull);
        if (B.is_lp && B.n) LAUNCH(k_scale, grid1d((size_t)B.n), (size_t)B.n, s, B.lp_cobj);
    }
    if (c->m) LAUNCH(k_scale, grid1d((size_t)c->m), (size_t)c->m, s, c->lambda);
    return 0;
}

int lorads_hip_set_mat(lorads_hip_ctx *c, int32_t which, int32_t k, const double *cm) {
    double *base = mat_base(c, which);
    if (!base || k < 0 || k >= c->nb) return fail_msg("set_mat: bad argument");
    Block &B = c->blk[k];
    B.t_uv_valid = false;
    std::vector<double> rm((size_t)B.n * B.r);
    for (int j = 0; j < B.r; ++j)
        for (int i = 0; i < B.n; ++i) rm[(size_t)i * B.r + j] = cm[(size_t)j * B.n + i];
    HC(hipMemcpyAsync(base + B.off, rm.data(), sizeof(double) * rm.size(), hipMemcpyHostToDevice, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_get_mat(lorads_hip_ctx *c, int32_t which, int32_t k, double *cm) {
    double *base = mat_base(c, which);
    if (!base || k < 0 || k >= c->nb) return fail_msg("get_mat: bad argument");
    Block &B = c->blk[k];
    std::vector<double> rm((size_t)B.n * B.r);
    HC(hipMemcpyAsync(rm.data(), base + B.off, sizeof(double) * rm.size(), hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    for (int j = 0; j < B.r; ++j)
        for (int i = 0; i < B.n; ++i) cm[(size_t)j * B.n + i] = rm[(size_t)i * B.r + j];
    return 0;
}

int lorads_hip_set_vec(lorads_hip_ctx *c, int32_t which, const double *v) {
    c->ls_np = 0;
    double *d = vec_base(c, which);
    if (!d) return fail_msg("set_vec: bad argument");
    HC(hipMemcpyAsync(d, v, sizeof(double) * (size_t)c->m, hipMemcpyHostToDevice, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_get_vec(lorads_hip_ctx *c, int32_t which, double *v) {
    double *d = vec_base(c, which);
    if (!d) return fail_msg("get_vec: bad argument");
    HC(hipMemcpyAsync(v, d, sizeof(double) * (size_t)c->m, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

int lorads_hip_resize_rank(lorads_hip_ctx *c, const int32_t *nr) {
    // AUG_RANK (data/lorads_solver.c:806-906): keep the old columns, new columns = 1/sqrt(k) on their
    // leading diagonal (lpRandomDiag :776-786), clear L-BFGS history and CG workspaces
    std::vector<std::vector<double>> keep[4];
    int whichs[4] = {LORADS_HIP_MAT_R, LORADS_HIP_MAT_U, LORADS_HIP_MAT_V, LORADS_HIP_MAT_GRAD};
    for (int a = 0; a < 4; ++a) {
        keep[a].resize(c->nb);
        for (int k = 0; k < c->nb; ++k) {
            Block &B = c->blk[k];
            if (nr[k] < B.r || nr[k] > 512) return fail_msg("resize_rank: bad rank");
            if (B.dense_c && nr[k] > 128)
                return fail_msg("resize_rank: a cone with a dense objective matrix supports rank <= 128 (MFMA tile kernel)");
            if (nblocks_for((size_t)B.n, TPB / lg_for(nr[k])) > MAXPART)
                return fail_msg("resize_rank: cone dimension too large for the partial-sum slots at this rank");
            std::vector<double> oldm((size_t)B.n * B.r);
            if (lorads_hip_get_mat(c, whichs[a], k, oldm.data())) return 1;
            std::vector<double> nw((size_t)B.n * nr[k], 0.0);
            std::copy(oldm.begin(), oldm.end(), nw.begin());
            int aug = nr[k] - B.r, rr = std::min(B.n, aug);
            for (int i = 0; i < rr; ++i) nw[(size_t)B.n * B.r + (size_t)i * B.n + i] = 1 / std::sqrt((double)rr);
            keep[a][k] = std::move(nw);
        }
    }
    free_factors(c);
    invalidate_t(c);
    for (int k = 0; k < c->nb; ++k) { c->blk[k].r = nr[k]; }
    refresh_merged(c);
    if (alloc_factors(c)) return 1;
    for (int a = 0; a < 4; ++a)
        for (int k = 0; k < c->nb; ++k)
            if (lorads_hip_set_mat(c, whichs[a], k, keep[a][k].data())) return 1;
    HC(hipMemset(c->ring_ab, 0, sizeof(double) * (size_t)2 * c->L));
    HC(hipDeviceSynchronize());
    return 0;
}

int lorads_hip_profile(lorads_hip_ctx *c, int32_t enable, int32_t every) {
    HC(hipStreamSynchronize(c->stream));
    drain_events(c);
    c->prof = enable;
    c->prof_every = std::max(1, every);
    if (enable && c->ev_pool.empty()) {
        c->ev_pool.resize(3 * 1024);
        for (auto &e : c->ev_pool) HC(hipEventCreate(&e));
    }
    c->n_matvec = c->n_cg_it = c->n_solves = c->n_samp = c->n_samp_spmm = c->n_resume = 0;
    c->ms_samp = c->ms_samp_spmm = 0;
    return 0;
}

int lorads_hip_profile_read(lorads_hip_ctx *c, double s[8]) {
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

int lorads_hip_operator_kind(lorads_hip_ctx *c, int32_t k, int32_t *kind) {
    if (k < 0 || k >= c->nb) return fail_msg("bad block");
    const Block &B = c->blk[k];
    *kind = B.diag_only ? 2 : B.entry_only ? 3 : B.use_cw ? 4 : B.has_gram ? 0 : 1;
    return 0;
}

int lorads_hip_algorithmic_bytes(lorads_hip_ctx *c, int32_t k, double *mv, double *cg) {
    if (k < 0 || k >= c->nb) return fail_msg("bad block");
    *mv = c->blk[k].bytes_mv;
    *cg = c->blk[k].bytes_cg;
    return 0;
}

} // extern "C"

#include "lanczos.inc"
