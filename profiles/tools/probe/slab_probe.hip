// VERDICT r3 item 3, measured before building: do column slabs with XCD affinity make the SpMM-type row gathers of the headline
// faster?  One SpMM  out_p = sum_{q in adj(p)} w_pq X_q  over a random pattern of the headline's shape (n = 20000, r = 40, ~19
// neighbours per row), three layouts of X:
//   rowmajor : X[q][40], an 8-lane group per row, three 128-byte column steps (the product's k_spmm_ell shape; table 6.4 MB)
//   slab4    : X[s][q][10], s = 0..3: workgroups of XCD x gather slab x % 4 only (1.6 MB per slab: L2-resident), 5 lanes per row
//   slab4 any: the same layout, slab chosen by workgroup index / 8 (every XCD sees every slab: the layout without the affinity)
//   slab2    : X[s][q][20], s = 0..1: XCD x gathers slab x % 2 (3.2 MB per slab), 10 lanes per row
// Each launch timed back to back (200 launches between one event pair); results of the layouts compared element by element.
// Build: hipcc -O3 --offload-arch=gfx950 slab_probe.hip -o slab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TPB = 256;

// 8 lanes per row, 32 rows per workgroup; U neighbours requested per trip
template <int U, int NST = 3>
__global__ __launch_bounds__(TPB) void k_rowmajor(int n, int r, const int *__restrict__ ptr, const int *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ X, double *__restrict__ out) {
    const int lane = threadIdx.x & 7, p = blockIdx.x * (TPB / 8) + threadIdx.x / 8;
    if (p >= n) return;
    double2 acc[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) acc[s] = double2{0, 0};
    const int e0 = ptr[p], e1 = ptr[p + 1];
    for (int e = e0; e < e1; e += U) {
        int q[U]; double w[U]; double2 x[U][NST];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int ee = e + u < e1 ? e + u : e0; q[u] = col[ee]; w[u] = e + u < e1 ? val[ee] : 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const int c = s * 16 + lane * 2;
                x[u][s] = c < r ? *(const double2 *)(X + (size_t)q[u] * r + c) : double2{0, 0};
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int s = 0; s < NST; ++s) { acc[s].x += w[u] * x[u][s].x; acc[s].y += w[u] * x[u][s].y; }
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int c = s * 16 + lane * 2;
        if (c < r) *(double2 *)(out + (size_t)p * r + c) = acc[s];
    }
}

// slabs of W doubles (W/2 lanes per row).  affine: workgroup b works on slab (b % 8) % S; rows dealt to the workgroups of a slab.
template <int S, int W, int U, bool AFFINE>
__global__ __launch_bounds__(TPB) void k_slab(int n, const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                              const double *__restrict__ Xs, double *__restrict__ outs, int *__restrict__ misplaced) {
    constexpr int L = W / 2, RPW = 64 / L, RPB = RPW * (TPB / 64);
    const int b = blockIdx.x, x8 = b & 7;
    int slab, w;
    if (AFFINE) { slab = x8 % S; w = (b >> 3) * (8 / S) + x8 / S; }           // 8 / S XCDs share a slab
    else { slab = (b >> 3) % S; w = ((b >> 3) / S) * 8 + x8; }
    if (AFFINE && threadIdx.x == 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        if ((xcc & 7) != x8) atomicAdd(misplaced, 1);
    }
    const int wl = threadIdx.x & 63, sub = wl % L, slot = wl / L;
    const int p = w * RPB + (threadIdx.x >> 6) * RPW + slot;
    if (slot >= RPW || p >= n) return;
    const double *T = Xs + (size_t)slab * n * W;
    double2 acc = {0, 0};
    const int e0 = ptr[p], e1 = ptr[p + 1];
    for (int e = e0; e < e1; e += U) {
        int q[U]; double wv[U]; double2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int ee = e + u < e1 ? e + u : e0; q[u] = col[ee]; wv[u] = e + u < e1 ? val[ee] : 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *(const double2 *)(T + (size_t)q[u] * W + sub * 2);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += wv[u] * x[u].x; acc.y += wv[u] * x[u].y; }
    }
    *(double2 *)(outs + (size_t)slab * n * W + (size_t)p * W + sub * 2) = acc;
}

template <class F> static double time_us(hipStream_t st, int reps, F f) {
    hipEvent_t a, b;
    HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    HC(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) f();
    HC(hipEventRecord(b, st));
    HC(hipEventSynchronize(b));
    float ms = 0;
    HC(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000, deg = argc > 2 ? atoi(argv[2]) : 19, r = argc > 3 ? atoi(argv[3]) : 40; // r = 40 or 64
    std::mt19937_64 g(1);
    std::vector<int> ptr(n + 1), col;
    std::vector<double> val;
    for (int p = 0; p < n; ++p) {
        ptr[p] = (int)col.size();
        const int d = deg - 4 + (int)(g() % 9);
        for (int k = 0; k < d; ++k) { col.push_back((int)(g() % n)); val.push_back(1.0 + (double)(g() % 1000) * 1e-3); }
    }
    ptr[n] = (int)col.size();
    const size_t ne = col.size();
    std::vector<double> X((size_t)n * r);
    for (auto &v : X) v = (double)(g() % 2001) * 1e-3 - 1.0;
    auto slabbed = [&](int S) { // X[s][q][W]
        const int W = r / S;
        std::vector<double> T((size_t)n * r);
        for (int s = 0; s < S; ++s) for (int q = 0; q < n; ++q) for (int c = 0; c < W; ++c) T[((size_t)s * n + q) * W + c] = X[(size_t)q * r + s * W + c];
        return T;
    };
    std::vector<double> X4 = slabbed(4), X2 = slabbed(2);
    int *d_ptr, *d_col, *d_mis; double *d_val, *d_X, *d_X4, *d_X2, *d_o, *d_o4, *d_o2;
    HC(hipMalloc(&d_ptr, sizeof(int) * (n + 1))); HC(hipMalloc(&d_col, sizeof(int) * ne)); HC(hipMalloc(&d_val, sizeof(double) * ne));
    HC(hipMalloc(&d_mis, sizeof(int))); HC(hipMemset(d_mis, 0, sizeof(int)));
    const size_t vb = sizeof(double) * (size_t)n * r;
    HC(hipMalloc(&d_X, vb)); HC(hipMalloc(&d_X4, vb)); HC(hipMalloc(&d_X2, vb)); HC(hipMalloc(&d_o, vb)); HC(hipMalloc(&d_o4, vb)); HC(hipMalloc(&d_o2, vb));
    HC(hipMemcpy(d_ptr, ptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HC(hipMemcpy(d_col, col.data(), sizeof(int) * ne, hipMemcpyHostToDevice));
    HC(hipMemcpy(d_val, val.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    HC(hipMemcpy(d_X, X.data(), vb, hipMemcpyHostToDevice)); HC(hipMemcpy(d_X4, X4.data(), vb, hipMemcpyHostToDevice)); HC(hipMemcpy(d_X2, X2.data(), vb, hipMemcpyHostToDevice));
    hipStream_t st; HC(hipStreamCreate(&st));
    const double mb = (double)ne * r * 8 / 1e6;
    printf("n = %d, r = %d, %zu row visits (%.1f per row), %.1f MB of rows gathered, table %.1f MB, indices + weights %.1f MB\n", n, r, ne,
           (double)ne / n, mb, vb / 1e6, ne * 12 / 1e6);
    auto report = [&](const char *name, double us, double idx_mb) {
        printf("%-44s %8.2f us   %6.2f TB/s of rows   (indices + weights read: %.1f MB)\n", name, us, mb / us, idx_mb);
    };
    auto slab_grid = [&](int S, int W) { const int rpb = (64 / (W / 2)) * 4; const int per = (n + rpb - 1) / rpb; return ((per + 8 / S - 1) / (8 / S)) * 8; };
    const int reps = 200;
    const int g_row = (n + TPB / 8 - 1) / (TPB / 8);
    if (r == 64) { // a table far larger than one L2 (cfg5's regime: 480-byte rows of a 12-24 MB table): 4 slabs of 128 bytes
        std::vector<double> X4b = slabbed(4);
        HC(hipMemcpy(d_X4, X4b.data(), vb, hipMemcpyHostToDevice));
        report("rowmajor r = 64, 4 neighbours per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_rowmajor<4, 4>), dim3(g_row), dim3(TPB), 0, st, n, r, d_ptr, d_col, d_val, d_X, d_o); }), ne * 12 / 1e6);
        report("rowmajor r = 64, 8 neighbours per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_rowmajor<8, 4>), dim3(g_row), dim3(TPB), 0, st, n, r, d_ptr, d_col, d_val, d_X, d_o); }), ne * 12 / 1e6);
        const int G = slab_grid(4, 16);
        report("slab4 (128-byte pieces) XCD-affine, 8 per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<4, 16, 8, true>), dim3(G), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis); }), 4.0 * ne * 12 / 1e6);
        const int G2 = ((n + 31) / 32 + 7) / 8 * 8 * 4;
        report("slab4 (128-byte pieces), any XCD, 8 per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<4, 16, 8, false>), dim3(G2), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis); }), 4.0 * ne * 12 / 1e6);
        std::vector<double> o((size_t)n * r), o4((size_t)n * r);
        HC(hipMemcpy(o.data(), d_o, vb, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL((k_slab<4, 16, 8, true>), dim3(G), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis);
        HC(hipMemcpy(o4.data(), d_o4, vb, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 4; ++s) for (int q = 0; q < n; ++q) for (int c = 0; c < 16; ++c)
            worst = std::fmax(worst, std::fabs(o4[((size_t)s * n + q) * 16 + c] - o[(size_t)q * r + s * 16 + c]));
        printf("    slab4 against rowmajor: largest difference %.2e\n", worst);
        int mis = 0;
        HC(hipMemcpy(&mis, d_mis, sizeof(int), hipMemcpyDeviceToHost));
        printf("workgroups whose XCC_ID was not blockIdx %% 8: %d\n", mis);
        return 0;
    }
    report("rowmajor, 4 neighbours per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_rowmajor<4>), dim3(g_row), dim3(TPB), 0, st, n, r, d_ptr, d_col, d_val, d_X, d_o); }), ne * 12 / 1e6);
    report("rowmajor, 8 neighbours per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_rowmajor<8>), dim3(g_row), dim3(TPB), 0, st, n, r, d_ptr, d_col, d_val, d_X, d_o); }), ne * 12 / 1e6);
    // slab4: 12 rows per wavefront, 48 per workgroup
    { const int G = slab_grid(4, 10);
      report("slab4 (80-byte pieces) XCD-affine, 8 per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<4, 10, 8, true>), dim3(G), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis); }), 4.0 * ne * 12 / 1e6);
      std::vector<double> o((size_t)n * r), o4((size_t)n * r);
      HC(hipMemcpy(o.data(), d_o, vb, hipMemcpyDeviceToHost)); HC(hipMemcpy(o4.data(), d_o4, vb, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int s = 0; s < 4; ++s) for (int q = 0; q < n; ++q) for (int c = 0; c < 10; ++c)
          worst = std::fmax(worst, std::fabs(o4[((size_t)s * n + q) * 10 + c] - o[(size_t)q * r + s * 10 + c]));
      printf("    slab4 against rowmajor: largest difference %.2e\n", worst);
      HC(hipMemset(d_o4, 0, vb));
      report("slab4 (80-byte pieces) XCD-affine, 16 per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<4, 10, 16, true>), dim3(G), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis); }), 4.0 * ne * 12 / 1e6);
      const int G2 = ((n + 47) / 48 + 7) / 8 * 8 * 4;
      report("slab4, any XCD (layout without affinity), 8", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<4, 10, 8, false>), dim3(G2), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X4, d_o4, d_mis); }), 4.0 * ne * 12 / 1e6);
    }
    { const int G = slab_grid(2, 20);
      report("slab2 (160-byte pieces) XCD-affine, 8 per trip", time_us(st, reps, [&] { hipLaunchKernelGGL((k_slab<2, 20, 8, true>), dim3(G), dim3(TPB), 0, st, n, d_ptr, d_col, d_val, d_X2, d_o2, d_mis); }), 2.0 * ne * 12 / 1e6);
      std::vector<double> o((size_t)n * r), o2((size_t)n * r);
      HC(hipMemcpy(o.data(), d_o, vb, hipMemcpyDeviceToHost)); HC(hipMemcpy(o2.data(), d_o2, vb, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int s = 0; s < 2; ++s) for (int q = 0; q < n; ++q) for (int c = 0; c < 20; ++c)
          worst = std::fmax(worst, std::fabs(o2[((size_t)s * n + q) * 20 + c] - o[(size_t)q * r + s * 20 + c]));
      printf("    slab2 against rowmajor: largest difference %.2e\n", worst);
    }
    int mis = 0;
    HC(hipMemcpy(&mis, d_mis, sizeof(int), hipMemcpyDeviceToHost));
    printf("workgroups whose XCC_ID was not blockIdx %% 8 (all affine launches together): %d\n", mis);
    return 0;
}
