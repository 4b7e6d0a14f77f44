// Variants of tools/micro/march.hip to find what limits a row-marching sweep: reads only, writes
// only, no per-row barrier, two columns per lane (16-byte accesses, 128-column strips), two rows of
// prefetch.   hipcc --offload-arch=gfx950 -O3 -o march2 march2.hip ; ./march2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Ptrs { const double *r[16]; double *w[8]; };

// V = columns per lane (1 or 2); PF = rows of prefetch (1 or 2); BAR = per-row barrier + LDS coupling
template <int NL, int NR, int NW, int V, int PF, bool BAR>
__global__ __launch_bounds__(64 * NL) void k_march(Ptrs p, int L, int M, int rows_per_seg, long long n1) {
    __shared__ double s_col[2][NL][64 * V];
    const int lane = threadIdx.x & 63, lay = threadIdx.x >> 6;
    constexpr int W = 64 * V;
    const int nstrip = (L + W - 1) / W;
    const int strip = blockIdx.x % nstrip, seg = blockIdx.x / nstrip;
    int i = strip * W + lane * V;
    if (i + V > L) i = L - V;                        // last strip: shifted left (overlap)
    const int j0 = seg * rows_per_seg, j1 = min(M, j0 + rows_per_seg);
    const long long base = (long long)lay * n1 + i;
    double buf[PF + 1][NR > 0 ? NR : 1][V];
    for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) buf[q][a][v] = p.r[a][base + (long long)min(j0 + q, M - 1) * L + v];
    double carry = 0.0;
    for (int j = j0; j < j1; ++j) {
        const int jn = min(j + PF, M - 1);
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) buf[PF][a][v] = p.r[a][base + (long long)jn * L + v];
        double s[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { s[v] = (double)j; for (int a = 0; a < NR; ++a) s[v] += buf[0][a][v]; }
        const double e = __shfl_down(s[0], 1, 64);
        double col = 0.0;
        if (BAR) {
#pragma unroll
            for (int v = 0; v < V; ++v) s_col[j & 1][lay][lane * V + v] = s[v];
            __syncthreads();
#pragma unroll
            for (int l = 0; l < NL; ++l) col += s_col[j & 1][l][lane * V];
        }
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) p.w[a][base + (long long)j * L + v] = s[v] - e + col + carry + (double)a;
        carry = s[0];
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int a = 0; a < NR; ++a)
#pragma unroll
                for (int v = 0; v < V; ++v) buf[q][a][v] = buf[q + 1][a][v];
    }
}

template <int NL, int NR, int NW, int V, int PF, bool BAR>
int run(const Ptrs &p, int L, int M, int rows_per_seg, const char *note) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int W = 64 * V, nstrip = (L + W - 1) / W, nseg = (M + rows_per_seg - 1) / rows_per_seg;
    const long long n1 = (long long)L * M;
    const unsigned blocks = (unsigned)(nstrip * nseg);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, V, PF, BAR>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, M, rows_per_seg, n1);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, V, PF, BAR>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, M, rows_per_seg, n1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(NR + NW) * 8.0 * n1 * NL;
    printf("NL %d R %2d W %d cols/lane %d prefetch %d barrier %d rows/seg %4d blocks %5u : %7.3f ms  %6.0f GB/s  %s\n", NL, NR, NW, V, PF,
           (int)BAR, rows_per_seg, blocks, ms, bytes / ms * 1e-6, note);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 4097, M = 4097;
    printf("row pitch %d doubles\n", L);
    const long long n = (long long)L * M * 4;
    Ptrs p;
    if (argc > 2) {      // tile-shaped only
        for (int a = 0; a < 16; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); CK(hipMemset(q, 0, n * 8)); p.r[a] = q; }
        for (int a = 0; a < 8; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); p.w[a] = q; }
        run<4, 13, 8, 1, 1, false>(p, L, M, 8, "segments 8 (= a 64x8 tile per wave)");
        run<4, 14, 6, 1, 1, false>(p, L, M, 8, "u+v shape, 64x8 tile per wave");
        run<4, 5, 2, 1, 1, false>(p, L, M, 8, "update_h shape, 64x8 tile per wave");
        run<4, 3, 4, 1, 1, false>(p, L, M, 8, "mont+visc shape, 64x8 tile per wave");
        run<4, 0, 8, 1, 1, false>(p, L, M, 8, "writes only");
        run<4, 13, 8, 1, 1, false>(p, L, M, 128, "march 128 rows");
        run<4, 13, 8, 1, 1, true>(p, L, M, 512, "march 512 rows, barrier");
        return 0;
    }
    for (int a = 0; a < 16; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); CK(hipMemset(q, 0, n * 8)); p.r[a] = q; }
    for (int a = 0; a < 8; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); p.w[a] = q; }
    run<4, 13, 8, 1, 1, true>(p, L, M, 128, "base");
    run<4, 13, 8, 1, 1, false>(p, L, M, 128, "no barrier");
    run<4, 13, 0, 1, 1, false>(p, L, M, 128, "reads only");
    run<4, 0, 8, 1, 1, false>(p, L, M, 128, "writes only");
    run<4, 13, 8, 1, 2, false>(p, L, M, 128, "prefetch 2 rows");
    run<4, 13, 8, 2, 1, false>(p, L, M, 128, "2 columns per lane");
    run<4, 13, 8, 2, 1, true>(p, L, M, 128, "2 columns per lane, barrier");
    run<4, 13, 8, 2, 1, false>(p, L, M, 32, "2 columns per lane, short segments");
    run<4, 13, 8, 1, 1, false>(p, L, M, 16, "short segments 16");
    run<4, 13, 8, 1, 1, false>(p, L, M, 8, "segments 8 (= a 64x8 tile per wave)");
    run<1, 13, 8, 1, 1, false>(p, L, M * 4, 8, "1 wave per block, segments 8");
    run<4, 13, 0, 2, 1, false>(p, L, M, 128, "reads only, 2 columns per lane");
    return 0;
}
