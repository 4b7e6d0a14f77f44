// Row-marching sweep, third micro-benchmark: the layout a real kernel would use.  Block = NL waves (the
// layers of one 64-lane strip, coupled per row through LDS + a barrier); lanes HALO..63-HALO own their
// column (stores), the others only load; strips overlap by 2*HALO columns; blocks are dealt to the XCDs
// in bands of row segments (blockIdx & 7 = XCD) so that neighbouring strips share an L2: their halo loads
// hit it and the cache lines two strips write in part are merged before they go to HBM.
//   hipcc --offload-arch=gfx950 -O3 -o march3 march3.hip ; ./march3 [pitch]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Ptrs { const double *r[16]; double *w[8]; };

template <int NL, int NR, int NW, int HALO, bool XCD>
__global__ __launch_bounds__(64 * NL) void k_march(Ptrs p, int L, int P, int M, int rows_per_seg, long long n1) {
    __shared__ double s_col[2][NL][64];
    const int lane = threadIdx.x & 63, lay = threadIdx.x >> 6;
    constexpr int OWN = 64 - 2 * HALO;
    const int nstrip = (L + OWN - 1) / OWN, nseg = (M + rows_per_seg - 1) / rows_per_seg;
    int strip, seg;
    if (XCD) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int spx = (nseg + 7) / 8;                     // segments per XCD
        const int sib = k / nstrip;
        strip = k - sib * nstrip;
        seg = xcd * spx + sib;
        if (sib >= spx || seg >= nseg) return;
    } else { strip = blockIdx.x % nstrip; seg = blockIdx.x / nstrip; }
    int i = strip * OWN - HALO + lane;                       // 0-based column
    const bool in = i >= 0 && i < L;
    const bool own = in && lane >= HALO && lane < 64 - HALO;
    const int j0 = seg * rows_per_seg, j1 = min(M, j0 + rows_per_seg);
    const long long base = (long long)lay * n1 + (in ? i : 0);
    double cur[NR], nxt[NR];
#pragma unroll
    for (int a = 0; a < NR; ++a) cur[a] = p.r[a][base + (long long)j0 * P];
    double carry1 = 0.0, carry2 = 0.0;
    for (int j = j0; j < j1; ++j) {
        const int jn = min(j + 1, M - 1);
#pragma unroll
        for (int a = 0; a < NR; ++a) nxt[a] = p.r[a][base + (long long)jn * P];
        double s = (double)j;
#pragma unroll
        for (int a = 0; a < NR; ++a) s += cur[a];
        const double e = __shfl_down(s, 1, 64), w = __shfl_up(s, 1, 64);
        s_col[j & 1][lay][lane] = s;
        __syncthreads();
        double col = 0.0;
#pragma unroll
        for (int l = 0; l < NL; ++l) col += s_col[j & 1][l][lane];
        const double v = (s - e) + (s - w) + col + carry1 - carry2;
        carry2 = carry1; carry1 = s;
        if (own) {
#pragma unroll
            for (int a = 0; a < NW; ++a) p.w[a][base + (long long)j * P] = v + (double)a;
        }
#pragma unroll
        for (int a = 0; a < NR; ++a) cur[a] = nxt[a];
    }
}

template <int NL, int NR, int NW, int HALO, bool XCD>
int run(const Ptrs &p, int L, int P, int M, int rows_per_seg, const char *note) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int OWN = 64 - 2 * HALO;
    const int nstrip = (L + OWN - 1) / OWN, nseg = (M + rows_per_seg - 1) / rows_per_seg;
    const long long n1 = (long long)P * M;
    const unsigned blocks = XCD ? (unsigned)(8 * ((nseg + 7) / 8) * nstrip) : (unsigned)(nstrip * nseg);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, HALO, XCD>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, P, M, rows_per_seg, n1);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, HALO, XCD>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, P, M, rows_per_seg, n1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(NR + NW) * 8.0 * (double)L * M * NL;
    printf("NL %d R %2d W %d halo %d xcd %d rows/seg %4d blocks %5u : %7.3f ms  %6.0f GB/s (owned bytes)  %s\n", NL, NR, NW, HALO, (int)XCD,
           rows_per_seg, blocks, ms, bytes / ms * 1e-6, note);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const int L = 4097, M = 4097;
    const int P = argc > 1 ? atoi(argv[1]) : 4112;
    printf("row pitch %d doubles\n", P);
    const long long n = (long long)P * M * 4;
    Ptrs p;
    for (int a = 0; a < 16; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 256)); CK(hipMemset(q, 0, n * 8)); p.r[a] = q; }
    for (int a = 0; a < 8; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 256)); p.w[a] = q; }
    run<4, 13, 8, 3, true>(p, L, P, M, 128, "whole step, halo 3, XCD bands");
    run<4, 13, 8, 3, false>(p, L, P, M, 128, "whole step, halo 3, plain order");
    run<4, 13, 8, 3, true>(p, L, P, M, 64, "");
    run<4, 13, 8, 3, true>(p, L, P, M, 256, "");
    run<4, 13, 8, 0, true>(p, L, P, M, 128, "no halo");
    run<4, 13, 8, 8, true>(p, L, P, M, 128, "halo 8 (48 owned: whole lines)");
    run<4, 15, 8, 3, true>(p, L, P, M, 128, "15 reads (with fcor, h_th)");
    run<4, 19, 8, 3, true>(p, L, P, M, 128, "19 reads (with nudging terms)");
    return 0;
}
