// Micro-benchmark for a ROW-MARCHING fused time step ("one sweep"): does the access
// pattern such a kernel would have reach streaming bandwidth?  One wave = one 64-column strip of one
// layer, marching north over a segment of rows; a block = NL waves (the layers of the same strip,
// coupled once per row through LDS + a barrier, as the Montgomery potential couples them); per row
// and lane NR coalesced 8-byte loads issued one row ahead, neighbour columns by wavefront shuffles,
// NW stores from the OWN lanes only (64 - 2*HALO columns).  No physics: sums and shuffles.
//   hipcc --offload-arch=gfx950 -O3 -o march march.hip ; ./march
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Ptrs { const double *r[16]; double *w[8]; };

template <int NL, int NR, int NW, int HALO>
__global__ __launch_bounds__(64 * NL) void k_march(Ptrs p, int L, int M, int rows_per_seg, long long n1) {
    __shared__ double s_col[2][NL][64];
    const int lane = threadIdx.x & 63, lay = threadIdx.x >> 6;
    constexpr int OWN = 64 - 2 * HALO;
    const int nstrip = (L + OWN - 1) / OWN;
    const int strip = blockIdx.x % nstrip, seg = blockIdx.x / nstrip;
    const int i = strip * OWN - HALO + lane;                    // column (0-based), halo lanes overlap the neighbours
    const int j0 = seg * rows_per_seg, j1 = min(M, j0 + rows_per_seg);
    const bool in = i >= 0 && i < L;
    const bool own = in && lane >= HALO && lane < 64 - HALO;
    const long long base = (long long)lay * n1 + (in ? i : 0);
    double cur[NR], nxt[NR];
#pragma unroll
    for (int a = 0; a < NR; ++a) cur[a] = p.r[a][base + (long long)j0 * L];
    double carry1 = 0.0, carry2 = 0.0;
    for (int j = j0; j < j1; ++j) {
        const int jn = min(j + 1, M - 1);
#pragma unroll
        for (int a = 0; a < NR; ++a) nxt[a] = p.r[a][base + (long long)jn * L];      // one row ahead
        double s = 0.0;
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
            for (int a = 0; a < NW; ++a) p.w[a][base + (long long)j * L] = v + (double)a;
        }
#pragma unroll
        for (int a = 0; a < NR; ++a) cur[a] = nxt[a];
    }
}

template <int NL, int NR, int NW, int HALO>
int run(const Ptrs &p, int L, int M, int rows_per_seg, const char *note) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int OWN = 64 - 2 * HALO;
    const int nstrip = (L + OWN - 1) / OWN, nseg = (M + rows_per_seg - 1) / rows_per_seg;
    const long long n1 = (long long)L * M;
    const unsigned blocks = (unsigned)(nstrip * nseg);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, HALO>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, M, rows_per_seg, n1);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_march<NL, NR, NW, HALO>), dim3(blocks), dim3(64 * NL), 0, 0, p, L, M, rows_per_seg, n1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(NR + NW) * 8.0 * n1 * NL;
    printf("NL %d reads %2d writes %d halo %d rows/seg %4d blocks %5u : %7.3f ms  %6.0f GB/s (owned bytes)  %s\n", NL, NR, NW, HALO,
           rows_per_seg, blocks, ms, bytes / ms * 1e-6, note);
    fflush(stdout);
    return 0;
}

int main() {
    const int L = 4097, M = 4097;
    const long long n = (long long)L * M * 4;
    Ptrs p;
    for (int a = 0; a < 16; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); CK(hipMemset(q, 0, n * 8)); p.r[a] = q; }
    for (int a = 0; a < 8; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); p.w[a] = q; }
    run<4, 13, 8, 3>(p, L, M, 512, "whole-step shape");
    run<4, 13, 8, 3>(p, L, M, 256, "");
    run<4, 13, 8, 3>(p, L, M, 128, "");
    run<4, 13, 8, 3>(p, L, M, 64, "");
    run<4, 13, 8, 3>(p, L, M, 1024, "");
    run<4, 13, 8, 4>(p, L, M, 256, "halo 4");
    run<4, 13, 8, 0>(p, L, M, 256, "no halo (upper bound)");
    run<4, 5, 2, 1>(p, L, M, 256, "update_h alone, marching");
    run<1, 13, 8, 3>(p, L, M * 4, 256, "one layer per block (no coupling), 4x rows");
    return 0;
}
