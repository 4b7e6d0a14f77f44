// Micro-benchmark: HBM bandwidth of a streaming kernel as a function of the number of separate
// arrays it reads and writes at once (same bytes per element in total or not), MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o streams streams.hip ; ./streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Ptrs { const double *r[24]; double *w[8]; };

template <int NR, int NW, int VEC>
__global__ __launch_bounds__(256) void k_streams(Ptrs p, long long n) {
    // VEC doubles per thread and per array, contiguous (VEC = 1, 2, 4)
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= n) return;
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += p.r[a][i + v];
    }
#pragma unroll
    for (int a = 0; a < NW; ++a) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) p.w[a][i + v] = acc[v] + (double)a;
    }
}

template <int NR, int NW, int VEC>
int run(const Ptrs &p, long long n, const char *note) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n / VEC + 255) / 256);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_streams<NR, NW, VEC>), dim3(blocks), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_streams<NR, NW, VEC>), dim3(blocks), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(NR + NW) * 8.0 * n;
    printf("reads %2d writes %2d vec %d : %7.3f ms  %6.0f GB/s  %s\n", NR, NW, VEC, ms, bytes / ms * 1e-6, note);
    fflush(stdout);
    return 0;
}

int main() {
    const long long n = 4097ll * 4097 * 4;           // cell-layers of the headline case
    Ptrs p;
    for (int a = 0; a < 24; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); CK(hipMemset(q, 0, n * 8)); p.r[a] = q; }
    for (int a = 0; a < 8; ++a) { double *q; CK(hipMalloc(&q, n * 8 + 64)); p.w[a] = q; }
    run<1, 1, 1>(p, n, "copy");
    run<5, 2, 1>(p, n, "update_h shape");
    run<3, 4, 1>(p, n, "mont+visc shape");
    run<8, 3, 1>(p, n, "");
    run<14, 6, 1>(p, n, "u+v shape");
    run<16, 6, 1>(p, n, "");
    run<20, 6, 1>(p, n, "");
    run<14, 6, 2>(p, n, "u+v shape, 16 B per lane and array");
    run<14, 6, 4>(p, n, "u+v shape, 32 B per lane and array");
    run<7, 3, 2>(p, n, "half the arrays, pairs interleaved (same bytes as 14/6 vec 1)");
    run<4, 2, 4>(p, n, "quads interleaved (close to 14/6 bytes)");
    run<5, 2, 2>(p, n, "");
    run<3, 4, 2>(p, n, "");
    return 0;
}
