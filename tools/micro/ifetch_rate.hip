// Development micro-benchmark: does a LARGE straight-line loop body (8 B per instruction, ~16 KB)
// slow down when several wavefronts per CU stream it? (instruction-fetch bandwidth)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/ifetch_rate tools/micro/ifetch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BODY>   // BODY x 4 independent f64 FMAs per loop iteration
__global__ void __launch_bounds__(64) big_body(double *out, int iters, double a, double b)
{
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    // desynchronise the wavefronts: a block-dependent delay before the big body
    for (int d = 0; d < (int)((blockIdx.x * 2654435761u) >> 22); d++) x0 = fma(x0, a, b);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < BODY; r++) {
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int BODY>
static void run(int nblocks)
{
    double *out;
    hipMalloc(&out, sizeof(double) * 64 * nblocks);
    const int iters = 400000 / BODY;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((big_body<BODY>), dim3(nblocks), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * BODY * 4;
    printf("body %5d FMAs  waves=%5d: %.3f ms, %.2f clocks per wave-instruction\n", BODY * 4, nblocks, ms,
           ms * 1e-3 * 2.4e9 / instr);
    hipFree(out);
}

int main()
{
    for (int nb : {256, 1024, 2048}) {
        run<8>(nb);
        run<128>(nb);
        run<512>(nb);
    }
    return 0;
}
