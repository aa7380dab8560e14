// Development micro-benchmark: FP64 / FP32 FMA issue rate per CU as a function of the number of
// wavefronts per CU (one per SIMD up to 4, then two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/dp_rate tools/micro/dp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T, int ILP>
__global__ void __launch_bounds__(64) fma_chain(T *out, int iters, T a, T b)
{
    T x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = (T)threadIdx.x * (T)1e-3 + (T)i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] = x[i] * a + b;
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class T, int ILP>
static void run(const char *name, int nblocks, int lds_pad)
{
    T *out;
    hipMalloc(&out, sizeof(T) * 64 * nblocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fma_chain<T, ILP>), dim3(nblocks), dim3(64), lds_pad, 0, out, iters, (T)1.0000001, (T)1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * ILP;
    const double clocks = ms * 1e-3 * 2.4e9;
    printf("%-6s ILP=%d waves=%5d: %.3f ms, %.2f clocks per wave-instruction (per wave), %.1f G wave-instr/s total\n",
           name, ILP, nblocks, ms, clocks / instr_per_wave, nblocks * instr_per_wave / (ms * 1e-3) / 1e9);
    hipFree(out);
}

int main()
{
    for (int nb : {256, 512, 1024, 2048, 4096}) {
        run<double, 1>("f64", nb, 0);
        run<double, 4>("f64", nb, 0);
        run<float, 1>("f32", nb, 0);
        run<float, 4>("f32", nb, 0);
    }
    return 0;
}
