// Development micro-benchmark: does the FP64 issue rate of a lone wavefront depend on how many of
// its lanes are active? (one wavefront per CU, `active` lanes run the chain, the others exit)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lanes_rate tools/micro/lanes_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, bool DPP>
__global__ void __launch_bounds__(64) chain(double *out, int iters, double a, double b, unsigned long long mask)
{
    if (!((mask >> threadIdx.x) & 1)) return;
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                x[i] = x[i] * a + b;
                if (DPP) {
                    int lo = __builtin_amdgcn_mov_dpp(__double2loint(x[i]), 0xB1, 0xf, 0xf, false);
                    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x[i]), 0xB1, 0xf, 0xf, false);
                    x[i] += __hiloint2double(hi, lo) * 1e-9;
                }
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, bool DPP>
static void run(int nblocks, unsigned long long active)
{
    double *out;
    hipMalloc(&out, sizeof(double) * 64 * nblocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chain<ILP, DPP>), dim3(nblocks), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, active);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * ILP * (DPP ? 4 : 1);
    printf("ILP=%d dpp=%d waves=%4d lane mask %016llx: %.3f ms, %.2f clocks per instruction\n", ILP, (int)DPP, nblocks,
           active, ms, ms * 1e-3 * 2.4e9 / n);
    hipFree(out);
}

int main()
{
    const unsigned long long R = 0xffffull;
    const unsigned long long masks[] = {
        0xfull, R, R << 16, R << 32, R << 48, R | R << 16, R | R << 32, R | R << 48, R << 16 | R << 32, R << 32 | R << 48,
        R | R << 16 | R << 32, R | R << 32 | R << 48, ~0ull,
        0x1ull | 1ull << 16 | 1ull << 32 | 1ull << 48,      // one lane per row
        0xfull | 0xfull << 16 | 0xfull << 32 | 0xfull << 48, // one quad per row
        0xfull | 0xfull << 32,                                // one quad in rows 0 and 2
        0x1ull | 1ull << 63, 0x1ull | 1ull << 33, 0x1ull << 40,
        0xffffffffull | 1ull << 32, 0xffffffffull | 1ull << 48, 0xffull | 0xffull << 16 | 0xffull << 32 | 0xffull << 48};
    for (unsigned long long m : masks) {
        run<1, false>(256, m);
        run<4, false>(256, m);
    }
    return 0;
}
