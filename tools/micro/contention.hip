// Development micro-benchmark: which instruction classes of the quad integrator slow down when
// four wavefronts (one per SIMD) share a CU instead of one? Each pattern is timed at 256 wavefronts
// (one per CU) and 1024 (one per SIMD), all 64 lanes active.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/micro/contention tools/micro/contention.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double dpp_xor1(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int P>
__global__ void __launch_bounds__(64) pattern(double *out, int iters, double a, double b, const int *flags)
{
    double x = threadIdx.x * 1e-3 + 1.0, y = x + 0.5;
    const int f = flags[threadIdx.x];      // all ones at run time, unknown at compile time
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (P == 0) {            // baseline: 4 dependent FMAs
                x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b);
            } else if (P == 1) {     // divergent-style branch around 4 FMAs (exec masking + s_cbranch)
                if ((f >> (r & 7)) & 1) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
                asm volatile("" : "+v"(x));
            } else if (P == 2) {     // DPP butterfly: 2 x (2 dpp movs + add)
                x = fma(x, a, b);
                asm volatile("" : "+v"(x));
                x += dpp_xor1(x);
                asm volatile("" : "+v"(x));
                x += dpp_xor1(x);
                x = fma(x, a, b);
            } else if (P == 3) {     // f64 reciprocal + 2 Newton steps
                double rr = __builtin_amdgcn_rcp(x);
                rr = fma(fma(-x, rr, 1.0), rr, rr);
                rr = fma(fma(-x, rr, 1.0), rr, rr);
                x = rr + a;
            } else if (P == 4) {     // selects on a per-lane condition
                const bool c = (f >> (r & 7)) & 1;
                x = fma(x, a, b);
                y = fma(y, b, a);
                const double t = c ? x : y;
                y = c ? y : x;
                x = t;
            } else if (P == 5) {     // f32 transcendental chain
                float s = (float)x;
                s = __builtin_amdgcn_rcpf(s) + 1.5f;
                s = __builtin_amdgcn_sqrtf(s) + 0.25f;
                s = __builtin_amdgcn_rsqf(s);
                x = fma(x, a, (double)s);
            } else if (P == 6) {     // v_cmp -> scalar mask arithmetic (vcc to SALU and back)
                const bool c1 = x > a, c2 = y < b;
                x = fma(x, a, b);
                y = fma(y, b, a);
                if (__builtin_amdgcn_ballot_w64(c1 && c2) == 0x12345ull) x += 1.0;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y;
}

// dependent gather from a 1.6 MB table (L2-resident) + 40 dependent FMAs per load, and one 64-B
// row store per lane-quad every iteration: the memory pattern of the integrator
__global__ void __launch_bounds__(64) gather(double *out, int iters, double a, double b, const double *tab,
                                             double *rows, int do_store)
{
    double x = threadIdx.x * 1e-3 + 1.0;
    unsigned idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    double *myrows = rows + (size_t)(blockIdx.x * 64 + threadIdx.x) * 8;
    for (int it = 0; it < iters; it++) {
        idx = idx * 1664525u + 1013904223u;
        const double2 v = *(const double2 *)(tab + ((idx >> 8) % 200000u) * 2);
        x = fma(x, a, v.x * 1e-30 + v.y * 1e-30);
#pragma unroll
        for (int r = 0; r < 40; r++) x = fma(x, a, b);
        if (do_store) { myrows[0] = x; myrows[1] = x; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

static void run_gather(const char *name, int do_store)
{
    double *tab, *rows;
    hipMalloc(&tab, 400000 * sizeof(double));
    hipMemset(tab, 0, 400000 * sizeof(double));
    for (int nb : {256, 1024, 2048}) {
        double *out;
        hipMalloc(&out, sizeof(double) * 64 * nb);
        hipMalloc(&rows, sizeof(double) * 64 * nb * 8);
        const int iters = 20000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3(nb), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, tab, rows, do_store);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-28s waves=%5d: %8.3f ms, %7.1f clocks per iteration\n", name, nb, ms,
               ms * 1e-3 * 2.4e9 / (double)iters);
        hipFree(out); hipFree(rows);
    }
}

// large branchy body: 512 guarded blocks (about 25 KB of code); `flags` decides which are skipped
__global__ void __launch_bounds__(64) branchy(double *out, int iters, double a, double b, const int *flags)
{
    double x = threadIdx.x * 1e-3 + 1.0;
    const int f0 = flags[threadIdx.x], f1 = flags[64 + threadIdx.x];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 512; r++) {
            const int f = (r & 32) ? f1 : f0;
            if ((f >> (r & 31)) & 1) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
            asm volatile("" : "+v"(x));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

static void run_branchy(const char *name, const int *flags)
{
    for (int nb : {256, 1024, 2048}) {
        double *out;
        hipMalloc(&out, sizeof(double) * 64 * nb);
        const int iters = 250;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(branchy, dim3(nb), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, flags);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-28s waves=%5d: %8.3f ms, %7.1f clocks per guarded block\n", name, nb, ms,
               ms * 1e-3 * 2.4e9 / ((double)iters * 512));
        hipFree(out);
    }
}

template <int P>
static void run(const char *name, const int *flags)
{
    for (int nb : {256, 1024, 2048}) {
        double *out;
        hipMalloc(&out, sizeof(double) * 64 * nb);
        const int iters = 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((pattern<P>), dim3(nb), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, flags);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-28s waves=%5d: %8.3f ms, %7.1f clocks per pattern\n", name, nb, ms,
               ms * 1e-3 * 2.4e9 / ((double)iters * 32));
        hipFree(out);
    }
}

int main()
{
    int h[64], *d;
    for (int i = 0; i < 64; i++) h[i] = 0xff;
    hipMalloc(&d, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    {
        int hb[128], *db;
        hipMalloc(&db, sizeof(hb));
        for (int i = 0; i < 128; i++) hb[i] = -1;
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        run_branchy("big body, all taken", db);
        for (int i = 0; i < 128; i++) hb[i] = 0x55555555;
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        run_branchy("big body, every 2nd skipped", db);
        for (int i = 0; i < 128; i++) hb[i] = 0x11111111;
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        run_branchy("big body, 3 of 4 skipped", db);
    }
    run_gather("gather + 40 FMAs", 0);
    run_gather("gather + 40 FMAs + store", 1);
    run<0>("4 dependent FMAs", d);
    run<1>("branch around 4 FMAs", d);
    run<2>("FMA + 2 DPP adds + FMA", d);
    run<3>("rcp_f64 + 2 Newton", d);
    run<4>("2 FMAs + 2 f64 selects", d);
    run<5>("f32 rcp/sqrt/rsq + FMA", d);
    run<6>("2 cmp + ballot + 2 FMAs", d);
    return 0;
}
