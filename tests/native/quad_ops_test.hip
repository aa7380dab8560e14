// tests/native/quad_ops_test.hip -- development check of the DPP backend of sonic_quad.hpp:
// evaluates quad_rhs, the butterfly sum, the neighbour swap, the error norm and store_row on the
// device backend and on the 4-array emulation backend for the same inputs and prints the largest
// differences.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/native/quad_ops_test tests/native/quad_ops_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../../pysonic_amd/csrc/sonic_quad.hpp"
using namespace sonic;

constexpr int NOUT = 12;   // fQ fg Jqq Jqg Jgq Dg kQ kg sum swap es spare

__global__ void quad_ops_kernel(const double *rec, CorticalParams P, double q, const double *x4,
                                double *out, double *rows)
{
    typedef QuadOpsDev O;
    QuadCell<O> S;
    S.xlo = rec[0]; S.xhi = rec[1]; S.vv = rec[2]; S.vs = rec[3];
    O::load_gate_lines(rec, S.av, S.as, S.bv, S.bs);
    const QuadConsts<O> C = quad_consts<O>(P);
    const int quad = threadIdx.x >> 2;
    const double x = x4[threadIdx.x & 3] + 1e-3 * quad;
    double fg, r, gpw, other, drive;
    quad_rhs<O>(S, C, q, x, fg, r, gpw, other, drive);
    const double cond = gpw * other;
    const double fQ = O::allsum(cond * drive), Jqq = S.vs * O::allsum(cond);
    const double Jqg = r, Jgq = gpw, Dg = other, kQ = drive, kg = fg;
    double *o = out + threadIdx.x * NOUT;
    o[0] = fQ; o[1] = fg; o[2] = Jqq; o[3] = Jqg; o[4] = Jgq; o[5] = Dg; o[6] = kQ; o[7] = kg;
    o[8] = O::allsum(x); o[9] = O::swap1(x);
    o[10] = O::errsum(kg, x, x + kg, 1e-8f, 1e-6f);
    o[11] = 0.0;
    O::store_row(rows + quad * 8, 1.5 + quad, 1.0, q, x, -71.9);
}

int main()
{
    double rec[QUAD_REC] = {-71.9e-5, -70.9e-5, -71.9, 1.0e5};
    for (int g = 0; g < 4; g++) {
        rec[4 + 4 * g] = 100.0 + 37.0 * g; rec[5 + 4 * g] = 1e6 * (g + 1);
        rec[6 + 4 * g] = 2500.0 - 11.0 * g; rec[7 + 4 * g] = -2e6 * (g + 1);
    }
    CorticalParams P{560.0, 50.0, 60.0, -90.0, 0.75, 0.205, -70.3};
    const double x4[4] = {0.03, 0.6, 0.05, 0.04}, q = -71.5e-5;
    double *d_rec, *d_x4, *d_out, *d_rows;
    hipMalloc(&d_rec, sizeof(rec)); hipMalloc(&d_x4, sizeof(x4));
    hipMalloc(&d_out, 64 * NOUT * sizeof(double)); hipMalloc(&d_rows, 16 * 8 * sizeof(double));
    hipMemcpy(d_rec, rec, sizeof(rec), hipMemcpyHostToDevice);
    hipMemcpy(d_x4, x4, sizeof(x4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(quad_ops_kernel, dim3(1), dim3(64), 0, 0, d_rec, P, q, d_x4, d_out, d_rows);
    double out[64 * NOUT], rows[16 * 8];
    if (hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { puts("hip error"); return 2; }
    hipMemcpy(rows, d_rows, sizeof(rows), hipMemcpyDeviceToHost);

    typedef QuadOpsHost O;
    double worst[NOUT] = {0}, worst_row = 0;
    for (int quad = 0; quad < 16; quad++) {
        QuadCell<O> S;
        S.xlo = rec[0]; S.xhi = rec[1]; S.vv = rec[2]; S.vs = rec[3];
        O::load_gate_lines(rec, S.av, S.as, S.bv, S.bs);
        const QuadConsts<O> C = quad_consts<O>(P);
        O::V x;
        for (int g = 0; g < 4; g++) x.v[g] = x4[g] + 1e-3 * quad;
        O::V fg, r, gpw, other, drive;
        quad_rhs<O>(S, C, q, x, fg, r, gpw, other, drive);
        const O::V cond = O::mul(gpw, other);
        const double fQ = O::allsum(O::mul(cond, drive)), Jqq = S.vs * O::allsum(cond);
        const O::V Jqg = r, Jgq = gpw, Dg = other, kg = fg;
        const O::V kQv = drive;
        const float es = O::errsum(kg, x, O::add(x, kg), 1e-8f, 1e-6f);
        const O::V sw = O::swap1(x);
        double hrow[8];
        O::store_row(hrow, 1.5 + quad, 1.0, q, x, -71.9);
        for (int g = 0; g < 4; g++) {
            const double *o = out + (quad * 4 + g) * NOUT;
            const double ref[NOUT] = {fQ, fg.v[g], Jqq, Jqg.v[g], Jgq.v[g], Dg.v[g], kQv.v[g], kg.v[g],
                                      O::allsum(x), sw.v[g], es, 0.0};
            for (int i = 0; i < NOUT; i++) {
                const double e = fabs(o[i] - ref[i]) / fmax(1e-300, fabs(ref[i]));
                if (!(e <= worst[i])) worst[i] = e;
            }
        }
        for (int i = 0; i < 8; i++) {
            const double e = fabs(rows[quad * 8 + i] - hrow[i]);
            if (!(e <= worst_row)) worst_row = e;
        }
    }
    const char *names[NOUT] = {"fQ", "fg", "Jqq", "r", "gpw", "other", "drive", "fg2", "allsum", "swap1",
                               "errsum", "-"};
    int bad = 0;
    for (int i = 0; i < NOUT - 1; i++) {
        printf("%-7s max rel diff %.3e\n", names[i], worst[i]);
        if (!(worst[i] < 2e-6)) bad = 1;
    }
    printf("store_row max abs diff %.3e\n", worst_row);
    if (!(worst_row < 1e-15)) bad = 1;
    puts(bad ? "FAIL" : "OK");
    return bad;
}
