// tests/native/oct_ops_test.hip -- the device backend of full_coop.hpp (OctOpsDev: DPP moves, bank-masked
// broadcasts, shared exponential, two sums in one butterfly) against its 8-array emulation (OctOpsHost), which
// tests/test_cpu_cores.py holds to the oracle's fullDerivatives: one right-hand side per octet on the states of
// the input file, eight octets per wavefront (different states side by side, as in a packed launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o oct_ops_test tests/native/oct_ops_test.hip
//   oct_ops_test in.bin out.bin
//   in.bin  (float64): n, neuron, membrane, fs, bls[9], cortical parameters [sizeof(CorticalParams) / 8],
//                      then n x (y[8], pac)
//   out.bin (float64): n x 8 device results, n x 8 emulation results, n clamp flags (device), n (emulation)
// Used by tests/test_gpu_full.py::test_cooperative_rhs_device_against_emulation.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../pysonic_amd/csrc/full_coop.hpp"
using namespace sonic;

#define HIP_OK(call) do { if ((call) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #call); return 3; } } while (0)

template <bool MEMBRANE>
__global__ void __launch_bounds__(64) oct_rhs_kernel(const double *in, long n, BLSParams p, CorticalParams P, int neuron,
                                                     double fs, double *out, double *flags)
{
    typedef OctOpsDev O;
    const long idx = (long)blockIdx.x * 8 + (threadIdx.x >> 3);
    const int l = threadIdx.x & 7;
    const long src = idx < n ? idx : n - 1;                 // the octets past the end repeat the last state
    const CoopConsts<O> C = coop_consts<O>(p, P, neuron, 0.0);
    const CoopScalars<O> S = coop_scalars<O>(p, fs, 0.0);
    const double y = in[src * 9 + l];
    // the pressure term is owed to lane 0 only (coop_rhs): the other quad gets a NaN that nothing may consume
    const double pterm = l < 4 ? S.p0r - in[src * 9 + 8] * S.inv_rho : NAN;
    bool clamped = false;
    const double dy = coop_rhs<O, MEMBRANE>(C, S, y, pterm, clamped);
    if (idx < n) {
        out[idx * 8 + l] = dy;
        if (l == 0) flags[idx] = clamped ? 1.0 : 0.0;
    }
}

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<double> buf;
    double tmp[512];
    size_t k;
    while ((k = std::fread(tmp, sizeof(double), 512, f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
    std::fclose(f);
    const long n = (long)buf[0];
    const int neuron = (int)buf[1], membrane = (int)buf[2];
    const double fs = buf[3];
    BLSParams p;
    CorticalParams P;
    std::memcpy(&p, &buf[4], sizeof(p));
    std::memcpy(&P, &buf[4 + sizeof(p) / 8], sizeof(P));
    const double *states = &buf[4 + sizeof(p) / 8 + sizeof(P) / 8];
    if ((long)buf.size() != 4 + (long)(sizeof(p) + sizeof(P)) / 8 + 9 * n) { std::fprintf(stderr, "bad input size\n"); return 2; }

    double *d_in, *d_out, *d_fl;
    HIP_OK(hipMalloc((void **)&d_in, 9 * n * sizeof(double)));
    HIP_OK(hipMalloc((void **)&d_out, 8 * n * sizeof(double)));
    HIP_OK(hipMalloc((void **)&d_fl, n * sizeof(double)));
    HIP_OK(hipMemcpy(d_in, states, 9 * n * sizeof(double), hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n + 7) / 8);
    if (membrane) hipLaunchKernelGGL(oct_rhs_kernel<true>, dim3(grid), dim3(64), 0, 0, d_in, n, p, P, neuron, fs, d_out, d_fl);
    else hipLaunchKernelGGL(oct_rhs_kernel<false>, dim3(grid), dim3(64), 0, 0, d_in, n, p, P, neuron, fs, d_out, d_fl);
    HIP_OK(hipDeviceSynchronize());
    std::vector<double> dev(8 * n), dfl(n), emu(8 * n), efl(n);
    HIP_OK(hipMemcpy(dev.data(), d_out, 8 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dfl.data(), d_fl, n * sizeof(double), hipMemcpyDeviceToHost));

    typedef OctOpsHost O;
    const CoopConsts<O> C = coop_consts<O>(p, P, neuron, 0.0);
    const CoopScalars<O> S = coop_scalars<O>(p, fs, 0.0);
    for (long i = 0; i < n; i++) {
        O::V y, dy;
        for (int l = 0; l < OCT; l++) y.v[l] = states[i * 9 + l];
        bool clamped = false;
        const O::V pterm = O::splat(S.p0r - states[i * 9 + 8] * S.inv_rho);
        dy = membrane ? coop_rhs<O, true>(C, S, y, pterm, clamped) : coop_rhs<O, false>(C, S, y, pterm, clamped);
        for (int l = 0; l < OCT; l++) emu[i * 8 + l] = dy.v[l];
        efl[i] = clamped ? 1.0 : 0.0;
    }
    f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    std::fwrite(dev.data(), sizeof(double), dev.size(), f);
    std::fwrite(emu.data(), sizeof(double), emu.size(), f);
    std::fwrite(dfl.data(), sizeof(double), dfl.size(), f);
    std::fwrite(efl.data(), sizeof(double), efl.size(), f);
    std::fclose(f);
    return 0;
}
