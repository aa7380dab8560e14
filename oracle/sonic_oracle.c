/* ---------------------------------------------------------------------------------------------
 * oracle/sonic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the right-hand sides of the reference's (tjjlemaire/PySONIC) batched
 * ODE hot path, so that oracle/oracle.py can drive scipy.integrate.odeint (ODEPACK LSODA,
 * scipy 1.15.3 -- the same third-party integrator the reference calls at
 * PySONIC/core/solvers.py:166-167) without the reference's ~50 us/call Python overhead.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (pysonic_amd/) never does; it shares no source with this file.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Parity status: pinned -- oracle/oracle.py is checked against golden vectors captured from
 * the reference itself (tests/golden/, generator scripts committed alongside).
 * ------------------------------------------------------------------------------------------- */
#include <math.h>
#include <stddef.h>

/* PySONIC/constants.py:12-17 */
#define FARADAY 9.64853e4
#define RG 8.31342
#define Z_CA 2.0
#define CELSIUS_2_KELVIN 273.15

enum { ORC_RS = 0, ORC_FS = 1, ORC_LTS = 2, ORC_RE = 3, ORC_TC = 4, ORC_STN = 5, ORC_IB = 6,
       ORC_NNEURONS = 7 };

/* ------------------------------------------------------------------------------------------
 * np.interp(x, xp, fp, left=nan, right=nan) for scalar x, as used by
 * Lookup.interpVar1D (PySONIC/core/lookups.py:309-322). numpy's kernel
 * (numpy/_core/src/multiarray/compiled_base.c, arr_interp) finds j with xp[j] <= x < xp[j+1] by
 * binary search and returns slope*(x - xp[j]) + fp[j], slope = (fp[j+1]-fp[j])/(xp[j+1]-xp[j]);
 * x == xp[n-1] returns fp[n-1]; outside -> left/right.
 * ---------------------------------------------------------------------------------------- */
static int orc_bsearch(double x, const double *xp, int n)
{
    /* largest j such that xp[j] <= x, assuming xp[0] <= x < xp[n-1] */
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

double orc_interp(double x, const double *xp, const double *fp, int n)
{
    if (isnan(x)) return NAN;
    if (x < xp[0] || x > xp[n - 1]) return NAN;
    if (x == xp[n - 1]) return fp[n - 1];
    int j = orc_bsearch(x, xp, n);
    if (x == xp[j]) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return slope * (x - xp[j]) + fp[j];
}

/* ==========================================================================================
 * Point-neuron definitions (PySONIC/neurons/cortical.py, thalamic.py, stn.py)
 * ======================================================================================== */

/* PySONIC/core/pneuron.py:351-354 */
static double vtrap(double x, double y) { return x / (exp(x / y) - 1.0); }

/* ---- Cortical / Thalamic shared m, h, n kinetics (cortical.py:36-58, thalamic.py:31-53) ---- */
static double hh_alpham(double Vm, double VT) { return 0.32 * vtrap(13 - (Vm - VT), 4) * 1e3; }
static double hh_betam(double Vm, double VT) { return 0.28 * vtrap((Vm - VT) - 40, 5) * 1e3; }
static double hh_alphah(double Vm, double VT) { return 0.128 * exp(-((Vm - VT) - 17) / 18) * 1e3; }
static double hh_betah(double Vm, double VT) { return 4 / (1 + exp(-((Vm - VT) - 40) / 5)) * 1e3; }
static double hh_alphan(double Vm, double VT) { return 0.032 * vtrap(15 - (Vm - VT), 5) * 1e3; }
static double hh_betan(double Vm, double VT) { return 0.5 * exp(-((Vm - VT) - 10) / 40) * 1e3; }

/* cortical.py:60-66 */
static double ctx_pinf(double Vm) { return 1.0 / (1 + exp(-(Vm + 35) / 10)); }
static double ctx_taup(double Vm, double TauMax)
{
    return TauMax / (3.3 * exp((Vm + 35) / 20) + exp(-(Vm + 35) / 20));
}

/* cortical.py:254-272 (LTS) and thalamic.py:289-307 (TC): same forms, Vx differs */
static double lts_sinf(double Vm, double Vx) { return 1.0 / (1.0 + exp(-(Vm + Vx + 57.0) / 6.2)); }
static double lts_taus(double Vm, double Vx)
{
    double x = exp(-(Vm + Vx + 132.0) / 16.7) + exp((Vm + Vx + 16.8) / 18.2);
    return 1.0 / 3.7 * (0.612 + 1.0 / x) * 1e-3;
}
static double lts_uinf(double Vm, double Vx) { return 1.0 / (1.0 + exp((Vm + Vx + 81.0) / 4.0)); }
static double lts_tauu(double Vm, double Vx)
{
    if (Vm + Vx < -80.0)
        return 1.0 / 3.7 * exp((Vm + Vx + 467.0) / 66.6) * 1e-3;
    else
        return 1.0 / 3.7 * (exp(-(Vm + Vx + 22) / 10.5) + 28.0) * 1e-3;
}

/* thalamic.py:164-179 (RE) */
static double re_sinf(double Vm) { return 1.0 / (1.0 + exp(-(Vm + 52.0) / 7.4)); }
static double re_taus(double Vm)
{
    return (1 + 0.33 / (exp((Vm + 27.0) / 10.0) + exp(-(Vm + 102.0) / 15.0))) * 1e-3;
}
static double re_uinf(double Vm) { return 1.0 / (1.0 + exp((Vm + 80.0) / 5.0)); }
static double re_tauu(double Vm)
{
    return (28.3 + 0.33 / (exp((Vm + 48.0) / 4.0) + exp(-(Vm + 407.0) / 50.0))) * 1e-3;
}

/* cortical.py:356-370 (IB iCaL gates q, r: alpha / beta forms) */
static double ib_alphaq(double Vm) { return 0.055 * vtrap(-(Vm + 27.0), 3.8) * 1e3; }
static double ib_betaq(double Vm) { return 0.94 * exp(-(Vm + 75.0) / 17.0) * 1e3; }
static double ib_alphar(double Vm) { return 0.000457 * exp(-(Vm + 13.0) / 50.0) * 1e3; }
static double ib_betar(double Vm) { return 0.0065 / (exp(-(Vm + 15.0) / 28.0) + 1.0) * 1e3; }

/* thalamic.py:309-323 (TC iH gate) */
static double tc_oinf(double Vm) { return 1.0 / (1.0 + exp((Vm + 75.0) / 5.5)); }
static double tc_tauo(double Vm)
{
    return 1 / (exp(-14.59 - 0.086 * Vm) + exp(-1.87 + 0.0701 * Vm)) * 1e-3;
}

typedef struct {
    /* common */
    double Cm0, Vm0, ENa, EK, ECa, ELeak, gNabar, gKdbar, gLeak, VT;
    /* cortical */
    double gMbar, TauMax, gCaTbar, Vx;
    /* TC */
    double EH, gKLeak, gHbar, taur_Cai, Cai_min, deff, nCa, k1, k2, k3, k4;
} orc_pospischil_t;

static const orc_pospischil_t P_RS = {   /* cortical.py:122-160 */
    .Cm0 = 1e-2, .Vm0 = -71.9, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -70.3,
    .gNabar = 560.0, .gKdbar = 60.0, .gLeak = 0.205, .VT = -56.2, .gMbar = 0.75, .TauMax = 0.608 };
static const orc_pospischil_t P_FS = {   /* cortical.py:163-201 */
    .Cm0 = 1e-2, .Vm0 = -71.4, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -70.4,
    .gNabar = 580.0, .gKdbar = 39.0, .gLeak = 0.38, .VT = -57.9, .gMbar = 0.787, .TauMax = 0.502 };
static const orc_pospischil_t P_LTS = {  /* cortical.py:204-250 */
    .Cm0 = 1e-2, .Vm0 = -54.0, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -50.0,
    .gNabar = 500.0, .gKdbar = 40.0, .gLeak = 0.19, .VT = -50.0, .gMbar = 0.28, .TauMax = 4.0,
    .gCaTbar = 4.0, .Vx = -7.0 };
static const orc_pospischil_t P_IB = {   /* cortical.py:307-342; gCaTbar holds gCaLbar */
    .Cm0 = 1e-2, .Vm0 = -71.4, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -70.0,
    .gNabar = 500.0, .gKdbar = 50.0, .gLeak = 0.1, .VT = -56.2, .gMbar = 0.3, .TauMax = 0.608,
    .gCaTbar = 1.0 };
static const orc_pospischil_t P_RE = {   /* thalamic.py:117-160 */
    .Cm0 = 1e-2, .Vm0 = -89.5, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -90.0,
    .gNabar = 2000.0, .gKdbar = 200.0, .gLeak = 0.5, .VT = -67.0, .gCaTbar = 30.0 };
static const orc_pospischil_t P_TC = {   /* thalamic.py:182-250 */
    .Cm0 = 1e-2, .Vm0 = -61.93, .ENa = 50.0, .EK = -90.0, .ECa = 120.0, .ELeak = -70.0,
    .gNabar = 900.0, .gKdbar = 100.0, .gLeak = 0.1, .VT = -52.0, .gCaTbar = 20.0, .Vx = 0.0,
    .EH = -40.0, .gKLeak = 0.138, .gHbar = 0.175, .taur_Cai = 5e-3, .Cai_min = 50e-9,
    .deff = 100e-9, .nCa = 4, .k1 = 2.5e22, .k2 = 0.4, .k3 = 100.0, .k4 = 1.0 };

static const orc_pospischil_t *pospischil(int id)
{
    switch (id) {
    case ORC_RS: return &P_RS;
    case ORC_FS: return &P_FS;
    case ORC_LTS: return &P_LTS;
    case ORC_IB: return &P_IB;
    case ORC_RE: return &P_RE;
    case ORC_TC: return &P_TC;
    }
    return NULL;
}

/* pneuron.py:328-337 */
static double currentToConcentrationRate(double z_ion, double depth)
{
    return 1e-6 / (z_ion * depth * FARADAY);
}

/* ---------------- STN (stn.py:14-456) ---------------- */
/* stn.py:52 comments out its own celsius, so PointNeuron.celsius = 36.0 applies (pneuron.py:27-28) */
static const double stn_celsius = 36.0;

static double stn_xinf(double var, double theta, double k) { return 1 / (1 + exp((var - theta) / k)); }
static double stn_taux1(double Vm, double theta, double sigma, double tau0, double tau1)
{
    return tau0 + tau1 / (1 + exp(-(Vm - theta) / sigma));
}
static double stn_taux2(double Vm, double th1, double th2, double s1, double s2, double tau0, double tau1)
{
    return tau0 + tau1 / (exp(-(Vm - th1) / s1) + exp(-(Vm - th2) / s2));
}
/* stn.py:52-136 parameters */
static double stn_ainf(double V) { return stn_xinf(V, -45, -14.7); }
static double stn_binf(double V) { return stn_xinf(V, -90, 7.5); }
static double stn_cinf(double V) { return stn_xinf(V, -30.6, -5); }
static double stn_d1inf(double V) { return stn_xinf(V, -60, 7.5); }
static double stn_d2inf(double Cai) { return stn_xinf(Cai, 0.1e-6, 0.02e-6); }
static double stn_minf(double V) { return stn_xinf(V, -40, -8); }
static double stn_hinf(double V) { return stn_xinf(V, -45.5, 6.4); }
static double stn_ninf(double V) { return stn_xinf(V, -41, -14); }
static double stn_pinf(double V) { return stn_xinf(V, -56, -6.7); }
static double stn_qinf(double V) { return stn_xinf(V, -85, 5.8); }
static double stn_rinf(double Cai) { return stn_xinf(Cai, 0.17e-6, -0.08e-6); }
static double stn_taua(double V) { return stn_taux1(V, -40, -0.5, 1e-3, 1e-3); }
static double stn_taum(double V) { return stn_taux1(V, -53, -0.7, 0.2e-3, 3e-3); }
static double stn_taub(double V) { return stn_taux2(V, -60, -40, -30, 10, 0e-3, 200e-3); }
static double stn_tauc(double V) { return stn_taux2(V, -27, -50, -20, 15, 45e-3, 10e-3); }
static double stn_taud1(double V) { return stn_taux2(V, -40, -20, -15, 20, 400e-3, 500e-3); }
static double stn_tauh(double V) { return stn_taux2(V, -50, -50, -15, 16, 0e-3, 24.5e-3); }
static double stn_taun(double V) { return stn_taux2(V, -40, -40, -40, 50, 0e-3, 11e-3); }
static double stn_taup(double V) { return stn_taux2(V, -27, -102, -10, 15, 5e-3, 0.33e-3); }
static double stn_tauq(double V) { return stn_taux2(V, -50, -50, -15, 16, 0e-3, 400e-3); }
#define STN_TAU_D2 130e-3
#define STN_TAU_R 2e-3
#define STN_CAO 2e-3
#define STN_TAUR_CAI 0.5e-3
#define STN_CAI0 5e-9
#define STN_VM0 -58.0
static const double STN_ENa = 60.0, STN_EK = -90.0, STN_ELeak = -60.0;
static const double STN_gNabar = 490.0, STN_gLeak = 3.5, STN_gKdbar = 570.0, STN_gCaTbar = 50.0,
                    STN_gCaLbar = 150.0, STN_gAbar = 50.0, STN_gKCabar = 10.0;

static double stn_T(void) { return stn_celsius + CELSIUS_2_KELVIN; }

/* pneuron.py:339-349 */
static double nernst(double z_ion, double Cin, double Cout, double T)
{
    return (RG * T) / (z_ion * FARADAY) * log(Cout / Cin) * 1e3;
}
/* stn.py:400-415 */
static double stn_iCaT(double p, double q, double Vm, double Cai)
{
    return STN_gCaTbar * (p * p) * q * (Vm - nernst(Z_CA, Cai, STN_CAO, stn_T()));
}
static double stn_iCaL(double c, double d1, double d2, double Vm, double Cai)
{
    return STN_gCaLbar * (c * c) * d1 * d2 * (Vm - nernst(Z_CA, Cai, STN_CAO, stn_T()));
}
/* stn.py:198-207 getEffectiveDepth, evaluated once (stn.py:172-175 __new__) */
double orc_stn_deff(void)
{
    double Vm = STN_VM0, Cai = STN_CAI0;
    double iCaT = stn_iCaT(stn_pinf(Vm), stn_qinf(Vm), Vm, Cai);
    double iCaL = stn_iCaL(stn_cinf(Vm), stn_d1inf(Vm), stn_d2inf(Cai), Vm, Cai);
    return -(iCaT + iCaL) / (Z_CA * FARADAY * Cai / STN_TAUR_CAI) * 1e-6;
}
/* stn.py:340-343 */
double orc_stn_derCai(double p, double q, double c, double d1, double d2, double Cai, double Vm)
{
    double iCa_tot = stn_iCaT(p, q, Vm, Cai) + stn_iCaL(c, d1, d2, Vm, Cai);
    return -currentToConcentrationRate(Z_CA, orc_stn_deff()) * iCa_tot - Cai / STN_TAUR_CAI;
}

/* ------------------------------------------------------------------------------------------
 * Dimensions (states in `states` dict order, tables = V + effRates order)
 * ---------------------------------------------------------------------------------------- */
int orc_nstates(int id)
{
    static const int n[ORC_NNEURONS] = {4, 4, 6, 5, 9, 12, 6};
    return (id >= 0 && id < ORC_NNEURONS) ? n[id] : -1;
}
int orc_nrates(int id)
{
    static const int n[ORC_NNEURONS] = {8, 8, 12, 10, 12, 18, 12};
    return (id >= 0 && id < ORC_NNEURONS) ? n[id] : -1;
}
double orc_Cm0(int id) { (void)id; return 1e-2; }
double orc_Vm0(int id)
{
    if (id == ORC_STN) return STN_VM0;
    return pospischil(id)->Vm0;
}

/* ------------------------------------------------------------------------------------------
 * True (voltage-dependent) rate constants in the order of the reference's `effRates()` dict
 * (PySONIC/core/translators.py:287-327, 396-419): used by getEffRates (pneuron.py:268-271).
 *   RS/FS : alpham betam alphah betah alphan betan alphap betap
 *   LTS   : ... + alphas betas alphau betau
 *   IB    : ... + alphaq betaq alphar betar
 *   RE    : alpham betam alphah betah alphan betan alphas betas alphau betau
 *   TC    : RE order + alphao betao
 *   STN   : alphaa betaa alphab betab alphac betac alphad1 betad1 alpham betam alphah betah
 *           alphan betan alphap betap alphaq betaq
 * x_inf/tau gates contribute alpha = xinf/tau, beta = (1 - xinf)/tau (translators.py:317-320).
 * ---------------------------------------------------------------------------------------- */
#define INF_TAU(out, k, inf, tau) do { double _i = (inf), _t = (tau); \
    (out)[k] = _i / _t; (out)[(k) + 1] = (1 - _i) / _t; } while (0)

void orc_rates(int id, double Vm, double *out)
{
    if (id == ORC_STN) {
        INF_TAU(out, 0, stn_ainf(Vm), stn_taua(Vm));
        INF_TAU(out, 2, stn_binf(Vm), stn_taub(Vm));
        INF_TAU(out, 4, stn_cinf(Vm), stn_tauc(Vm));
        INF_TAU(out, 6, stn_d1inf(Vm), stn_taud1(Vm));
        INF_TAU(out, 8, stn_minf(Vm), stn_taum(Vm));
        INF_TAU(out, 10, stn_hinf(Vm), stn_tauh(Vm));
        INF_TAU(out, 12, stn_ninf(Vm), stn_taun(Vm));
        INF_TAU(out, 14, stn_pinf(Vm), stn_taup(Vm));
        INF_TAU(out, 16, stn_qinf(Vm), stn_tauq(Vm));
        return;
    }
    const orc_pospischil_t *p = pospischil(id);
    out[0] = hh_alpham(Vm, p->VT); out[1] = hh_betam(Vm, p->VT);
    out[2] = hh_alphah(Vm, p->VT); out[3] = hh_betah(Vm, p->VT);
    out[4] = hh_alphan(Vm, p->VT); out[5] = hh_betan(Vm, p->VT);
    switch (id) {
    case ORC_RS: case ORC_FS:
        INF_TAU(out, 6, ctx_pinf(Vm), ctx_taup(Vm, p->TauMax));
        break;
    case ORC_LTS:
        INF_TAU(out, 6, ctx_pinf(Vm), ctx_taup(Vm, p->TauMax));
        INF_TAU(out, 8, lts_sinf(Vm, p->Vx), lts_taus(Vm, p->Vx));
        INF_TAU(out, 10, lts_uinf(Vm, p->Vx), lts_tauu(Vm, p->Vx));
        break;
    case ORC_IB:
        INF_TAU(out, 6, ctx_pinf(Vm), ctx_taup(Vm, p->TauMax));
        out[8] = ib_alphaq(Vm); out[9] = ib_betaq(Vm);
        out[10] = ib_alphar(Vm); out[11] = ib_betar(Vm);
        break;
    case ORC_RE:
        INF_TAU(out, 6, re_sinf(Vm), re_taus(Vm));
        INF_TAU(out, 8, re_uinf(Vm), re_tauu(Vm));
        break;
    case ORC_TC:
        INF_TAU(out, 6, lts_sinf(Vm, p->Vx), lts_taus(Vm, p->Vx));
        INF_TAU(out, 8, lts_uinf(Vm, p->Vx), lts_tauu(Vm, p->Vx));
        /* thalamic.py:317-323: alphao = oinf/tauo, betao = (1-oinf)/tauo (explicit methods) */
        INF_TAU(out, 10, tc_oinf(Vm), tc_tauo(Vm));
        break;
    }
}

/* Vectorised helper for getEffRates: out[k*n + i] = rate_k(Vm[i]) */
void orc_rates_vec(int id, const double *Vm, int n, double *out)
{
    double r[32];
    int nr = orc_nrates(id);
    for (int i = 0; i < n; i++) {
        orc_rates(id, Vm[i], r);
        for (int k = 0; k < nr; k++) out[(size_t)k * n + i] = r[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * Net membrane current iNet(Vm, states) (pneuron.py:288-296; currents() of each neuron), mA/m2
 * ---------------------------------------------------------------------------------------- */
double orc_iNet(int id, double Vm, const double *x)
{
    if (id == ORC_STN) {   /* states: m h n a b p q c d1 d2 r Cai (stn.py:157-170, 417-430) */
        double m = x[0], h = x[1], n = x[2], a = x[3], b = x[4], p = x[5], q = x[6], c = x[7],
               d1 = x[8], d2 = x[9], r = x[10], Cai = x[11];
        double iNa = STN_gNabar * (m * m * m) * h * (Vm - STN_ENa);
        double iKd = STN_gKdbar * (n * n * n * n) * (Vm - STN_EK);
        double iA = STN_gAbar * (a * a) * b * (Vm - STN_EK);
        double iCaT = stn_iCaT(p, q, Vm, Cai);
        double iCaL = stn_iCaL(c, d1, d2, Vm, Cai);
        double iKCa = STN_gKCabar * (r * r) * (Vm - STN_EK);
        double iLeak = STN_gLeak * (Vm - STN_ELeak);
        /* python sum([...]) starts at 0 and adds left to right */
        return 0 + iNa + iKd + iA + iCaT + iCaL + iKCa + iLeak;
    }
    const orc_pospischil_t *P = pospischil(id);
    double m = x[0], h = x[1], n = x[2];
    double iNa = P->gNabar * (m * m * m) * h * (Vm - P->ENa);
    double iKd = P->gKdbar * (n * n * n * n) * (Vm - P->EK);
    double iLeak = P->gLeak * (Vm - P->ELeak);
    switch (id) {
    case ORC_RS: case ORC_FS: {   /* cortical.py:112-119 : iNa iKd iM iLeak */
        double iM = P->gMbar * x[3] * (Vm - P->EK);
        return 0 + iNa + iKd + iM + iLeak;
    }
    case ORC_IB:                  /* cortical.py:392-400 : + iCaL = gCaLbar q^2 r (Vm - ECa) */
    case ORC_LTS: {               /* cortical.py:299-303 : + iCaT */
        double iM = P->gMbar * x[3] * (Vm - P->EK);
        double iCaT = P->gCaTbar * (x[4] * x[4]) * x[5] * (Vm - P->ECa);
        return 0 + iNa + iKd + iM + iLeak + iCaT;
    }
    case ORC_RE: {                /* thalamic.py:107-114 : iNa iKd iCaT iLeak */
        double iCaT = P->gCaTbar * (x[3] * x[3]) * x[4] * (Vm - P->ECa);
        return 0 + iNa + iKd + iCaT + iLeak;
    }
    case ORC_TC: {                /* thalamic.py:352-366 : + iKLeak iH */
        double iCaT = P->gCaTbar * (x[3] * x[3]) * x[4] * (Vm - P->ECa);
        double O = x[7], C = x[8];
        double iKLeak = P->gKLeak * (Vm - P->EK);
        double iH = P->gHbar * (O + 2 * (1 - O - C)) * (Vm - P->EH);
        return 0 + iNa + iKd + iCaT + iLeak + iKLeak + iH;
    }
    }
    return NAN;
}

/* ------------------------------------------------------------------------------------------
 * Effective state derivatives given interpolated lookup values `lk` (lk[0] = V, lk[1..] rates
 * in effRates order) -- the translated lambdas of derEffStates (translators.py:355-372) with
 * tau = 1/(alpha+beta), xinf = alpha*tau derived on the fly (lookups.py:488-512).
 * ---------------------------------------------------------------------------------------- */
#define AB(k) (lk[1 + (k)] * (1 - x[i]) - lk[2 + (k)] * x[i])
static double inf_tau_der(double alpha, double beta, double xi)
{
    double tau = 1 / (alpha + beta);      /* lookups.py:494-495 */
    double inf = alpha * tau;             /* lookups.py:497-498 */
    return (inf - xi) / tau;
}

static void eff_dstates(int id, const double *lk, const double *x, double *dx)
{
    int i;
    if (id == ORC_STN) {
        /* state order m h n a b p q c d1 d2 r Cai ; table order a b c d1 m h n p q */
        double V = lk[0];
        dx[0] = inf_tau_der(lk[9], lk[10], x[0]);    /* m */
        dx[1] = inf_tau_der(lk[11], lk[12], x[1]);   /* h */
        dx[2] = inf_tau_der(lk[13], lk[14], x[2]);   /* n */
        dx[3] = inf_tau_der(lk[1], lk[2], x[3]);     /* a */
        dx[4] = inf_tau_der(lk[3], lk[4], x[4]);     /* b */
        dx[5] = inf_tau_der(lk[15], lk[16], x[5]);   /* p */
        dx[6] = inf_tau_der(lk[17], lk[18], x[6]);   /* q */
        dx[7] = inf_tau_der(lk[5], lk[6], x[7]);     /* c */
        dx[8] = inf_tau_der(lk[7], lk[8], x[8]);     /* d1 */
        dx[9] = (stn_d2inf(x[11]) - x[9]) / STN_TAU_D2;   /* d2 */
        dx[10] = (stn_rinf(x[11]) - x[10]) / STN_TAU_R;   /* r */
        dx[11] = orc_stn_derCai(x[5], x[6], x[7], x[8], x[9], x[11], V);
        return;
    }
    const orc_pospischil_t *P = pospischil(id);
    i = 0; dx[0] = AB(0);
    i = 1; dx[1] = AB(2);
    i = 2; dx[2] = AB(4);
    switch (id) {
    case ORC_RS: case ORC_FS:
        dx[3] = inf_tau_der(lk[7], lk[8], x[3]);
        break;
    case ORC_LTS:
        dx[3] = inf_tau_der(lk[7], lk[8], x[3]);
        dx[4] = inf_tau_der(lk[9], lk[10], x[4]);
        dx[5] = inf_tau_der(lk[11], lk[12], x[5]);
        break;
    case ORC_IB:                  /* q, r are alpha / beta gates (cortical.py:374-380) */
        dx[3] = inf_tau_der(lk[7], lk[8], x[3]);
        i = 4; dx[4] = AB(8);
        i = 5; dx[5] = AB(10);
        break;
    case ORC_RE:
        dx[3] = inf_tau_der(lk[7], lk[8], x[3]);
        dx[4] = inf_tau_der(lk[9], lk[10], x[4]);
        break;
    case ORC_TC: {
        double V = lk[0], alphao = lk[11], betao = lk[12];
        double s = x[3], u = x[4], Cai = x[5], P0 = x[6], O = x[7], C = x[8];
        dx[3] = inf_tau_der(lk[7], lk[8], s);
        dx[4] = inf_tau_der(lk[9], lk[10], u);
        double iCaT = P->gCaTbar * (s * s) * u * (V - P->ECa);
        dx[5] = (P->Cai_min - Cai) / P->taur_Cai - currentToConcentrationRate(Z_CA, P->deff) * iCaT;
        dx[6] = P->k2 * (1 - P0) - P->k1 * P0 * pow(Cai, P->nCa);
        dx[7] = alphao * C - betao * O - P->k3 * O * (1 - P0) + P->k4 * (1 - O - C);
        dx[8] = betao * O - alphao * C;
        break;
    }
    }
}

/* ------------------------------------------------------------------------------------------
 * NeuronalBilayerSonophore.effDerivatives (nbls.py:280-315) with qss_vars = [].
 *   y = [Qm, states...]; tabs[k*nQ + j], k = 0 is 'V'.
 * ---------------------------------------------------------------------------------------- */
void orc_eff_rhs(int id, const double *y, const double *Qref, int nQ, const double *tabs, double *dy)
{
    double lk[32];
    int ntab = 1 + orc_nrates(id);
    for (int k = 0; k < ntab; k++)
        lk[k] = orc_interp(y[0], Qref, tabs + (size_t)k * nQ, nQ);
    dy[0] = -orc_iNet(id, lk[0], y + 1) * 1e-3;
    eff_dstates(id, lk, y + 1, dy + 1);
}

/* ------------------------------------------------------------------------------------------
 * PointNeuron.derivatives (pneuron.py:485-505) for the `full` method: true rate functions.
 * derStates of each neuron: alpha/beta gates use alpha*(1-x) - beta*x, inf/tau gates
 * (xinf - x)/taux with the true functions (no alpha/beta detour).
 * ---------------------------------------------------------------------------------------- */
void orc_hh_rhs(int id, const double *y, double Cm, double *dy)
{
    double Vm = y[0] / Cm * 1e3;
    const double *x = y + 1;
    double *dx = dy + 1;
    dy[0] = -orc_iNet(id, Vm, x) * 1e-3;
    if (id == ORC_STN) {
        dx[0] = (stn_minf(Vm) - x[0]) / stn_taum(Vm);
        dx[1] = (stn_hinf(Vm) - x[1]) / stn_tauh(Vm);
        dx[2] = (stn_ninf(Vm) - x[2]) / stn_taun(Vm);
        dx[3] = (stn_ainf(Vm) - x[3]) / stn_taua(Vm);
        dx[4] = (stn_binf(Vm) - x[4]) / stn_taub(Vm);
        dx[5] = (stn_pinf(Vm) - x[5]) / stn_taup(Vm);
        dx[6] = (stn_qinf(Vm) - x[6]) / stn_tauq(Vm);
        dx[7] = (stn_cinf(Vm) - x[7]) / stn_tauc(Vm);
        dx[8] = (stn_d1inf(Vm) - x[8]) / stn_taud1(Vm);
        dx[9] = (stn_d2inf(x[11]) - x[9]) / STN_TAU_D2;
        dx[10] = (stn_rinf(x[11]) - x[10]) / STN_TAU_R;
        dx[11] = orc_stn_derCai(x[5], x[6], x[7], x[8], x[9], x[11], Vm);
        return;
    }
    const orc_pospischil_t *P = pospischil(id);
    dx[0] = hh_alpham(Vm, P->VT) * (1 - x[0]) - hh_betam(Vm, P->VT) * x[0];
    dx[1] = hh_alphah(Vm, P->VT) * (1 - x[1]) - hh_betah(Vm, P->VT) * x[1];
    dx[2] = hh_alphan(Vm, P->VT) * (1 - x[2]) - hh_betan(Vm, P->VT) * x[2];
    switch (id) {
    case ORC_RS: case ORC_FS:
        dx[3] = (ctx_pinf(Vm) - x[3]) / ctx_taup(Vm, P->TauMax);
        break;
    case ORC_LTS:
        dx[3] = (ctx_pinf(Vm) - x[3]) / ctx_taup(Vm, P->TauMax);
        dx[4] = (lts_sinf(Vm, P->Vx) - x[4]) / lts_taus(Vm, P->Vx);
        dx[5] = (lts_uinf(Vm, P->Vx) - x[5]) / lts_tauu(Vm, P->Vx);
        break;
    case ORC_IB:
        dx[3] = (ctx_pinf(Vm) - x[3]) / ctx_taup(Vm, P->TauMax);
        dx[4] = ib_alphaq(Vm) * (1 - x[4]) - ib_betaq(Vm) * x[4];
        dx[5] = ib_alphar(Vm) * (1 - x[5]) - ib_betar(Vm) * x[5];
        break;
    case ORC_RE:
        dx[3] = (re_sinf(Vm) - x[3]) / re_taus(Vm);
        dx[4] = (re_uinf(Vm) - x[4]) / re_tauu(Vm);
        break;
    case ORC_TC: {
        double s = x[3], u = x[4], Cai = x[5], P0 = x[6], O = x[7], C = x[8];
        double alphao = tc_oinf(Vm) / tc_tauo(Vm), betao = (1 - tc_oinf(Vm)) / tc_tauo(Vm);
        dx[3] = (lts_sinf(Vm, P->Vx) - s) / lts_taus(Vm, P->Vx);
        dx[4] = (lts_uinf(Vm, P->Vx) - u) / lts_tauu(Vm, P->Vx);
        double iCaT = P->gCaTbar * (s * s) * u * (Vm - P->ECa);
        dx[5] = (P->Cai_min - Cai) / P->taur_Cai - currentToConcentrationRate(Z_CA, P->deff) * iCaT;
        dx[6] = P->k2 * (1 - P0) - P->k1 * P0 * pow(Cai, P->nCa);
        dx[7] = alphao * C - betao * O - P->k3 * O * (1 - P0) + P->k4 * (1 - O - C);
        dx[8] = betao * O - alphao * C;
        break;
    }
    }
}

/* ==========================================================================================
 * Bilayer sonophore mechanics (PySONIC/core/bls.py)
 * ======================================================================================== */
typedef struct {
    double a;          /* sonophore radius (m) */
    double Cm0;        /* resting capacitance (F/m2) */
    double Delta;      /* equilibrium gap (m), bls_lookups.json 'Delta_eq' */
    double LJ_x0, LJ_C, LJ_nrep, LJ_nattr;   /* bls_lookups.json 'LJ_approx' */
    double kA_tissue;  /* bls.py:583-586 (0 when embedding depth d = 0) */
    double ng0;        /* bls.py:136-137 */
} orc_bls_t;

/* bls.py:88-110 */
static const double BLS_T = 309.15, BLS_delta0 = 2.0e-9, BLS_rhoL = 1075.0, BLS_muL = 7.0e-4,
    BLS_muS = 0.035, BLS_kA = 0.24, BLS_C0 = 0.62, BLS_kH = 1.613e5, BLS_P0 = 1.0e5,
    BLS_Dgl = 3.68e-9, BLS_xi = 0.5e-9, BLS_epsilon0 = 8.854e-12, BLS_epsilonR = 1.0,
    BLS_rel_Zmin = -0.49;

/* bls.py:286-296 */
static double bls_curvrad(const orc_bls_t *p, double Z)
{
    if (Z == 0.0) return INFINITY;
    return (p->a * p->a + Z * Z) / (2 * Z);
}
/* bls.py:302-309 */
static double bls_surface(const orc_bls_t *p, double Z) { return M_PI * (p->a * p->a + Z * Z); }
/* bls.py:311-319 */
static double bls_volume(const orc_bls_t *p, double Z)
{
    return M_PI * (p->a * p->a) * p->Delta * (1 + (Z / (3 * p->Delta) * (3 + (Z * Z) / (p->a * p->a))));
}
/* bls.py:334-345 */
double orc_bls_capacitance(const orc_bls_t *p, double Z)
{
    if (Z == 0.0) return p->Cm0;
    double Z2 = (p->a * p->a - Z * Z - Z * p->Delta) / (2 * Z);
    return p->Cm0 * p->Delta / (p->a * p->a) * (Z + Z2 * log((2 * Z + p->Delta) / p->Delta));
}
void orc_bls_capacitance_vec(const orc_bls_t *p, const double *Z, int n, double *Cm)
{
    for (int i = 0; i < n; i++) Cm[i] = orc_bls_capacitance(p, Z[i]);
}
/* bls.py:29-41, 472-480 */
double orc_bls_PMavgpred(const orc_bls_t *p, double Z)
{
    double r = p->LJ_x0 / (2 * Z + p->Delta);
    return p->LJ_C * (pow(r, p->LJ_nrep) - pow(r, p->LJ_nattr));
}
/* bls.py:482-491 */
static double bls_Pelec(const orc_bls_t *p, double Z, double Qm)
{
    double relS = (M_PI * p->a * p->a) / bls_surface(p, Z);
    double abs_perm = BLS_epsilon0 * BLS_epsilonR;
    return -relS * (Qm * Qm) / (2 * abs_perm);
}
/* bls.py:518-526 */
static double bls_gasmol2Pa(double ng, double V) { return ng * RG * BLS_T / V; }

/* bls.py:538-553 (predict method) */
double orc_bls_PtotQS(const orc_bls_t *p, double Z, double ng, double Qm, double Pac)
{
    double Pm = orc_bls_PMavgpred(p, Z);
    return Pm + bls_gasmol2Pa(ng, bls_volume(p, Z)) - BLS_P0 - Pac + bls_Pelec(p, Z, Qm);
}

/* bls.py:681-718 BilayerSonophore.derivatives; drive = AcousticDrive.compute (drives.py:303-304).
 * Returns 1 in *clamped if Z was clamped at Zmin (the reference logs a warning there). */
void orc_bls_rhs(const orc_bls_t *p, double t, const double *y, double f, double A, double phi,
                 double Qm, double *dy, int *clamped)
{
    double U = y[0], Z = y[1], ng = y[2];
    double Zmin = BLS_rel_Zmin * p->Delta;
    if (Z < Zmin) { Z = Zmin; if (clamped) *clamped = 1; }
    double R = bls_curvrad(p, Z);
    double Pg = bls_gasmol2Pa(ng, bls_volume(p, Z));
    double Pm = orc_bls_PMavgpred(p, Z);
    double Pac = A * sin(2 * M_PI * f * t - phi);
    /* PVleaflet (bls.py:613-621) + PVfluid (623-631) */
    double Pv = -12 * U * BLS_delta0 * BLS_muS / (R * R) + (-4 * U * BLS_muL / fabs(R));
    /* PEtot (bls.py:575-611): -(kA + kA_tissue) * (Z/a)^2 / R, summed as TEleaflet + TEtissue */
    double strain = (Z / p->a) * (Z / p->a);
    double PE = -(BLS_kA * strain + p->kA_tissue * strain) / R;
    double Ptot = Pm + Pg - BLS_P0 - Pac + PE + Pv + bls_Pelec(p, Z, Qm);
    /* accP (bls.py:633-641) + accNL (643-655) */
    dy[0] = Ptot / (BLS_rhoL * fabs(R)) + (-(3 * U * U) / (2 * R));
    dy[1] = U;
    /* gasFlux (bls.py:508-516) */
    double dC = BLS_C0 - Pg / BLS_kH;
    dy[2] = 2 * bls_surface(p, Z) * BLS_Dgl * dC / BLS_xi;
}

/* NeuronalBilayerSonophore.fullDerivatives (nbls.py:265-278): y = [U, Z, ng, Qm, states...] */
void orc_full_rhs(int id, const orc_bls_t *p, double t, const double *y, double f, double A,
                  double phi, double fs, double *dy, int *clamped)
{
    orc_bls_rhs(p, t, y, f, A, phi, y[3], dy, clamped);
    double Cm = fs * orc_bls_capacitance(p, y[1]) + (1 - fs) * p->Cm0;   /* nbls.py:148-151 */
    orc_hh_rhs(id, y + 3, Cm, dy + 3);
}
