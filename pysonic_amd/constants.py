# -*- coding: utf-8 -*-
''' Numerical constants of the model family (values of PySONIC/constants.py:12-73, which are
    part of the numerical contract: time steps, cycle limits, spike and titration thresholds). '''

# biophysics
FARADAY = 9.64853e4
Rg = 8.31342
Z_Ca = 2
Z_Na = 1
Z_K = 1
CELSIUS_2_KELVIN = 273.15

# intermolecular pressure fitting
LJFIT_PM_MAX = 1e8
PNET_EQ_MAX = 1e-1
PMAVG_STD_ERR_MAX = 5e3

# lookups
DQ_LOOKUP = 1e-5

# simulations
MAX_RMSE_PTP_RATIO = 1e-4
Z_ERR_MAX = 1e-11
NG_ERR_MAX = 1e-24
NCYCLES_MAX = 10
CHARGE_RANGE = (-300e-5, 150e-5)
SOLVER_NSTEPS = 1000
CLASSIC_TARGET_DT = 1e-8
NPC_DENSE = 1000
NPC_SPARSE = 40
MIN_SPARSE_DT = 1e-12
HYBRID_UPDATE_INTERVAL = 5e-4
DT_EFFECTIVE = 5e-5
MIN_SAMPLES_PER_PULSE_INTERVAL = 1
MAX_NSAMPLES_EFFECTIVE = 1e5

# post-processing
DT_MAX_REL_TOL = 1e-5
SPIKE_MIN_DT = 5e-4
SPIKE_MIN_QAMP = 3e-5
SPIKE_MIN_QPROM = 20e-5
SPIKE_MIN_VAMP = 3.0
SPIKE_MIN_VPROM = 20.0
MIN_NSPIKES_SPECTRUM = 3

# titrations
ESTIM_AMP_UPPER_BOUND = 1e5
ESTIM_AMP_INITIAL = 1e0
ESTIM_REL_CONV_THR = 1e-2
ASTIM_AMP_INITIAL = 1e4
ASTIM_ABS_CONV_THR = 1e2
ASTIM_REL_CONV_THR = 1e0

# QSS analysis
QSS_REL_OFFSET = .05
QSS_HISTORY_INTERVAL = 30e-3
QSS_INTEGRATION_INTERVAL = 1e-3
QSS_MAX_INTEGRATION_DURATION = 1000e-3
QSS_Q_CONV_THR = 1e-7
QSS_Q_DIV_THR = 1e-4
TMIN_STABILIZATION = 500e-3


def getConstantsDict():
    return {k: v for k, v in globals().items()
            if not k.startswith('__') and k != 'getConstantsDict'}
