''' Detailed model, BASELINE config 5: how far two independent step sequences of the SAME trajectory drift apart.
    Configurations of one amplitude share their first 0.5 ms when their duty cycles are 0.52 and 1.0 (PRF 1 kHz);
    the probe runs such pairs at a list of tolerances, twice each, and prints per amplitude and variable the RMS
    distance over the shared prefix relative to the variable's range, plus whether the two launches gave the same
    bits (the kernels have no atomics: they must).

    usage (GPU box): python tools/full_prefix_probe.py [--amps 10,15] [--rtols 0,3e-8]
'''
import os
import sys
import json
import argparse
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron  # noqa: E402
from pysonic_amd import _native as N  # noqa: E402


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--amps', default='10,15')       # indices into the 16 amplitudes of config 5
    ap.add_argument('--rtols', default='0,3e-8')     # 0: the library's default
    args = ap.parse_args()
    N.require_gpu()
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 16)
    DCs = np.linspace(0.1, 1.0, 16)
    ias = [int(x) for x in args.amps.split(',')]
    cfgs = [(AcousticDrive(500e3, float(amps[ia])), PulsedProtocol(1e-3, 0.25e-3, 1e3, float(DCs[j])), 1.)
            for ia in ias for j in (7, 15)]
    for rtol in [float(x) for x in args.rtols.split(',')]:
        opts = {'rtol': rtol} if rtol > 0 else None
        runs = []
        for _ in range(2):
            frames, status, ms = nbls.runFullBatch(cfgs, opts=opts)
            runs.append(([f.values.copy() for f in frames], ms))
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(runs[0][0], runs[1][0]))
        cols = list(frames[0].columns)
        t = frames[0]['t'].values
        n = int(np.searchsorted(t, 0.5e-3))
        for k, ia in enumerate(ias):
            a, b = runs[0][0][2 * k], runs[0][0][2 * k + 1]
            rel = {}
            for c in ('Z', 'ng', 'Qm', 'm', 'h', 'n', 'p'):
                j = cols.index(c)
                x, y = a[1:n, j], b[1:n, j]
                rel[c] = float(np.sqrt(np.mean((x - y)**2)) / np.ptp(y))
            print(json.dumps({'rtol': rtol, 'amp_index': ia, 'A_kPa': float(amps[ia]) * 1e-3, 'kernel_ms': runs[0][1],
                              'two_launches_same_bits': bool(same), 'status': [int(s) for s in status],
                              'prefix_rms_over_range': rel}), flush=True)
