''' Detailed-model kernels (method='full'): time per integration step of the lane-per-configuration
    and of the octet-cooperative kernel, for batches of 1 .. 256 configurations of a short protocol.
    Prints one JSON line per (kernel, batch size): kernel ms, steps of the slowest configuration,
    microseconds per step of the slowest configuration (= latency of one step), total steps per second.

    usage (GPU box): python tools/full_probe.py [--neuron RS] [--tstim 4e-5] [--sizes 1,8,64,256] [--rtol 1e-8]
'''
import os
import sys
import json
import argparse
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron  # noqa: E402
from pysonic_amd import _native as N  # noqa: E402


def run(nbls, n, tstim, kernel, rtol):
    amps = np.logspace(np.log10(10e3), np.log10(600e3), max(n, 2))[:n] if n > 1 else np.array([100e3])
    pp = PulsedProtocol(tstim, tstim / 4)
    cfgs = [(AcousticDrive(500e3, float(a)), pp) for a in amps]
    A, tstop, _, ev_t, ev_x, ev_off = nbls._packConfigs(cfgs)
    o = N.full_default_opts(kernel=kernel, rtol=rtol)
    tr, row_off, status, nsteps, ms = N.full_batch_run(
        nbls.pneuron.name, nbls.pneuron.device_params(), nbls.device_params(), [500e3] * n, A, [1.] * n,
        tstop, ev_t, ev_x, ev_off, nbls.initialConditionsSonic(), o)
    return {'kernel': {1: 'lane dopri5', 2: 'coop dop853', 3: 'coop dopri5'}[kernel], 'configs': n,
            'tstim_us': tstim * 1e6, 'rtol': rtol,
            'kernel_ms': ms, 'max_steps': int(nsteps.max()), 'mean_steps': float(nsteps.mean()),
            'us_per_step_slowest': ms * 1e3 / float(nsteps.max()),
            'steps_per_s_total': float(nsteps.sum()) / (ms * 1e-3), 'bad_status': int(np.count_nonzero(status)),
            'checksum_Qm': float(np.nansum(tr[:, 4]))}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--neuron', default='RS')
    ap.add_argument('--tstim', type=float, default=4e-5)
    ap.add_argument('--sizes', default='1,8,64,256')
    ap.add_argument('--rtol', type=float, default=0.)      # 0: the library's default for the method
    ap.add_argument('--kernels', default='1,3,2')
    args = ap.parse_args()
    N.require_gpu()
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(args.neuron))
    for k in [int(x) for x in args.kernels.split(',')]:
        for n in [int(x) for x in args.sizes.split(',')]:
            print(json.dumps(run(nbls, n, args.tstim, k, args.rtol)), flush=True)
