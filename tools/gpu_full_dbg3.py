''' Development script (GPU box): the detailed-model kernel with 1 or 64 active lanes per wavefront. '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import _native as N
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
name = sys.argv[1] if len(sys.argv) > 1 else 'TC'
pn = getPointNeuron(name)
nbls = NeuronalBilayerSonophore(32e-9, pn)
nbls.setTissueModulus(AcousticDrive(500e3, 120e3))
def run(ncopies, ipw):
    os.environ['PYSONIC_AMD_IPW'] = str(ipw)
    cfgs = [(AcousticDrive(500e3, 120e3), PulsedProtocol(4e-6, 1e-6))] * ncopies
    A, tstop, _, ev_t, ev_x, ev_off = nbls._packConfigs(cfgs)
    traces, row_off, status, nsteps, ms = N.full_batch_run(
        name, pn.device_params(), nbls.device_params(), [500e3] * ncopies, A, [1.] * ncopies, tstop, ev_t, ev_x,
        ev_off, nbls.initialConditionsSonic())
    return [traces[row_off[i]:row_off[i + 1]] for i in range(ncopies)], nsteps, status
ref, ns, st = run(1, 64)
print('1 config, 1 active lane: steps', ns, 'status', st)
for ncopies, ipw, label in [(1, 1, '1 config + 63 shadow lanes'), (64, 64, '64 copies, one per lane'), (2, 2, '2 copies + shadows'), (1, 64, 'again 1 active lane')]:
    tr, ns, st = run(ncopies, ipw)
    d = [np.abs(t - ref[0]).max(axis=0) for t in tr]
    print(f'{label}: steps {sorted(set(ns.tolist()))} status {sorted(set(st.tolist()))}; max |diff| vs 1-lane run, per column (worst copy): '
          + ' '.join(f'{x:.1e}' for x in np.max(d, axis=0)), flush=True)
    if ncopies > 1:
        print('   copies identical to each other:', all(np.array_equal(tr[0], t) for t in tr[1:]))
