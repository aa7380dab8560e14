''' Development (GPU box): the group-cooperative sonic kernel -- us per step of the costliest configuration
    alone, and the 2000-configuration sweep of one frequency, for 1 / 2 / 4 configurations per wavefront
    and for the lane-per-configuration kernel.  usage: [TSTIM=0.1] python tools/group_probe.py [neurons...]
    (TSTIM: stimulus duration in s, the offset is half of it; the neurons with a 5 us or 0.5 us output step want 0.01) '''
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
PRFs = np.logspace(1, 3, 10); DCs = np.linspace(0.05, 1.0, 10)
tstim = float(os.environ.get('TSTIM', 100e-3))
PRFs = np.maximum(PRFs, 1. / tstim)
out = {}
for name in (sys.argv[1:] or ['LTS', 'RE', 'TC', 'STN']):
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(tstim, tstim / 2, float(prf), float(dc)))
            for a in amps for prf in PRFs for dc in DCs]
    res = {}

    def run(c, env):
        for k in ['PYSONIC_AMD_GROUP', 'PYSONIC_AMD_GPW']:
            os.environ.pop(k, None)
        os.environ.update(env)
        (_, met, st, ms), = nbls.runSonicBatches([(500e3, 1., c, None)], traces=False)
        return met, ms
    met, ms = run(cfgs, {})
    worst = int(np.argmax(met[:, N.M_NSTEPS]))
    res['sweep_2000'] = {'auto': ms}
    for label, env in (('gpw1', {'PYSONIC_AMD_GPW': '1'}), ('gpw2', {'PYSONIC_AMD_GPW': '2'}),
                       ('gpw4', {'PYSONIC_AMD_GPW': '4'}), ('lane', {'PYSONIC_AMD_GROUP': '0'})):
        res['sweep_2000'][label] = run(cfgs, env)[1]
    res['worst'] = {'steps': float(met[worst, N.M_NSTEPS])}
    for label, env, k in (('gpw1_x256', {'PYSONIC_AMD_GPW': '1'}, 256), ('gpw4_x1024', {'PYSONIC_AMD_GPW': '4'}, 1024),
                          ('gpw4_x4096', {'PYSONIC_AMD_GPW': '4'}, 4096), ('lane_x256', {'PYSONIC_AMD_GROUP': '0'}, 256)):
        m2, ms2 = run([cfgs[worst]] * k, env)
        res['worst'][label] = {'ms': ms2, 'us_per_step': ms2 * 1e3 / float(m2[0, N.M_NSTEPS])}
    out[name] = res
    print(name, json.dumps(res), flush=True)
os.makedirs('gpurun_out/r02e', exist_ok=True)
json.dump(out, open(f'gpurun_out/r02e/group_probe_{"_".join(out)}.json', 'w'), indent=1)
