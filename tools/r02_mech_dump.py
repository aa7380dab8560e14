''' Development: device-made LTS 2 MHz table + cycle counts, and the A = 0 row of config 3, to gpurun_out '''
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
d = np.load('tests/golden/devtables_LTS_32nm_2000kHz.npz')
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('LTS'))
A, Q = np.meshgrid(d['A'], d['Q'], indexing='ij')
eff, ncyc, status, ms = nbls.runMechBatch(np.full(A.size, float(d['f'])), A.ravel(), Q.ravel(), [1.0])
out = {'eff': eff, 'ncyc': ncyc, 'status': status}
pn = getPointNeuron('RS')
freqs = np.array([20., 100., 500., 1e3, 2e3, 3e3, 4e3]) * 1e3
charges = np.arange(pn.Qbounds[0], pn.Qbounds[1] + 1e-5, 1e-5)
for a in [16e-9, 32e-9, 64e-9]:
    nb = NeuronalBilayerSonophore(a, pn)
    F, Qq = np.meshgrid(freqs, charges, indexing='ij')
    e, n, s, _ = nb.runMechBatch(F.ravel(), np.zeros(F.size), Qq.ravel(), [1.0])
    out[f'a0_{a*1e9:.0f}_ncyc'] = n.reshape(F.shape); out[f'a0_{a*1e9:.0f}_status'] = s.reshape(F.shape)
os.makedirs('gpurun_out/r02d', exist_ok=True)
np.savez('gpurun_out/r02d/mech_dump.npz', **out)
print('done', ms)
