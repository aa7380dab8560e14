''' Development (GPU box): the RS sonic kernel on the 65 536-configuration sweep of bench.py (`saturated`) under the
    library's work-queue switch PYSONIC_AMD_WPS (wavefronts per SIMD that hold configurations at the start; 0: no
    queue, every configuration placed by the host). Kernel ms per setting; rows and metrics must not depend on it. '''
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
name = sys.argv[1] if len(sys.argv) > 1 else 'RS'
nA = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
model, _ = nbls._sonicModel(500e3, 1.)
amps = np.logspace(np.log10(10e3), np.log10(600e3), nA); DCs = np.linspace(0.05, 1.0, 256)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
packed, y0 = nbls._packConfigs(cfgs), nbls.initialConditionsSonic()
ref = None
out = {}
# (wavefronts per SIMD, costs: 0 the host's a-priori estimate / 1 the measured step counts of the first run)
for wps, known in ((0, 0), (1, 0), (2, 0), (3, 0), (2, 1)):
    os.environ["PYSONIC_AMD_WPS"] = str(wps)
    if known:
        os.environ["PYSONIC_AMD_COST_FILE"] = 'gpurun_out/sat_costs.f64'
    else:
        os.environ.pop("PYSONIC_AMD_COST_FILE", None)
    b = model.prepare(*packed, y0)
    b.launch(); b.sync()
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    tr, met, st = b.fetch()
    assert np.all(st == 0), np.unique(st, return_counts=True)
    sel = np.arange(0, len(cfgs), 997)
    rows = [tr[b.row_off[i]:b.row_off[i + 1]].copy() for i in sel]
    if ref is None:
        ref = (rows, met[:, :11].copy())
        os.makedirs('gpurun_out', exist_ok=True)
        met[:, 0].astype(np.float64).tofile('gpurun_out/sat_costs.f64')
    else:
        for a, r in zip(rows, ref[0]):
            assert np.array_equal(a, r), 'rows depend on the schedule'
        assert np.array_equal(met[:, :11], ref[1], equal_nan=True)
    label = 'measured steps' if known else 'a-priori estimate'
    out[f'wps{wps}_{"known" if known else "estimate"}'] = float(np.mean(ms))
    print(f'{name} {len(cfgs)} configurations, WPS {wps}, costs = {label}: kernel {np.mean(ms):.2f} ms ({len(cfgs) / np.mean(ms) * 1e3:.3e} configs/s), '
          f'steps mean {met[:, 0].mean():.0f} max {met[:, 0].max():.0f}', flush=True)
    b.close(); del tr
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'neuron': name, 'configs': len(cfgs), 'kernel_ms_by_wps': out}, open(f'gpurun_out/sat_probe_{name}.json', 'w'))
