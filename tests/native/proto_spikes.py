''' Development: the device spike tracker (CPU build) against the reference procedure (oracle.detect_spikes)
    on every golden sonic trace. '''
import ctypes, sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
lib = ctypes.CDLL('/root/repo/tests/native/libharness.so')
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
bad = 0
for name in ['RS', 'FS', 'LTS', 'RE', 'TC', 'STN']:
    g = np.load(f'/root/repo/tests/golden/golden_sonic_{name}.npz')
    for i in range(len(g['configs'])):
        ref = g[f'c{i}_default']
        t = np.ascontiguousarray(ref[:, 0]); q = np.ascontiguousarray(ref[:, 2])
        out = np.zeros(4); cand = np.zeros(512 * 5); stack = np.zeros(512, dtype=np.int32)
        flags = lib.harness_spikes(t.ctypes.data_as(dp), q.ctypes.data_as(dp), ctypes.c_long(t.size),
                                   out.ctypes.data_as(dp), cand.ctypes.data_as(dp), stack.ctypes.data_as(ip), 512)
        isp, _ = O.detect_spikes(t, q)
        ok = int(out[0]) == isp.size and (isp.size == 0 or (out[1] == t[isp[0]] and out[2] == t[isp[-1]]))
        if not ok and not (flags & 2):
            bad += 1
            print(name, i, 'MISMATCH', out, isp.size, flags)
print('mismatches', bad)
