# -*- coding: utf-8 -*-
''' Extract the cached bilayer-sonophore parameters (Delta_eq and Lennard-Jones fit) that the
    reference keeps in its data file PySONIC/core/bls_lookups.json (read at bls.py:44-77) for the
    sonophore radii of BASELINE.json's configs (16, 32, 64 nm), all resting charges present.

    These are inputs of the hot path, not results: the LJ fit itself is out
    of scope, so both the oracle and pysonic_amd read them from this JSON (same key format).

    Output: tests/golden/bls_params.json   (build container only)
'''
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/PySONIC/core/bls_lookups.json'

with open(SRC) as fh:
    d = json.load(fh)
sub = {k: d[k] for k in ('16.0', '32.0', '64.0')}
with open(os.path.join(HERE, 'bls_params.json'), 'w') as fh:
    json.dump(sub, fh, indent=1)
print({k: len(v) for k, v in sub.items()})
