# -*- coding: utf-8 -*-
''' Point-neuron registry (API of PySONIC/neurons/__init__.py:24-44): the six neurons of
    BASELINE.json's configurations, the cortical intrinsically bursting neuron and three axon
    membrane models (Hodgkin-Huxley segment, Sweeney node, MRG node, Sundt segment,
    Frankenhaeuser-Huxley node). '''
from .cortical import CorticalRS, CorticalFS, CorticalLTS, CorticalIB
from .thalamic import ThalamicRE, ThalamoCortical
from .stn import OtsukaSTN
from .pas import passiveNeuron, getDefaultPassiveNeuron  # noqa: F401
from .axons import (HodgkinHuxleySegment, SweeneyNode, MRGNode, SundtSegment,
                    FrankenhaeuserHuxleyNode)

_CLASSES = [CorticalRS, CorticalFS, CorticalLTS, CorticalIB, ThalamicRE, ThalamoCortical, OtsukaSTN,
            HodgkinHuxleySegment, SweeneyNode, MRGNode, SundtSegment, FrankenhaeuserHuxleyNode]


def getNeuronsDict():
    return {c.name: c for c in _CLASSES}


def getPointNeuron(name):
    if name.startswith('pas_'):                    # neurons/__init__.py:40-41
        return passiveNeuron(name)
    classes = getNeuronsDict()
    try:
        return classes[name]()
    except KeyError:
        raise ValueError('"{}" neuron not found. Implemented neurons are: {}'.format(
            name, ', '.join(classes.keys())))
