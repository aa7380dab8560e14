# -*- coding: utf-8 -*-
''' pysonic_amd: MI355X-native batched integration of the SONIC / NICE models behind the
    PySONIC Python API. Host code is Python; all numerics of the hot path run in hand-written HIP
    kernels reached through a C ABI (include/pysonic_amd.h) with ctypes. See DESIGN.md. '''
__version__ = '0.1.0'

from . import constants  # noqa: F401
from .core import *  # noqa: F401,F403
from .neurons import getPointNeuron, getNeuronsDict  # noqa: F401
