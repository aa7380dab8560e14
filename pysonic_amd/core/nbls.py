# -*- coding: utf-8 -*-
''' NeuronalBilayerSonophore -- the reference's public model class for acoustic stimulation
    (PySONIC/core/nbls.py:24-671), with simulate() executed by the HIP kernels of
    libpysonic_amd.so instead of scipy's LSODA.

    Kept verbatim-compatible: constructor, simulate(drive, pp, fs=1., method='sonic',
    qss_vars=None) -> (TimeSeries, meta) incl. input validation and meta keys, simQueue,
    filecodes / filecode, getLookup / getLookup2D, the output column order
    (t, stimstate, Qm, states..., Vm, Z, ng) and Batch(nbls.simulate, queue).run(mpi=True).

    New: `_batched_simulate` runs a whole queue in ONE kernel launch per (f, fs) group;
    `Batch.run(mpi=True)` routes to it. There is no CPU integrator here: without the native
    library / a GPU every simulate() raises NativeLibraryError.
'''
import os

import logging

import numpy as np

from .bls import BilayerSonophore
from .pneuron import PointNeuron
from .model import Model
from .drives import Drive, AcousticDrive
from .protocols import TimeProtocol, PulsedProtocol, BurstProtocol
from .lookups import EffectiveVariablesLookup
from .timeseries import TimeSeries
from ..utils import logger, isIterable, si_format, LOOKUP_DIR, timer, LogCache, methodCallSignature
from ..constants import MAX_NSAMPLES_EFFECTIVE
from .. import _native

# thresholds found by titrate() are logged here, signature -> amplitude (the reference logs next to its module).
# The reference's key (the call signature) says nothing about the integrator or the tables behind a threshold, so the
# default file name carries the digest of the native sources: a new build starts a new log instead of serving the old
# build's thresholds. PYSONIC_AMD_TITRATIONS=<path> names a file explicitly (e.g. the reference's own log).
TITRATION_LOG = os.environ.get('PYSONIC_AMD_TITRATIONS')


def default_titration_log():
    if TITRATION_LOG is not None:
        return TITRATION_LOG
    from ..build import source_hash
    try:
        tag = source_hash()[:10]
    except OSError:
        tag = f'abi{_native.ABI_VERSION}'
    return os.path.join(os.path.expanduser('~'), '.cache', 'pysonic_amd', f'astim_titrations_{tag}.log')


# tables generated on the device on demand are cached here (never in the package directory)
GENERATED_LOOKUP_DIR = os.environ.get(
    'PYSONIC_AMD_CACHE', os.path.join(os.path.expanduser('~'), '.cache', 'pysonic_amd', 'lookups'))


import collections.abc
import operator


class RowBlocks(collections.abc.Sequence):
    ''' The row arrays of the configurations of one launch: views of the launch's ONE host block (page-locked,
        filled by the device), made when asked for -- a sweep of thousands of configurations does not pay for
        thousands of array objects it may never look at. `masks`: {index: boolean row mask} for the few
        configurations whose rows are filtered (progress-log events, _rowsKeptWithLogEvents). '''

    def __init__(self, block, row_start, n_rows, masks=None):
        self.block, self.row_start, self.n_rows, self.masks = block, row_start, n_rows, masks or {}

    def __len__(self):
        return len(self.n_rows)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = operator.index(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError('configuration index out of range')
        a = self.row_start[i]
        r = self.block[a:a + self.n_rows[i]]
        m = self.masks.get(i)
        return r if m is None else r[m]


class SimResults(collections.abc.Sequence):
    ''' What a batched simulate() returns: the (TimeSeries, meta) pairs of the queue, in queue order, each built
        the first time it is asked for (then kept). The reference's Batch.run returns a list; this is a sequence
        with the same indexing, slicing, iteration and length, whose frames are views of the launches' host
        blocks. `list(results)` builds them all. Entries are None where a simulation has no output (unresolved
        drive without a threshold). '''

    def __init__(self, n):
        self._n = n
        self._entries = [None] * n        # (maker, argument) | ('done', value)

    def _set(self, i, maker, arg):
        self._entries[i] = (maker, arg)

    def _set_value(self, i, value):
        self._entries[i] = ('done', value)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        i = operator.index(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError('batch output index out of range')
        e = self._entries[i]
        if e is None:
            return None
        maker, arg = e
        if maker == 'done':
            return arg
        value = maker(arg)
        self._entries[i] = ('done', value)
        return value

    def __iter__(self):
        for i in range(self._n):
            yield self[i]

    def __eq__(self, other):
        try:
            return len(other) == self._n and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __reduce__(self):
        return (list, (list(self),))      # pickles (all_gather_object, multiprocessing) as the plain list


class _SonicLaunchOutputs:
    ''' outputs of the configurations of one launch, made on demand (SimResults) '''

    def __init__(self, nbls, rows, params, qss, lkp, tcomp_each):
        self.nbls, self.rows, self.params, self.qss, self.lkp, self.tcomp_each = nbls, rows, params, qss, lkp, tcomp_each

    def __call__(self, j):
        nbls, p = self.nbls, self.params[j]
        drive, pp, fs, method, qss_vars = p
        meta = {'simkey': nbls.simkey, 'model': nbls.meta, 'drive': drive, 'pp': pp, 'fs': fs, 'method': method,
                'qss_vars': qss_vars, 'tcomp': self.tcomp_each}
        return nbls._toTimeSeries(self.rows[j], self.qss, self.lkp, drive.A), meta


class NeuronalBilayerSonophore(BilayerSonophore):

    tscale = 'ms'
    simkey = 'ASTIM'

    def __init__(self, a, pneuron, embedding_depth=0.0):
        if not isinstance(pneuron, PointNeuron):
            raise ValueError(f'{pneuron} is not a valid PointNeuron instance')
        self.pneuron = pneuron
        self._models = {}       # (f, fs, device) -> _native.SonicModel
        self.device = None      # GPU of this process: None = _native.default_device() (LOCAL_RANK)
        self.titration_cache = LogCache(default_titration_log())      # None: no threshold cache
        self.solver_opts = {}   # overrides of the native integrator options (rtol, atol, ...)
        self.full_opts = {}     # the same for the detailed-model kernels (full, hybrid)
        self.max_full_dense_points = 5e6   # guard for method='full' (10 ms at 500 kHz)
        super().__init__(a, pneuron.Cm0, pneuron.Qm0, embedding_depth=embedding_depth)

    @staticmethod
    def _queueCosts(calls):
        ''' Rough relative cost of each call of a queue (for the split of a queue over the GPUs of a
            process group, parallel.weighted_bounds): simulations cost their stimulated time times a
            saturating function of the amplitude (a spiking neuron crosses many table cells) plus a
            little for the offset; detailed simulations cost ~100x per unit of time, more at high
            amplitude; lookup cells cost one acoustic period (~ 1 / f). Items it does not recognise cost 1. '''
        costs = []
        for args, kwargs in calls:
            drive = args[0] if args else kwargs.get('drive')
            c = 1.0
            try:
                if len(args) > 1 and hasattr(args[1], 'tstop'):
                    pp = args[1]
                    method = args[3] if len(args) > 3 else kwargs.get('method', 'sonic')
                    A = drive.A or 0.
                    ton = getattr(pp, 'tstim', pp.tstop) * getattr(pp, 'DC', 1.)
                    c = ton * A / (A + 50e3) + 0.05 * pp.tstop
                    if method in ('full', 'hybrid'):
                        c = 100. * pp.tstop * (1. + A / 100e3)
                elif drive is not None and hasattr(drive, 'f'):
                    c = 1. / drive.f
            except (AttributeError, TypeError):
                c = 1.0
            costs.append(max(float(c), 1e-12))
        return costs

    def _device(self):
        ''' GPU index the native calls of this object run on: `self.device`, or the process default
            (one process per GPU: LOCAL_RANK, see _native.default_device). '''
        return _native.default_device() if self.device is None else int(self.device)

    @property
    def a_str(self):
        return f'{self.a * 1e9:.1f} nm'

    def __repr__(self):
        s = f'{self.__class__.__name__}({self.a_str}, {self.pneuron}'
        if self.d > 0.:
            s += f', d={si_format(self.d, precision=1)}m'
        return f'{s})'

    def copy(self):
        return self.__class__(self.a, self.pneuron, embedding_depth=self.d)

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return False
        return self.a == other.a and self.pneuron == other.pneuron and self.d == other.d

    def __hash__(self):
        return hash((self.a, self.pneuron.name, self.d))

    @property
    def meta(self):
        return {'neuron': self.pneuron.name, 'a': self.a, 'd': self.d}

    @classmethod
    def initFromMeta(cls, meta):
        from ..neurons import getPointNeuron
        return cls(meta['a'], getPointNeuron(meta['neuron']), embedding_depth=meta['d'])

    def filecode(self, *args):
        return Model.filecode(self, *args)

    def filecodes(self, drive, pp, fs, method, qss_vars):
        codes = {'simkey': self.simkey, 'neuron': self.pneuron.name, 'nature': pp.nature,
                 'a': f'{self.a * 1e9:.0f}nm', **drive.filecodes, **pp.filecodes}
        codes['fs'] = f'fs{fs * 1e2:.0f}%' if fs < 1 else None
        codes['method'] = method
        codes['qss_vars'] = qss_vars
        return codes

    # ------------------------------------------------------------------------------------------
    # lookups (nbls.py:224-263)
    # ------------------------------------------------------------------------------------------
    def getLookupFileName(self, a=None, f=None, A=None, fs=None, novertones=0.):
        if all(x is None for x in [a, f, A, fs]):
            fs = 1.
        fname = f'{getattr(self.pneuron, "lookup_name", self.pneuron.name)}_lookups'
        if a is not None:
            fname += f'_{a * 1e9:.0f}nm'
        if f is not None:
            fname += f'_{f * 1e-3:.0f}kHz'
        if A is not None:
            fname += f'_{A * 1e-3:.0f}kPa'
        if fs is not None:
            fname += f'_fs{fs:.2f}'
        if novertones > 0:
            fname += f'_{novertones}overtones'
        return f'{fname}.pkl'

    def getLookupFilePath(self, *args, **kwargs):
        return os.path.join(LOOKUP_DIR, self.getLookupFileName(*args, **kwargs))

    def _packagedLookup(self):
        ''' 5-D lookup (a, f, A, Q, fs) assembled from the shipped 2-D .npz tables. '''
        refs_a, refs_f, per_af = [], [], {}
        prefix = f'tables_{self.pneuron.name}_'
        if os.path.isdir(LOOKUP_DIR):
            for fname in sorted(os.listdir(LOOKUP_DIR)):
                if fname.startswith(prefix) and fname.endswith('.npz'):
                    d = np.load(os.path.join(LOOKUP_DIR, fname))
                    per_af[(float(d['a']), float(d['f']))] = d
        if not per_af:
            return None
        refs_a = sorted({k[0] for k in per_af})
        refs_f = sorted({k[1] for k in per_af})
        if len(per_af) != len(refs_a) * len(refs_f):
            raise ValueError(f'incomplete (a, f) grid of packaged lookups for {self.pneuron.name}')
        d0 = next(iter(per_af.values()))
        keys = [str(k) for k in d0['keys']]
        A, Q = d0['A'], d0['Q']
        tables = {k: np.empty((len(refs_a), len(refs_f), A.size, Q.size, 1)) for k in keys}
        for ia, a in enumerate(refs_a):
            for i_f, f in enumerate(refs_f):
                d = per_af[(a, f)]
                for k in keys:
                    tables[k][ia, i_f, :, :, 0] = d[f'tab_{k}']
        refs = {'a': np.array(refs_a), 'f': np.array(refs_f), 'A': A, 'Q': Q,
                'fs': np.array([1.])}
        return EffectiveVariablesLookup(refs, tables)

    def getLookup(self, *args, **kwargs):
        keep_tcomp = kwargs.pop('keep_tcomp', False)
        lookup_path = self.getLookupFilePath(*args, **kwargs)
        if os.path.isfile(lookup_path):
            lkp = EffectiveVariablesLookup.fromPickle(lookup_path)
        else:
            lkp = self._packagedLookup() if not args and kwargs in ({}, {'fs': 1.}) else None
            if lkp is None:
                raise FileNotFoundError(f'Missing lookup file: "{lookup_path}"')
        if not keep_tcomp and 'tcomp' in lkp.tables:
            del lkp.tables['tcomp']
        return lkp

    def getLookup2D(self, f, fs):
        proj_kwargs = {'a': self.a, 'f': f, 'fs': fs}
        if fs < 1.:
            kwargs = proj_kwargs.copy()
            kwargs['fs'] = None
        else:
            kwargs = {'fs': fs}
        try:
            return self.getLookup(**kwargs).projectN(proj_kwargs)
        except (FileNotFoundError, ValueError) as err:
            # The reference needs pre-computed files here (downloaded, or run_lookups.py with
            # --spanFs for fs < 1) and raises for a radius / frequency outside their grids.
            # Without them the (A, Q) table of this (a, f, fs) is generated on the device -- one
            # mech_batch_run launch: about a second at 500 kHz, ~15 s at 20 kHz -- on the
            # reference's standard grids (scripts/run_lookups.py:183-199), cached in memory and,
            # if the lookup directory is writable, on disk.
            # Only inside the ranges the reference's lookups span (16-64 nm, 20 kHz-4 MHz): outside
            # them the ValueError of the projection stands. NB for a frequency BETWEEN the
            # reference's grid values the reference interpolates its tables linearly in f; the
            # table generated here is the one of that exact frequency.
            if not (16e-9 <= self.a <= 64e-9 and 20e3 <= f <= 4e6 and 0. < fs <= 1.):
                raise
            if isinstance(err, ValueError) and 'interval' not in str(err):
                raise
            return self._generatedLookup2D(f, fs)

    def _generatedLookup2D(self, f, fs):
        key = (float(f), float(fs))
        cache = self.__dict__.setdefault('_lkp2d_cache', {})
        if key in cache:
            return cache[key]
        keys = ['V'] + list(self.pneuron.rates)
        # amplitudes: 0 and 50 log-spaced values up to 600 kPa; charges: Qbounds in 1 nC/cm2 steps
        amps = np.insert(np.logspace(np.log10(0.1), np.log10(600), num=50), 0, 0.0) * 1e3
        Qmin, Qmax = self.pneuron.Qbounds
        charges = np.arange(Qmin, Qmax + 1e-5, 1e-5)
        try:        # same grids as an existing full-coverage lookup of this neuron, if any
            ref = self.getLookup()
            amps, charges = ref.refs['A'], ref.refs['Q']
        except FileNotFoundError:
            pass
        # The file name is for humans; what identifies a cached table is the digest of everything
        # it depends on, at full precision (a 32.4 nm sonophore must not pick up the 32 nm table),
        # and the stored parameters are checked again on load.
        import hashlib
        from ..build import source_hash
        try:
            build_tag = source_hash()        # a change of the kernels (method, tolerances) starts new tables
        except OSError:
            build_tag = f'abi{_native.ABI_VERSION}'
        ident = repr((self.pneuron.name, float(self.a), float(self.d), float(f), float(fs),
                      build_tag)).encode() + amps.tobytes() + charges.tobytes() + \
            np.asarray(self.pneuron.device_params(), dtype=float).tobytes()
        fname = (f'generated_{self.pneuron.name}_{self.a * 1e9:.0f}nm_{f * 1e-3:.0f}kHz_'
                 f'fs{fs:.2f}_{hashlib.sha256(ident).hexdigest()[:12]}.npz')
        fpath = os.path.join(GENERATED_LOOKUP_DIR, fname)

        def load():
            # the cached table, or None: a missing, truncated or foreign file is a cache miss
            import zipfile
            try:
                with np.load(fpath) as d:
                    if (float(d['a']) == self.a and float(d['f']) == f and float(d['fs']) == fs and
                            np.array_equal(d['A'], amps) and np.array_equal(d['Q'], charges)):
                        return EffectiveVariablesLookup({'A': d['A'], 'Q': d['Q']}, {k: d[f'tab_{k}'] for k in keys})
            except (OSError, ValueError, KeyError, EOFError, zipfile.BadZipFile):
                pass
            return None

        def generate():
            logger.info('generating the %s lookup for a = %.0f nm, f = %.0f kHz, fs = %.0f%% on the '
                        'device', self.pneuron.name, self.a * 1e9, f * 1e-3, fs * 1e2)
            return self.computeLookup([f], amps, charges, fs=fs).project('f', f)

        lkp = load()
        if lkp is None:
            # One writer at a time per table: the processes of a one-process-per-GPU launch all miss the same
            # table at once; the first to take the lock generates it, the others wait and then read the file.
            # The file appears under its name only when complete (temporary file + rename). No collective
            # here: the ranks of a sharded sweep do not all need the same tables.
            try:
                os.makedirs(GENERATED_LOOKUP_DIR, exist_ok=True)
                lock = open(fpath + '.lock', 'a')
            except OSError:
                lock = None
            if lock is None:
                lkp = generate()             # cache directory not writable: in memory only
            else:
                from ..utils import file_lock
                with lock, file_lock(lock):
                    lkp = load()
                    if lkp is None:
                        lkp = generate()
                        tmp = f'{fpath}.{os.getpid()}.tmp.npz'
                        try:
                            np.savez_compressed(tmp, A=lkp.refs['A'], Q=lkp.refs['Q'], a=self.a, f=f, fs=fs,
                                                keys=np.array(keys), **{f'tab_{k}': lkp[k] for k in keys})
                            os.replace(tmp, fpath)
                        except OSError:
                            try:
                                os.remove(tmp)
                            except OSError:
                                pass
        cache[key] = lkp
        return lkp

    def getArange(self, drive):
        return (0., self.getLookup().refs['A'].max())

    # ------------------------------------------------------------------------------------------
    # queues (nbls.py:447-476)
    # ------------------------------------------------------------------------------------------
    @classmethod
    @Model.checkOutputDir
    def simQueue(cls, freqs, amps, durations, offsets, PRFs, DCs, fs, methods, qss_vars, **kwargs):
        ''' [[drive, pp, fs, method, qss_vars], ...]: drives (f outer, A inner) x protocols
            (CW not repeated over PRFs) x fs x methods. '''
        if ('full' in methods or 'hybrid' in methods) and kwargs.get('outputdir') is None:
            logger.warning('Running cumbersome simulation(s) without file saving')
        if amps is None:
            amps = [None]
        drives = AcousticDrive.createQueue(freqs, amps)
        protocols = PulsedProtocol.createQueue(durations, offsets, PRFs, DCs)
        queue = []
        for drive in drives:
            for pp in protocols:
                for cov in fs:
                    for method in methods:
                        queue.append([drive, pp, cov, method, qss_vars])
        return queue

    @classmethod
    @Model.checkOutputDir
    def simQueueBurst(cls, freqs, amps, durations, PRFs, DCs, BRFs, nbursts, fs, methods, qss_vars,
                      **kwargs):
        ''' simQueue for burst protocols (nbls.py:478-494) '''
        if ('full' in methods or 'hybrid' in methods) and kwargs.get('outputdir') is None:
            logger.warning('Running cumbersome simulation(s) without file saving')
        if amps is None:
            amps = [None]
        drives = AcousticDrive.createQueue(freqs, amps)
        protocols = BurstProtocol.createQueue(durations, PRFs, DCs, BRFs, nbursts)
        return [[drive, pp, cov, method, qss_vars] for drive in drives for pp in protocols
                for cov in fs for method in methods]

    # ------------------------------------------------------------------------------------------
    # input validation (nbls.py:496-511, pneuron.py:469-479)
    # ------------------------------------------------------------------------------------------
    def intMethods(self):
        return {'full': None, 'hybrid': None, 'sonic': self._batched_simulate}

    def checkInputs(self, drive, pp, fs, method, qss_vars, _checked_protocols=None):
        ''' `_checked_protocols` (a set, batched calls only): protocols of the queue whose events have been
            validated already -- a sweep repeats a few protocols over many amplitudes '''
        if not isinstance(drive, Drive):
            raise TypeError('Invalid "drive" parameter (must be an "Drive" object)')
        if not isinstance(pp, TimeProtocol):
            raise TypeError('Invalid time protocol (must be "TimeProtocol" instance)')
        known = False
        if _checked_protocols is not None:
            try:
                key = (pp, getattr(pp, 'modfactor', None))
                known = key in _checked_protocols       # StimObject: hashed and compared by class and parameters
            except TypeError:                           # array-valued parameters (CustomProtocol)
                known = False
        if not known:
            _, xevents = zip(*pp.stimEvents())
            if np.any(np.array([xevents]) < 0.):
                raise ValueError('Invalid time protocol: contains negative modulators')
            if _checked_protocols is not None:
                try:
                    _checked_protocols.add((pp, getattr(pp, 'modfactor', None)))
                except TypeError:
                    pass
        if not isinstance(fs, float):
            raise TypeError('Invalid "fs" parameter (must be float typed)')
        if qss_vars is not None:
            if not isIterable(qss_vars) or not isinstance(qss_vars[0], str):
                raise ValueError('Invalid QSS variables: must be None or an iterable of strings')
            sn = self.pneuron.statesNames()
            for item in qss_vars:
                if item not in sn:
                    raise ValueError(f'Invalid QSS variable: {item} (must be in {sn}')
        if method not in list(self.intMethods().keys()):
            raise ValueError(f'Invalid integration method: "{method}"')

    def desc(self, meta):
        method = meta['method'] if 'method' in meta else meta['model']['method']
        fs = meta['fs'] if 'fs' in meta else meta['model']['fs']
        s = f'{self}: {method} simulation @ {meta["drive"].desc}, {meta["pp"].desc}'
        if fs < 1.0:
            s += f', fs = {(fs * 1e2):.2f}%'
        if 'qss_vars' in meta and meta['qss_vars'] is not None:
            s += f" - QSS ({', '.join(meta['qss_vars'])})"
        return s

    @staticmethod
    def getNSpikes(data):
        return PointNeuron.getNSpikes(data)

    @property
    def titrationFunc(self):
        return self.pneuron.titrationFunc

    # ------------------------------------------------------------------------------------------
    # device models + batched execution
    # ------------------------------------------------------------------------------------------
    def _sonicModel(self, f, fs):
        ''' Native model (neuron parameters + 2-D lookup resident on the GPU) for (f, fs). '''
        key = (f, fs, self._device())
        if key not in self._models:
            lkp = self.getLookup2D(f, fs)
            if lkp.inputs != ['A', 'Q']:
                lkp.move('A', 0)
            names = ['V'] + list(self.pneuron.rates)
            missing = [k for k in names if k not in lkp.tables]
            if missing:
                raise KeyError(f'lookup is missing tables {missing}')
            tables = np.array([lkp[k] for k in names] +
                              [np.zeros_like(lkp['V'])] * (len(self._devRates()) - len(self.pneuron.rates)))
            self._models[key] = (_native.SonicModel(
                self.pneuron.name, self.pneuron.device_params(), tables,
                lkp.refs['A'], lkp.refs['Q'], device=self._device()), lkp)
        return self._models[key]

    # A passive neuron has no state; the device models have at least one gate, so it runs on the
    # one-gate data-driven model with a padding gate (rates 0, conductance 0) that the host strips.
    _PAD = '_pad'

    def _devStates(self):
        return [self._PAD] if self.pneuron.is_passive else self.pneuron.statesNames()

    def _devRates(self):
        return [f'alpha{self._PAD}', f'beta{self._PAD}'] if self.pneuron.is_passive else list(self.pneuron.rates)

    def initialConditionsSonic(self):
        ''' y0 = (Qm0, x_inf(Vm0)) (nbls.py:408-411) '''
        if self.pneuron.is_passive:
            return np.array([self.Qm0, 0.])
        ss = self.pneuron.steadyStates()
        return np.array([self.Qm0] + [ss[k](self.pneuron.Vm0) for k in self.pneuron.statesNames()])

    @staticmethod
    def _withLogEvents(events, tstop):
        ''' Progress-log events of EventDrivenSolver.solve (solvers.py:452-457: one every tstop / 100)
            merged into the stimulus events. A log event changes nothing but splits the integration
            into one more segment with its own np.linspace grid, which moves the resampled rows of a
            detailed simulation by ~5e-5 of their range: it is carried as an event that repeats
            the current modulation factor. '''
        tlogs = np.arange(0., tstop, tstop / 100)[1:]
        if tstop not in tlogs:
            tlogs = np.hstack((tlogs, [tstop]))
        merged = sorted(list(events) + [(t, 'log') for t in tlogs], key=lambda e: e[0])
        out, xcur = [], 0.
        for t, x in merged:
            if x != 'log':
                xcur = x
            out.append((float(t), xcur))
        return out

    @staticmethod
    def _sonicLogEvents(pp):
        ''' effective simulations of 5 s and more carry progress-log events whatever the logging
            level (nbls.py:422) '''
        return pp.tstop >= 5

    def _rowsKeptWithLogEvents(self, pp):
        ''' The segment that follows a log event loses its first row (solvers.py:475-478:
            remove_first), the one after a stimulus event keeps it. The device emits every segment
            in full: boolean mask of the rows the reference keeps. '''
        dt = self.pneuron.chooseTimeStep()
        events = sorted(pp.stimEvents(), key=lambda e: e[0])
        tlogs = np.arange(0., pp.tstop, pp.tstop / 100)[1:]
        if pp.tstop not in tlogs:
            tlogs = np.hstack((tlogs, [pp.tstop]))
        merged = sorted(list(events) + [(t, 'log') for t in tlogs], key=lambda e: e[0])
        keep, tnow, prev_log = [True], 0., False
        for t, x in merged:
            n = max(int(np.round((t - tnow) / dt)), 2)
            keep += [not prev_log] + [True] * (n - 1)
            tnow, prev_log = t, x == 'log'
        # the device adds the segment up to tstop after the last event (zero length here)
        n = max(int(np.round((pp.tstop - tnow) / dt)), 2)
        keep += [False] * n
        return np.array(keep)

    def _packConfigs(self, configs, log_events=False):
        ''' (drive, pp) list -> CSR arrays of the C ABI (include/pysonic_amd.h). '''
        A, tstop, ev_t, ev_x, ev_off = [], [], [], [], [0]
        step = self.pneuron.chooseTimeStep()
        # A sweep repeats a few protocols over many amplitudes: the event list of a protocol is built once
        # per distinct (class, parameters) -- 100 times instead of 10 000 in BASELINE config 4.
        memo, slots_of, nev = {}, {}, 0
        for drive, pp in configs:
            cls = type(pp)
            slots = slots_of.get(cls, 0)
            if slots == 0:
                # the parameters of a StimObject live in its __dict__ under '_<name>' (stimobj.Param): read there,
                # not through the descriptors -- this loop runs once per configuration of a sweep
                names = tuple(pp.inputs()) if hasattr(pp, 'inputs') else None
                slots = slots_of[cls] = None if names is None else tuple('_' + k for k in names) + ('modfactor',)
            key = ev = None
            if slots is not None:
                try:
                    key = (cls, *map(pp.__dict__.get, slots))
                    ev = memo.get(key)
                except TypeError:       # array-valued parameters (CustomProtocol): not memoised
                    key = ev = None
            if ev is None:
                events = sorted(pp.stimEvents(), key=lambda e: e[0])   # solvers.py:441-443
                if log_events(pp) if callable(log_events) else log_events:
                    events = self._withLogEvents(events, pp.tstop)
                ev = (np.array([e[0] for e in events], dtype=float),
                      np.array([e[1] for e in events], dtype=float), pp.tstop)
                if key is not None:
                    memo[key] = ev
            A.append(drive.A)
            tstop.append(ev[2])
            ev_t.append(ev[0])
            ev_x.append(ev[1])
            nev += ev[0].size
            ev_off.append(nev)
        cat = lambda parts: np.concatenate(parts) if parts else np.empty(0)
        return (np.array(A, dtype=float), np.array(tstop, dtype=float), np.full(len(A), step, dtype=float),
                cat(ev_t), cat(ev_x), np.array(ev_off, dtype=np.int64))

    @staticmethod
    def _resampleRows(rows, lkp, A):
        ''' More than MAX_NSAMPLES_EFFECTIVE rows (the neurons with a 0.5 us output step, long
            protocols): the reference resamples the solution to ptp(t) / MAX_NSAMPLES_EFFECTIVE before
            it adds Vm (nbls.py:423, solvers.py:172-191, 213-221): linear interpolation of the
            variables, nearest neighbour for the stimulus state, Vm from the lookup at the resampled
            charge (nbls.py:426-428). '''
        from scipy.interpolate import interp1d
        t = rows[:, 0]
        target_dt = np.ptp(t) / MAX_NSAMPLES_EFFECTIVE
        n = max(int(np.round((t[-1] - t[0]) / target_dt)), 2)
        tnew = np.linspace(t[0], t[-1], n)
        out = np.empty((n, rows.shape[1]))
        out[:, 0] = tnew
        out[:, 1] = interp1d(t, rows[:, 1], kind='nearest', assume_sorted=True)(tnew)
        for j in range(2, rows.shape[1] - 1):
            out[:, j] = np.interp(tnew, t, rows[:, j])
        Vm = np.empty(n)
        for sv in np.unique(out[:, 1] * A):
            sel = out[:, 1] * A == sv
            Vm[sel] = lkp.project('A', sv).interpVar1D(out[sel, 2], 'V')
        out[:, -1] = Vm
        return out

    def _toTimeSeries(self, rows, qss_vars=None, lkp=None, A=None):
        ''' Device rows (t, stimstate, Qm, states..., Vm) -> reference DataFrame layout:
            differential variables, Vm, then the quasi-steady-state variables interpolated from
            the lookup of x_inf = alpha / (alpha + beta) on the (A, Q) grid (interpEffVariable on
            lkp_QSS, nbls.py:402-404, 426-430), + Z, ng = NaN columns (nbls.py:432-434). '''
        states = self._devStates()
        if rows.shape[0] > MAX_NSAMPLES_EFFECTIVE and lkp is not None:
            rows = self._resampleRows(rows, lkp, A)
        qss_vars = list(qss_vars or [])
        if not qss_vars and self._PAD not in states:
            # the device block already is the reference's table, minus its two NaN columns: the frame is a view of
            # the launch's host block plus a small block of NaN
            return TimeSeries.from_block(rows, ['Qm'] + states + ['Vm'], nan_columns=('Z', 'ng'))
        cols = {k: rows[:, 2 + i] for i, k in enumerate(['Qm'] + states + ['Vm'])}
        if qss_vars:
            lkp_qss = EffectiveVariablesLookup(
                lkp.refs, {k: lkp[f'alpha{k}'] / (lkp[f'alpha{k}'] + lkp[f'beta{k}'])
                           for k in qss_vars})
            stim, Qm = rows[:, 1], rows[:, 2]
            for k in qss_vars:
                x = np.zeros(stim.size)
                for sv in np.unique(stim * A):
                    sel = stim * A == sv
                    x[sel] = lkp_qss.project('A', sv).interpVar1D(Qm[sel], k)
                cols[k] = x
        order = ['Qm'] + [k for k in states if k not in qss_vars and k != self._PAD] + ['Vm'] + qss_vars
        data = TimeSeries(rows[:, 0], rows[:, 1], {k: cols[k] for k in order})
        for key in ['Z', 'ng']:
            data[key] = np.full(rows.shape[0], np.nan)
        return data

    def _qssMask(self, qss_vars):
        ''' qss_vars -> bit mask over statesNames() (sonic_opts_t.qss_mask). The device handles
            voltage-gated states (x = alpha / (alpha + beta) from the interpolated rates). '''
        mask = 0
        states = self.pneuron.statesNames()
        for k in (qss_vars or []):
            if f'alpha{k}' not in self.pneuron.rates:
                raise NotImplementedError(
                    f'QSS variable "{k}" is not voltage-gated: not supported on the device')
            mask |= 1 << states.index(k)
        return mask

    def runSonicBatch(self, f, fs, configs, traces=True, opts=None, qss_vars=None):
        ''' Integrate a list of (drive, pp) configurations sharing (f, fs) -- and the same
            quasi-steady-state variables, if any -- in one launch.
            :return: (list of row arrays or None, metrics, status, kernel_ms) '''
        return self.runSonicBatches([(f, fs, configs, qss_vars)], traces=traces, opts=opts)[0]

    # sonic_opts_t.chunks of the launches with traces. 0: one launch, its rows copied to the host behind the kernel
    # on the same stream. Cutting the 4096-cell map into 2 - 16 launches on streams of their own was measured and
    # gains nothing (profiles/r03d_e2e_probe.txt): pack_wavefronts equalises the wavefronts -- the cheap
    # configurations share theirs sixteen at a time -- so every part of the batch ends within 15 % of the whole, and
    # beyond three streams the launches take turns on the process's hardware queues.
    PIPELINE_CHUNKS = 0

    def runSonicBatches(self, groups, traces=True, opts=None):
        ''' Several launches IN FLIGHT TOGETHER: groups = [(f, fs, configs, qss_vars), ...], one launch per
            group, each on its own stream(s). A launch lasts as long as its slowest configuration whatever
            its size, so the groups of a sweep over frequencies cost the longest of
            them, not their sum (five 2000-configuration launches one after the other: 109 ms for RS,
            together: the time of one). With traces, the rows are copied to ONE page-locked host block per launch
            behind the kernels, on their streams; a large launch is pipelined.
            :return: [(rows or None, metrics, status, kernel_ms), ...] in group order; `rows` is a sequence of
                row arrays (RowBlocks: views of the host block, made on demand) '''
        batches = []
        try:
            for f, fs, configs, qss_vars in groups:
                model, _ = self._sonicModel(f, fs)
                chunks = self.PIPELINE_CHUNKS if traces else 0
                o = _native.default_opts(**{'chunks': chunks, **self.solver_opts, **(opts or {}),
                                            'write_traces': int(bool(traces)),
                                            'qss_mask': self._qssMask(qss_vars)})
                batches.append(model.prepare(*self._packConfigs(configs, log_events=self._sonicLogEvents),
                                             self.initialConditionsSonic(), o))
            for batch in batches:
                batch.launch(to_host=bool(traces))
            out = []
            for (f, fs, configs, qss_vars), batch in zip(groups, batches):
                kernel_ms = batch.sync()
                _, metrics, status = batch.fetch(traces=False)
                rows = None
                if traces:
                    masks = {}
                    for i, (_, pp) in enumerate(configs):
                        if pp.tstop >= 5:                               # _sonicLogEvents
                            keep = self._rowsKeptWithLogEvents(pp)
                            assert keep.size == batch.n_rows[i], (keep.size, batch.n_rows[i])
                            masks[i] = keep
                    rows = RowBlocks(batch.host_traces, batch.row_start, batch.n_rows, masks)
                if np.any(status & (_native.ST_MAX_STEPS | _native.ST_STEP_UNDERFLOW)):
                    logger.warning('%d configuration(s) hit the integrator step limits',
                                   int(np.count_nonzero(status & 6)))
                out.append((rows, metrics, status, kernel_ms))
        finally:
            for batch in batches:
                batch.close()
        return out

    def _resolveSimulateCalls(self, calls):
        ''' queue items -> [drive, pp, fs, method, qss_vars] lists, validated (checkInputs) and logged like
            Model.logDesc. Positional items -- what simQueue builds -- are read directly; anything else goes
            through the signature of simulate(). '''
        import inspect
        sig, resolved, checked = None, [], set()
        methods = self.intMethods()
        info = logger.isEnabledFor(logging.INFO)
        for args, kwargs in calls:
            n = len(args)
            if not kwargs and 2 <= n <= 5:
                p = [args[0], args[1], args[2] if n > 2 else 1., args[3] if n > 3 else 'sonic',
                     args[4] if n > 4 else None]
            else:
                if sig is None:
                    sig = inspect.signature(self.simulate)
                ba = sig.bind(*args, **kwargs)
                ba.apply_defaults()
                p = [ba.arguments[k] for k in ('drive', 'pp', 'fs', 'method', 'qss_vars')]
            drive, pp, fs, method, qss_vars = p
            # the checks of checkInputs that need nothing but the types; the events of the protocols are validated
            # once per launch by the library (negative modulators, order: sonic_batch_prepare raises the same
            # ValueError), quasi-steady-state variables by the full method
            if not isinstance(drive, Drive) or not isinstance(pp, TimeProtocol) or type(fs) is not float or \
                    qss_vars is not None or method not in methods or method != 'sonic':
                self.checkInputs(drive, pp, fs, method, qss_vars, _checked_protocols=checked)
            if info:       # one line per simulation, as Model.logDesc (model.py:136-148)
                logger.info(self.desc({'simkey': self.simkey, 'model': self.meta, 'drive': drive, 'pp': pp,
                                       'fs': fs, 'method': method, 'qss_vars': qss_vars}))
            resolved.append(p)
        return resolved

    def _batched_simulate(self, calls, strict=False):
        ''' Execute a queue of simulate() calls (list of (args, kwargs)) on the device, one launch
            per (f, fs) group, and return the (data, meta) pairs in queue order (SimResults). '''
        resolved = self._resolveSimulateCalls(calls)
        # unresolved drives (A is None): titrate them all together first (model.py:187-215)
        iunres = [i for i, p in enumerate(resolved) if p[0].A is None and p[0].is_searchable]
        dead = set()
        if iunres:
            thrs = self._batched_titrate([([resolved[i][0], resolved[i][1]],
                                           {'fs': resolved[i][2], 'method': resolved[i][3],
                                            'qss_vars': resolved[i][4]}) for i in iunres])
            for i, xthr in zip(iunres, thrs):
                if np.isnan(xthr):
                    logger.error('Could not find threshold US pressure amplitude')
                    dead.add(i)
                else:
                    resolved[i][0] = resolved[i][0].updatedX(xthr)
        if not dead:
            return self._simulate_resolved(resolved, strict=strict)
        live = [i for i in range(len(resolved)) if i not in dead]
        part = self._simulate_resolved([resolved[i] for i in live], strict=strict)
        out = SimResults(len(resolved))
        for k, i in enumerate(live):
            out._set(i, part.__getitem__, k)
        return out

    def _simulate_resolved(self, resolved, strict=False):
        ''' resolved: [drive, pp, fs, method, qss_vars] per simulation -> SimResults '''
        out = SimResults(len(resolved))
        # detailed (full) simulations: one launch for all of them
        ifull = [i for i, p in enumerate(resolved) if p[3] == 'full']
        if ifull:
            (frames, _, _), tcomp = timer(self.runFullBatch)([(resolved[i][0], resolved[i][1], resolved[i][2]) for i in ifull])
            for j, i in enumerate(ifull):
                drive, pp, fs, method, qss_vars = resolved[i]
                meta = {'simkey': self.simkey, 'model': self.meta, 'drive': drive, 'pp': pp, 'fs': fs, 'method': method,
                        'qss_vars': qss_vars, 'tcomp': tcomp / len(ifull)}
                out._set_value(i, (frames[j], meta))
        # hybrid simulations (dense periods + sparse phases): one launch as well
        ihyb = [i for i, p in enumerate(resolved) if p[3] == 'hybrid']
        if ihyb:
            (frames, _, _, _), tcomp = timer(self.runHybridBatch)(
                [(resolved[i][0], resolved[i][1], resolved[i][2]) for i in ihyb])
            for j, i in enumerate(ihyb):
                drive, pp, fs, method, qss_vars = resolved[i]
                meta = {'simkey': self.simkey, 'model': self.meta, 'drive': drive, 'pp': pp, 'fs': fs, 'method': method,
                        'qss_vars': qss_vars, 'tcomp': tcomp / len(ihyb)}
                out._set_value(i, (frames[j], meta))
        groups = {}
        for i, p in enumerate(resolved):
            if p[3] == 'sonic':
                qss = tuple(p[4]) if p[4] is not None else None
                key = (p[0].f, p[2], qss)
                g = groups.get(key)
                if g is None:
                    g = groups[key] = []
                g.append(i)
        glist = list(groups.items())
        if glist:
            self.setTissueModulus(resolved[glist[0][1][0]][0])
        results, tcomp_all = timer(self.runSonicBatches)(
            [(f, fs, [(resolved[i][0], resolved[i][1]) for i in idxs], qss) for (f, fs, qss), idxs in glist]) \
            if glist else ([], 0.)
        nsonic = max(1, sum(len(idxs) for _, idxs in glist))
        for ((f, fs, qss), idxs), (rows, metrics, status, _) in zip(glist, results):
            lkp = self._sonicModel(f, fs)[1]
            bad = np.flatnonzero(status & _native.ST_Q_OUT_OF_RANGE)
            if bad.size:
                # the reference ends such a simulation with this ValueError (isWithin inside
                # the right-hand side, lookups.py:320-321, utils.py:348); here the rows from the
                # exit on are NaN and the error is raised by simulate() / logged by a batch
                Qlo, Qhi = lkp.refs['Q'][[0, -1]]
                for j in bad:
                    Qbad = metrics[j, _native.M_QMIN] if metrics[j, _native.M_QMIN] < Qlo else \
                        metrics[j, _native.M_QMAX]
                    msg = f'Q value ({Qbad}) out of [{Qlo}, {Qhi}] interval'
                    if strict:
                        raise ValueError(msg)
                    logger.error('%s: %s', resolved[idxs[j]][0].desc, msg)
            maker = _SonicLaunchOutputs(self, rows, [resolved[i] for i in idxs], qss, lkp, tcomp_all / nsonic)
            for j, i in enumerate(idxs):
                out._set(i, maker, j)
        return out

    def runFullBatch(self, configs, opts=None, loglevel=None):
        ''' Detailed NICE model (method='full', nbls.py:331-354) for a list of (drive, pp, fs)
            sharing this sonophore, in one launch. Like the reference, the integration is split
            at 100 progress-log events when the logger level (or `loglevel`) is INFO or lower.
            :return: (list of TimeSeries with columns t, stimstate, Z, ng, Qm, states..., Vm;
                      status array; kernel_ms) '''
        # dense grid = 1000 points per acoustic period (drives.py:276-279): a 100 ms protocol at
        # 500 kHz is 5e7 dense points per configuration (minutes of GPU time, hours on the CPU)
        npts = max(pp.tstop * d.f * 1e3 for d, pp, _ in configs)
        if npts > self.max_full_dense_points:
            raise ValueError(
                f'full simulation of {npts:.2g} dense points per configuration exceeds '
                f'max_full_dense_points = {self.max_full_dense_points:.2g}: raise that attribute '
                'to run it anyway')
        freqs = {d.f for d, _, _ in configs}
        if len(freqs) > 1 and self.d > 0.:
            raise NotImplementedError('mixed frequencies with an embedding depth need one launch '
                                      'per frequency')
        self.setTissueModulus(configs[0][0])
        level = logger.getEffectiveLevel() if loglevel is None else loglevel
        A, tstop, _, ev_t, ev_x, ev_off = self._packConfigs(
            [(d, pp) for d, pp, _ in configs], log_events=level <= logging.INFO)
        phis = {d.phi for d, _, _ in configs}
        if len(phis) > 1:
            raise NotImplementedError('mixed drive phases need one launch per phase')
        o = _native.full_default_opts(**{**self.full_opts, **(opts or {}), 'phi': phis.pop()})
        traces, row_off, status, nsteps, ms = _native.full_batch_run(
            self.pneuron.name, self.pneuron.device_params(), self.device_params(),
            [d.f for d, _, _ in configs], A, [fs for _, _, fs in configs], tstop, ev_t, ev_x,
            ev_off, self.initialConditionsSonic(), o, device=self._device())
        if np.any(status & 2):
            raise ValueError('P_QS not changing sign within deflection interval')
        if np.any(status & 4):
            logger.warning('%d configuration(s) hit the step budget', int(np.count_nonzero(status & 4)))
        names = ['Z', 'ng', 'Qm'] + self._devStates() + ['Vm']
        frames = []
        for i in range(len(configs)):
            r = traces[row_off[i]:row_off[i + 1]]
            frames.append(TimeSeries(r[:, 0], r[:, 1], {k: r[:, 2 + j] for j, k in enumerate(names) if k != self._PAD}))
        return frames, status, ms

    def runHybridBatch(self, configs, opts=None):
        ''' Hybrid scheme (method='hybrid', nbls.py:356-387 / HybridSolver, solvers.py:483-633)
            for a list of (drive, pp, fs) sharing this sonophore, in one launch: per interval of
            HYBRID_UPDATE_INTERVAL the detailed model runs whole acoustic periods until Z and ng
            are periodically stable, then only (Qm, states) advance with U, Z, ng replayed from
            the last period. Rows are resampled to CLASSIC_TARGET_DT like the reference.
            :return: (list of TimeSeries t, stimstate, Z, ng, Qm, states..., Vm; status array;
                      number of dense periods per configuration; kernel_ms) '''
        # at best 2 of the 250 periods of an update interval run dense, at worst all of them (OFF
        # phases that never become periodically stable): guard like method='full', 10 x looser
        npts = max(pp.tstop * d.f * 1e3 for d, pp, _ in configs)
        if npts > 10 * self.max_full_dense_points:
            raise ValueError(
                f'hybrid simulation of {npts:.2g} dense-equivalent points per configuration exceeds '
                f'10 x max_full_dense_points = {10 * self.max_full_dense_points:.2g}: raise that '
                'attribute to run it anyway')
        freqs = {d.f for d, _, _ in configs}
        if len(freqs) > 1 and self.d > 0.:
            raise NotImplementedError('mixed frequencies with an embedding depth need one launch '
                                      'per frequency')
        self.setTissueModulus(configs[0][0])
        # (the reference adds progress-log events to hybrid runs below INFO only, nbls.py:377)
        A, tstop, _, ev_t, ev_x, ev_off = self._packConfigs([(d, pp) for d, pp, _ in configs])
        phis = {d.phi for d, _, _ in configs}
        if len(phis) > 1:
            raise NotImplementedError('mixed drive phases need one launch per phase')
        o = _native.full_default_opts(**{**self.full_opts, **(opts or {}), 'phi': phis.pop()})
        traces, row_off, status, nsteps, ncycles, ms = _native.hybrid_batch_run(
            self.pneuron.name, self.pneuron.device_params(), self.device_params(),
            [d.f for d, _, _ in configs], A, [fs for _, _, fs in configs], tstop, ev_t, ev_x,
            ev_off, self.initialConditionsSonic(), o, device=self._device())
        if np.any(status & 2):
            raise ValueError('P_QS not changing sign within deflection interval')
        if np.any(status & 16):
            raise AssertionError('incorrect bounds for number of cycles (min > max)')   # solvers.py:347
        if np.any(status & 32):
            raise ValueError('Invalid index')                                           # solvers.py:307
        if np.any(status & 4):
            logger.warning('%d configuration(s) hit the step budget', int(np.count_nonzero(status & 4)))
        names = ['Z', 'ng', 'Qm'] + self._devStates() + ['Vm']
        frames = []
        for i in range(len(configs)):
            r = traces[row_off[i]:row_off[i + 1]]
            frames.append(TimeSeries(r[:, 0], r[:, 1], {k: r[:, 2 + j] for j, k in enumerate(names) if k != self._PAD}))
        return frames, status, ncycles, ms

    # ------------------------------------------------------------------------------------------
    # titration (threshold.py:335-363, nbls.py:559-571)
    # ------------------------------------------------------------------------------------------
    def _titrate_uncached(self, calls):
        ''' Queue of titrate(drive, pp, fs=1., method='sonic', qss_vars=None, xfunc=None,
            Arange=None) calls -> list of threshold amplitudes (Pa, nan if none), queue order.
            All searches advance together: one metrics-only launch per bisection round. '''
        import inspect
        from ..threshold import threshold_search, titrate_many
        sig = inspect.signature(self.titrate)
        items = []
        for args, kwargs in calls:
            ba = sig.bind(*args, **kwargs)
            ba.apply_defaults()
            p = dict(ba.arguments)
            if p['method'] != 'sonic' or p['qss_vars'] is not None:
                raise NotImplementedError('titration is implemented for the sonic method only')
            self.checkInputs(p['drive'].updatedX(0.), p['pp'], p['fs'], p['method'], p['qss_vars'])
            items.append(p)
        searches = []
        for p in items:
            drive = p['drive']
            Arange = p['Arange'] if p['Arange'] is not None else self.getArange(drive)
            searches.append(threshold_search(
                Arange, x0=drive.xvar_initial, rel_eps_thr=drive.xvar_rel_thr,
                eps_thr=drive.xvar_thr, precheck=drive.xvar_precheck))
        # excitation predicate = "at least one spike" (pneuron.py:324-326,578-585) unless the
        # neuron overrides titrationFunc (STN: isSilenced) -> then traces are analysed on the host
        default_pred = (type(self.pneuron).titrationFunc.__func__ is
                        PointNeuron.titrationFunc.__func__)

        def evaluate_round(pending):
            ''' pending: [(search index, amplitude)] -> [is_above] '''
            res = [None] * len(pending)
            groups = {}
            for k, (i, A) in enumerate(pending):
                p = items[i]
                fast = p['xfunc'] is None and default_pred
                groups.setdefault((p['drive'].f, p['fs'], fast), []).append((k, i, float(A)))
            for (f, fs, fast), members in groups.items():
                configs = [(items[i]['drive'].updatedX(A), items[i]['pp']) for _, i, A in members]
                rows, metrics, status, _ = self.runSonicBatch(f, fs, configs, traces=not fast)
                redo = []
                for j, (k, i, A) in enumerate(members):
                    if fast and metrics[j, _native.M_SPKFLAGS] == 0 and status[j] == 0:
                        res[k] = metrics[j, _native.M_NSPIKES] > 0          # isExcited
                    elif fast:
                        redo.append((j, k, i))
                    else:
                        xfunc = items[i]['xfunc'] or self.titrationFunc
                        res[k] = bool(xfunc(self._toTimeSeries(rows[j])))
                if redo:
                    rows2, _, _, _ = self.runSonicBatch(f, fs, [configs[j] for j, _, _ in redo])
                    for (j, k, i), r in zip(redo, rows2):
                        res[k] = bool(self.titrationFunc(self._toTimeSeries(r)))
            return res

        thresholds, nrounds = titrate_many(evaluate_round, searches)
        logger.info(f'{len(items)} titration(s) completed in {nrounds} batched round(s)')
        return [float(x) for x in thresholds]

    def _batched_titrate(self, calls):
        ''' The batched titration (_titrate_uncached) behind the titration log (the reference's @logCache around nbls.titrate,
            nbls.py:559, utils.py:457-497): calls found in the log return their logged threshold, the
            others are titrated together and appended. Keys are the reference's call signatures, so its
            own astim_titrations.log can be used as the log file (PYSONIC_AMD_TITRATIONS=<path>). '''
        cache = self.titration_cache
        if cache is None:
            return self._titrate_uncached(calls)
        sigs = [methodCallSignature(self.titrate, a, k) for a, k in calls]
        out = [cache.get(s) for s in sigs]
        todo = [i for i, v in enumerate(out) if v is None]
        if todo:
            for i, thr in zip(todo, self._titrate_uncached([calls[i] for i in todo])):
                out[i] = thr
                cache.put(sigs[i], thr)
        return [float(x) for x in out]

    def titrate(self, drive, pp, fs=1., method='sonic', qss_vars=None, xfunc=None, Arange=None):
        ''' Threshold amplitude (Pa) for neural excitation by binary search (nbls.py:559-571);
            nan if no threshold lies within the amplitude range of the lookup. '''
        return self._batched_titrate([([drive, pp], dict(fs=fs, method=method, qss_vars=qss_vars,
                                                        xfunc=xfunc, Arange=Arange))])[0]

    def getQuasiSteadyStates(self, f, amps=None, charges=None, DC=1.0, squeeze_output=False):
        ''' Quasi-steady states of the neuron's states over (amplitude, charge) at one frequency and
            duty cycle (nbls.py:573-603): duty-cycle-averaged lookup (projectDC), projected at this
            radius and `f` (and at `charges` if given), amplitude first; the states evaluated on it.
            :return: (lookup of the averaged effective variables, lookup of the quasi-steady states) '''
        lkp = self.getLookup().projectDC(amps=amps, DC=DC).projectN({'a': self.a, 'f': f})
        if 'fs' in lkp.refs and lkp.refs['fs'].size == 1:
            lkp = lkp.project('fs', lkp.refs['fs'][0])     # the packaged lookup carries a one-point fs axis
        if charges is not None:
            lkp = lkp.project('Q', charges)
        lkp.move('A', 0)
        QSS = EffectiveVariablesLookup(
            lkp.refs, {k: v(lkp) for k, v in self.pneuron.quasiSteadyStates().items()})
        if squeeze_output:
            QSS, lkp = QSS.squeeze(), lkp.squeeze()
        return lkp, QSS

    def simulate(self, drive, pp, fs=1., method='sonic', qss_vars=None):
        ''' Simulate one configuration; returns (TimeSeries, meta) like nbls.py:513-536
            (None if the drive is unresolved and no threshold is found). Batch of one on the GPU. '''
        out = self._batched_simulate([([drive, pp, fs, method, qss_vars], {})], strict=True)[0]
        if out is None:
            return None
        data, meta = out
        nspikes = self.getNSpikes(data)
        logger.debug(f'{nspikes} spike{"s" if nspikes != 1 else ""} detected')
        return data, meta

    def _batched_simAndSave(self, calls):
        ''' Queue of simAndSave() calls: simulate the missing outputs in one batch, then write
            one pickle per configuration (utils.simAndSave semantics incl. overwrite=False). '''
        import pickle
        paths, todo = [None] * len(calls), []
        for i, (args, kwargs) in enumerate(calls):
            kwargs = dict(kwargs)
            outputdir = kwargs.pop('outputdir', '.')
            overwrite = kwargs.pop('overwrite', True)
            full_output = kwargs.pop('full_output', True)
            fpath = os.path.join(outputdir, f'{self.filecode(*args)}.pkl')
            paths[i] = fpath
            if os.path.isfile(fpath) and not overwrite:
                logger.warning(f'File "{os.path.basename(fpath)}" already present in directory '
                               f'"{outputdir}" -> preserving')
                continue
            todo.append((i, args, kwargs, full_output))
        results = self._batched_simulate([(a, k) for _, a, k, _ in todo])
        for (i, _, _, full_output), (data, meta) in zip(todo, results):
            if not full_output:
                data.dumpOutputsOtherThan(['Qm', 'Vm'])
            with open(paths[i], 'wb') as fh:
                pickle.dump({'meta': meta, 'data': data}, fh)
        return paths

    # ------------------------------------------------------------------------------------------
    # effective-variable computation = lookup generation (nbls.py:153-222)
    # ------------------------------------------------------------------------------------------
    def runMechBatch(self, f, A, Qm, fs, opts=None, overtones=None):
        ''' computeEffVars for arrays of cells (f, A, Qm) sharing this sonophore, in one launch.
            :param overtones: optional (n, n_overtones, 2) amplitudes and phases of the charge overtones
            :return: effvars (n, n_fs, 1 + n_rates) with columns ['V'] + pneuron.rates,
                (with overtones: (n, n_fs, n_overtones, 2) amplitudes and phases of Vm,)
                ncycles (n,), status (n,), kernel_ms '''
        f = np.atleast_1d(np.asarray(f, dtype=float))
        drive_f = float(f[0])
        if np.any(f != drive_f) and self.d > 0.:
            raise NotImplementedError('mixed frequencies with an embedding depth need one launch '
                                      'per frequency (tissue modulus depends on f)')
        self.kA_tissue = 2 * (self.alpha * drive_f) * self.d      # setTissueModulus, bls.py:583-586
        o = _native.mech_default_opts(**(opts or {}))
        return _native.mech_batch_run(self.pneuron.name, self.device_params(), f, A, Qm, fs, o,
                                      device=self._device(), overtones=overtones)

    def _batched_computeEffVars(self, calls):
        ''' Queue of computeEffVars(drive, fs, Qm0) calls -> [(effvars_list, tcomp), ...] in queue
            order, one launch per distinct fs array. '''
        import inspect
        sig = inspect.signature(self.computeEffVars)
        items = []
        for args, kwargs in calls:
            ba = sig.bind(*args, **kwargs)
            ba.apply_defaults()
            p = dict(ba.arguments)
            BilayerSonophore.checkInputs(p['drive'], p['Qm0'])
            if p['drive'].A is None:
                raise ValueError('computeEffVars needs a resolved drive amplitude')
            fs = np.atleast_1d(np.asarray(p['fs'], dtype=float))
            ov = None if p['Qm_overtones'] is None else [tuple(map(float, x)) for x in p['Qm_overtones']]
            items.append((p['drive'], fs, float(p['Qm0']), ov))
        groups = {}
        for i, (drive, fs, Qm0, ov) in enumerate(items):
            groups.setdefault((tuple(fs), drive.phi, 0 if ov is None else len(ov)), []).append(i)
        out = [None] * len(items)
        keys = ['V'] + list(self.pneuron.rates)
        for (fs, phi, nov), idxs in groups.items():
            f = [items[i][0].f for i in idxs]
            A = [items[i][0].A for i in idxs]
            Q = [items[i][2] for i in idxs]
            ovs = np.array([items[i][3] for i in idxs]) if nov > 0 else None
            res, tcomp = timer(self.runMechBatch)(f, A, Q, np.array(fs), {'phi': phi}, ovs)
            eff, ncyc, status = res[0], res[-3], res[-2]
            ovout = res[1] if nov > 0 else None
            if np.any(status & 2):
                raise ValueError('P_QS not changing sign within deflection interval')
            if np.any(status & 1):
                logger.warning('Deflection out of range in %d cell(s)', int(np.count_nonzero(status & 1)))
            for j, i in enumerate(idxs):
                effvars_list = []
                for k in range(len(fs)):
                    # key order of the reference: V, (A_V1, phi_V1, ...), rates (nbls.py:191-204)
                    ev = {'V': eff[j, k, 0]}
                    for m in range(nov):
                        ev[f'A_V{m + 1}'], ev[f'phi_V{m + 1}'] = ovout[j, k, m]
                    ev.update(zip(keys[1:], eff[j, k, 1:1 + len(keys) - 1]))
                    effvars_list.append(ev)
                out[i] = (effvars_list, tcomp / len(idxs))
        return out

    def computeEffVars(self, drive, fs, Qm0, Qm_overtones=None):
        ''' Effective (cycle-averaged) membrane potential and rate constants for an imposed charge
            density: returns (list of dicts -- one per coverage fraction --, computation time), like
            the reference's @timer-decorated method (nbls.py:153-222). Batch of one on the GPU. '''
        return self._batched_computeEffVars([([drive, fs, Qm0, Qm_overtones], {})])[0]

    def computeLookup(self, freqs, amps, charges, fs=1., overtones=None):
        ''' All (f, A, Q) cells of a lookup in one launch (the inner part of
            scripts/run_lookups.py:99-175 for one radius and one coverage fraction).
            :param overtones: optional list of (AQ_ref, phiQ_ref) reference vectors, one pair per
                charge overtone (run_lookups.py:105-127: 5 amplitudes up to 100 nC/cm2 x 5 phases):
                the lookup gains the dimensions AQ1, phiQ1, ... and the tables A_V1, phi_V1, ...
            :return: EffectiveVariablesLookup with refs (f, A, Q[, AQi, phiQi]) and tables
                V [+ A_Vi, phi_Vi] + rates + ncycles '''
        freqs, amps, charges = [np.atleast_1d(np.asarray(x, dtype=float))
                                for x in (freqs, amps, charges)]
        refs = {'f': freqs, 'A': amps, 'Q': charges}
        nov = 0 if overtones is None else len(overtones)
        for i in range(nov):
            refs[f'AQ{i + 1}'] = np.atleast_1d(np.asarray(overtones[i][0], dtype=float))
            refs[f'phiQ{i + 1}'] = np.atleast_1d(np.asarray(overtones[i][1], dtype=float))
        grids = np.meshgrid(*refs.values(), indexing='ij')
        shape = grids[0].shape
        F, A, Q = [g.ravel() for g in grids[:3]]
        if nov > 0:
            ov = np.stack([np.stack([grids[3 + 2 * i].ravel(), grids[4 + 2 * i].ravel()], axis=-1)
                           for i in range(nov)], axis=1)               # (n, nov, 2)
            eff, ovout, ncyc, status, ms = self.runMechBatch(F, A, Q, [fs], overtones=ov)
        else:
            eff, ncyc, status, ms = self.runMechBatch(F, A, Q, [fs])
        if np.any(status & 2):
            raise ValueError('P_QS not changing sign within deflection interval')
        keys = ['V'] + list(self.pneuron.rates)
        tables = {'V': eff[:, 0, 0].reshape(shape)}
        for i in range(nov):
            tables[f'A_V{i + 1}'] = ovout[:, 0, i, 0].reshape(shape)
            tables[f'phi_V{i + 1}'] = ovout[:, 0, i, 1].reshape(shape)
        for i, k in enumerate(keys[1:]):
            tables[k] = eff[:, 0, 1 + i].reshape(shape)
        lkp = EffectiveVariablesLookup(refs, tables)
        lkp.ncycles = ncyc.reshape(shape)
        lkp.kernel_ms = ms
        return lkp


class DrivenNeuronalBilayerSonophore(NeuronalBilayerSonophore):
    ''' Sonophore model with a constant injected current (nbls.py:674-721): Idrive (mA/m2) adds
        Idrive * 1e-3 to dQm/dt of the effective and of the detailed system -- carried to the device
        as the `idrive` integrator option. '''

    simkey = 'DASTIM'

    def __init__(self, Idrive, *args, **kwargs):
        self.Idrive = Idrive
        super().__init__(*args, **kwargs)
        self.solver_opts['idrive'] = float(Idrive)
        self.full_opts['idrive'] = float(Idrive)

    def __repr__(self):
        return super().__repr__()[:-1] + f', Idrive = {self.Idrive:.2f} mA/m2)'

    def copy(self):
        return self.__class__(self.Idrive, self.a, self.pneuron, embedding_depth=self.d)

    def __eq__(self, other):
        return super().__eq__(other) and self.Idrive == other.Idrive

    def __hash__(self):
        return hash((self.a, self.pneuron.name, self.d, self.Idrive))

    @classmethod
    def initFromMeta(cls, meta):
        from ..neurons import getPointNeuron
        return cls(meta['Idrive'], meta['a'], getPointNeuron(meta['neuron']),
                   embedding_depth=meta['d'])

    @property
    def meta(self):
        return {**super().meta, 'Idrive': self.Idrive}

    def filecodes(self, *args):
        return {**super().filecodes(*args), 'Idrive': f'Idrive{self.Idrive:.1f}mAm2'}
