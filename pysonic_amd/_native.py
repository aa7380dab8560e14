# -*- coding: utf-8 -*-
''' ctypes binding of libpysonic_amd.so (C ABI: include/pysonic_amd.h).

    The library is built in-tree by `python -m pysonic_amd.build` (or __graft_entry__.build()):
    hipcc --offload-arch=gfx950. There is NO CPU fallback: if the shared object is missing or no
    GPU is usable, the calls below raise -- loudly -- instead of computing something else.
'''
import ctypes
import threading
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PYSONIC_AMD_LIB', os.path.join(PKG_DIR, '_lib', 'libpysonic_amd.so'))

ABI_VERSION = 6
SONIC_OK = 0
SONIC_EINVAL = -1
SONIC_ERANGE = -2
SONIC_EHIP = -3
SONIC_ENODEV = -4
SONIC_NMETRICS = 16
M_NSTEPS, M_NREJ, M_NROWS, M_QMIN, M_QMAX, M_QLAST, M_NSPIKES, M_TFIRST, M_TLAST, M_SUMINVISI, \
    M_SPKFLAGS = range(11)
M_NCAPPED, M_NREJ_NODE, M_NCROSS = 12, 13, 14       # what set the steps (include/pysonic_amd.h)

ST_Q_OUT_OF_RANGE = 1
ST_STEP_UNDERFLOW = 2
ST_MAX_STEPS = 4

NEURON_IDS = {'RS': 0, 'FS': 1, 'LTS': 2, 'RE': 3, 'TC': 4, 'STN': 5, 'IB': 6, 'HHseg': 7, 'SWnode': 8,
              'MRGnode': 9, 'SUseg': 10, 'FHnode': 11}
PASSIVE_NEURON_ID = 12     # names are parametric: pas_Cm0_<..>uF_cm2_gLeak_<..>S_m2_ELeak_<..>mV


def neuron_id(name):
    if name in NEURON_IDS:
        return NEURON_IDS[name]
    if name.startswith('pas_'):
        return PASSIVE_NEURON_ID
    raise KeyError(f'neuron "{name}" has no device model')


class NativeLibraryError(RuntimeError):
    ''' The HIP extension is missing / failed: there is no fallback path. '''


class SonicOpts(ctypes.Structure):
    _fields_ = [('rtol', ctypes.c_double), ('atol', ctypes.c_double), ('h0', ctypes.c_double),
                ('hmin', ctypes.c_double), ('max_steps', ctypes.c_int),
                ('write_traces', ctypes.c_int), ('qss_mask', ctypes.c_int), ('idrive', ctypes.c_double),
                ('chunks', ctypes.c_int)]


class MechOpts(ctypes.Structure):
    _fields_ = [('rtol', ctypes.c_double), ('max_steps', ctypes.c_int),
                ('ncycles_max', ctypes.c_int), ('phi', ctypes.c_double)]


class FullOpts(ctypes.Structure):
    _fields_ = [('rtol', ctypes.c_double), ('max_steps', ctypes.c_int),
                ('target_dt', ctypes.c_double), ('phi', ctypes.c_double), ('idrive', ctypes.c_double),
                ('kernel', ctypes.c_int), ('stiff', ctypes.c_int)]


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_llp = ctypes.POINTER(ctypes.c_longlong)
_vp = ctypes.c_void_p

# symbol -> (restype, argtypes): every entry point include/pysonic_amd.h declares
SIGNATURES = {
    'sonic_abi_version': (ctypes.c_int, []),
    'sonic_device_count': (ctypes.c_int, []),
    'sonic_last_error': (ctypes.c_char_p, []),
    'sonic_default_opts': (None, [ctypes.POINTER(SonicOpts)]),
    'sonic_neuron_nstates': (ctypes.c_int, [ctypes.c_int]),
    'sonic_neuron_ntables': (ctypes.c_int, [ctypes.c_int]),
    'sonic_neuron_nparams': (ctypes.c_int, [ctypes.c_int]),
    'sonic_model_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp,
                                          ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(_vp)]),
    'sonic_model_destroy': (None, [_vp]),
    'sonic_count_rows': (ctypes.c_int, [_dp, _dp, _dp, _llp, ctypes.c_longlong, _llp]),
    'sonic_batch_fetch_strided': (ctypes.c_int, [_vp, _dp, ctypes.c_longlong, _dp, _ip]),
    'sonic_batch_fetch_padded': (ctypes.c_int, [_vp, _dp, ctypes.c_longlong, _dp, _ip]),
    'sonic_host_alloc': (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    'sonic_host_free': (ctypes.c_int, [_vp]),
    'sonic_batch_prepare': (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _llp, ctypes.c_longlong,
                                           _dp, ctypes.POINTER(SonicOpts), ctypes.POINTER(_vp)]),
    'sonic_batch_total_rows': (ctypes.c_longlong, [_vp]),
    'sonic_batch_row_offsets': (ctypes.c_int, [_vp, _llp]),
    'sonic_batch_launch': (ctypes.c_int, [_vp]),
    'sonic_batch_launch_to_host': (ctypes.c_int, [_vp, _dp]),
    'sonic_batch_row_blocks': (ctypes.c_int, [_vp, _llp, _llp]),
    'sonic_batch_n_chunks': (ctypes.c_int, [_vp]),
    'sonic_batch_tolerances': (ctypes.c_int, [_vp, _dp, _dp]),
    'sonic_batch_chunk_times': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    'sonic_release_device_memory': (ctypes.c_int, []),
    'sonic_batch_sync': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    'sonic_batch_fetch': (ctypes.c_int, [_vp, _dp, _dp, _ip]),
    'sonic_batch_device_ptrs': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                               ctypes.POINTER(_vp)]),
    'sonic_batch_destroy': (None, [_vp]),
    'sonic_batch_run': (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _llp, ctypes.c_longlong, _dp,
                                       ctypes.POINTER(SonicOpts), _dp, _dp, _ip]),
    'mech_default_opts': (None, [ctypes.POINTER(MechOpts)]),
    'mech_neuron_nrates': (ctypes.c_int, [ctypes.c_int]),
    'mech_batch_run': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp, _dp,
                                      ctypes.c_longlong, _dp, ctypes.c_int,
                                      ctypes.POINTER(MechOpts), _dp, _ip, _ip,
                                      ctypes.POINTER(ctypes.c_float)]),
    'mech_batch_run_overtones': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp,
                                                _dp, ctypes.c_longlong, _dp, ctypes.c_int, ctypes.c_int,
                                                _dp, _dp, ctypes.POINTER(MechOpts), _dp, _dp, _ip, _ip,
                                                ctypes.POINTER(ctypes.c_float)]),
    'full_default_opts': (None, [ctypes.POINTER(FullOpts)]),
    'full_count_rows': (ctypes.c_int, [_dp, ctypes.c_longlong, ctypes.c_double, _llp]),
    'full_batch_run': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp,
                                      ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _llp,
                                      ctypes.c_longlong, _dp, ctypes.POINTER(FullOpts), _dp, _ip,
                                      _ip, ctypes.POINTER(ctypes.c_float)]),
    'hybrid_batch_run': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp,
                                        ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _llp,
                                        ctypes.c_longlong, _dp, ctypes.POINTER(FullOpts), _dp, _ip,
                                        _ip, _ip, ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


def load():
    ''' Load libpysonic_amd.so and declare every prototype. Raises NativeLibraryError. '''
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(
            f'{LIB_PATH} not found: build it with `python -m pysonic_amd.build` '
            '(hipcc --offload-arch=gfx950). pysonic_amd has no CPU fallback.')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as err:
        raise NativeLibraryError(f'cannot load {LIB_PATH}: {err}') from err
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as err:
            raise NativeLibraryError(f'{LIB_PATH} does not export {name}') from err
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.sonic_abi_version() != ABI_VERSION:
        raise NativeLibraryError('ABI version mismatch between pysonic_amd and its native library')
    _lib = lib
    return lib


def last_error():
    return load().sonic_last_error().decode('utf-8', 'replace')


def check(rc):
    ''' Convert a negative return code into the reference's exception types. '''
    if rc == SONIC_OK:
        return
    msg = last_error()
    if rc in (SONIC_EINVAL, SONIC_ERANGE):
        raise ValueError(msg)      # reference: ValueError from isWithin / checkInputs
    raise NativeLibraryError(f'native library error {rc}: {msg}')


def require_gpu():
    lib = load()
    n = lib.sonic_device_count()
    if n <= 0:
        raise NativeLibraryError('no HIP device visible: pysonic_amd needs an AMD GPU (gfx950)')
    return n


def default_device():
    ''' GPU of this process when a model does not name one: LOCAL_RANK under a one-process-per-GPU
        launcher (torchrun, the reference's `--mpi` has no equivalent), wrapped to the visible devices;
        0 otherwise. '''
    try:
        r = int(os.environ.get('LOCAL_RANK', '0'))
    except ValueError:
        r = 0
    n = load().sonic_device_count()
    return r % n if n > 0 else 0


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ=_dp):
    return a.ctypes.data_as(typ)


def default_opts(**overrides):
    o = SonicOpts()
    load().sonic_default_opts(ctypes.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f'unknown solver option {k}')
        setattr(o, k, v)
    return o


class SonicModel:
    ''' Device-resident SONIC model: neuron parameters + 2-D (A, Q) lookup (sonic_model_t). '''

    def __init__(self, neuron, params, tables, A_grid, Q_grid, device=0):
        lib = load()
        require_gpu()
        self.neuron = neuron
        self.neuron_id = neuron_id(neuron)
        params = _f64(params)
        tables = _f64(tables)
        A_grid = _f64(A_grid)
        Q_grid = _f64(Q_grid)
        if tables.ndim != 3 or tables.shape[1:] != (A_grid.size, Q_grid.size):
            raise ValueError(f'tables shape {tables.shape} does not match (ntab, {A_grid.size}, '
                             f'{Q_grid.size})')
        self.nstates = lib.sonic_neuron_nstates(self.neuron_id)
        if self.nstates < 0:
            raise NotImplementedError(f'{neuron} neuron not available in the native library')
        self.ncol = self.nstates + 4
        h = _vp()
        check(lib.sonic_model_create(device, self.neuron_id, _ptr(params), params.size,
                                     _ptr(tables), _ptr(A_grid), A_grid.size, _ptr(Q_grid),
                                     Q_grid.size, tables.shape[0], ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, '_h', None):
            load().sonic_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, A, tstop, dt, ev_t, ev_x, ev_off, y0, opts=None):
        return SonicBatch(self, A, tstop, dt, ev_t, ev_x, ev_off, y0, opts)


class SonicBatch:
    ''' A prepared batch of configurations resident in HBM (sonic_batch_t). '''

    def __init__(self, model, A, tstop, dt, ev_t, ev_x, ev_off, y0, opts=None):
        lib = load()
        self.model = model
        A, tstop, dt = _f64(A), _f64(tstop), _f64(dt)
        ev_t, ev_x = _f64(ev_t), _f64(ev_x)
        ev_off = np.ascontiguousarray(ev_off, dtype=np.int64)
        y0 = _f64(y0)
        n = A.size
        if tstop.size != n or dt.size != n or ev_off.size != n + 1:
            raise ValueError('inconsistent batch array sizes')
        if y0.size != model.nstates + 1:
            raise ValueError("Initial conditions do not match system's dimensions")
        if opts is None:
            opts = default_opts()
        self.opts = opts
        self.n_cfg = n
        h = _vp()
        check(lib.sonic_batch_prepare(model._h, _ptr(A), _ptr(tstop), _ptr(dt), _ptr(ev_t),
                                      _ptr(ev_x), _ptr(ev_off, _llp), n, _ptr(y0),
                                      ctypes.byref(opts), ctypes.byref(h)))
        self._h = h
        self.total_rows = lib.sonic_batch_total_rows(h)
        # rows of configuration i: [row_start[i], row_start[i] + n_rows[i]) of the trace block (queue order unless
        # the batch is pipelined: opts.chunks > 1)
        self.row_start = np.empty(n, dtype=np.int64)
        self.n_rows = np.empty(n, dtype=np.int64)
        check(lib.sonic_batch_row_blocks(h, _ptr(self.row_start, _llp), _ptr(self.n_rows, _llp)))
        self.n_chunks = lib.sonic_batch_n_chunks(h)
        self._host = None
        # the tolerances in force (the kernel's own where the options left them at 0)
        rt, at = ctypes.c_double(), ctypes.c_double()
        check(lib.sonic_batch_tolerances(h, ctypes.byref(rt), ctypes.byref(at)))
        self.rtol, self.atol = rt.value, at.value

    @property
    def row_off(self):
        ''' [n_cfg + 1] row offsets in queue order (not defined for a pipelined batch) '''
        if self.n_chunks:
            raise ValueError('pipelined batch: rows are not in queue order, use row_start / n_rows')
        if getattr(self, '_row_off', None) is None:
            self._row_off = np.concatenate((self.row_start, [self.total_rows])).astype(np.int64)
        return self._row_off

    def rows_of(self, block, i):
        ''' view of configuration i's rows in a fetched trace block '''
        return block[self.row_start[i]:self.row_start[i] + self.n_rows[i]]

    def launch(self, to_host=False):
        ''' :param to_host: the rows are copied to a page-locked host block (self.host_traces) behind each kernel, on
                its stream -- with a pipelined batch while the other launches still integrate '''
        if to_host and self.opts.write_traces:
            self._host = host_block((self.total_rows, self.model.ncol))
            check(load().sonic_batch_launch_to_host(self._h, _ptr(self._host)))
        else:
            check(load().sonic_batch_launch(self._h))

    @property
    def host_traces(self):
        ''' trace block of the last launch(to_host=True), valid after sync() '''
        return self._host

    def chunk_times(self):
        ''' pipelined batch, after sync(): (kernel_ms, done_ms) per launch '''
        k = np.zeros(self.n_chunks, dtype=np.float32)
        d = np.zeros(self.n_chunks, dtype=np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        check(load().sonic_batch_chunk_times(self._h, k.ctypes.data_as(fp), d.ctypes.data_as(fp)))
        return k, d

    def sync(self):
        ''' :return: HIP-event duration of the last launch (ms) '''
        ms = ctypes.c_float()
        check(load().sonic_batch_sync(self._h, ctypes.byref(ms)))
        return ms.value

    def fetch(self, traces=True, nan_columns=0):
        ''' :param nan_columns: extra columns of NaN after the device's (the Z / ng columns the reference appends
                to the table of an effective simulation): the padded table is assembled on the device
            :return: traces (total_rows, ncol + nan_columns) or None, metrics (n_cfg, 12), status (n_cfg,)
            The traces land in page-locked host memory (host_block): one contiguous transfer at the speed of the
            link; the memory goes back to a small pool when the last view of it is dropped. '''
        tr = None
        metrics = np.empty((self.n_cfg, SONIC_NMETRICS))
        status = np.empty(self.n_cfg, dtype=np.int32)
        if traces and self.opts.write_traces:
            ncol = self.model.ncol
            tr = host_block((self.total_rows, ncol + nan_columns))
            check(load().sonic_batch_fetch_padded(self._h, _ptr(tr), ncol + nan_columns, _ptr(metrics),
                                                  _ptr(status, _ip)))
        else:
            check(load().sonic_batch_fetch(self._h, None, _ptr(metrics), _ptr(status, _ip)))
        return tr, metrics, status

    def device_ptrs(self):
        ''' (traces, metrics, status) device addresses (int or None) '''
        t, m, s = _vp(), _vp(), _vp()
        check(load().sonic_batch_device_ptrs(self._h, ctypes.byref(t), ctypes.byref(m),
                                             ctypes.byref(s)))
        return t.value, m.value, s.value

    def run(self, traces=True):
        self.launch()
        self.sync()
        return self.fetch(traces=traces)

    def close(self):
        if getattr(self, '_h', None):
            load().sonic_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- page-locked host blocks for the outputs -------------------------------------------------------------
# A fresh pageable buffer of 0.5 GB costs the transfer a staging copy and 1.6e5 page faults (87 ms for the traces
# of the 4096-cell map against 10 ms into memory that is already mapped). The blocks below are hipHostMalloc'ed,
# handed out as numpy arrays and returned to a pool when the last array / frame viewing them is collected, so a
# caller that sweeps repeatedly (activation maps, titrations with traces) pays the mapping once.
_POOL_MAX_BLOCKS = 2
_POOL_MAX_BYTES = 4 << 30
_PINNED_MAX_OUTSTANDING = 8 << 30      # page-locked memory is physical memory: beyond this, pageable buffers
_pool = []                       # [(capacity in bytes, address)]
_pool_lock = threading.RLock()   # sweeps may run from several host threads (tools/bench_configs.py 4)
_outstanding = [0]               # bytes of page-locked memory handed out and not yet collected


class _PinnedBlock:
    __slots__ = ('addr', 'capacity')

    def __init__(self, nbytes):
        self.addr = None
        with _pool_lock:
            if _outstanding[0] + nbytes > _PINNED_MAX_OUTSTANDING:
                raise MemoryError('page-locked budget exhausted')
            for i, (cap, addr) in enumerate(_pool):
                if nbytes <= cap <= max(2 * nbytes, 1 << 20):
                    del _pool[i]
                    self.addr, self.capacity = addr, cap
                    break
            else:
                out = _vp()
                check(load().sonic_host_alloc(nbytes, ctypes.byref(out)))
                self.addr, self.capacity = out.value, nbytes
            _outstanding[0] += self.capacity

    def __del__(self):
        try:
            if self.addr is None:
                return
            with _pool_lock:
                _outstanding[0] -= self.capacity
                if len(_pool) < _POOL_MAX_BLOCKS and sum(c for c, _ in _pool) + self.capacity <= _POOL_MAX_BYTES:
                    _pool.append((self.capacity, self.addr))
                else:
                    load().sonic_host_free(self.addr)
                self.addr = None
        except Exception:
            pass


def host_block(shape):
    ''' float64 array of `shape` in page-locked host memory (contents undefined); pageable memory if the
        allocation fails (a machine that refuses to lock that much) '''
    n = int(np.prod(shape))
    if n == 0:
        return np.empty(shape)
    try:
        blk = _PinnedBlock(n * 8)
    except (RuntimeError, ValueError, MemoryError, OSError):
        return np.empty(shape)
    carr = (ctypes.c_double * n).from_address(blk.addr)
    carr._owner = blk                         # the ctypes array is the numpy array's base: it keeps the block alive
    return np.ctypeslib.as_array(carr).reshape(shape)


def release_host_pool():
    ''' free the pooled blocks (those still viewed by arrays are freed when the arrays go) '''
    with _pool_lock:
        while _pool:
            _, addr = _pool.pop()
            load().sonic_host_free(addr)


def count_rows(tstop, dt, ev_t, ev_off):
    tstop, dt, ev_t = _f64(tstop), _f64(dt), _f64(ev_t)
    ev_off = np.ascontiguousarray(ev_off, dtype=np.int64)
    out = np.empty(tstop.size, dtype=np.int64)
    check(load().sonic_count_rows(_ptr(tstop), _ptr(dt), _ptr(ev_t), _ptr(ev_off, _llp),
                                  tstop.size, _ptr(out, _llp)))
    return out


def mech_default_opts(**overrides):
    o = MechOpts()
    load().mech_default_opts(ctypes.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f'unknown mech option {k}')
        setattr(o, k, v)
    return o


def mech_batch_run(neuron, bls_params, f, A, Q, fs, opts=None, device=0, overtones=None):
    ''' Batched computeEffVars. :return: effvars (n, n_fs, 1 + n_rates), ncycles, status, ms
        With overtones (n, n_ov, 2) -- amplitude, phase of the charge overtones of every cell --
        also returns, before ncycles, the (n, n_fs, n_ov, 2) amplitudes and phases of the
        overtones of the membrane potential. '''
    lib = load()
    require_gpu()
    nid = neuron_id(neuron)
    f, A, Q, fs = _f64(f), _f64(A), _f64(Q), _f64(np.atleast_1d(fs))
    bls_params = _f64(bls_params)
    n = f.size
    if A.size != n or Q.size != n:
        raise ValueError('inconsistent cell array sizes')
    nv = 1 + lib.mech_neuron_nrates(nid)
    eff = np.empty((n, fs.size, nv))
    ncyc = np.empty(n, dtype=np.int32)
    status = np.empty(n, dtype=np.int32)
    ms = ctypes.c_float()
    if opts is None:
        opts = mech_default_opts()
    if overtones is not None:
        ov = np.asarray(overtones, dtype=float)
        if ov.ndim != 3 or ov.shape[0] != n or ov.shape[2] != 2 or ov.shape[1] < 1:
            raise ValueError('overtones must have shape (n, n_overtones, 2)')
        ovA, ovphi = _f64(ov[:, :, 0]), _f64(ov[:, :, 1])
        ovout = np.empty((n, fs.size, ov.shape[1], 2))
        check(lib.mech_batch_run_overtones(
            device, nid, _ptr(bls_params), bls_params.size, _ptr(f), _ptr(A), _ptr(Q), n, _ptr(fs),
            fs.size, ov.shape[1], _ptr(ovA), _ptr(ovphi), ctypes.byref(opts), _ptr(eff), _ptr(ovout),
            _ptr(ncyc, _ip), _ptr(status, _ip), ctypes.byref(ms)))
        return eff, ovout, ncyc, status, ms.value
    check(lib.mech_batch_run(device, nid, _ptr(bls_params), bls_params.size, _ptr(f), _ptr(A),
                             _ptr(Q), n, _ptr(fs), fs.size, ctypes.byref(opts), _ptr(eff),
                             _ptr(ncyc, _ip), _ptr(status, _ip), ctypes.byref(ms)))
    return eff, ncyc, status, ms.value


def full_default_opts(**overrides):
    o = FullOpts()
    load().full_default_opts(ctypes.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f'unknown option {k}')
        setattr(o, k, v)
    return o


def full_batch_run(neuron, neuron_params, bls_params, f, A, fs, tstop, ev_t, ev_x, ev_off, y0,
                   opts=None, device=0):
    ''' Batched method='full' simulations.
        :return: traces (rows, n_states + 6), row_off (n + 1), status, nsteps, kernel_ms '''
    lib = load()
    require_gpu()
    nid = neuron_id(neuron)
    f, A, fs, tstop = _f64(f), _f64(A), _f64(fs), _f64(tstop)
    ev_t, ev_x, y0 = _f64(ev_t), _f64(ev_x), _f64(y0)
    ev_off = np.ascontiguousarray(ev_off, dtype=np.int64)
    neuron_params, bls_params = _f64(neuron_params), _f64(bls_params)
    n = f.size
    if opts is None:
        opts = full_default_opts()
    nrows = np.empty(n, dtype=np.int64)
    check(lib.full_count_rows(_ptr(tstop), n, opts.target_dt, _ptr(nrows, _llp)))
    row_off = np.concatenate(([0], np.cumsum(nrows)))
    ncol = y0.size + 5
    traces = np.empty((int(row_off[-1]), ncol))
    status = np.empty(n, dtype=np.int32)
    nsteps = np.empty(n, dtype=np.int32)
    ms = ctypes.c_float()
    check(lib.full_batch_run(device, nid, _ptr(neuron_params), neuron_params.size,
                             _ptr(bls_params), bls_params.size, _ptr(f), _ptr(A), _ptr(fs),
                             _ptr(tstop), _ptr(ev_t), _ptr(ev_x), _ptr(ev_off, _llp), n, _ptr(y0),
                             ctypes.byref(opts), _ptr(traces), _ptr(status, _ip),
                             _ptr(nsteps, _ip), ctypes.byref(ms)))
    return traces, row_off, status, nsteps, ms.value


def hybrid_batch_run(neuron, neuron_params, bls_params, f, A, fs, tstop, ev_t, ev_x, ev_off, y0,
                     opts=None, device=0):
    ''' Batched method='hybrid' simulations.
        :return: traces (rows, n_states + 6), row_off (n + 1), status, nsteps, ncycles, kernel_ms '''
    lib = load()
    require_gpu()
    nid = neuron_id(neuron)
    f, A, fs, tstop = _f64(f), _f64(A), _f64(fs), _f64(tstop)
    ev_t, ev_x, y0 = _f64(ev_t), _f64(ev_x), _f64(y0)
    ev_off = np.ascontiguousarray(ev_off, dtype=np.int64)
    neuron_params, bls_params = _f64(neuron_params), _f64(bls_params)
    n = f.size
    if opts is None:
        opts = full_default_opts()
    nrows = np.empty(n, dtype=np.int64)
    check(lib.full_count_rows(_ptr(tstop), n, opts.target_dt, _ptr(nrows, _llp)))
    row_off = np.concatenate(([0], np.cumsum(nrows)))
    ncol = y0.size + 5
    traces = np.empty((int(row_off[-1]), ncol))
    status = np.empty(n, dtype=np.int32)
    nsteps = np.empty(n, dtype=np.int32)
    ncycles = np.empty(n, dtype=np.int32)
    ms = ctypes.c_float()
    check(lib.hybrid_batch_run(device, nid, _ptr(neuron_params), neuron_params.size,
                               _ptr(bls_params), bls_params.size, _ptr(f), _ptr(A), _ptr(fs),
                               _ptr(tstop), _ptr(ev_t), _ptr(ev_x), _ptr(ev_off, _llp), n,
                               _ptr(y0), ctypes.byref(opts), _ptr(traces), _ptr(status, _ip),
                               _ptr(nsteps, _ip), _ptr(ncycles, _ip), ctypes.byref(ms)))
    return traces, row_off, status, nsteps, ncycles, ms.value
