# -*- coding: utf-8 -*-
''' Multi-GPU sharding of a configuration queue: one process per GPU, static partition of the
    independent configurations (no collective during integration), and ONE all-gather of the
    per-configuration metric rows at the end (RCCL over xGMI when the backend is "nccl"; the same
    code runs on "gloo" for CPU tests).

    The reference's equivalent is the single-node `multiprocess` pool of Batch.run(mpi=True)
    (PySONIC/core/batches.py:86-153), which returns results re-ordered to queue order; the
    gather below preserves queue order the same way.

    A launcher (torchrun: one process per GPU) calls `init_process_group()` once, BEFORE the first
    GPU call of the process; everything else (Batch.run(mpi=True), run_sharded, the map classes)
    finds the group by itself. The buffers of the collective live where the backend needs them:
    on cuda:<LOCAL_RANK> for nccl (= RCCL), on the host for gloo.
'''
import os

import numpy as np


def shard_bounds(n, rank, world):
    ''' Contiguous, balanced [start, stop) block of `n` items for `rank` of `world`. '''
    if world < 1 or not (0 <= rank < world):
        raise ValueError('invalid rank / world size')
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_queue(queue, rank, world):
    start, stop = shard_bounds(len(queue), rank, world)
    return queue[start:stop]


def local_rank():
    try:
        return int(os.environ.get('LOCAL_RANK', '0'))
    except ValueError:
        return 0


def init_process_group(backend=None, timeout_s=1800):
    ''' Join the process group of a one-process-per-GPU launch (torchrun exports RANK, WORLD_SIZE,
        LOCAL_RANK, MASTER_ADDR, MASTER_PORT). Call it before anything touches the GPU: it selects
        cuda:<LOCAL_RANK> for this process, which is also the device the native library defaults to
        (_native.default_device).
        :param backend: 'nccl' (RCCL; default when this process sees a GPU), 'gloo' otherwise
        :return: torch.distributed, or None for a single process without a launcher (WORLD_SIZE unset
            or 1): the callers' single-process path needs no group. '''
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist
    if world <= 1 and backend is None:
        return None
    import datetime
    import torch
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', str(world))
    if backend is None:
        backend = 'nccl' if torch.cuda.device_count() > 0 else 'gloo'
    kwargs = {}
    if backend == 'nccl':
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise RuntimeError('nccl backend without a visible GPU')
        dev = torch.device('cuda', local_rank() % ndev)
        torch.cuda.set_device(dev)
        kwargs['device_id'] = dev
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    return dist


def collective_device(dist):
    ''' Where the buffers of a collective must live for the group's backend: cuda:<LOCAL_RANK> for nccl
        (RCCL moves device memory only), None = host memory for gloo. '''
    backend = str(dist.get_backend()).lower()
    if 'nccl' not in backend:
        return None
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError('nccl process group without a visible GPU')
    dev = torch.device('cuda', local_rank() % ndev)
    torch.cuda.set_device(dev)
    return dev


def _gather_blocks(local, bounds, dist, device=None):
    ''' all-gather of the ranks' row blocks (unequal lengths -> equal padded blocks, ONE
        all_gather_into_tensor: a few hundred KB, latency-bound) back into item order '''
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    k = local.shape[1]
    nmax = max(b - a for a, b in bounds)
    pad = np.zeros((nmax, k))
    pad[:local.shape[0]] = local
    if device is None:
        device = collective_device(dist)
    t_local = torch.from_numpy(pad)
    if device is not None:
        t_local = t_local.to(device)
    out = torch.empty((world * nmax, k), dtype=torch.float64, device=t_local.device)
    dist.all_gather_into_tensor(out, t_local)
    out = out.cpu().numpy().reshape(world, nmax, k)
    return np.concatenate([out[r, :b - a] for r, (a, b) in enumerate(bounds)], axis=0)


def all_gather_rows(local_rows, n_total, dist=None, device=None):
    ''' Gather row blocks of unequal length (shard_bounds order) from all ranks into queue order.

        :param local_rows: (n_local, k) float64 array of this rank's shard
        :param n_total: total number of rows over all ranks
        :param dist: torch.distributed module with an initialised process group (None: 1 rank)
        :param device: torch device of the collective buffers (default: what the backend needs)
        :return: (n_total, k) numpy array, identical on every rank
    '''
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local_rows.shape[0] == n_total
        return local_rows
    world = dist.get_world_size()
    return _gather_blocks(local_rows, [shard_bounds(n_total, r, world) for r in range(world)], dist, device)


def weighted_bounds(costs, world):
    ''' Contiguous blocks of a queue with (nearly) equal summed cost: list of `world` (start, stop).
        The integration kernels are latency-bound -- a launch lasts as long as its costliest
        configurations -- so ranks are balanced by estimated cost, not by count. '''
    costs = np.asarray(costs, dtype=float)
    n = costs.size
    if world < 1:
        raise ValueError('invalid world size')
    if n == 0:
        return [(0, 0)] * world
    csum = np.concatenate(([0.], np.cumsum(np.maximum(costs, 0.) + 1e-300)))
    # greedy: rank r takes items until it holds its share of what is LEFT (so that one costly item does
    # not starve the ranks after it), and at least one item while there are more items than ranks left
    cuts = [0]
    for r in range(world - 1):
        start = cuts[-1]
        target = csum[start] + (csum[-1] - csum[start]) / (world - r)
        stop = int(np.searchsorted(csum, target, side='left'))
        # the cut nearest to the target
        if stop > start + 1 and abs(csum[stop - 1] - target) <= abs(csum[stop] - target):
            stop -= 1
        stop = min(max(stop, start + 1), n - min(world - 1 - r, n - start - 1)) if start < n else n
        cuts.append(min(max(stop, start), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def dealt_shards(costs, world):
    ''' Split of a queue over `world` ranks that does not lean on the quality of the cost estimate: the items are
        DEALT in order of falling estimated cost, back and forth over the ranks (0 .. W-1, W-1 .. 0, ...), so every
        rank gets the same share of every cost class. Returns `world` ascending index arrays (queue order within a
        rank, as the reference's pool returns its results: batches.py:135-153).

        Measured on the 65 536-configuration (A x DC) sweep of bench.py with the kernel's step counts as the true
        cost: contiguous blocks of equal ESTIMATED cost (weighted_bounds) leave the costliest rank with 1.33 / 1.43 /
        1.49 x the mean at 2 / 4 / 8 ranks -- the estimate knows the amplitude and the duty cycle but not where the
        neuron starts to fire -- dealt shards 1.0001 / 1.002 / 1.004 x. '''
    costs = np.asarray(costs, dtype=float)
    n = costs.size
    if world < 1:
        raise ValueError('invalid world size')
    order = np.argsort(-costs, kind='stable')
    r = np.arange(n) % (2 * world)
    rank_of = np.where(r < world, r, 2 * world - 1 - r)
    return [np.sort(order[rank_of == k]) for k in range(world)]


def _gather_dealt(local, shards, n_items, dist, device=None):
    ''' all-gather of the ranks' row blocks of dealt shards (one padded all_gather_into_tensor) into item order '''
    import torch
    world = dist.get_world_size()
    k = local.shape[1]
    nmax = max(max(len(s) for s in shards), 1)
    pad = np.zeros((nmax, k))
    pad[:local.shape[0]] = local
    if device is None:
        device = collective_device(dist)
    t_local = torch.from_numpy(pad)
    if device is not None:
        t_local = t_local.to(device)
    out = torch.empty((world * nmax, k), dtype=torch.float64, device=t_local.device)
    dist.all_gather_into_tensor(out, t_local)
    out = out.cpu().numpy().reshape(world, nmax, k)
    rows = np.empty((n_items, k))
    for r_, idx in enumerate(shards):
        rows[idx] = out[r_, :len(idx)]
    return rows


def _group(dist):
    if dist is None:
        # a process group can only exist if the caller has imported torch: a single process that never did
        # is spared the import (a second of start-up on the first Batch.run of a session)
        import sys
        if 'torch' not in sys.modules:
            return None, 0, 1
        try:
            import torch.distributed as dist
        except ImportError:
            return None, 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(), dist.get_world_size()


def barrier(dist=None):
    ''' all ranks of the group (if any) meet here '''
    dist, _, world = _group(dist)
    if dist is not None and world > 1:
        dist.barrier()


def run_sharded(launch, n_items, costs=None, dist=None, device=None, force_collective=False, dealt=False):
    ''' Execute a sweep of `n_items` independent work items on all ranks of the process group (one
        process per GPU) and return the (n_items, k) result rows, in item order, on every rank.

        dealt=True: the items are dealt over the ranks by estimated cost (dealt_shards) instead of cut into
        contiguous blocks, and `launch(indices)` receives this rank's ascending index array -- the split to use
        when the cost estimate is rough (simulations: it is).

        :param launch: launch(start, stop) -> (stop - start, k) float64 rows of items [start, stop),
            computed on THIS rank's GPU (e.g. nbls.runSonicBatch(..., traces=False) metric rows,
            nbls.runMechBatch effective variables). Called once, with this rank's block.
        :param costs: optional per-item cost estimates (default: equal) for the split
        :param dist: torch.distributed (default: the initialised default group, if any)
        :param device: torch device of the collective buffers; default: cuda:<LOCAL_RANK> when the
            group's backend is nccl (= RCCL over xGMI), host memory for gloo
        :param force_collective: run the gather even in a group of one rank (test hook: the RCCL
            path on a one-GPU box)
        No collective runs during the integration; ONE all-gather of the rows ends the sweep
        (the rows are a few hundred KB: latency-bound). '''
    dist, rank, world = _group(dist)
    if dealt:
        shards = dealt_shards(np.ones(n_items) if costs is None else costs, world)
        local = np.ascontiguousarray(launch(shards[rank]), dtype=np.float64)
        if local.ndim == 1:
            local = local[:, None]
        if local.shape[0] != len(shards[rank]):
            raise ValueError(f'launch returned {local.shape[0]} rows for {len(shards[rank])} items')
        if dist is None or (world == 1 and not force_collective):
            return local
        return _gather_dealt(local, shards, n_items, dist, device)
    bounds = weighted_bounds(np.ones(n_items) if costs is None else costs, world)
    start, stop = bounds[rank]
    local = np.ascontiguousarray(launch(start, stop), dtype=np.float64)
    if local.ndim == 1:
        local = local[:, None]
    if local.shape[0] != stop - start:
        raise ValueError(f'launch returned {local.shape[0]} rows for items [{start}, {stop})')
    if dist is None or (world == 1 and not force_collective):
        return local
    return _gather_blocks(local, bounds, dist, device)


def run_sharded_objects(launch, n_items, costs=None, dist=None, gather=True, dealt=False):
    ''' Like run_sharded for results that are Python objects (thresholds, dicts of effective variables,
        file paths ...): every rank runs launch(start, stop) -> list of (stop - start) objects.
        gather=True: the lists are exchanged with all_gather_object and returned concatenated in item
        order on every rank -- for SMALL results. gather=False: no exchange; the returned list has one
        entry per item of the whole queue, `None` where another rank holds the result (what
        Batch.run(mpi=True) does for simulate(): the traces of a 4096-cell map are 0.5 GB of pickles;
        sweeps that need every rank's results ask for metric rows through run_sharded instead). '''
    dist, rank, world = _group(dist)
    if dealt:            # launch(indices), as run_sharded
        shards = dealt_shards(np.ones(n_items) if costs is None else costs, world)
        local = list(launch(shards[rank]))
        if len(local) != len(shards[rank]):
            raise ValueError(f'launch returned {len(local)} results for {len(shards[rank])} items')
        if world == 1:
            return local
        out = [None] * n_items
        if gather:
            collective_device(dist)
            gathered = [None] * world
            dist.all_gather_object(gathered, local)
            for idx, part in zip(shards, gathered):
                for i, x in zip(idx, part):
                    out[i] = x
        else:
            for i, x in zip(shards[rank], local):
                out[i] = x
        return out
    bounds = weighted_bounds(np.ones(n_items) if costs is None else costs, world)
    start, stop = bounds[rank]
    local = list(launch(start, stop))
    if len(local) != stop - start:
        raise ValueError(f'launch returned {len(local)} results for items [{start}, {stop})')
    if world == 1:
        return local
    if not gather:
        return [None] * start + local + [None] * (n_items - stop)
    collective_device(dist)          # nccl pickles through device memory: select this rank's GPU first
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    return [x for part in gathered for x in part]
