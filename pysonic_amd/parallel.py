# -*- coding: utf-8 -*-
''' Multi-GPU sharding of a configuration queue: one process per GPU, static partition of the
    independent configurations (no collective during integration), and ONE all-gather of the
    per-configuration metric rows at the end (RCCL over xGMI when the backend is "nccl"; the same
    code runs on "gloo" for CPU tests).

    The reference's equivalent is the single-node `multiprocess` pool of Batch.run(mpi=True)
    (PySONIC/core/batches.py:86-153), which returns results re-ordered to queue order; the
    gather below preserves queue order the same way.
'''
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


def all_gather_rows(local_rows, n_total, dist=None, device=None):
    ''' Gather row blocks of unequal length from all ranks into queue order.

        :param local_rows: (n_local, k) float64 array of this rank's shard (shard_bounds order)
        :param n_total: total number of rows over all ranks
        :param dist: torch.distributed module with an initialised process group (None: 1 rank)
        :param device: torch device for the collective buffers (cuda:<local_rank> with nccl)
        :return: (n_total, k) numpy array, identical on every rank
    '''
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local_rows.shape[0] == n_total
        return local_rows
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    k = local_rows.shape[1]
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    nmax = max(b - a for a, b in sizes)
    # equal-sized padded blocks -> a single all_gather (latency-bound: a few hundred KB at most)
    pad = np.zeros((nmax, k))
    pad[:local_rows.shape[0]] = local_rows
    t_local = torch.from_numpy(pad)
    if device is not None:
        t_local = t_local.to(device)
    out = torch.empty((world * nmax, k), dtype=torch.float64, device=t_local.device)
    dist.all_gather_into_tensor(out, t_local)
    out = out.cpu().numpy().reshape(world, nmax, k)
    return np.concatenate([out[r, :b - a] for r, (a, b) in enumerate(sizes)], axis=0)
