"""Z-slab multi-GPU driver: one process per GPU, torch.distributed for rendezvous, RCCL for the data path.

What is sharded: the dense per-sweep recount (every voxel: 4 B intensity + 2 class bits), cut into
contiguous Z-slabs, one per rank.  What is replicated: the label volume and the O(band) relabel, which
is deterministic, so all ranks hold identical labels without exchanging halo planes.  The only
per-sweep exchange is a 32-byte all-reduce of the region statistics {n_in, n_out, sum_in, sum_out},
issued by the library on its own HIP stream through RCCL (vrg_comm_init) - or, for CPU tests and other
transports, through a host callback (`reduce='callback'`, here torch.distributed.all_reduce).
All ranks must make the same sequence of calls with the same arguments; use maxTime=None (a wall-clock cap
would let ranks stop at different sweeps).
"""
from __future__ import annotations

import time

import numpy as np

from ._capi import Session, VrgError


def partition(nz, world):
    """Contiguous Z-slabs [z0, z1) per rank, sizes differing by at most one plane."""
    base, rem = divmod(nz, world)
    out, z = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((z, z + n))
        z += n
    return out


def make_slab_session(shape, rank, world, device=0, lib=None, reduce='rccl', group=None, observer=None):
    """Session restricted to this rank's Z-slab with the cross-rank statistics reduction wired up.
    Collective: every rank of `group` must call it.  `observer(partial, total)` (callback reduction only) sees this
    rank's slab statistics {n_in, n_out, sum_in, sum_out} and their sum over the ranks, once per vrg_init and once per
    sweep in sweep order - tests use it to check the partition."""
    import torch
    import torch.distributed as dist
    if shape[2] < world:
        raise ValueError('fewer Z planes than ranks')
    s = Session(shape, device=device, lib=lib)
    z0, z1 = partition(shape[2], world)[rank]
    s.set_slab(z0, z1)
    s.slab = (z0, z1)
    s.reduce_mode = 'none'
    s.comm_ranks = 0
    if reduce == 'none':
        return s
    if world > 1 or reduce == 'rccl-always':
        ok = False
        if reduce in ('rccl', 'rccl-always'):
            ident = [s.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0, group=group)
            try:
                s.comm_init(world, rank, ident[0])
                s.comm_ranks = world
                ok = True
            except VrgError:
                ok = False
            flags = [None] * world                      # every rank must take the same path
            dist.all_gather_object(flags, ok, group=group)
            ok = all(flags)
            s.reduce_mode = 'rccl' if ok else 'callback'
        if not ok:
            on_gpu = dist.get_backend(group) == 'nccl'
            dev = torch.device('cuda', device) if on_gpu else torch.device('cpu')

            def allreduce(v):
                t = torch.tensor(v, dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                out = t.tolist()
                if observer is not None:
                    observer(list(v), out)
                return out
            s.set_reduce_callback(allreduce)
            if s.reduce_mode == 'none':
                s.reduce_mode = 'callback'
    s.slab = (z0, z1)
    return s


def bench_slabs(shape, args, dev, rank, world, roofline):
    """bench.py body for N > 1 ranks: every rank generates the same synthetic volume in its HBM, recounts
    its own Z-slab; barrier + synchronize around exactly K sweeps; MAX over ranks; whole-job throughput.
    `roofline` is bench.py's roofline(shape, planes, kernel_ms, launches, traffic, storage16, dense_bytes)."""
    import torch
    import torch.distributed as dist
    from . import phantoms
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels)
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    s = make_slab_session(shape, rank, world, device=dev.index, reduce='rccl-always')
    if args.sweep_blocks:
        s.set_option('sweep_blocks', args.sweep_blocks)
    s.set_option('events', 4)            # HIP events around every 4th dense launch
    s.set_option('batch', 64)
    s.set_option('nt_loads', getattr(args, 'nt_loads', -1))
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(args.H)
    big = 10 ** 15
    r0 = s.run(args.warmup, big, None)
    db0 = s.stats()['dense_bytes']
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    r = s.run(args.warmup + args.steps, big, None)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    dense_bytes = (db0 + s.stats()['dense_bytes']) / 2.0
    dense_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    t = torch.tensor([dt, dense_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, kern_ms = float(t[0]), float(t[1])
    per_rank = [None] * world
    valid = (r.sweeps == args.steps) and (r0.sweeps == args.warmup)
    tr = s.trace()
    chain = s.chain_timing(args.H)               # (last: the session has to be re-initialised after it)
    dist.all_gather_object(per_rank, {'rank': rank, 'slab': list(s.slab), 'dense_ms': round(dense_ms, 4),
                                      'band_chain_ms': chain.get('band_chain_ms'), 'seconds': round(dt, 4)})
    z0, z1 = s.slab
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape),
        'value': round(V * r.sweeps / dt_max / 1e6, 1), 'unit': 'Mvoxel-iter/s', 'n_gpus': world,
        'steps': int(r.sweeps), 'warmup': args.warmup, 'ms_per_step': round(dt_max / max(1, r.sweeps) * 1e3, 4),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic', 'valid': bool(valid),
        'config': {'workload': '{} synthetic MRA tube volume ({} intensity levels stored fp32, brain-mask excluded '
                               'voxels), H={}, {} incremental VRG sweeps'.format(args.shape, args.levels, args.H, r.sweeps),
                   'parallelism': 'zslab{} (dense recount sharded into {} Z-slabs, band relabel replicated, one '
                                  '32-byte RCCL all-reduce per sweep)'.format(world, world),
                   'reduction': s.reduce_mode, 'rccl_ranks': s.comm_ranks,
                   'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
                   'dense_ms': round(kern_ms, 4), 'dense_events_every': 4, 'band_chain_ms': chain.get('band_chain_ms'), 'ranks': per_rank},
        'roofline': roofline(shape, z1 - z0, kern_ms, int(r.sweep_launches), None, False, dense_bytes),
    }
    s.close()
    return out
