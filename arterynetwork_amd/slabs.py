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


def one_gpu_step_ms(shape, world, planes, args):
    """ms/step of the committed one-GPU bench line of the volume this slab is a share of (profiles/r0x_bench_<nx>.json, newest
    round first): with one forced rank (--force-dist on a slab shape) the volume is planes * 8 deep by convention of the slab
    lines (880x880x80 stands for an eighth of 880x880x640); None when no such line is committed."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nz = shape[2] if world > 1 else {80: 640, 160: 640, 320: 640}.get(shape[2], shape[2])
    want = '{}x{}x{}'.format(shape[0], shape[1], nz)
    if getattr(args, 'storage16', False) or getattr(args, 'no_brain_mask', False) or getattr(args, 'levels', 255) != 255:
        return None
    # (the headline line of a round is profiles/r<NN>_bench_<nx>.json - the variants carry a suffix)
    for f in sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9][0-9]_bench_{}.json'.format(shape[0]))), reverse=True):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            if d.get('n_gpus') == 1 and d.get('valid') and want in d.get('metric', '') and 'single GPU' in d['config'].get('parallelism', '') \
                    and 'fp32' in d['config'].get('intensity_storage', '') and 'no excluded' not in d['config'].get('workload', ''):
                return d['ms_per_step']
        except Exception:
            continue
    return None


def bench_slabs(shape, args, dev, rank, world, roofline, configure, load_traffic):
    """bench.py body for N > 1 ranks (the twin of its one-GPU body: same options through `configure`): every rank generates
    the same synthetic volume in its HBM, recounts its own Z-slab; barrier + synchronize around exactly K sweeps; MAX
    over ranks; whole-job throughput.  `roofline`, `configure`, `load_traffic` are bench.py's."""
    import torch
    import torch.distributed as dist
    from . import phantoms
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask,
                                        integer_values=getattr(args, 'integer_values', False),   # (H already scaled by bench.main)
                                        tubes=getattr(args, 'tubes', 1), seed_mode=getattr(args, 'seed_mode', 'planes'))
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    s = make_slab_session(shape, rank, world, device=dev.index, reduce='rccl-always')
    configure(s, args)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    t0 = time.perf_counter()
    s.init(args.H)
    t_init = time.perf_counter() - t0
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
    st = s.stats()
    dense_bytes = (db0 + st['dense_bytes']) / 2.0 if args.skip_excluded else None
    dense_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    chain_beside = r.chain_kernel_ms / r.chain_launches if r.chain_launches else 0.0
    t = torch.tensor([dt, dense_ms, chain_beside], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, kern_ms, chain_max = float(t[0]), float(t[1]), float(t[2])
    per_rank = [None] * world
    valid = (r.sweeps == args.steps) and (r0.sweeps == args.warmup)
    tr = s.trace()
    nlev = s.nlevels()
    chain = s.chain_timing(args.H)               # (last: the session has to be re-initialised after it)
    z0, z1 = s.slab
    dist.all_gather_object(per_rank, {'rank': rank, 'slab': [z0, z1], 'dense_ms': round(dense_ms, 4),
                                      'band_chain_beside_dense_ms': round(chain_beside, 4), 'band_chain_ms': chain.get('band_chain_ms'),
                                      'seconds': round(dt, 4), 'dense_bytes': int(dense_bytes) if dense_bytes else None,
                                      'rccl_ranks': s.comm_ranks, 'reduction': s.reduce_mode})
    # what bounds the step of one rank, and the whole-job ratio it allows: every rank repeats the whole band chain (replicated
    # labels), so the ratio to one GPU cannot exceed (one-GPU step) / (band chain) however many ranks share the dense pass
    chain_floor = chain_max if chain_max else (chain.get('band_chain_ms') or 0.0)
    t1 = one_gpu_step_ms(shape, world, z1 - z0, args)
    scaling_floor = {'chain_ms': round(chain_floor, 4) if chain_floor else None, 'slab_recount_ms': round(kern_ms, 4),
                     'one_gpu_step_ms': t1, 'amdahl_max': round(t1 / chain_floor, 2) if (t1 and chain_floor) else None,
                     'note': 'chain_ms = band chain of a sweep beside the dense pass (every rank repeats it); slab_recount_ms = this rank\'s dense pass; '
                             'one_gpu_step_ms = the committed one-GPU line of the whole volume (profiles/), null when there is none; '
                             'amdahl_max = one_gpu_step_ms / chain_ms = the ratio to one GPU no number of ranks can exceed'}
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape),
        'value': round(V * r.sweeps / dt_max / 1e6, 1), 'unit': 'Mvoxel-iter/s', 'n_gpus': world,
        'steps': int(r.sweeps), 'warmup': args.warmup, 'ms_per_step': round(dt_max / max(1, r.sweeps) * 1e3, 4),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic', 'valid': bool(valid),
        'config': {'workload': '{} synthetic MRA tube volume ({} distinct intensities stored {}, {}), H={}, {} incremental VRG sweeps'.format(
                       args.shape, nlev, 'as u16 level indices' if args.storage16 else 'fp32',
                       'no excluded voxels' if args.no_brain_mask else 'brain-mask excluded voxels', args.H, r.sweeps),
                   'parallelism': 'zslab{} (dense recount sharded into {} Z-slabs; labels and the O(band) relabel replicated on every rank - no halo '
                                  'plane travels; one RCCL all-reduce of the slab statistics per 8 sweeps)'.format(world, world),
                   'reduction': s.reduce_mode, 'rccl_ranks': min(p['rccl_ranks'] for p in per_rank),
                   'intensity_storage': 'u16 level index (2 B/voxel)' if args.storage16 else 'fp32 (4 B/voxel)',
                   'init_seconds': round(t_init, 3),
                   'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
                   'dense_ms': round(kern_ms, 4), 'dense_events_every': args.events,
                   'band_chain_beside_dense_ms': round(chain_max, 4) if chain_max else None,
                   'band_chain_ms': chain.get('band_chain_ms'), 'band_chain_note': chain.get('band_chain_note'),
                   'dense_pass_loads': 'non-temporal' if st['dense_nt_loads'] else 'ordinary', 'dense_workgroups': st['dense_workgroups'],
                   'scaling_floor': scaling_floor,
                   'ranks': per_rank},
        'roofline': roofline(shape, z1 - z0, kern_ms, int(r.sweep_launches), load_traffic(shape, world, args.storage16, z1 - z0, dense_bytes),
                             args.storage16, dense_bytes, st['dense_kernel']),
    }
    s.close()
    return out
