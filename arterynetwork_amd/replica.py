"""Leader / follower multi-GPU driver: one process per GPU, torch.distributed for rendezvous, the change log over RCCL,
hipIpc or host callbacks (include/vrg.h "leader / follower replication").

Rank 0 - the leader - runs the band chain (decisions, update(), densities: variationalRegionGrowing.py:58-117) exactly as one GPU
does and logs what every sweep did to the labels.  Every other rank - a follower - holds the intensities and the labels, applies
the log and counts the sweeps assigned to it (the reference's dense recount, :113-116) over the whole volume, round robin.  All
ranks make the same calls with the same arguments and end with the same labels, `segmented` order, trace and result.
"""
from __future__ import annotations

import numpy as np

from ._capi import Session, VrgError


def leader_verifies_default(world):
    """Small groups let the leader count a share (its chain then runs beside a dense pass, as on one GPU: 0.037 ms per sweep
    against 0.17 / N for the share); from five ranks on the followers' shares are smaller than that and the leader only leads."""
    return world <= 4


def make_replica_session(shape, rank, world, device=0, lib=None, transport='rccl', group=None, leader_verifies=None, options=None):
    """Session of one rank of a leader / follower group with the log's transport wired up.  Collective: every rank of `group`
    calls it.  transport: 'rccl' (ncclBroadcast through the library's own communicator; falls back to 'callback' when it cannot
    be created), 'ipc' (followers map the leader's log buffers: ranks of one node), 'callback' (torch.distributed broadcast of
    host buffers: the CPU tests, any other fabric)."""
    import torch
    import torch.distributed as dist
    s = Session(shape, device=device, lib=lib)
    for k, v in (options or {}).items():
        s.set_option(k, v)
    lv = leader_verifies_default(world) if leader_verifies is None else bool(leader_verifies)
    s.repl_init(world, rank, lv)
    s.replica = {'rank': rank, 'world': world, 'leader_verifies': lv, 'transport': transport}
    if transport == 'none':                         # (the caller wires a transport of its own: bench_proxy's recorded log)
        return s
    if world == 1 and transport != 'rccl':
        transport = 'callback'
    if transport == 'rccl':
        ident = [s.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        ok = True
        try:
            s.comm_init(world, rank, ident[0])
            s.repl_use_rccl()
        except VrgError:
            ok = False
        if world > 1:
            flags = [None] * world                      # every rank must take the same path
            dist.all_gather_object(flags, ok, group=group)
            ok = all(flags)
        if not ok:
            transport = 'callback'
    if transport == 'ipc':
        blob = [s.repl_ipc_export() if rank == 0 else None]
        dist.broadcast_object_list(blob, src=0, group=group)
        if rank != 0:
            s.repl_ipc_import(blob[0])
    if transport == 'callback':
        on_gpu = world > 1 and dist.get_backend(group) == 'nccl'
        dev = torch.device('cuda', device) if on_gpu else torch.device('cpu')

        def bcast(buf, root):
            if world == 1:
                return
            a = np.frombuffer(buf, dtype=np.uint8)
            t = torch.from_numpy(a).to(dev) if on_gpu else torch.from_numpy(a)
            dist.broadcast(t, src=root, group=group)
            if on_gpu and rank != root:
                a[:] = t.cpu().numpy()

        def allsum(v):
            if world == 1:
                return v
            t = torch.tensor(v, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t.tolist()
        s.repl_set_callbacks(bcast, allsum)
    s.replica['transport'] = transport
    return s


# ---- bench.py bodies ---------------------------------------------------------------------------------------------------------
def _setup(s, I, vm, args, configure):
    configure(s, args)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))


def bench_replicas(shape, args, dev, rank, world, roofline, configure, load_traffic):
    """bench.py body for N > 1 ranks: one leader, N - 1 followers (the leader counts a share when N <= 4).  Every rank generates the
    same synthetic volume in its HBM; barrier + synchronize around exactly K sweeps of the collective vrg_run (which ends when
    every sweep has been applied and counted everywhere); MAX over ranks; whole-job throughput."""
    import time
    import torch
    import torch.distributed as dist
    from . import phantoms
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask, integer_values=getattr(args, 'integer_values', False),
                                        tubes=getattr(args, 'tubes', 1), seed_mode=getattr(args, 'seed_mode', 'planes'))
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    lv = leader_verifies_default(world) if args.leader_verifies < 0 else bool(args.leader_verifies)
    s = make_replica_session(shape, rank, world, device=dev.index, transport=args.transport, leader_verifies=lv)
    _setup(s, I, vm, args, configure)
    s.set_option('batch', args.repl_batch)
    s.set_option('chain_events', 0)
    if rank == 0 and not lv:
        s.set_option('events', 0)
    t0 = time.perf_counter()
    s.init(args.H)
    t_init = time.perf_counter() - t0
    big = 10 ** 15
    r0 = s.run(args.warmup, big, None)
    db0 = s.stats()['dense_bytes'] if rank == 0 else 0
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    r = s.run(args.warmup + args.steps, big, None)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    st, rs = s.stats(), s.repl_stats()
    dense_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    t = torch.tensor([dt, dense_ms], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else torch.device('cpu'))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, kern_ms = float(t[0]), float(t[1])
    tr = s.trace()
    nlev = s.nlevels()
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {'rank': rank, 'role': 'leader' if rank == 0 else 'follower', 'seconds': round(dt, 4), 'dense_ms': round(dense_ms, 4),
                                      'dense_launches': int(r.sweep_launches), 'sweeps_counted': rs['verified'] if rank else None, 'log_batches': rs['batches'],
                                      'log_records': rs['records'], 'transport': rs['transport']})
    valid = (r.sweeps == args.steps) and (r0.sweeps == args.warmup)
    dense_bytes = (db0 + st['dense_bytes']) / 2.0 if (args.skip_excluded and rank == 0) else None
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape),
        'value': round(V * r.sweeps / dt_max / 1e6, 1), 'unit': 'Mvoxel-iter/s', 'n_gpus': world,
        'steps': int(r.sweeps), 'warmup': args.warmup, 'ms_per_step': round(dt_max / max(1, r.sweeps) * 1e3, 4),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'valid': bool(valid),
        'config': {'workload': '{} synthetic MRA tube volume ({} distinct intensities stored fp32, {}), H={}, {} incremental VRG sweeps'.format(
                       args.shape, nlev, 'no excluded voxels' if args.no_brain_mask else 'brain-mask excluded voxels', args.H, r.sweeps),
                   'parallelism': 'replica{} (rank 0 leads: band chain + change log; {} verifier(s) apply the log and count the sweeps round robin over the WHOLE '
                                  'volume - the partition by role and time of DESIGN.md section 7, not north_star\'s Z-slabs + halo)'.format(world, rs['verifiers']),
                   'transport': rs['transport'], 'rccl_ranks': world if rs['transport'] == 'rccl' else 0, 'leader_verifies': bool(lv),
                   'log_batch_trips': args.repl_batch, 'init_seconds': round(t_init, 3),
                   'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]), 'flips_per_sweep_mean': round(float(tr['nflip'][args.warmup + 1:].mean()), 1),
                   'dense_ms': round(kern_ms, 4), 'dense_events_every': args.events,
                   'dense_pass_loads': 'non-temporal' if st['dense_nt_loads'] else 'ordinary', 'dense_workgroups': st['dense_workgroups'],
                   'ranks': per_rank},
        'roofline': roofline(shape, shape[2], kern_ms, int(max(p['dense_launches'] for p in per_rank)), load_traffic(shape, 1, args.storage16, None, dense_bytes),
                             args.storage16, dense_bytes, st['dense_kernel']),
    }
    s.close()
    return out


def bench_proxy(shape, args, dev, roofline, configure, load_traffic):
    """bench.py --force-dist: the two roles of an N-rank group measured one after the other on ONE GPU (N = --proxy-world, default 8) -
    what a step of each rank costs, and the whole-job ratio to one GPU that allows:
      one_gpu_step_ms        a plain handle, the same K sweeps (the ratio's numerator)
      leader_step_ms         rank 0 of a group whose leader only leads: band chain + change log + publish (RCCL, one-rank communicator)
      verifier_step_ms       a follower's work per sweep: every sweep's records applied, every (N-1)-th sweep counted over the whole volume -
                             a follower handle fed the leader's recorded log through the callback transport
    amdahl_max = one_gpu_step_ms / max(leader_step_ms, verifier_step_ms)."""
    import time
    import torch
    from . import phantoms
    from ._capi import Session
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask, integer_values=getattr(args, 'integer_values', False),
                                        tubes=getattr(args, 'tubes', 1), seed_mode=getattr(args, 'seed_mode', 'planes'))
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    N = max(2, args.proxy_world)
    big = 10 ** 15
    W, K = args.warmup, args.steps

    def timed(s):
        r0 = s.run(W, big, None)
        assert r0.sweeps == W, 'warm-up stopped early'
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = s.run(W + K, big, None)
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0
    # (1) one GPU
    s = Session(shape, device=dev.index)
    _setup(s, I, vm, args, configure)
    s.init(args.H)
    r1, dt1 = timed(s)
    db = s.stats()['dense_bytes']
    st1 = s.stats()
    ref_tr = s.trace()
    dense1 = r1.sweep_kernel_ms / max(1, r1.sweep_launches)
    s.close()
    # (2) the leader of a group whose leader only leads, over RCCL
    s = make_replica_session(shape, 0, 1, device=dev.index, transport='rccl', leader_verifies=False)
    _setup(s, I, vm, args, configure)
    s.set_option('batch', args.repl_batch)
    s.set_option('events', 0)                          # (nothing is timed inside the leader's run: no dense pass, and the chain's time is the run's)
    s.set_option('chain_events', 0)
    s.init(args.H)
    rl, dtl = timed(s)
    lst = s.repl_stats()
    leader_tr = s.trace()
    leader_transport = s.replica['transport']
    s.close()
    # ... and once more recording its log (callback transport), for the follower below
    log = []
    s = make_replica_session(shape, 0, 1, device=dev.index, transport='callback', leader_verifies=False)
    s.repl_set_callbacks(lambda buf, root: log.append(bytes(buf)), lambda v: v)
    _setup(s, I, vm, args, configure)
    s.set_option('batch', args.repl_batch)
    s.init(args.H)
    s.run(W, big, None)
    n_warm = len(log)
    s.run(W + K, big, None)
    s.close()
    # (3) a follower of an N-rank group fed that log: rank 1, N - 1 verifiers
    feed = iter(log)
    s = make_replica_session(shape, 1, N, device=dev.index, transport='none', leader_verifies=False)

    def replay(buf, root):
        b = next(feed)
        assert len(b) == len(buf)
        np.frombuffer(buf, dtype=np.uint8)[:] = np.frombuffer(b, dtype=np.uint8)

    def fake_allsum(v):                    # (the other ranks' contributions: they ended on the same sweep)
        v = list(v)
        v[-5] *= N
        return v
    s.repl_set_callbacks(replay, fake_allsum)
    _setup(s, I, vm, args, configure)
    s.init(args.H)
    rf, dtf = timed(s)
    fst = s.repl_stats()
    ftr = s.trace()
    densef = rf.sweep_kernel_ms / max(1, rf.sweep_launches)
    s.close()
    ok = all(np.array_equal(leader_tr[f], ref_tr[f]) and np.array_equal(ftr[f], ref_tr[f]) for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'))
    counted = ~np.isnan(ftr['sum_in'][W + 1:])
    ok = ok and bool(np.array_equal(ftr['sum_in'][W + 1:][counted], ref_tr['sum_in'][W + 1:][counted]))
    t1, tl, tf = dt1 / K * 1e3, dtl / max(1, rl.sweeps) * 1e3, dtf / max(1, rf.sweeps) * 1e3
    floor = {'one_gpu_step_ms': round(t1, 4), 'leader_step_ms': round(tl, 4), 'verifier_step_ms': round(tf, 4), 'ranks_modelled': N,
             'verifier_counts_every': N - 1, 'verifier_sweeps_counted': fst['verified'], 'verifier_dense_ms': round(densef, 4),
             'amdahl_max': round(t1 / max(tl, tf), 2),
             'note': 'one GPU, roles measured one after the other: leader = band chain + change log + publish over {} (it counts nothing); verifier = a follower '
                     'applying every sweep of the recorded log and counting every {}th over the whole volume (log fed through host callbacks); amdahl_max = '
                     'one_gpu_step_ms / max(leader, verifier) = the whole-job ratio to one GPU an {}-rank group can reach'.format(leader_transport, N - 1, N)}
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape), 'value': round(V * K / dtl / 1e6, 1), 'unit': 'Mvoxel-iter/s', 'n_gpus': 1,
        'steps': int(rl.sweeps), 'warmup': W, 'ms_per_step': round(tl, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic', 'valid': bool(ok and rl.sweeps == K and rf.sweeps == K and r1.sweeps == K),
        'config': {'workload': '{} synthetic MRA tube volume, H={}, {} incremental VRG sweeps - the LEADER role of an {}-rank leader / follower group alone on one GPU '
                               '(value = its rate; the group\'s roles: scaling_floor)'.format(args.shape, args.H, K, N),
                   'parallelism': 'replica proxy: roles of an {}-rank group on one GPU, one after the other'.format(N), 'transport': leader_transport, 'rccl_ranks': 1,
                   'log_batch_trips': args.repl_batch, 'log_records_per_sweep': round(lst['records'] / max(1, lst['sweeps']), 1),
                   'flips_per_sweep_mean': round(float(ref_tr['nflip'][W + 1:].mean()), 1), 'dense_ms': round(dense1, 4),
                   'scaling_floor': floor, 'proxy_parity': bool(ok)},
        'roofline': roofline(shape, shape[2], densef, int(rf.sweep_launches), load_traffic(shape, 1, args.storage16, None, db), args.storage16, db, st1['dense_kernel']),
    }
    return out
