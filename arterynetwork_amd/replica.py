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
    """Groups of up to three ranks let the leader count a share (a rank that counts every n-th sweep over the whole volume needs (0.178 + (n-1) x 0.029) / n
    ms per sweep at 880x880x640: 2 ranks 0.103, 3 ranks 0.079 - against one verifier 0.178 / two verifiers 0.089 behind a leader that only leads); from four
    ranks on the followers' shares are the smaller ones and the leader only leads (4 ranks: three verifiers at 0.062 against 0.066 with the leader counting)."""
    return world <= 3


def make_replica_session(shape, rank, world, device=0, lib=None, transport='rccl', group=None, leader_verifies=None, options=None, allow_fallback=False):
    """Session of one rank of a leader / follower group with the log's transport wired up.  Collective: every rank of `group`
    calls it.  transport: 'rccl' (ncclBroadcast through the library's own communicator), 'ipc' (followers map the leader's log
    buffers: ranks of one node), 'callback' (torch.distributed broadcast of host buffers: the CPU tests, any other fabric).
    A communicator that cannot be created on some rank raises VrgError on EVERY rank - unless allow_fallback is set, in which case
    the group takes the 'callback' transport instead (s.replica['transport'] says what carries the log)."""
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
        ok, why = True, ''
        try:
            s.comm_init(world, rank, ident[0])
            s.repl_use_rccl()
        except VrgError as e:
            ok, why = False, str(e)
        if world > 1:
            flags = [None] * world                      # every rank must take the same path
            dist.all_gather_object(flags, (ok, why), group=group)
            ok = all(f[0] for f in flags)
            why = '; '.join('rank {}: {}'.format(i, f[1]) for i, f in enumerate(flags) if not f[0])
        if not ok:
            if not allow_fallback:
                s.close()
                raise VrgError(-8, "replication: the RCCL transport could not be set up ({}); pass allow_fallback=True to let the group use host callbacks instead".format(why))
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
    """bench.py body for N > 1 ranks: one leader, N - 1 followers (the leader counts a share when N <= 3).  Every rank generates the
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
    try:
        s = make_replica_session(shape, rank, world, device=dev.index, transport=args.transport, leader_verifies=lv)
    except VrgError as e:                              # (raised on EVERY rank when the RCCL communicator cannot be set up on some rank: the ranks of one node then map the
        if args.transport != 'rccl':                   #  leader's log buffers instead - hipIpc, xGMI peer copies - and the line says so: config.transport)
            raise
        import sys
        if rank == 0:
            sys.stderr.write('bench.py: RCCL transport of the change log not available ({}); using hipIpc\n'.format(e))
        s = make_replica_session(shape, rank, world, device=dev.index, transport='ipc', leader_verifies=lv)
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


# what the projection assumes for the time between the leader publishing a sweep and a follower's kernels for it starting (a poll of the
# progress word + two small copies + a launch over hipIpc; three small broadcasts over RCCL): NOT measured - no multi-GPU node - and stated in the line
ASSUMED_LAG_MS = 0.1


def project_whole_run(steps, t1, tl, tv, dense_ms, lag_ms=ASSUMED_LAG_MS):
    """Whole-run time of a leader / follower group for `steps` sweeps from the roles' per-sweep times (ms): the leader emits a sweep every tl,
    a verifier needs tv per sweep on average (every sweep's records applied, its share counted); the last sweep's count starts `lag_ms`
    after the leader's last sweep and takes one dense pass.  Fill (the first sweep has to exist before anybody can follow) and drain (the
    last count) are IN the figure - they are what a short run pays."""
    group = max(steps * tl, tl + steps * tv) + lag_ms + dense_ms
    return {'steps': steps, 'one_gpu_ms': round(steps * t1, 3), 'group_ms': round(group, 3), 'ratio': round(steps * t1 / group, 2)}


def bench_proxy(shape, args, dev, roofline, configure, load_traffic):
    """bench.py --force-dist: the roles of leader / follower groups of 2, 4 and 8 ranks measured one after the other on ONE GPU, and the
    whole-run ratio to one GPU those times allow - a PROJECTION (no multi-GPU node has run this), kept under config.scaling_floor.  The
    line's own value / ms_per_step are the ONE-GPU run's (a whole-job throughput that was really measured).  Per group size N:
      leader_step_ms     rank 0's step: band chain + change log + sending it (N = 8: it only leads, log over RCCL with a one-rank
                         communicator; N <= 3: it also counts every N-th sweep, its chain beside that pass, log through host callbacks)
      verifier_step_ms   a follower's work per sweep: every sweep's records applied, every n-th sweep counted over the whole volume
                         (n = the group's verifiers) - a follower handle fed the leader's recorded log through the callback transport
      whole_run          steps x max(leader, verifier) + fill + drain for the line's own --steps, for 500 and for 20 (project_whole_run)."""
    import time
    import torch
    from . import phantoms
    from ._capi import Session
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask, integer_values=getattr(args, 'integer_values', False),
                                        tubes=getattr(args, 'tubes', 1), seed_mode=getattr(args, 'seed_mode', 'planes'))
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
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
    t1 = dt1 / K * 1e3
    ok_all = r1.sweeps == K
    fields = ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no')
    groups = {}
    worlds = sorted({2, 4, max(2, args.proxy_world)})
    rf_dense, rf_launches = dense1, int(r1.sweep_launches)
    for N in worlds:
        lv = leader_verifies_default(N)
        # (2) the leader: timed, its log recorded for the follower below
        log = []
        leader_transport = 'callback'
        if not lv:                                      # it only leads: the log over RCCL (one-rank communicator) - timed; then once more, recorded
            s = make_replica_session(shape, 0, 1, device=dev.index, transport='rccl', leader_verifies=False)
            _setup(s, I, vm, args, configure)
            s.set_option('batch', args.repl_batch); s.set_option('events', 0); s.set_option('chain_events', 0)
            s.init(args.H)
            rl, dtl = timed(s)
            lst = s.repl_stats()
            leader_tr = s.trace()
            leader_transport = s.replica['transport']
            s.close()
            s = make_replica_session(shape, 0, 1, device=dev.index, transport='none', leader_verifies=False)
            s.repl_set_callbacks(lambda buf, root: log.append(bytes(buf)), lambda v: v)
            _setup(s, I, vm, args, configure)
            s.set_option('batch', args.repl_batch)
            s.init(args.H)
            s.run(W, big, None)
            s.run(W + K, big, None)
            s.close()
        else:                                           # it counts every N-th sweep too: one run, timed and recorded (host callbacks)
            s = make_replica_session(shape, 0, N, device=dev.index, transport='none', leader_verifies=True)

            def fake_allsum_leader(v, N=N):             # (the other ranks' contributions: they ended on the same sweep)
                v = list(v)
                v[-5] *= N
                return v
            s.repl_set_callbacks(lambda buf, root: log.append(bytes(buf)), fake_allsum_leader)
            _setup(s, I, vm, args, configure)
            s.set_option('batch', args.repl_batch); s.set_option('chain_events', 0)
            s.init(args.H)
            rl, dtl = timed(s)
            lst = s.repl_stats()
            leader_tr = s.trace()
            s.close()
        # (3) a follower of the N-rank group fed that log: rank 1
        feed = iter(log)
        s = make_replica_session(shape, 1, N, device=dev.index, transport='none', leader_verifies=lv)

        def replay(buf, root):
            b = next(feed)
            assert len(b) == len(buf)
            np.frombuffer(buf, dtype=np.uint8)[:] = np.frombuffer(b, dtype=np.uint8)

        def fake_allsum(v, N=N):
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
        ok = all(np.array_equal(leader_tr[f], ref_tr[f]) and np.array_equal(ftr[f], ref_tr[f]) for f in fields)
        counted = ~np.isnan(ftr['sum_in'][W + 1:])
        ok = ok and bool(np.array_equal(ftr['sum_in'][W + 1:][counted], ref_tr['sum_in'][W + 1:][counted])) and rl.sweeps == K and rf.sweeps == K
        ok_all = ok_all and ok
        tl, tf = dtl / max(1, rl.sweeps) * 1e3, dtf / max(1, rf.sweeps) * 1e3
        nver = N if lv else N - 1
        groups[str(N)] = {'leader_verifies': bool(lv), 'leader_step_ms': round(tl, 4), 'leader_log_transport': leader_transport, 'verifier_step_ms': round(tf, 4),
                          'verifier_counts_every': nver, 'verifier_sweeps_counted': fst['verified'], 'verifier_dense_ms': round(densef, 4),
                          'log_records_per_sweep': round(lst['records'] / max(1, lst['sweeps']), 1), 'log_chunks_per_batch': round(lst['chunks'] / max(1, lst['batches']), 1),
                          'steady_state_ratio': round(t1 / max(tl, tf), 2), 'parity': bool(ok),
                          'whole_run': [project_whole_run(k, t1, tl, tf, densef) for k in sorted({K, 500, 20})]}
        if N == worlds[-1]:
            rf_dense, rf_launches = densef, int(rf.sweep_launches)
    best = groups[str(worlds[-1])]
    floor = {'one_gpu_step_ms': round(t1, 4), 'groups': groups, 'assumed_lag_ms': ASSUMED_LAG_MS,
             'where_n_gpus_buy_nothing': 'a group divides the dense pass only: a volume whose one-GPU step is already the band chain (512x512x170: ~0.034 vs a 0.027 chain) gains ~1.2x at '
                                         'any N; a sweep with thousands of flips (leader chain 0.3-0.5 ms > dense pass) gains ~1.0x (DESIGN.md section 7)',
             'note': 'PROJECTION from roles measured one after the other on ONE GPU - no multi-GPU node has run this.  whole_run = max(steps x leader, leader + steps x verifier) + '
                     'assumed_lag_ms + one dense pass (the last sweep\'s count): fill and drain included; xGMI / RCCL latency enters only through assumed_lag_ms'}
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape), 'value': round(V * K / dt1 / 1e6, 1), 'unit': 'Mvoxel-iter/s', 'n_gpus': 1,
        'steps': int(r1.sweeps), 'warmup': W, 'ms_per_step': round(t1, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic', 'valid': bool(ok_all), 'proxy': True,
        'config': {'workload': '{} synthetic MRA tube volume, H={}, {} incremental VRG sweeps on ONE GPU (value / ms_per_step: that run); beside it the roles of 2-, 4- and {}-rank '
                               'leader / follower groups measured one after the other on the same GPU: config.scaling_floor, a projection'.format(args.shape, args.H, K, worlds[-1]),
                   'parallelism': 'single GPU (+ replica role proxies under scaling_floor)', 'transport': best['leader_log_transport'], 'rccl_ranks': 1,
                   'log_batch_trips': args.repl_batch, 'flips_per_sweep_mean': round(float(ref_tr['nflip'][W + 1:].mean()), 1), 'dense_ms': round(dense1, 4),
                   'scaling_floor': floor, 'proxy_parity': bool(ok_all)},
        'roofline': roofline(shape, shape[2], dense1, int(r1.sweep_launches), load_traffic(shape, 1, args.storage16, None, db), args.storage16, db, st1['dense_kernel']),
    }
    return out
