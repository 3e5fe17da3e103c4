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
