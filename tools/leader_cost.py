#!/usr/bin/env python3
"""Development tool: what the leader's side of the change log costs per sweep - the band chain alone (no log), with a log that travels once
per batch (repl_stream=0), and streamed sweep by sweep in chunks of n sweeps (repl_chunk) - on one GPU, one-rank RCCL communicator.
usage: python3 tools/leader_cost.py [SHAPE] [sweeps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from arterynetwork_amd import phantoms, replica
from arterynetwork_amd._capi import Session
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '880x880x640').split('x'))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device('cuda', 0)
I, vm = phantoms.bench_volume_torch(shape, dev)
torch.cuda.synchronize()
def run(make, opts):
    s = make()
    for k, v in opts.items(): s.set_option(k, v)
    s.set_option('batch', 128); s.set_option('events', 0); s.set_option('chain_events', 0)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride())); s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    s.run(20, 10 ** 15, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = s.run(20 + K, 10 ** 15, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = s.repl_stats() if opts.get('_repl', True) else {}
    s.close()
    return round(dt / r.sweeps * 1e3, 4), st.get('chunks'), st.get('batches')
rows = []
def plain():
    s = Session(shape); s.set_option('dense_off', 1); return s
rows.append(('band chain alone (plain handle, dense_off)', run(plain, {'_repl': False}) if False else None))
leader = lambda: replica.make_replica_session(shape, 0, 1, transport='rccl', leader_verifies=False)
for name, o in (('leader, log once per batch (repl_stream=0)', {'repl_stream': 0}), ('leader, streamed, chunks of >= 8 sweeps', {'repl_chunk': 8}),
                ('leader, streamed, chunks of >= 32 sweeps', {'repl_chunk': 32}), ('leader, streamed, chunks of >= 1 sweep', {'repl_chunk': 1}),
                ('leader, log once per batch again', {'repl_stream': 0})):
    rows.append((name, run(leader, o)))
for name, v in rows:
    if v: print('%-50s %s ms/sweep  chunks %s batches %s' % (name, v[0], v[1], v[2]))
