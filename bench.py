#!/usr/bin/env python3
"""bench.py - throughput of the VRG sweep on MI355X (BASELINE.json metric).

A "step" is one incremental update() sweep of the hot path (decide -> relabel stencil + region
statistics over every voxel -> band/density bookkeeping) over the synthetic volume.  Inputs are
generated on the GPU with torch (plumbing only) and are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape 880x880x640]

N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process is one
rank; started plainly with --gpus N it starts the N ranks itself (a child `python -m torch.distributed.run`,
before anything here touches the GPU) and passes their JSON line through.  The dense recount is cut into
Z-slabs (arterynetwork_amd/slabs.py).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_VOXEL_ITER = 6     # 4 B fp32 intensity + 1 B label read + 1 B label write
REFERENCE_ITSELF = {'value': 2.43, 'unit': 'Mvoxel-iter/s', 'cores': 1, 'config': '128x128x64 tube, 50 sweeps (BASELINE configs[0])',
                    'source': 'the reference imported in the build container (SURVEY.md §6; it cannot run the larger configs)'}


def device_source_sha():
    """Identifies the kernel sources a committed PMC traffic figure belongs to."""
    h = hashlib.sha256()
    for f in ('vrg_device.hip', 'vrg_items.h', 'vrg_types.h'):
        with open(os.path.join(ROOT, 'arterynetwork_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def side_line(shape, dev, args, configure, roofline, kind):
    """Two more workloads measured inside the default one-GPU run, so that the driver's line carries them itself:
    'nomask' - the headline volume WITHOUT its brain mask (nothing for the dense pass to skip: every voxel's intensity is fetched): the
    dense kernel's bytes counted on the device, its HIP-event time, the fraction of peak;
    'many_flip' - 128 disjoint tubes at 512x512x170 (SURVEY 8(d) config 5's "several disjoint tubes": ~13 000 flips per sweep, the regime
    a whole-mask seeding as refine() does creates): ms per sweep, and how the trips ran (fused / four-launch / host-driven)."""
    import torch
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    if kind == 'refine_like':
        return refine_like_line(dev, args, configure)
    if kind == 'nomask':
        shp, kw, W, K = shape, dict(brain_mask=False), 10, 100
    else:
        shp, kw, W, K = (512, 512, 170), dict(tubes=128), 10, 60
    I, vm = phantoms.bench_volume_torch(shp, dev, levels=args.levels, **kw)
    torch.cuda.synchronize()
    s = Session(shp, device=dev.index)
    configure(s, args)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(args.H)
    r0 = s.run(W, 10 ** 15, None)
    db0 = s.stats()['dense_bytes']
    st0 = s.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = s.run(W + K, 10 ** 15, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = s.stats()
    tr = s.trace()
    kern_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    out = {'workload': '{}x{}x{}, {}'.format(shp[0], shp[1], shp[2], 'no brain mask' if kind == 'nomask' else '128 disjoint tubes'), 'sweeps': int(r.sweeps),
           'valid': bool(r.sweeps == K and r0.sweeps == W), 'ms_per_step': round(dt / max(1, r.sweeps) * 1e3, 4), 'dense_ms': round(kern_ms, 4)}
    if kind == 'nomask':
        db = (db0 + st['dense_bytes']) / 2.0
        rf = roofline(shp, shp[2], kern_ms, int(r.sweep_launches), None, False, db, st['dense_kernel'])
        out.update({k: rf[k] for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'kernel_ms_avg', 'launches', 'bytes_per_launch', 'bytes_counted_on_device')})
        out['value'] = round(shp[0] * shp[1] * shp[2] * r.sweeps / dt / 1e6, 1)
    else:
        out.update({'flips_per_sweep_mean': round(float(tr['nflip'][W + 1:].mean()), 1), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
                    'host_driven_trips': st['host_driven_trips'] - st0['host_driven_trips'], 'fused_trips': st['fused_trips'] - st0['fused_trips'],
                    'value': round(shp[0] * shp[1] * shp[2] * r.sweeps / dt / 1e6, 1), 'unit': 'Mvoxel-iter/s'})
    s.close()
    del I, vm
    torch.cuda.empty_cache()
    return out


def refine_like_line(dev, args, configure, shp=(512, 512, 170), tubes=16, iter_max=400):
    """What the pipeline uses this stage for (README.md:69-71, :209 of the reference: VRG smooths an existing vessel mask): the seed is a
    perturbed mask of ALL vessels at once (phantoms.bench_volume_torch seed_mode='noisy-mask': half the surface eroded + 2 % salt), run
    to CONVERGENCE (the reference's first stop test, :91).  Reported: flips of the first ten sweeps, sweeps and seconds to convergence
    (vrg_init + the sweeps; inputs resident in HBM), how the trips ran.  Not the headline metric - a time-to-solution figure."""
    import torch
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    I, vm = phantoms.bench_volume_torch(shp, dev, levels=args.levels, tubes=tubes, seed_mode='noisy-mask')
    nseed = int((vm == 0).sum().item())
    torch.cuda.synchronize()
    s = Session(shp, device=dev.index)
    configure(s, args)
    s.set_option('events', 0); s.set_option('chain_events', 0)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.init(args.H)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    st0 = s.stats()
    t0 = time.perf_counter()
    r = s.run(iter_max, 10 ** 15, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st, tr = s.stats(), s.trace()
    nfl = tr['nflip'][1:]
    sweeps = int(r.sweeps)
    fused = st['fused_trips'] - st0['fused_trips']
    host = st['host_driven_trips'] - st0['host_driven_trips']
    out = {'workload': '{}x{}x{}, {} tubes, seed = perturbed mask of all of them (half the surface eroded + 2 % salt): {} seed voxels'.format(shp[0], shp[1], shp[2], tubes, nseed),
           'stop_reason': {1: 'converged', 2: 'time', 3: 'size', 4: 'itermax'}.get(int(r.stop_reason), str(int(r.stop_reason))), 'converged': bool(r.stop_reason == 1),
           'sweeps_to_convergence': sweeps, 'seconds_to_convergence': round(t_init + dt, 4), 'init_seconds': round(t_init, 4), 'sweeps_seconds': round(dt, 4),
           'ms_per_sweep_mean': round(dt / max(1, sweeps) * 1e3, 4), 'flips_first_10_sweeps': [int(v) for v in nfl[:10]], 'flips_total': int(nfl.sum()),
           'nseg_start': int(tr['nseg'][0]), 'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
           'trips': {'fused': int(fused), 'host_driven': int(host), 'four_launch': int(max(0, sweeps - fused - host)), 'handed_back': int(st['bail_flips'] + st['bail_fuse'] - st0['bail_flips'] - st0['bail_fuse'])},
           'ties': int(r.ties), 'valid': bool(r.stop_reason == 1)}
    s.close()
    del I, vm
    torch.cuda.empty_cache()
    return out


def load_traffic(shape, n_gpus, storage16, planes=None, design_bytes=None):
    """HBM bytes per dense launch from the committed rocprofv3 PMC passes (profiles/traffic.json) - only when an entry was
    measured on these very kernel sources (src_sha) and workload (shape, ranks, storage, slab planes - and, `design_bytes`
    given, a pass that has to fetch the same bytes within 2 %: a volume without its brain mask, say, is another workload);
    None otherwise (never a stale or foreign figure)."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        for e in t.get('entries', [t]):
            if (e.get('shape') == list(shape) and e.get('n_gpus', 1) == n_gpus and bool(e.get('storage16', False)) == bool(storage16)
                    and e.get('planes', shape[2]) == (planes if planes is not None else shape[2]) and e.get('src_sha') == device_source_sha()):
                ref = e.get('design_bytes_per_launch_counted_on_device')
                if design_bytes and ref and abs(float(design_bytes) / float(ref) - 1.0) > 0.02:
                    continue
                return e.get('hbm_bytes_per_launch')
    except Exception:
        pass
    return None


def configure(s, args):
    """The option set of a bench session, the same for one GPU and for a rank of N."""
    s.set_option('sweep_variant', args.variant)
    if args.sweep_blocks:
        s.set_option('sweep_blocks', args.sweep_blocks)
    if args.prio_mode >= 0:
        s.set_option('prio_mode', args.prio_mode)
    if args.storage16:
        s.set_option('storage16', 1)
    s.set_option('events', args.events)
    s.set_option('chain_events', args.events)
    s.set_option('batch', 64)
    if args.serial:
        s.set_option('serial_streams', 1)
    s.set_option('skip_excluded', args.skip_excluded)
    s.set_option('nt_loads', args.nt_loads)
    s.set_option('dense_pipe', getattr(args, 'dense_pipe', 1))


def padded_slab_voxels(shape, planes):
    """Voxels the dense pass addresses for `planes` Z planes: padded rows (PX = roundup(nx+2,16)) x (ny+4) rows."""
    nx, ny, _ = shape
    return ((nx + 2 + 15) // 16 * 16) * (ny + 4) * planes


def roofline(shape, planes, kern_ms, launches, traffic, storage16=False, dense_bytes=None, kernel=None):
    """Dominant kernel = k_recount_bits (dense region recount, one launch per sweep), HBM-bound.
    `achieved` = the bytes the kernel has to fetch by its own design divided by the HIP-event time of the launch;
    `frac` = achieved / 8 TB/s.  The design bytes are `dense_bytes` when given: counted on the device from the class
    bits (vrg_get_stats out[8]; mean of the count before and after the timed sweeps) - 2 class bits per voxel of the
    padded slab + every 128-byte intensity line that holds an included voxel; runs of excluded voxels (the brain mask)
    are not fetched.  Without it (option skip_excluded = 0) every voxel is streamed: 4 B fp32 intensity (2 B level
    index with 16-bit storage) + 2 class bits per padded voxel.  The label bytes are NOT streamed (labels are updated
    in place at the ~10^3 marked voxels).  Two figures are kept beside it and are never `frac`:
    `streamed_equiv_gbs` (what a kernel that streams every voxel of the slab would have to reach for the same time)
    and `algorithmic_equiv_gbs` (SURVEY.md section 8(d)'s 6 B/voxel-iteration accounting)."""
    bpv = 2.25 if storage16 else 4.25
    streamed = bpv * padded_slab_voxels(shape, planes)
    design = float(dense_bytes) if dense_bytes else streamed
    V = shape[0] * shape[1] * planes
    achieved = design / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
    out = {'bound': 'hbm', 'kernel': kernel or 'k_recount_bits<3,true,{},{}>'.format(1 if storage16 else 0, 'true' if dense_bytes else 'false'),
           'achieved': round(achieved, 1) if achieved else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
           'kernel_ms_avg': round(kern_ms, 4), 'launches': launches,
           'bytes_per_launch': int(design), 'bytes_per_voxel': round(design / padded_slab_voxels(shape, planes), 4),
           'bytes_counted_on_device': bool(dense_bytes),
           'streamed_equiv_gbs': round(streamed / (kern_ms * 1e-3) / 1e9, 1) if kern_ms > 0 else None,
           'algorithmic_equiv_gbs': round((4 if storage16 else BYTES_PER_VOXEL_ITER) * V / (kern_ms * 1e-3) / 1e9, 1) if kern_ms > 0 else None,
           # SURVEY.md section 8(d)'s accounting as the driver would compute it: 6 B x voxels / kernel time / peak.  NOT a
           # fraction of anything (it exceeds 1): the kernel does not move those bytes - no label write-back, 2 class bits
           # instead of a label byte, excluded runs not fetched
           'section8d_frac': round((4 if storage16 else BYTES_PER_VOXEL_ITER) * V / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern_ms > 0 else None,
           'section8d_note': 'algorithmic 6 B/voxel-iteration of SURVEY 8(d) / kernel time / 8 TB/s; not a fraction (the kernel fetches bytes_per_launch, not 6 B per voxel)',
           'traffic': traffic}
    if traffic and kern_ms > 0:
        out['traffic_gbs'] = round(traffic / (kern_ms * 1e-3) / 1e9, 1)
        out['traffic_frac_of_peak'] = round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def host_memory_available_gb():
    """MemAvailable of /proc/meminfo (no third-party module)."""
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                return int(line.split()[1]) / 2 ** 20
    except Exception:
        pass
    return 0.0


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(I_t, vm_t, H, levels_note, min_free_gb=48.0, crop_vox=70e6):
    """The oracle (oracle/vrg_oracle.c, a C port of the reference; level-histogram mode, OpenMP build of the same file)
    timed on this host on the WHOLE workload volume (880x880x640: about 20 GB of host memory; when the host has less than
    `min_free_gb` available, on a crop around the seeds - the line then says so).  One thread first, then candidate thread
    counts probed for >= 10 sweeps and >= 1 s each (the dense loops are memory-bound and a container may own fewer cores
    than it sees), then the best one again for >= 20 sweeps: `value` is THAT run's own rate, so it can never sit far below
    `threads_tried`.  The sweeps the oracle made are then repeated by the HIP path on the same volume and compared
    (labels, segmented order, integer trace): `parity`.  Outside the timed region."""
    from oracle import vrg_oracle as O
    from arterynetwork_amd._capi import Session
    nx, ny, nz = I_t.shape
    free_gb = host_memory_available_gb()
    whole = nx * ny * nz <= crop_vox or free_gb >= min_free_gb
    if whole:
        cx, cyy, czz, y0, z0 = nx, ny, nz, 0, 0
        what = 'the whole {}x{}x{} volume'.format(nx, ny, nz)
    else:
        cx, cyy, czz = min(nx, 448), min(ny, 448), min(nz, 320)
        y0 = max(0, min(ny - cyy, ny // 2 - cyy // 2))
        z0 = max(0, min(nz - czz, int(nz / 2.0 + 0.18 * nz) - czz // 2))
        what = 'the {}x{}x{} crop around the seeds of the same volume (only {:.0f} GB of host memory available)'.format(cx, cyy, czz, free_gb)
    t_prep = time.perf_counter()
    Ic32 = I_t[:cx, y0:y0 + cyy, z0:z0 + czz].contiguous().cpu().numpy()
    vm = vm_t[:cx, y0:y0 + cyy, z0:z0 + czz].contiguous().cpu().numpy()
    if not (vm == 0).any():
        return None
    o = O.Oracle(Ic32.astype(np.float64), vm, H, density_mode=1, omp=True)
    try:
        navail = len(os.sched_getaffinity(0))
    except Exception:
        navail = os.cpu_count() or 1
    o.set_threads(min(navail, 64))
    o.init()
    t_prep = time.perf_counter() - t_prep
    sweeps = 0

    def timed(threads, min_sweeps, min_s, cap_s, max_sweeps):
        """>= min_sweeps sweeps and >= min_s seconds (one discarded sweep first), at most cap_s seconds and max_sweeps sweeps
        (on a small volume a second is hundreds of sweeps: the probes together must not use up the run - the region stops
        growing at some point, and a sampled rate has to be the rate of sweeps that still do work)."""
        nonlocal sweeps
        got = o.set_threads(threads)
        if o.step(10 ** 6, 10 ** 12, -1.0) != 0:
            return got, 0, 0.0
        sweeps += 1
        t0 = time.perf_counter()
        n = 0
        while True:
            el = time.perf_counter() - t0
            if (n >= min_sweeps and el >= min_s) or el >= cap_s or n >= max_sweeps:
                break
            if o.step(10 ** 6, 10 ** 12, -1.0) != 0:
                break
            n += 1
        sweeps += n
        return got, n, time.perf_counter() - t0
    t1 = timed(1, 3, 1.0, 6.0, 10)
    cands = sorted({t for t in (8, 16, 32, 64, 128, navail) if t <= navail})
    probe = {t: timed(t, 10, 1.0, 5.0, 40) for t in cands}
    rate = {t: (n / dt if n and dt > 0 else 0.0) for t, (got, n, dt) in probe.items()}
    best = max(rate, key=rate.get) if rate else 1
    tn = timed(best, 20, 2.0, 10.0, 80)
    if sweeps == 0 or tn[1] == 0:
        o.close()
        return None
    # the same sweeps by the HIP path on the same volume
    s = Session(Ic32.shape)
    s.set_volume(Ic32)
    s.set_labels(vm)
    s.init(H)
    r = s.run(sweeps, 10 ** 15, None)
    tr, otr = s.trace(), o.trace()
    lab = np.empty(Ic32.shape, np.uint8)
    s.labels(out=lab)
    parity = bool(r.sweeps == sweeps and r.ties == 0 and np.array_equal(lab, o.labels())
                  and np.array_equal((lambda c: (c[:, 0] * Ic32.shape[1] + c[:, 1]) * Ic32.shape[2] + c[:, 2])(s.segmented()), o.segmented_lex())
                  and all(np.array_equal(tr[f], otr[f]) for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no')))
    s.close()
    o.close()
    (g1, n1, d1), (gn, nn, dn) = t1, tn
    V = Ic32.size
    value = V * nn / dn / 1e6
    best_probe = V * rate[best] / 1e6
    return {'value': round(value, 1), 'unit': 'Mvoxel-iter/s', 'cores': gn, 'kind': 'port',
            'single_core': {'value': round(V * n1 / d1 / 1e6, 1) if n1 else None, 'cores': 1, 'sweeps': n1, 'seconds': round(d1, 1)},
            'cpu': cpu_model(), 'cpus_visible': navail,
            'threads_tried': {str(t): {'value': round(V * rate[t] / 1e6, 1), 'sweeps': probe[t][1], 'seconds': round(probe[t][2], 2)} for t in cands},
            'probe_to_final': round(value / best_probe, 3) if best_probe else None,
            'whole_volume': bool(whole), 'prepare_seconds': round(t_prep, 1),
            'parity': parity, 'parity_sweeps': sweeps,
            'sample': '{} sweeps with 1 thread ({:.1f} s), 10- to 40-sweep probes of {} threads, then {} sweeps with {} threads ({:.1f} s) of '
                      'the oracle (C port of the reference, level-histogram mode, OpenMP build) on {} ({}); the HIP path repeated the {} '
                      'sweeps on the same volume and was compared with it'.format(n1, d1, '/'.join(str(t) for t in cands), nn, gn, dn, what, levels_note, sweeps),
            'reference_itself': REFERENCE_ITSELF}


# Test aid (tests/test_gpu_parity.py): VRG_BENCH_SHARE_GPU=1 puts every rank of --gpus N on GPU 0 - N processes on the one GPU of a test
# box, so that the N-rank body of this file runs there; RCCL refuses two ranks on one device, so the change log travels over hipIpc or
# host callbacks and torch.distributed runs on gloo.  Never a measurement.
SHARE_GPU = os.environ.get('VRG_BENCH_SHARE_GPU') == '1'


def visible_gpus():
    """Devices this process could use (torch.cuda.device_count() does not initialise the GPU on this image)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def preflight(n, rank=None, local_rank=None):
    """--gpus N needs N visible devices: say so and leave with exit code 2 BEFORE anything is spawned or initialised
    (a rank that finds no device of its own would otherwise fail somewhere inside the rendezvous)."""
    have = visible_gpus()
    if have < n or (local_rank is not None and local_rank >= have):
        who = '' if rank is None else '[rank {}] '.format(rank)
        sys.stderr.write('{}bench.py: --gpus {} needs {} visible GPUs, this node shows {} (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); '
                         'nothing was run\n'.format(who, n, n, have))
        raise SystemExit(2)


def spawn_ranks(args_list, n):
    """--gpus N without a launcher: start the N ranks as a child torch.distributed.run (this process has not touched the
    GPU), pass the child's output through, exit with its code.  Every rank's stderr is also kept in a log directory; when
    the launch fails the tail of each is printed with its rank, so a failure of one rank cannot hide behind the
    launcher's summary."""
    import glob
    import socket
    import tempfile
    if not SHARE_GPU:
        preflight(n)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix='bench_ranks_')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), '--log-dir', logdir, '--tee', '2', os.path.abspath(__file__)] + args_list
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write('bench.py: the {}-rank launch failed with exit code {}; stderr of every rank (last 25 lines each):\n'.format(n, rc))
        for f in sorted(glob.glob(os.path.join(logdir, '**', 'stderr.log'), recursive=True)):
            try:
                tail = open(f, errors='replace').read().strip().splitlines()[-25:]
            except OSError:
                continue
            sys.stderr.write('---- {}\n{}\n'.format(os.path.relpath(f, logdir), '\n'.join(tail)))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shape', default='880x880x640')
    ap.add_argument('--levels', type=int, default=255, help='intensity levels of the synthetic volume; 0 = continuous float32 noise')
    ap.add_argument('--H', type=float, default=2.25)
    ap.add_argument('--integer-values', action='store_true', help='the same volume with integer intensities 0..levels (as a scanner delivers them) and H / levels^2: the same run, '
                    'with the level of a voxel looked up directly instead of searched for')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side-lines', action='store_true', help='skip the two extra workloads of the default line (roofline_nomask, many_flip)')
    ap.add_argument('--variant', type=int, default=0)
    ap.add_argument('--sweep-blocks', type=int, default=0)
    ap.add_argument('--prio-mode', type=int, default=-1)
    ap.add_argument('--events', type=int, default=4, help='HIP events around every n-th dense launch (they cost the dense stream '
                    'a few us each); 0: none (no roofline then)')
    ap.add_argument('--skip-excluded', type=int, default=1, help='0: the dense pass fetches the intensities of excluded voxels too')
    ap.add_argument('--no-brain-mask', action='store_true', help='experiment: no excluded voxels (the dense pass then has nothing to skip)')
    ap.add_argument('--nt-loads', type=int, default=-1, help='dense pass: -1 = non-temporal loads when the pass exceeds the Infinity Cache (default), 0 / 1 = never / always')
    ap.add_argument('--dense-pipe', type=int, default=1, help='0: the dense pass as k_recount_bits instead of the two-trips-deep k_recount_pipe (A/B)')
    ap.add_argument('--serial', type=int, default=0, help='1: option serial_streams (needed under rocprofv3 --pmc, which runs one kernel at a time)')
    ap.add_argument('--storage16', action='store_true', help='16-bit intensity storage (level indices): config 5 style; 4 B/voxel-iter algorithmic')
    ap.add_argument('--tubes', type=int, default=1, help='disjoint tubes of the synthetic volume (SURVEY 8(d) config 5: "several disjoint tubes"): about 100 flips per tube and sweep')
    ap.add_argument('--seed-mode', default='planes', choices=['planes', 'whole', 'noisy-mask'], help="'whole': every tube voxel is a seed; 'noisy-mask': a perturbed mask of all "
                    "tubes (half the surface eroded + 2 %% salt) - what refine() hands this stage")
    ap.add_argument('--refine-like', action='store_true', help='only the refine()-shaped workload: perturbed mask of 16 tubes at 512x512x170 run to convergence; prints its record')
    ap.add_argument('--force-dist', action='store_true', help='one GPU: measure the roles of an N-rank group one after the other (replica partition), or run the '
                    'N>1 code path with a one-rank communicator (zslab partition)')
    ap.add_argument('--partition', default='replica', choices=['replica', 'zslab'], help="N > 1: 'replica' = one leader (band chain + change log), the other ranks apply the "
                    "log and count the sweeps round robin over the whole volume (DESIGN.md section 7); 'zslab' = every rank repeats the band chain, the dense pass is cut into Z-slabs")
    ap.add_argument('--transport', default='rccl', choices=['rccl', 'ipc', 'callback'], help='replica partition: how the change log travels')
    ap.add_argument('--leader-verifies', type=int, default=-1, help='replica partition: 1 / 0 = the leader counts a share of the sweeps / only leads; -1 = by the number of ranks (<= 3: it counts)')
    ap.add_argument('--repl-batch', type=int, default=128, help='replica partition: trips per batch of the change log')
    ap.add_argument('--proxy-world', type=int, default=8, help='--force-dist with the replica partition: ranks of the group whose roles are measured')
    args = ap.parse_args()
    shape = tuple(int(s) for s in args.shape.lower().split('x'))
    assert len(shape) == 3

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(sys.argv[1:], args.gpus))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus {} but the launcher started {} ranks'.format(args.gpus, world))
    if world > 1 and not SHARE_GPU:
        preflight(world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    if SHARE_GPU:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    from arterynetwork_amd import phantoms

    if args.integer_values:                                 # (both paths: the same run as the fractional volume with H / levels^2)
        if not args.levels:
            raise SystemExit('--integer-values needs --levels > 0')
        args.H = args.H / float(args.levels) ** 2
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        from arterynetwork_amd import slabs
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if SHARE_GPU and world > 1:
            dist.init_process_group('gloo')
            if args.transport == 'rccl':
                args.transport = 'ipc'
        else:
            dist.init_process_group('nccl', device_id=dev)
        try:
            if args.partition == 'zslab':
                out = slabs.bench_slabs(shape, args, dev, rank, world, roofline, configure, load_traffic)
            elif world == 1:
                from arterynetwork_amd import replica
                out = replica.bench_proxy(shape, args, dev, roofline, configure, load_traffic)
            else:
                from arterynetwork_amd import replica
                out = replica.bench_replicas(shape, args, dev, rank, world, roofline, configure, load_traffic)
        except Exception:
            import traceback
            sys.stderr.write('[rank {} of {}] bench_slabs failed:\n{}'.format(rank, world, traceback.format_exc()))
            sys.stderr.flush()
            raise
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        if out['config']['rccl_ranks'] != world and not (args.partition == 'replica' and out['config'].get('transport') == 'ipc') and not SHARE_GPU:   # the data path must be RCCL (or hipIpc) over all ranks, or the line is not the N-GPU line
            raise SystemExit(3)
        return

    from arterynetwork_amd._capi import Session
    if args.refine_like:
        print(json.dumps({'refine_like': refine_like_line(dev, args, configure, tubes=max(16, args.tubes) if args.tubes > 1 else 16)}), flush=True)
        return
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask, integer_values=args.integer_values,
                                        tubes=args.tubes, seed_mode=args.seed_mode)
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    s = Session(shape, device=local_rank)
    configure(s, args)
    s.set_volume_ptr(I.data_ptr(), np.float32, [st for st in I.stride()])
    s.set_labels_ptr(vm.data_ptr(), np.uint8, [st for st in vm.stride()])
    t0 = time.perf_counter()
    s.init(args.H)
    t_init = time.perf_counter() - t0
    big = 10 ** 15
    r0 = s.run(args.warmup, big, None)                      # W untimed warm-up sweeps
    assert r0.sweeps == args.warmup, 'warm-up stopped early: {}'.format(r0.stop_reason)
    db0 = s.stats()['dense_bytes']                          # bytes a dense pass has to fetch with the labels as they are now
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = s.run(args.warmup + args.steps, big, None)          # EXACTLY K timed sweeps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dense_bytes = (db0 + s.stats()['dense_bytes']) / 2.0 if args.skip_excluded else None
    valid = (r.sweeps == args.steps)
    ms_per_step = dt / max(1, r.sweeps) * 1e3
    value = V * r.sweeps / dt / 1e6
    kern_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    chain_beside_ms = r.chain_kernel_ms / r.chain_launches if r.chain_launches else None
    tr = s.trace()
    nlev = s.nlevels()
    st = s.stats()
    lev_note = '{} distinct intensities'.format(nlev)
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape), 'value': round(value, 1),
        'unit': 'Mvoxel-iter/s', 'n_gpus': 1, 'steps': int(r.sweeps), 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64',                # region sums, densities and decisions in float64; labels u8
        'data': 'synthetic', 'valid': bool(valid),
        'config': {'workload': '{} synthetic MRA tube volume ({} stored {}, {}), H={}, {} incremental VRG sweeps'.format(
                                   args.shape, lev_note, 'as u16 level indices' if args.storage16 else 'fp32',
                                   'no excluded voxels' if args.no_brain_mask else 'brain-mask excluded voxels', args.H, r.sweeps),
                   'parallelism': 'single GPU', 'sweep_variant': args.variant,
                   'intensity_storage': 'u16 level index (2 B/voxel)' if args.storage16 else 'fp32 (4 B/voxel)',
                   'init_seconds': round(t_init, 3), 'nseg_start': int(tr['nseg'][args.warmup]),
                   'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
                   'flips_per_sweep_mean': round(float(tr['nflip'][args.warmup + 1:].mean()), 1),
                   'init_note': 'vrg_init of a fresh handle in a fresh process: allocation, code-object load, level table, band, histograms, class bits, unit list, first recount',
                   'dense_ms': round(kern_ms, 4), 'dense_events_every': args.events,
                   # the band chain of a sweep (k_band's start to k_close's end) as it ran BESIDE the dense pass in the timed
                   # sweeps; where the dense pass bounds the step it includes k_close's wait for the pass of two sweeps ago
                   'band_chain_beside_dense_ms': round(chain_beside_ms, 4) if chain_beside_ms else None,
                   # (vrg.h option nt_loads: a pass of up to ~300 MB is read with ordinary loads and stays in the Infinity Cache)
                   'dense_pass_loads': 'non-temporal' if st['dense_nt_loads'] else 'ordinary', 'dense_workgroups': st['dense_workgroups'],
                   'dense_listed_units': st['dense_listed_units']},
        'roofline': roofline(shape, shape[2], kern_ms, int(r.sweep_launches), load_traffic(shape, 1, args.storage16, None, dense_bytes), args.storage16, dense_bytes, st['dense_kernel']),
    }
    out['config']['engine'] = st                            # trips handed back to the host / array growth during the run
    out['config'].update(s.chain_timing(args.H))
    # init once more on the warm handle (labels set again): what a second volume of the same shape costs
    s.set_option('dense_off', 0)
    s.set_labels_ptr(vm.data_ptr(), np.uint8, [st_ for st_ in vm.stride()])
    t0 = time.perf_counter()
    s.init(args.H)
    out['config']['reinit_seconds'] = round(time.perf_counter() - t0, 4)
    s.close()
    if out['roofline'].get('traffic') is not None:
        out['roofline']['traffic_source'] = 'profiles/traffic.json@{} (rocprofv3 PMC passes of these kernel sources, FETCH_SIZE x 2 + WRITE_SIZE; not measured by this run)'.format(device_source_sha())
    if not args.no_side_lines and args.shape == '880x880x640' and not (args.storage16 or args.no_brain_mask or args.tubes > 1):
        del I, vm
        torch.cuda.empty_cache()
        out['roofline_nomask'] = side_line(shape, dev, args, configure, roofline, 'nomask')
        out['many_flip'] = side_line(shape, dev, args, configure, roofline, 'many_flip')
        out['refine_like'] = side_line(shape, dev, args, configure, roofline, 'refine_like')
        I, vm = phantoms.bench_volume_torch(shape, dev, levels=args.levels, brain_mask=not args.no_brain_mask, integer_values=args.integer_values)   # (the cpu_baseline's volume)
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(I, vm, args.H, lev_note)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
