#!/usr/bin/env python3
"""bench.py - throughput of the VRG sweep on MI355X (BASELINE.json metric).

A "step" is one incremental update() sweep of the hot path (decide -> relabel stencil + region
statistics over every voxel -> band/density bookkeeping) over the synthetic volume.  Inputs are
generated on the GPU with torch (plumbing only) and are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape 880x880x640]

N > 1 is launched by torch.distributed.run (one rank per GPU): the volume is cut into Z-slabs with a
per-sweep halo exchange (arterynetwork_amd/slabs.py).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_VOXEL_ITER = 6     # 4 B fp32 intensity + 1 B label read + 1 B label write


def make_volume_torch(shape, device, seed=3, levels=255, radius=4.0, noise=0.1, seed_planes=3, H=2.25):
    """Configs 2-4 recipe generated directly in HBM, x-fastest layout.
    Returns (I, vm) as torch tensors of logical shape (nx,ny,nz) with element strides (1,nx,nx*ny)."""
    import torch
    nx, ny, nz = shape
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    xs = torch.arange(nx, device=device, dtype=torch.float32)[None, None, :]
    ys = torch.arange(ny, device=device, dtype=torch.float32)[None, :, None]
    zs = torch.arange(nz, device=device, dtype=torch.float32)[:, None, None]
    cy = ny / 2.0 + 0.18 * ny * torch.sin(2 * math.pi * xs / nx)
    cz = nz / 2.0 + 0.18 * nz * torch.cos(2 * math.pi * xs / nx)
    tube = ((ys - cy) ** 2 + (zs - cz) ** 2) <= radius ** 2            # (nz,ny,nx)
    I = torch.randn((nz, ny, nx), generator=g, device=device, dtype=torch.float32)
    I.mul_(noise).add_(tube.to(torch.float32))
    I = torch.round(I * levels) / levels
    ell = (((xs - (nx - 1) / 2.0) / (0.48 * nx)) ** 2 + ((ys - (ny - 1) / 2.0) / (0.48 * ny)) ** 2
           + ((zs - (nz - 1) / 2.0) / (0.48 * nz)) ** 2) <= 1.0
    vm = torch.full((nz, ny, nx), 3, dtype=torch.uint8, device=device)
    vm[~ell.expand(nz, ny, nx)] = 4
    vm[tube & (xs < seed_planes)] = 0
    return I.permute(2, 1, 0), vm.permute(2, 1, 0)


def cpu_baseline(I_t, vm_t, H, budget_s=20.0):
    """Time the oracle (a scalar C port of the reference, density_mode 1) on this host's cores, on a
    bounded crop of the same volume around the seeds. Returns the cpu_baseline JSON object."""
    from oracle import vrg_oracle as O
    nx, ny, nz = I_t.shape
    cx, cyy, czz = min(nx, 448), min(ny, 448), min(nz, 320)
    y0 = max(0, min(ny - cyy, ny // 2 - cyy // 2))
    z0 = max(0, min(nz - czz, int(nz / 2.0 + 0.18 * nz) - czz // 2))
    I = I_t[:cx, y0:y0 + cyy, z0:z0 + czz].contiguous().cpu().numpy().astype(np.float64)
    vm = vm_t[:cx, y0:y0 + cyy, z0:z0 + czz].contiguous().cpu().numpy()
    if not (vm == 0).any():
        return None
    o = O.Oracle(I, vm, H, density_mode=1)
    o.init()
    t0 = time.perf_counter()
    sweeps = 0
    while time.perf_counter() - t0 < budget_s and sweeps < 300:
        if o.step(10 ** 6, 10 ** 12, -1.0) != 0:
            break
        sweeps += 1
    dt = time.perf_counter() - t0
    o.close()
    if sweeps == 0:
        return None
    return {'value': round(I.size * sweeps / dt / 1e6, 3), 'unit': 'Mvoxel-iter/s', 'cores': 1, 'kind': 'port',
            'sample': '{} sweeps of the oracle (oracle/vrg_oracle.c, level-histogram mode) on the {}x{}x{} crop '
                      'around the seeds of the same volume, {:.1f} s'.format(sweeps, cx, cyy, czz, dt)}


def load_traffic(shape, n_gpus):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/traffic.json)."""
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        t = json.load(open(p))
        if t.get('shape') == list(shape) and t.get('n_gpus', 1) == n_gpus:
            return t.get('hbm_bytes_per_launch')
    except Exception:
        pass
    return None


def roofline(V_per_launch, kern_ms, launches, traffic, bytes_per_voxel=BYTES_PER_VOXEL_ITER):
    """Dominant kernel = k_recount_bits (dense region recount, one launch per sweep).  `achieved` uses the
    ALGORITHMIC 6 B/voxel-iter of SURVEY.md 8(d).  The kernel itself moves 4.25 B/voxel: labels are updated in
    place (no 1 B/voxel write-back) and the recount reads a 2-bit class volume instead of the label bytes, so
    `achieved` can exceed the HBM peak; `traffic` is the rocprofv3 PMC figure of what really crossed the bus and
    `traffic_frac_of_peak` the fraction of the HBM peak the kernel sustains."""
    achieved = bytes_per_voxel * V_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
    out = {'bound': 'hbm', 'kernel': 'k_recount_bits<3,true,true>' if bytes_per_voxel == 4 else 'k_recount_bits<3,true,false>', 'achieved': round(achieved, 1) if achieved else None,
           'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
           'kernel_ms_avg': round(kern_ms, 4), 'launches': launches,
           'algorithmic_bytes_per_launch': bytes_per_voxel * V_per_launch, 'traffic': traffic}
    if traffic and kern_ms > 0:
        out['traffic_gbs'] = round(traffic / (kern_ms * 1e-3) / 1e9, 1)
        out['traffic_frac_of_peak'] = round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shape', default='880x880x640')
    ap.add_argument('--levels', type=int, default=255)
    ap.add_argument('--H', type=float, default=2.25)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--variant', type=int, default=0)
    ap.add_argument('--sweep-blocks', type=int, default=0)
    ap.add_argument('--prio-mode', type=int, default=-1)
    ap.add_argument('--events', type=int, default=1, help='0: no HIP events around the dense launches (no roofline then)')
    ap.add_argument('--graph', type=int, default=0, help='replay the band kernels of each sweep from captured hipGraphs')
    ap.add_argument('--storage16', action='store_true', help='16-bit intensity storage (level indices): config 5 style; 4 B/voxel-iter algorithmic')
    ap.add_argument('--force-dist', action='store_true', help='use the N>1 code path (RCCL comm) even with one rank')
    args = ap.parse_args()
    shape = tuple(int(s) for s in args.shape.lower().split('x'))
    assert len(shape) == 3

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N '
                             '--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    if world > 1 or args.force_dist:
        import torch.distributed as dist
        from arterynetwork_amd import slabs
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=dev)
        out = slabs.bench_slabs(shape, args, dev, rank, world)
        if rank == 0:
            print(json.dumps(out))
        dist.barrier()
        dist.destroy_process_group()
        return

    from arterynetwork_amd._capi import Session
    I, vm = make_volume_torch(shape, dev, levels=args.levels, H=args.H)
    torch.cuda.synchronize()
    V = shape[0] * shape[1] * shape[2]
    s = Session(shape, device=local_rank)
    s.set_option('sweep_variant', args.variant)
    if args.sweep_blocks:
        s.set_option('sweep_blocks', args.sweep_blocks)
    if args.prio_mode >= 0:
        s.set_option('prio_mode', args.prio_mode)
    if args.storage16:
        s.set_option('storage16', 1)
    s.set_option('events', args.events)
    if args.graph:
        s.set_option('graph', 1)
    s.set_option('batch', 64)
    s.set_volume_ptr(I.data_ptr(), np.float32, [st for st in I.stride()])
    s.set_labels_ptr(vm.data_ptr(), np.uint8, [st for st in vm.stride()])
    t0 = time.perf_counter()
    s.init(args.H)
    t_init = time.perf_counter() - t0
    big = 10 ** 15
    r0 = s.run(args.warmup, big, None)                      # W untimed warm-up sweeps
    assert r0.sweeps == args.warmup, 'warm-up stopped early: {}'.format(r0.stop_reason)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = s.run(args.warmup + args.steps, big, None)          # EXACTLY K timed sweeps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    valid = (r.sweeps == args.steps)
    ms_per_step = dt / max(1, r.sweeps) * 1e3
    value = V * r.sweeps / dt / 1e6
    kern_ms = r.sweep_kernel_ms / max(1, r.sweep_launches)
    tr = s.trace()
    out = {
        'metric': 'Mvoxel-iters/sec, VRG sweep, {} volume'.format(args.shape), 'value': round(value, 1),
        'unit': 'Mvoxel-iter/s', 'n_gpus': 1, 'steps': int(r.sweeps), 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'u8 labels + f32 intensities (f64 region sums / densities)',
        'data': 'synthetic', 'valid': bool(valid),
        'config': {'workload': '{} synthetic MRA tube volume ({} intensity levels stored {}, brain-mask excluded '
                               'voxels), H={}, {} incremental VRG sweeps'.format(
                                   args.shape, args.levels, 'as u16 level indices' if args.storage16 else 'fp32', args.H, r.sweeps),
                   'parallelism': 'single GPU', 'sweep_variant': args.variant,
                   'intensity_storage': 'u16 level index (2 B/voxel)' if args.storage16 else 'fp32 (4 B/voxel)',
                   'init_seconds': round(t_init, 3), 'nseg_start': int(tr['nseg'][args.warmup]),
                   'nseg_end': int(tr['nseg'][-1]), 'band_end': int(tr['ni'][-1] + tr['no'][-1]),
                   'flips_per_sweep_mean': round(float(tr['nflip'][args.warmup + 1:].mean()), 1),
                   'hbm_gbs_whole_step': round(BYTES_PER_VOXEL_ITER * V / (ms_per_step * 1e-3) / 1e9, 1)},
        'roofline': roofline(V, kern_ms, int(r.sweep_launches), None if args.storage16 else load_traffic(shape, 1),
                             4 if args.storage16 else BYTES_PER_VOXEL_ITER),
    }
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(I, vm, args.H)
    s.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
