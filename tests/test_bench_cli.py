"""bench.py plumbing that needs no GPU: `--gpus N` started plainly launches the N ranks itself (a child
torch.distributed.run, before this process touches the GPU); the roofline arithmetic; the traffic stamp."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench


def test_gpus_n_spawns_a_launcher(monkeypatch):
    calls = []
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(bench, 'visible_gpus', lambda: 8)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '7', '--warmup', '3'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '7', '--warmup', '3'] and os.path.basename(cmd[-7]) == 'bench.py'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert '--log-dir' in cmd and cmd[cmd.index('--tee') + 1] == '2'          # every rank's stderr is shown and kept (stdout - the JSON line - passes through untouched)


def test_gpus_n_preflight_and_rank_stderr(monkeypatch, capsys):
    """--gpus N on a node with fewer devices: exit code 2 and a message BEFORE anything is spawned; a launch that fails
    prints the tail of every rank's stderr log."""
    calls = []
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: calls.append(cmd) or 0)
    monkeypatch.setattr(bench, 'visible_gpus', lambda: 1)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and not calls
    assert 'needs 4 visible GPUs, this node shows 1' in capsys.readouterr().err

    def failing(cmd, env=None):
        d = os.path.join(cmd[cmd.index('--log-dir') + 1], 'attempt_0', '1')
        os.makedirs(d)
        open(os.path.join(d, 'stderr.log'), 'w').write('Traceback ...\nVrgError: RCCL all-reduce of the slab statistics failed\n')
        return 1
    monkeypatch.setattr(bench.subprocess, 'call', failing)
    monkeypatch.setattr(bench, 'visible_gpus', lambda: 4)
    with pytest.raises(SystemExit) as e:
        bench.main()
    err = capsys.readouterr().err
    assert e.value.code == 1 and 'the 4-rank launch failed with exit code 1' in err and 'RCCL all-reduce of the slab statistics failed' in err


def test_roofline_is_a_fraction_of_peak():
    r = bench.roofline((880, 880, 640), 640, 0.32, 500, None)
    assert r['bytes_per_launch'] == int(4.25 * 896 * 884 * 640) == 2154414080
    assert 0.8 < r['frac'] < 0.86 and r['frac'] == round(r['achieved'] / 8000.0, 4)
    assert r['algorithmic_equiv_gbs'] > r['achieved']            # 6 B/voxel accounting is kept apart, never as frac
    r16 = bench.roofline((880, 880, 640), 640, 0.24, 500, None, storage16=True)
    assert r16['bytes_per_voxel'] == 2.25 and r16['frac'] < 1
    # bytes counted on the device (excluded runs not fetched) replace the streamed figure in frac, which stays <= 1
    rs = bench.roofline((880, 880, 640), 640, 0.19, 500, None, dense_bytes=1.06e9)
    assert rs['bytes_per_launch'] == 1060000000 and rs['bytes_counted_on_device'] and 0.6 < rs['frac'] < 0.75
    assert rs['streamed_equiv_gbs'] > 8000 > rs['achieved']


def test_traffic_only_for_the_sources_it_was_measured_on():
    t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    entries = t.get('entries', [t])
    assert entries
    for e in entries:                                  # one entry per profiled workload (shape, ranks, storage, slab planes)
        got = bench.load_traffic(tuple(e['shape']), e.get('n_gpus', 1), bool(e.get('storage16')), e.get('planes'))
        assert got == (e['hbm_bytes_per_launch'] if e.get('src_sha') == bench.device_source_sha() else None)
        assert bench.load_traffic(tuple(e['shape']), e.get('n_gpus', 1), not e.get('storage16'), e.get('planes')) is None or len(entries) > 1
        # ... and only for a pass that has to fetch the same bytes (within 2 %): the same shape without its brain mask is another workload
        ref = e.get('design_bytes_per_launch_counted_on_device')
        if ref and e.get('src_sha') == bench.device_source_sha():
            args = (tuple(e['shape']), e.get('n_gpus', 1), bool(e.get('storage16')), e.get('planes'))
            assert bench.load_traffic(*args, design_bytes=ref * 1.01) == e['hbm_bytes_per_launch']
            assert bench.load_traffic(*args, design_bytes=ref * 2.0) is None
    assert bench.load_traffic((1, 2, 3), 1, False) is None


def test_section8d_figure_is_kept_apart_from_frac():
    r = bench.roofline((880, 880, 640), 640, 0.178, 125, None, dense_bytes=1.0616e9, kernel='k_recount_bits<3,true,0,true>')
    assert r['kernel'] == 'k_recount_bits<3,true,0,true>' and 0.7 < r['frac'] < 0.8
    assert r['section8d_frac'] > 1.5 and 'not a fraction' in r['section8d_note']      # 6 B x voxels / time / peak: the driver's own figure
