#!/usr/bin/env python3
"""Run bench.py over a grid of (variant, sweep_blocks) in one process tree and print a compact table."""
import json, subprocess, sys
variants = [int(v) for v in sys.argv[1].split(',')]
blocks = [int(b) for b in sys.argv[2].split(',')]
extra = sys.argv[3:]
for v in variants:
    for b in blocks:
        cmd = [sys.executable, 'bench.py', '--no-cpu-baseline', '--variant', str(v), '--sweep-blocks', str(b)] + extra
        out = subprocess.run(cmd, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print('variant', v, 'blocks', b, 'FAILED', out.stderr[-300:])
            continue
        d = json.loads(line[-1])
        r = d['roofline']
        print('variant %3d blocks %5d  ms/step %.4f  kernel_ms %.4f  GB/s %7.1f  frac %.3f  value %.0f valid %s' % (
            v, b, d['ms_per_step'], r['kernel_ms_avg'], r['achieved'] or 0, r['frac'] or 0, d['value'], d['valid']), flush=True)
