"""The kernels of the band chain and of the dense pass must not use scratch memory (and stay within a register budget).

Round 5 lost 2 us per launch - the chain alone 0.030 -> 0.035 ms - when padding inside VrgState made the compiler park a by-value copy of
the state in scratch; nothing failed, only timing on a GPU showed it.  This check needs no GPU: hipcc cross-compiles
the device code to gfx950 assembly and the kernels' resource records are read from it."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, 'arterynetwork_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# kernel name fragment (mangled) -> most VGPRs (arch + accumulation) it may use
BUDGET = {'6k_bandILi4E': 170, '6k_bandILi8E': 170, '6k_bandILi16E': 170, '7k_sweepILb0E': 150, '7k_sweepILb1E': 130, '7k_orderE': 64, '14k_mark_relabelILi1E': 144,
          '14k_mark_relabelILi4E': 144, '14k_mark_compactE': 168, '7k_closeE': 110, '6k_gateE': 64, '14k_recount_pipeILi3ELb0E': 144, '14k_recount_pipeILi3ELb1E': 144, '11k_rank_wideE': 64}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_chain_kernels_use_no_scratch(tmp_path):
    out = tmp_path / 'vrg_device.s'
    p = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-o', str(out), 'vrg_device.hip'],
                       cwd=CSRC, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    text = out.read_text()
    recs = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', text, re.S):          # one metadata record per kernel
        body = m.group(2)
        f = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, body).group(1))
        recs[m.group(1)] = (f('private_segment_fixed_size'), f('vgpr_count'))
    assert recs, 'no kernel records in the assembly'
    seen = set()
    for name, (scratch, vgpr) in recs.items():
        for frag, budget in BUDGET.items():
            if frag in name:
                seen.add(frag)
                assert scratch == 0, '%s uses %d bytes of scratch per thread' % (name, scratch)
                assert vgpr <= budget, '%s uses %d VGPRs (budget %d)' % (name, vgpr, budget)
    assert seen == set(BUDGET), 'kernels not found: %s' % sorted(set(BUDGET) - seen)
