"""The cross-workgroup hand-off protocols of the kernels, pinned by a litmus program that can fail (tools/xcdbench.hip).

The product hands data between workgroups of ONE launch in two places - the last-workgroup reduction of the dense recount
(sweep_finish) and k_close's closing ticket - without release/acquire fences: payload stored write-through (sc1), drained
with s_waitcnt vmcnt(0), then an agent-scope ticket; the workgroup whose ticket came last reads with sc1 loads.  The litmus
runs exactly that pattern 10^6 times on 256 workgroups spread over all XCDs, alone and beside a kernel that saturates HBM,
checking every total - and its NEGATIVE twin (one plain store in the hand-off set), which may be caught (reported, not
asserted: a race is allowed to stay hidden in one run).  The same for the barrier + hand-off forms the persistent-kernel
study of round 3 used.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_handoff_litmus_and_its_negative_variants():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    exe = G.build_litmus()
    out = subprocess.run([exe, '4000', '1000000'], capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        open(os.path.join(d, 'litmus.log'), 'w').write(text)
    assert out.returncode == 0 and 'LITMUS OK' in text, text[-3000:]
    lines = text.splitlines()
    tick = [l for l in lines if l.startswith('TICKET sc1-stored')]
    assert len(tick) == 2 and all('1000000 rounds checked, 0 wrong totals' in l for l in tick), tick
    # the NEGATIVE variants (a plain store in the hand-off set; plain stores across XCDs) are ALLOWED to be stale, not obliged
    # to: whether a race shows within one run depends on timing, so they are reported (gpurun_out/litmus.log), never asserted
    neg = [l for l in lines if l.startswith('TICKET PLAIN-stored')]
    cross = [l for l in lines if 'ALL XCDs (cross-XCD: expect errors)' in l]
    assert len(neg) == 2 and cross, 'the negative variants did not run'
    seen = sum(' 0 wrong totals' not in l for l in neg) + sum('errors 0 of' not in l for l in cross)
    print('negative litmus variants that showed the race this run: {} of {}'.format(seen, len(neg) + len(cross)))
    ok = [l for l in lines if l.startswith('LITMUS sc1 store')]
    assert ok and all('errors 0 of' in l for l in ok)
    # ... but a box on which NO negative variant ever fails proves nothing about the positive result: one longer retry, then the test
    # says so (xfail, with the count on record) instead of passing silently
    if seen == 0:
        again = subprocess.run([exe, '8000', '3000000'], capture_output=True, text=True, timeout=1500)
        text2 = again.stdout + again.stderr
        if os.path.isdir(d):
            open(os.path.join(d, 'litmus.log'), 'a').write('\n---- retry (no negative variant showed the race) ----\n' + text2)
        l2 = text2.splitlines()
        seen = sum(' 0 wrong totals' not in l for l in l2 if l.startswith('TICKET PLAIN-stored')) + sum('errors 0 of' not in l for l in l2 if 'ALL XCDs (cross-XCD: expect errors)' in l)
        if seen == 0:
            pytest.xfail('no negative litmus variant showed its race in two runs on this box: the positive result is not shown to be falsifiable here')
