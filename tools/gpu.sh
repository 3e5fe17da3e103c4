#!/bin/bash
# One parameterised GPU-box session script (run through gpurun from the repo root).
#   tools/gpu.sh <tag> <step> [<step> ...]
# steps:
#   smoke            build() + smoke()
#   tests            pytest -m gpu (whole suite)            tests:<expr>   pytest -m gpu -k <expr>
#   bench:<name>:<bench.py args separated by commas>        one bench line -> gpurun_out/<tag>/bench_<name>.json
#   lines            the round's standard bench lines (headline, driver args, 512, replica role proxies, Z-slabs 80/320, level regimes, 16-bit storage, tubes)
#   manyflip         ms per sweep against flips per sweep (tools/manyflip.py: 1 .. 512 tubes, both volumes)
#   slabs            only the slab lines + 512
#   levels           only the level-regime lines
#   repeat           rare-race hunt: tools/repeat_case.py 8 seeds x 500, repeat_batched, repeat_stress
#   chaos            interleaving campaign on the -DVRG_CHAOS build (tools/build_chaos.sh: random delays at every kernel entry and
#                    hand-off): the parity tests of the GPU suite, repeat_case / repeat_batched / repeat_stress, a fuzz campaign
#   fenced           product vs its fenced twin (-DVRG_FENCES): fuzz campaign on both, parity tests on the twin, the fences' cost
#   stamps[:shape]   in-kernel stamps of the band chain (diagnostic build, tools/build_stamps.sh)
#   prof:<tag2>:<bench args,comma separated>   rocprofv3 kernel stats + PMC passes (tools/profile_bench.sh)
#   traffic:<tag2>:<bench args>                FETCH_SIZE / WRITE_SIZE passes only
set -u
tag=${1:?tag}; shift
out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
summ() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-40s value %10.1f  ms/step %.4f  dense %.4f  chain alone %s beside %s  frac %s  valid %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), r.get('frac'), d['valid']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> "$out/bench_$name.err" | grep '^{' > "$out/bench_$name.json"; summ "$out/bench_$name.json"; }
manyflip() { for sh in 512x512x170 880x880x640; do timeout 900 python tools/manyflip.py $sh 1,16,64,128,512 --sweeps 60 2>/dev/null | grep '^{' >> "$out/manyflip_$sh.jsonl"; done; python3 - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/manyflip_*.jsonl')):
    for l in open(f):
        d = json.loads(l); print('%-14s tubes %4d flips/sweep %8.0f band %8d ms/sweep %.4f dense %.4f host-driven %d' % (d['shape'], d['tubes'], d['flips_mean'] or 0, d['band_end'], d['ms_per_sweep'], d['dense_ms'], d['host_driven_trips']))
PY
}
for step in "$@"; do
  case "$step" in
    smoke) python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log" ;;
    tests) ( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 ) > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -16 "$out/pytest_gpu.log" | cut -c1-300 ;;
    tests:*) ( time timeout 3000 python -m pytest tests -m gpu -x -q -k "${step#tests:}" ) > "$out/pytest_k.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_k.log"; tail -12 "$out/pytest_k.log" | cut -c1-300 ;;
    bench:*) IFS=: read -r _ name args <<< "$step"; IFS=, read -r -a a <<< "$args"; run "$name" "${a[@]}" ;;
    lines)
      run 880; run 880_driver --steps 20 --warmup 5
      run 512 --shape 512x512x170 --steps 200
      run proxy_880 --force-dist --no-cpu-baseline --steps 500                       # replica partition: the roles of 2-, 4- and 8-rank groups, one GPU; whole-run projection
      run proxy_880_driver_args --force-dist --no-cpu-baseline --steps 20 --warmup 5
      run proxy_512 --force-dist --no-cpu-baseline --steps 300 --shape 512x512x170
      run refine_like --refine-like
      for nz in 320 80; do run dist1_zslab_880x880x$nz --force-dist --partition zslab --shape 880x880x$nz --no-cpu-baseline --steps 300; done
      for lv in 4095 65535 0; do run 512_levels$lv --no-cpu-baseline --no-side-lines --shape 512x512x170 --steps 100 --levels $lv; done
      run 880_s16 --storage16 --no-cpu-baseline --no-side-lines
      run 1024_s16 --shape 1024x1024x1024 --storage16 --no-cpu-baseline --no-side-lines --steps 200
      for tb in 16 128; do run 880_tubes$tb --tubes $tb --no-cpu-baseline --no-side-lines --steps 100; done ;;
    manyflip) manyflip ;;
    slabs)
      run 512 --shape 512x512x170 --steps 200 --no-cpu-baseline
      for nz in 320 160 80; do run dist1_880x880x$nz --force-dist --shape 880x880x$nz --no-cpu-baseline --steps 300; done ;;
    levels) for lv in 4095 65535 0; do run 512_levels$lv --no-cpu-baseline --shape 512x512x170 --steps 100 --levels $lv; done ;;
    repeat)
      for sd in 3 11 19 27 42 77 101 202; do timeout 600 python tools/repeat_case.py $sd 500 2>&1 | grep -v amdgpu.ids | tail -1; done > "$out/repeat_case.log" 2>&1
      for sd in 3 11 19 27 42 77; do timeout 600 python tools/repeat_batched.py $sd 300 4096 8 2>&1 | grep -v amdgpu.ids | tail -1; done >> "$out/repeat_case.log" 2>&1
      timeout 900 python tools/repeat_stress.py 30 2>&1 | grep -v amdgpu.ids | tail -2 >> "$out/repeat_case.log"
      cat "$out/repeat_case.log" ;;
    chaos)
      bash tools/build_chaos.sh > /dev/null && ( export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_chaos.so
        timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_float32_input.py -m gpu -x -q -k "not event_sampling and not bench and not litmus" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -4 | cut -c1-200 > "$out/chaos.log"
        for sd in 3 11 19 27 42 77 101 202; do timeout 900 python tools/repeat_case.py $sd 150 2>&1 | grep -v amdgpu.ids | tail -1; done >> "$out/chaos.log" 2>&1
        for sd in 3 11 19 27; do timeout 900 python tools/repeat_batched.py $sd 100 4096 8 2>&1 | grep -v amdgpu.ids | tail -1; done >> "$out/chaos.log" 2>&1
        timeout 900 python tools/repeat_stress.py 10 2>&1 | grep -v amdgpu.ids | tail -2 >> "$out/chaos.log"
        timeout 1500 python tests/fuzz_gpu.py 1000 100 4000 2>&1 | grep -v amdgpu.ids | tail -1 >> "$out/chaos.log" ); cat "$out/chaos.log" ;;
    fenced)   # A/B of the product against its FENCED TWIN (tools/build_fenced.sh: acquire / release on every hand-off): the same fuzz ranges on both builds, the
              # parity tests of the many-flip and replica paths on the twin, and what the fences cost (512x512x170 step, both builds)
      bash tools/build_fenced.sh > /dev/null && {
        : > "$out/fenced_ab.log"
        for lib in libvrg_hip.so libvrg_hip_fenced.so; do
          echo "---- $lib" >> "$out/fenced_ab.log"
          ( export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/$lib
            timeout 1500 python tests/fuzz_gpu.py 2500 250 60000 2>&1 | grep -v amdgpu.ids | tail -3 >> "$out/fenced_ab.log"
            timeout 600 python bench.py --shape 512x512x170 --steps 300 --no-cpu-baseline --no-side-lines 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('512x512x170: ms/step', d['ms_per_step'], 'band chain alone', d['config'].get('band_chain_ms'), 'dense', d['config'].get('dense_ms'))" >> "$out/fenced_ab.log" )
        done
        ( export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_fenced.so
          timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_flip or refine_like or kinds or replica or n_rank or stepwise or golden" 2>&1 | grep -E "passed|failed|error" | tail -3 | cut -c1-200 >> "$out/fenced_ab.log" )
        cat "$out/fenced_ab.log"; } ;;
    stamps*) shp=${step#stamps}; shp=${shp#:}; shp=${shp:-512x512x170}
      bash tools/build_stamps.sh > /dev/null && ( export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
        python tools/chain_stamps.py $shp 1 60 2>&1 | grep -v amdgpu.ids > "$out/chain_stamps.log"
        python tools/chain_stamps.py $shp 0 60 2>&1 | grep -v amdgpu.ids >> "$out/chain_stamps.log" ); cat "$out/chain_stamps.log" ;;
    prof:*) IFS=: read -r _ t2 args <<< "$step"; IFS=, read -r -a a <<< "$args"; bash tools/profile_bench.sh "$t2" "${a[@]}" 2>&1 | tail -14 ;;
    traffic:*) IFS=: read -r _ t2 args <<< "$step"; IFS=, read -r -a a <<< "$args"; bash tools/profile_traffic.sh "$t2" "${a[@]}" 2>&1 | tail -4 ;;
    *) echo "unknown step $step" ;;
  esac
done
