#!/bin/bash
# Profile session (run on the GPU box through gpurun): kernel-trace stats + PMC passes (each counter set in a
# run of its own, with --kernel-trace/--stats only in the first; the PMC runs use --serial 1: counter collection runs one
# kernel at a time and two of the kernels wait on the device for a kernel of the other stream).
# (--no-side-lines: the default bench line also runs a no-mask and a many-flip volume through the same kernels; here every launch belongs to the workload named)
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:?usage: profile_bench.sh <tag> [bench args...]}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-lines "$@" > "$OUT/bench_trace.log" 2>&1
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
run_pmc() { n=$1; ctrs=$2; shift 2; rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$n" -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-side-lines --serial 1 "$@" > "$OUT/bench_$n.log" 2>&1; echo "pmc $n rc=$?"; }
run_pmc fetch "FETCH_SIZE" "$@"
run_pmc write "WRITE_SIZE" "$@"
run_pmc sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "$@"
run_pmc sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_INSTS_SMEM" "$@"
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@"
run_pmc tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "$@"
run_pmc grbm "GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-side-lines "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-300
python3 tools/summarize_profile.py "$TAG" 2>&1 | tail -40
