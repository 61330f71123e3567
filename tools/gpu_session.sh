#!/bin/bash
# One GPU-box session per invocation (run through gpurun from the repo root): tools/gpu_session.sh <session> [args...]
# Everything lands in gpurun_out/<tag>_*; the summaries worth keeping are copied into profiles/ by hand afterwards.
# (Round 4's twelve tools/r4_run_{c..p}.sh are this script's history: git log -- tools/r4_run_c.sh)
set -u
S=${1:?session name}; shift || true
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O
export PYTHONPATH=$REPO

bench_ab() {  # tag, rounds, then pairs "name=libpath|env" ... : interleaved short bench runs, one JSON per run
  local tag=$1 rounds=$2; shift 2
  for i in $(seq 1 $rounds); do
    for spec in "$@"; do
      local name=${spec%%=*} rest=${spec#*=}
      ( for kv in ${rest//|/ }; do [ -n "$kv" ] && export "$kv"; done
        python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --breakdown $O/${tag}_bd_${name}_$i.txt \
          > $O/${tag}_ab_${name}_$i.json 2> $O/${tag}_ab_${name}_$i.err )
    done
  done
  python - "$O" "$tag" <<'PY'
import glob, json, sys
o, tag = sys.argv[1:3]
for f in sorted(glob.glob("%s/%s_ab_*.json" % (o, tag))):
    try:
        d = json.load(open(f))
        tb = d["roofline"]["transformer_block"]
        print(f.split("/")[-1], "value %.0f  ms/step %.3f  rollout %.3f  update-only %.0f  tblock mfma_frac %.4f" %
              (d["value"], d["ms_per_step"], d["rollout_inference_ms_per_step"], d["update_only_env_steps_per_s"], tb["mfma_frac"]))
    except Exception as e:
        print(f, "ERR", e)
PY
}

pmc() {  # tag, script, counter groups... (one rocprofv3 run per group; summarised per kernel)
  local tag=$1 script=$2; shift 2
  local n=0
  for grp in "$@"; do
    n=$((n+1))
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${tag}_pmc/g$n -- \
        python $REPO/$script > $O/${tag}_pmc_g$n.log 2>&1 ); echo "pmc $tag g$n rc=$?"
  done
  python tools/pmc_kernels.py $O/${tag}_pmc > $O/${tag}_pmc_summary.txt 2>&1
  find $O/${tag}_pmc -name "*.csv" -size +1M -delete; find $O/${tag}_pmc -name "*.db" -delete
}

case $S in
a)  # round 5, session A: today's baseline line, the rolled-layer-loop build (I-cache probe) A/B, I-fetch / wait counters, ckpt tests
  python bench.py > $O/r5a_bench_base.json 2> $O/r5a_bench_base.err; tail -c 400 $O/r5a_bench_base.json
  (timeout 600 python -m pytest tests/test_gpu_ckpt.py -q -m gpu --tb=short 2>&1 | tail -15) > $O/r5a_ckpt_tests.log; tail -5 $O/r5a_ckpt_tests.log
  bench_ab r5a 3 "base=" "roll=V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_roll.so"
  G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL"
  G2="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES"
  G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL"
  pmc r5a_base tools/probe/wps_run.py "$G1" "$G2" "$G3"
  V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_roll.so pmc r5a_roll tools/probe/wps_run.py "$G1" "$G2" "$G3"
  head -120 $O/r5a_base_pmc_summary.txt
  ;;
b)  # round 5, session B: the extended default line (f32 / dp1 / option-variant legs), the rollout floor probe, scheduling variants
  python bench.py > $O/r5b_bench_full.json 2> $O/r5b_bench_full.err; tail -c 3000 $O/r5b_bench_full.json; tail -5 $O/r5b_bench_full.err
  [ -x tools/probe/rollout_floor ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/rollout_floor.hip -o tools/probe/rollout_floor
  tools/probe/rollout_floor > $O/r5b_rollout_floor.txt 2>&1; cat $O/r5b_rollout_floor.txt
  L=$REPO/vision4leg_amd
  bench_ab r5b 3 "base=" "eu1=V4L_LIB=$L/libv4l_hip_eu1.so" "ilp=V4L_LIB=$L/libv4l_hip_ilp.so" "bias0=V4L_LIB=$L/libv4l_hip_bias0.so"
  ;;
c)  # round 5, session C: runtime knobs — where the kernel arguments live (device memory vs host-coherent memory)
  bench_ab r5c 3 "base=" "devkernarg1=HIP_FORCE_DEV_KERNARG=1" "devkernarg0=HIP_FORCE_DEV_KERNARG=0"
  ;;
full)  # the whole GPU suite + the default bench line (what the driver runs at round end)
  (timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -60) > $O/${TAG:-r6}_gpu_tests.log; tail -8 $O/${TAG:-r6}_gpu_tests.log
  cp $O/parity.json $O/${TAG:-r6}_parity.json 2>/dev/null
  python bench.py > $O/${TAG:-r6}_bench_full.json 2> $O/${TAG:-r6}_bench_full.err; tail -c 600 $O/${TAG:-r6}_bench_full.json
  ;;
e)  # round 5, session E: the collector's hand-over variants (cast included) and fresh phase stamps of the rollout step kernels
  python tools/probe/collector_pipe.py > $O/r5e_collector_pipe.txt 2> $O/r5e_collector_pipe.err; cat $O/r5e_collector_pipe.txt
  python tools/probe/stamps_rollout.py > $O/r5e_stamps_rollout.txt 2>&1; tail -32 $O/r5e_stamps_rollout.txt
  ;;
f)  # round 5, session F: what the padding token tile costs (timing-only one-tile build: WRONG results, never shipped)
  bench_ab r5f 3 "base=" "onetile=V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_onetile.so"
  for v in base onetile; do for i in 1 2 3; do printf "%s %d: " $v $i; grep -E "wps_layer|wps_wgrad" $O/r5f_bd_${v}_$i.txt | awk '{printf "%s=%.1f ", $1, $4}'; echo; done; done
  ;;
h)  # round 5, session H: launch-chain trims (update opening in registers, Adam operands ahead of the norm, 16-byte slab loads
    # in the reduce) against the build before them (vision4leg_amd/libv4l_hip_base.so = a copy of that build)
  (timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp2.py -q -m gpu --tb=short -x -k "ppo_update or graph or dp or backward" 2>&1 | tail -8) > $O/r5h_tests.log; tail -4 $O/r5h_tests.log
  bench_ab r5h 3 "base=V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_base.so" "new="
  for v in base new; do for i in 1 2 3; do printf "%s %d: " $v $i; grep -E "begin_pack|clip_adam|wgrad_reduce|pack " $O/r5h_bd_${v}_$i.txt | awk '{printf "%s=%.1f ", $1, $4}'; echo; done; done
  ;;
evidence)  # the round's records for profiles/: tools/gpu_session.sh evidence <tag, e.g. r5>
  TAG=${1:-r5}; EV=$O/${TAG}_ev; mkdir -p $EV
  python bench.py --steps 20 --warmup 5 --breakdown $EV/breakdown.txt > $EV/bench_full.json 2> $EV/bench_full.err
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$EV/trace" -- \
      python $REPO/bench.py --no-cpu-baseline --no-parity > "$EV/bench_traced.json" 2> "$EV/trace.log" )
  python tools/update_timeline.py $EV/trace $EV/update_timeline.txt > /dev/null
  python tools/kernel_stats.py $EV/trace "rocprofv3 --kernel-trace of \`python bench.py --no-cpu-baseline --no-parity\` (1 warm-up + 3 timed epochs + 1 profiled rollout + update pass)" > $EV/kernel_stats.txt
  tools/pmc_pass.sh "$EV/pmc" > "$EV/pmc.log" 2>&1
  python tools/pmc_traffic.py "$EV/pmc/summary.txt" "$EV/pmc_traffic.json" > /dev/null
  for w in cnn mlp loco64 loco_vis cnn_vis; do
    python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > $EV/bench_$w.json 2> $EV/bench_$w.err
  done
  find "$EV" -name "*.csv" -size +1M -delete; find "$EV" -name "*.db" -delete
  du -sh $EV
  python - "$EV" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_incl_transfers_ratio"))
    except Exception as e: print(f, "ERR", e)
PY
  ;;
g)  # round 5, session G: rows per block of the grouped dense weight-grads (probe builds -DV4L_TN_SMALL_ROWS=64|256 vs 128)
  L=$REPO/vision4leg_amd
  bench_ab r5g 3 "base=" "tn64=V4L_LIB=$L/libv4l_hip_tn64.so" "tn256=V4L_LIB=$L/libv4l_hip_tn256.so"
  for v in base tn64 tn256; do for i in 1 2 3; do printf "%s %d: " $v $i; grep -E "gemm_tn_group|wps_wgrad|wgrad_reduce|fused_conv3" $O/r5g_bd_${v}_$i.txt | awk '{printf "%s=%.1f ", $1, $4}'; echo; done; done
  ;;
*) echo "unknown session $S"; exit 2 ;;
esac
