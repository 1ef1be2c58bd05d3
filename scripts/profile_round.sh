#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag> [part ...]   parts: trace traffic pmc variants bench (default: all)
# Everything the profiles/ directory holds for one state of the kernels, left under gpurun_out/ for
# scripts/assemble_profiles.py <tag>:
#   trace     rocprofv3 --kernel-trace --stats of the default bench.py command (one process: --count-work 0)
#   traffic   FETCH_SIZE / WRITE_SIZE per launch (scripts/traffic.sh) of configs 1 (both flavours), 2, 3, 4s
#   pmc       SQ / TCP / TCC counter passes of config 1 in both flavours and of the 1M-triangle scene (scripts/pmc_cfg.sh)
#   variants  kernel-trace stats of pipelines 2 (wavefront) and 4 (path waves + walker waves) on configs 1 and 4s
#   bench     the bench line of every configuration
TAG=$1; shift
PARTS=${@:-trace traffic pmc variants bench}
cd $GRAFT_REPO_ROOT
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has trace; then
  OUT=$G/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --count-work 0 > $OUT/bench.json 2> $OUT/bench.err) || { echo "trace failed"; tail -3 $OUT/bench.err; exit 1; }
  find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  head -6 $OUT/kernel_stats.csv | cut -c1-160
fi
if has traffic; then
  for ct in "1 1 0" "1 0 0" "2 1 0" "2 0 0" "4s 1 0" "4s 0 0" "3 1 64"; do
    set -- $ct
    scripts/traffic.sh $1 $2 $3 > $G/traffic_$1_$2.log 2>&1 || { echo "traffic $ct failed"; tail -5 $G/traffic_$1_$2.log; exit 1; }
    echo "traffic $ct done"
  done
fi
if has pmc; then
  A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  B="SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
  C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum"
  for ct in "1 1" "1 0" "4s 1"; do
    set -- $ct
    scripts/pmc_cfg.sh ${TAG}_$1_$2_a "$A" $1 $2 > /dev/null || { echo "pmc a $ct failed"; exit 1; }
    scripts/pmc_cfg.sh ${TAG}_$1_$2_b "$B" $1 $2 > /dev/null || { echo "pmc b $ct failed"; exit 1; }
    scripts/pmc_cfg.sh ${TAG}_$1_$2_c "$C" $1 $2 > /dev/null || { echo "pmc c $ct failed"; exit 1; }
    echo "pmc $ct done"
  done
fi
if has variants; then
  for cp in "1 2" "1 4" "4s 2" "4s 4" "1 0"; do
    set -- $cp
    OUT=$G/wf_${TAG}_$1_p$2; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/render_cfg.py $1 $([ $2 = 2 ] && echo 0 || echo 1) 16 $2 > $OUT/run.log 2>&1) || { echo "variant $cp failed"; tail -3 $OUT/run.log; exit 1; }
    find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
    grep "^config" $OUT/run.log | tail -1
  done
fi
if has bench; then
  for c in 1 2 3 4s; do
    timeout -k 10 600 python3 bench.py --config $c --steps 3 --warmup 1 > $G/bench_${TAG}_cfg$c.json 2> $G/bench_${TAG}_cfg$c.err || { echo "bench $c failed"; tail -3 $G/bench_${TAG}_cfg$c.err; exit 1; }
    python3 -c "import json;d=json.load(open('$G/bench_${TAG}_cfg$c.json'));print('config $c', round(d['value']), 'Mrays/s', round(d['ms_per_frame'],3), 'ms/frame; other flavour', round(d['other_flavour']['value']), round(d['other_flavour']['ms_per_frame'],3))"
  done
fi
