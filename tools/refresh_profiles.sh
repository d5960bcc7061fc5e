#!/bin/bash
# Runs on the MI355X box (gpurun): counter passes, bench line, rocprofv3 kernel stats of the same command.
# usage: tools/refresh_profiles.sh TAG [all|net|kstep|bench|train|latency]   -> gpurun_out/TAG_*  (tools/collect_profiles.py condenses; copy into profiles/)
# Counter passes first (bench.py reads profiles/traffic.json, mfma_counters.json, kstep_counters.json).  Every pass is
# its own rocprofv3 run with --kernel-trace only (no --stats, no other trace domain), the program directly after `--`.
set -o pipefail
TAG=${1:-rXX}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters, program args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${TAG}_$name -o k -- python3 "$@" > $OUT/${TAG}_$name.log 2>&1 || echo "pass $name failed (see $OUT/${TAG}_$name.log)"
}
ONLY=${2:-all}   # all | net | kstep | bench | train | latency
if [ $ONLY = all ] || [ $ONLY = net ]; then
for W in "othello 32768" "othello 4096" "connect4 8192"; do
  set -- $W; g=$1; b=$2
  pass pmc_fetch_${g}_$b FETCH_SIZE $R/tools/prof_net.py $g $b 3
  pass pmc_write_${g}_$b WRITE_SIZE $R/tools/prof_net.py $g $b 3
  pass pmc_mfma_${g}_$b "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $R/tools/prof_net.py $g $b 3
  pass pmc_mops_${g}_$b "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU" $R/tools/prof_net.py $g $b 3
done
fi
if [ $ONLY = all ] || [ $ONLY = kstep ]; then
# the tree kernel inside self-play (12 plies from the start position): bytes and where its waves wait
for W in "othello 32768 100" "othello 4096 100" "connect4 8192 200"; do
  set -- $W; g=$1; b=$2; s=$3
  pass kstep_fetch_${g}_$b FETCH_SIZE $R/tools/prof_selfplay.py $g $b $s 12
  pass kstep_write_${g}_$b WRITE_SIZE $R/tools/prof_selfplay.py $g $b $s 12
  pass kstep_sq_${g}_$b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $R/tools/prof_selfplay.py $g $b $s 12
done
fi
cd $R
python3 tools/collect_profiles.py $TAG
cp $OUT/${TAG}_traffic.json profiles/traffic.json 2>/dev/null
cp $OUT/${TAG}_mfma_counters.json profiles/mfma_counters.json 2>/dev/null
cp $OUT/${TAG}_kstep_counters.json profiles/kstep_counters.json 2>/dev/null
if [ $ONLY = all ] || [ $ONLY = bench ]; then
python3 bench.py > $OUT/${TAG}_bench.log 2>&1
grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json   # the line of record (scalars only)
cp $R/bench_detail.json $OUT/${TAG}_bench_detail.json                   # every nested object of the same run
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o k -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-literal-configs > $OUT/${TAG}_prof.log 2>&1
# the literal BASELINE configs, one process each, so that their kernels do not mix in one table
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -o k -- python3 $R/tools/run_config.py othello 4096 > $OUT/${TAG}_prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c4 -o k -- python3 $R/tools/run_config.py connect4 8192 > $OUT/${TAG}_prof_c4.log 2>&1
fi
if [ $ONLY = all ] || [ $ONLY = train ] || [ $ONLY = bench ]; then
# the hand-written training step: rocprofv3 stats of 500 steps at the reference's batch size and at 512
cd /tmp
for W in "othello8 64 500" "othello8 512 200" "connect4 64 500" "tictactoe 64 500"; do
  set -- $W
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train_$1_$2 -o k -- python3 $R/tools/train_step_bench.py $1 $2 $3 > $OUT/${TAG}_train_$1_$2.log 2>&1
  f=$(find $OUT/${TAG}_prof_train_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_train_$1_$2_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_train_$1_$2
done
fi
if [ $ONLY = all ] || [ $ONLY = latency ]; then
# the latency regime: rocprofv3 stats of a one-game wave, a 64-game wave (k_trunk_q, k_dense_frag, k_heads2, k_step) and stage times over row counts
cd /tmp
for G in 1 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_lat_$G -o k -- python3 $R/tools/run_config.py othello $G > $OUT/${TAG}_lat_$G.log 2>&1
  f=$(find $OUT/${TAG}_prof_lat_$G -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_lat_${G}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_lat_$G
done
cd $R
python3 tools/dense_bench.py 1 64 256 512 1024 2048 4096 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stage_times.txt
python3 tools/stage_bench.py connect4 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_stage_times.txt
fi
cd $R
python3 tools/collect_profiles.py $TAG
# gpurun copies back at most 64 MiB: keep the condensed files, drop the raw per-dispatch tables
rm -rf $OUT/${TAG}_pmc_* $OUT/${TAG}_kstep_fetch_* $OUT/${TAG}_kstep_write_* $OUT/${TAG}_kstep_sq_* $OUT/${TAG}_prof $OUT/${TAG}_prof_c2 $OUT/${TAG}_prof_c4
ls -la $OUT | grep ${TAG}_
