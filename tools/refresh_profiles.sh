#!/bin/bash
# Runs on the MI355X box (gpurun): HBM counter passes, bench line, rocprofv3 kernel stats of the same command.
# usage: tools/refresh_profiles.sh TAG [GAMES]   -> gpurun_out/TAG_*  (copy what should be judged into profiles/)
# The counter passes come first: bench.py reads profiles/traffic.json for the roofline's `traffic` field.
set -eo pipefail
TAG=${1:-rXX}; G=${2:-32768}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o k -- python $R/tools/prof_net.py $G 3 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o k -- python $R/tools/prof_net.py $G 3 > $OUT/${TAG}_pmc_write.log 2>&1
cd $R
python tools/collect_profiles.py $TAG $G
cp $OUT/${TAG}_traffic.json profiles/traffic.json
python bench.py --steps 2 --warmup 1 --games $G > $OUT/${TAG}_bench.log 2>&1
grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o k -- python $R/bench.py --steps 1 --warmup 0 --games $G --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
cd $R
python tools/collect_profiles.py $TAG $G
