set -o pipefail
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; cd /tmp
for rb in 64 128; do
  AZ_TRAIN_RB=$rb rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04c_prof_$rb -o k -- python3 $R/tools/train_step_bench.py othello8 512 300 > $OUT/r04c_$rb.log 2>&1
  f=$(find $OUT/r04c_prof_$rb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04c_train_512_rb${rb}_kernel_stats.csv
  rm -rf $OUT/r04c_prof_$rb
done
AZ_TRAIN_RB=64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04c_prof_256 -o k -- python3 $R/tools/train_step_bench.py othello8 256 300 > $OUT/r04c_256.log 2>&1
f=$(find $OUT/r04c_prof_256 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04c_train_256_rb64_kernel_stats.csv
rm -rf $OUT/r04c_prof_256
ls $OUT | grep r04c
