set -o pipefail
python -m pytest tests/test_gpu_train_step.py -x -q > gpurun_out/r04d_train_tests.log 2>&1; echo rc=$? >> gpurun_out/r04d_train_tests.log
rm -f gpurun_out/r04d_train_bench.txt
for rb in 64 128; do for b in 512; do AZ_TRAIN_RB=$rb python tools/train_step_bench.py othello8 $b 600 2>&1 | grep -v amdgpu.ids | sed "s/^/RB=$rb /" >> gpurun_out/r04d_train_bench.txt; done; done
for b in 64 128; do python tools/train_step_bench.py othello8 $b 1500 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04d_train_bench.txt; done
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; cd /tmp
for b in 64 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04d_prof_$b -o k -- python3 $R/tools/train_step_bench.py othello8 $b 300 > $OUT/r04d_$b.log 2>&1
  f=$(find $OUT/r04d_prof_$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04d_train_${b}_kernel_stats.csv
  rm -rf $OUT/r04d_prof_$b
done
cd $R
tail -3 gpurun_out/r04d_train_tests.log; cat gpurun_out/r04d_train_bench.txt
