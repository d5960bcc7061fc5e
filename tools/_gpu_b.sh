set -o pipefail
python -m pytest tests/test_gpu_train_step.py -x -q > gpurun_out/r04h_train_tests.log 2>&1; echo rc=$? >> gpurun_out/r04h_train_tests.log
rm -f gpurun_out/r04h_train_bench.txt
for f in 0 1; do for b in 144 256 512; do AZ_TRAIN_FORK=$f python tools/train_step_bench.py othello8 $b 600 2>&1 | grep -v amdgpu.ids | sed "s/^/FORK=$f /" >> gpurun_out/r04h_train_bench.txt; done; done
AZ_TRAIN_FORK=1 python tools/train_step_bench.py connect4 512 600 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04h_train_bench.txt
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04h_prof -o k -- python3 $R/tools/train_step_bench.py othello8 512 300 > $OUT/r04h_512.log 2>&1
f=$(find $OUT/r04h_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04h_train_512_kernel_stats.csv
rm -rf $OUT/r04h_prof
cd $R
tail -3 gpurun_out/r04h_train_tests.log; cat gpurun_out/r04h_train_bench.txt
