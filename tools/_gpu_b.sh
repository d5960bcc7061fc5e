python -m pytest tests/test_gpu_train_step.py -x -q > gpurun_out/r04p_train_tests.log 2>&1; echo rc=$? >> gpurun_out/r04p_train_tests.log
rm -f gpurun_out/r04p_train_bench.txt
for rep in 1 2; do for b in 384 512; do python tools/train_step_bench.py othello8 $b 1000 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04p_train_bench.txt; done; done
for b in 384 512; do AZ_TRAIN_RB=64 python tools/train_step_bench.py othello8 $b 1000 2>&1 | grep -v amdgpu.ids | sed "s/^/RB=64 /" >> gpurun_out/r04p_train_bench.txt; done
tail -3 gpurun_out/r04p_train_tests.log; cut -c1-80 gpurun_out/r04p_train_bench.txt
