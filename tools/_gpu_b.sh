rm -f gpurun_out/r04s.txt
for pf in 1 2; do for b in 32 64; do AZ_TRAIN_PF=$pf python tools/train_step_bench.py othello8 $b 1500 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r04s.txt; done; done
AZ_TRAIN_PF=2 python -m pytest tests/test_gpu_train_step.py -x -q -k "autograd and 64" 2>&1 | tail -2 >> gpurun_out/r04s.txt
cut -c1-90 gpurun_out/r04s.txt
