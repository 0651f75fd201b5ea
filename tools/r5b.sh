cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "train or pointwise or sync_bn or conv1x1" > gpurun_out/r5b/train_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/train_tests.log; tail -15 gpurun_out/r5b/train_tests.log
for t in iassd pagnet; do for p in fp32 fp16x2; do timeout -k 10 300 python3 tools/train_step_kernels.py $t $p > gpurun_out/r5b/kernels_${t}_${p}.txt 2>&1; grep -E "LIBRARY|library kernels|launches" gpurun_out/r5b/kernels_${t}_${p}.txt | cut -c1-150; done; done
timeout -k 10 600 python3 tools/backbone_train_time.py > gpurun_out/r5b/backbone_train_time_fp32.txt 2>&1; tail -5 gpurun_out/r5b/backbone_train_time_fp32.txt
SPS_TRAIN_PRECISION=fp16x2 timeout -k 10 600 python3 tools/backbone_train_time.py > gpurun_out/r5b/backbone_train_time_fp16x2.txt 2>&1; tail -5 gpurun_out/r5b/backbone_train_time_fp16x2.txt
