#!/bin/bash
# Functional run of bench.py's N > 1 code path on a 1-GPU box: N ranks on device 0, gloo as the carrier (SSIM_BENCH_SHARED_DEVICE=1, a test mode).
# usage: tools/shared_device_run.sh <out-subdir-of-gpurun_out>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-shared_device}; mkdir -p $OUT
export SSIM_BENCH_SHARED_DEVICE=1
run() {   # name, ranks, bench args...
  local name=$1 n=$2; shift 2
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --steps 20 --warmup 2 --sustain 0 --no-configs --no-cold-start "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/summary.txt
  grep '"metric"' $OUT/$name.log > $OUT/$name.json
  grep -E "^bench.py: |native exchange|Error|error" $OUT/$name.log | head -20 >> $OUT/summary.txt
}
rm -f $OUT/summary.txt
run n2_auto 2 --exchange auto
run n2_torch 2 --exchange torch
run n4_torch_strong_1080p 4 --exchange torch --workload 1080p --scaling strong
run n8_torch 8 --exchange torch --pairs 4
# the same through bench.py's own launcher (python bench.py --gpus N without torch.distributed.run around it)
timeout 600 python3 bench.py --gpus 2 --steps 5 --warmup 1 --sustain 0 --no-configs --no-cold-start --no-cpu-baseline --pairs 4 --exchange torch > $OUT/n2_self_launch.log 2>&1
echo "n2_self_launch rc=$?" >> $OUT/summary.txt
grep '"metric"' $OUT/n2_self_launch.log > $OUT/n2_self_launch.json
grep -E "^bench.py: " $OUT/n2_self_launch.log | head -4 >> $OUT/summary.txt
cat $OUT/summary.txt
