#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench lines, the rocprofv3 kernel trace of the bench command and the
# PMC passes (one rocprofv3 run per counter set, torch-free target) that profiles/ summarises.
# usage: tools/collect_profiles.sh <out-subdir-of-gpurun_out>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-prof_final}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
# PMC passes FIRST: profiles/traffic.json is regenerated from them on this box, for the kernel source in this tree, so that the bench lines below carry roofline.traffic
pmc() {   # name, counters, target args...
  local name=$1 ctrs=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/$name" -o t -- python3 tools/profile_target.py "$@" > "$OUT/$name.log" 2>&1
}
for cfg in "4k 32 3 0 0 4096 4096" "8kmap 2 3 0 1 8192 8192" "1080p 128 3 0 0 1920 1080" "1080p32 32 3 0 0 1920 1080" "4kfast 8 3 1 0 4096 4096" "4ksep 8 3 4 0 4096 4096" "4kdouble 4 3 2 1 4096 4096"; do
  set -- $cfg; tag=$1; shift
  pmc ${tag}_fetch FETCH_SIZE "$@"
  pmc ${tag}_write WRITE_SIZE "$@"
  pmc ${tag}_sq  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "$@"
  pmc ${tag}_sq2 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "$@"
done
python3 tools/make_traffic_json.py "$OUT" r06 profiles/traffic.json > "$OUT/traffic_ratios.json" && cp profiles/traffic.json "$OUT/traffic.json"
for wl in 4k 8k-map 1080p; do
  timeout 600 python3 bench.py --workload $wl > "$OUT/bench_$wl.log" 2>&1
  grep '"metric"' "$OUT/bench_$wl.log" > "$OUT/bench_$wl.json"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --no-cpu-baseline --no-cold-start --sustain 0 > "$OUT/bench_under_rocprof.log" 2>&1
# the exchange step on a 1-rank communicator: the product's own carrier (--exchange native), torch's, and configs[3] whole
SSIM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-cold-start --no-configs --sustain 0 --exchange native > "$OUT/bench_rccl_1rank.log" 2>&1
grep '"metric"' "$OUT/bench_rccl_1rank.log" > "$OUT/bench_rccl_1rank.json"
SSIM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-cold-start --no-configs --sustain 0 --exchange torch > "$OUT/bench_rccl_1rank_torch.log" 2>&1
grep '"metric"' "$OUT/bench_rccl_1rank_torch.log" > "$OUT/bench_rccl_1rank_torch.json"
SSIM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-cold-start --no-configs --sustain 0 --workload 1080p --scaling strong > "$OUT/bench_c4_strong.log" 2>&1
grep '"metric"' "$OUT/bench_c4_strong.log" > "$OUT/bench_c4_strong.json"
for m in single absent-peer shards; do RMGR_SSIM_HIP_COMM_TIMEOUT_S=10 timeout 100 python3 tools/rccl_selftest.py $m 2>&1 | grep "rccl_selftest\|rmgr-ssim comm"; done > "$OUT/rccl_selftest.txt"
timeout 150 python3 tools/rccl_selftest.py single --with-torch 2>&1 | grep "rccl_selftest\|rmgr-ssim comm" >> "$OUT/rccl_selftest.txt"
timeout 300 python3 tools/host_call_probe.py 4096 1,8 > "$OUT/host_call_probe.txt" 2>&1
timeout 300 python3 tools/latency_probe.py > "$OUT/latency_probe.txt" 2>&1
grep '"metric"' "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json"
# kernel-only speeds per mode (HIP events, tools/ab.py): batch of 8, single pair, with map
{
  for m in 0 3 1 4 2; do timeout 300 python3 tools/ab.py 8 4096 $m 0 0 3 0 | tail -1 | sed "s/^/8 x 4096^2 mode $m:/"; done
  for m in 0 1 4 2; do timeout 300 python3 tools/ab.py 1 4096 $m 0 0 3 0 | tail -1 | sed "s/^/1 x 4096^2 mode $m:/"; done
  for m in 0 1 4 2; do timeout 300 python3 tools/ab.py 2 8192 $m 0 0 3 1 | tail -1 | sed "s/^/2 x 8192^2 + map mode $m:/"; done
  timeout 300 python3 tools/ab.py 8 4096 0 0 1 3 0 | tail -1 | sed "s/^/8 x 4096^2 mode 0 one-column kernel:/"
} > "$OUT/mode_speeds.txt" 2>&1
timeout 900 python3 tests/tools/error_table.py > "$OUT/error_table.md" 2> "$OUT/error_table.err"
# round 5: what a first call costs, what concurrent callers get, and the long parity runs of the final kernels
{ for w in plain split cli plain split; do timeout 120 python3 tools/cold_start_probe.py $w; done; } > "$OUT/cold_start.txt" 2>&1
{ for cfg in "4096 1" "4096 0" "1920 1"; do echo "# size map: $cfg"; timeout 120 python3 tools/concurrent_callers.py $cfg 1,2,4,6 1.5; done; echo "# RMGR_SSIM_HIP_POOL=1"; RMGR_SSIM_HIP_POOL=1 timeout 120 python3 tools/concurrent_callers.py 4096 1 1,4 1.5; } > "$OUT/concurrent_callers.txt" 2>&1
timeout 600 python3 tools/tune_sweep.py 0 0 quick > "$OUT/tune_sweep_quick.txt" 2>&1
{ time timeout 900 python3 tests/tools/soak.py 30000 20261004; } > "$OUT/soak.txt" 2>&1
timeout 900 python3 tests/tools/fullsize_check.py > "$OUT/fullsize_check.txt" 2>&1
timeout 300 python3 tools/balanced_check.py > "$OUT/balanced_check.txt" 2>&1
find "$OUT" -name "*.csv" | wc -l
du -sh "$OUT"
