#!/bin/bash
# Turns a directory written by tools/collect_profiles.sh (merged back under gpurun_out/) into the committed
# summaries under profiles/.   usage: tools/publish_profiles.sh gpurun_out/<dir> [round tag, default r02]
set -e
cd "$(dirname "$0")/.."
P=$1
R=${2:-r06}
note="-- PMC passes (one rocprofv3 run per counter set). FETCH_SIZE is KiB and reads 1/2 of the true bytes (profiles/r01_fetch_size_calibration.md)"
pmc() { python3 tools/summarize_prof.py "round ${R#r0}, final kernel: tools/profile_target.py $2 $note" $P/$1_fetch/t_kernel_trace.csv $P/$1_fetch/t_counter_collection.csv $P/$1_write/t_counter_collection.csv $P/$1_sq/t_counter_collection.csv $P/$1_sq2/t_counter_collection.csv > profiles/$3; }
pmc 4k     "32 x 4096^2 (no map: the headline batch), MODE_EXACT"        ${R}_final_exact_4k_pmc.md
pmc 8kmap  "2 x 8192^2 with map, MODE_EXACT"        ${R}_final_exact_8k_map_pmc.md
pmc 1080p  "128 x 1920x1080 (no map: configs[3] per-GPU share), MODE_EXACT"    ${R}_final_exact_1080p_pmc.md
[ -d $P/1080p32_sq ] && pmc 1080p32 "32 x 1920x1080 (no map), MODE_EXACT"    ${R}_final_exact_1080p_32pairs_pmc.md
pmc 4kfast "8 x 4096^2 (no map), MODE_FAST (reference-order E planes, separable mu)" ${R}_final_fast_4k_pmc.md
[ -d $P/4ksep_sq ] && pmc 4ksep "8 x 4096^2 (no map), MODE_SEPARABLE" ${R}_final_separable_4k_pmc.md
[ -d $P/4kdouble_sq ] && pmc 4kdouble "4 x 4096^2 with map, MODE_DOUBLE (ssim_strip1_kernel<2,true>)" ${R}_final_double_4k_map_pmc.md
cp $P/bench_4k.json profiles/${R}_final_bench.json
cp $P/bench_8k-map.json profiles/${R}_final_bench_8k_map.json
cp $P/bench_1080p.json profiles/${R}_final_bench_1080p.json
cp $P/bench_under_rocprof.json profiles/${R}_final_bench_under_rocprof.json
cp $P/trace/t_kernel_stats.csv profiles/${R}_final_bench_kernel_stats.csv
python3 tools/summarize_prof.py "round ${R#r0}, final kernel: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (32 x 4096^2 per step; 20 timed steps + gate, clock-settle and warm-up launches; also the fast-mode and single-pair legs)" $P/trace/t_kernel_trace.csv > profiles/${R}_final_bench_kernel_trace.md
cp $P/mode_speeds.txt profiles/${R}_final_mode_speeds.txt
cp $P/error_table.md profiles/${R}_error_table.md
for f in bench_rccl_1rank.json bench_rccl_1rank_torch.json bench_c4_strong.json host_call_probe.txt latency_probe.txt rccl_selftest.txt cold_start.txt concurrent_callers.txt; do [ -f $P/$f ] && cp $P/$f profiles/${R}_final_$f; done
for f in soak.txt fullsize_check.txt balanced_check.txt tune_sweep_quick.txt; do [ -f $P/$f ] && cp $P/$f profiles/${R}_$f; done
# traffic.json as the GPU box wrote it (stamped with the kernel source the passes ran); regenerated here only for collections that predate that step
if [ -f $P/traffic.json ]; then cp $P/traffic.json profiles/traffic.json; else python3 tools/make_traffic_json.py "$P" "$R"; fi
