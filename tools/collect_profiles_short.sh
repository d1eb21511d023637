# Kernel-trace summaries + steady-state table + bench line of the current build (no PMC passes).  On the GPU box, from the repo root:
#   bash tools/collect_profiles_short.sh r05 v0
R=${1:-r05}; V=${2:-v0}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
STEPS=20 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_step --output-format csv -- python3 tools/bench_step.py > gpurun_out/prof_step.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 < /dev/null
STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_ss --output-format csv -- python3 tools/bench_step.py > gpurun_out/trace_ss.log 2>&1 < /dev/null
T=$(find gpurun_out/trace_ss -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && timeout -k 10 120 python3 tools/trace_gaps.py $T 0.5 --table --aten > gpurun_out/${R}_steady_state_${V}.txt 2>&1 < /dev/null
rm -rf gpurun_out/trace_ss
A=$(find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1); [ -n "$A" ] && cp $A gpurun_out/${R}_train_step_${V}_kernel_stats.csv
B=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$B" ] && cp $B gpurun_out/${R}_bench_${V}_kernel_stats.csv
rm -rf gpurun_out/prof_step gpurun_out/prof_bench
timeout -k 10 500 python3 bench.py > gpurun_out/${R}_bench_line_${V}.json 2> gpurun_out/${R}_bench_line_${V}.err < /dev/null
head -24 gpurun_out/${R}_steady_state_${V}.txt; cut -c1-300 gpurun_out/${R}_bench_line_${V}.json
