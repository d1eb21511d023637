cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
STEPS=20 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_step --output-format csv -- python3 tools/bench_step.py > gpurun_out/prof_step.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 < /dev/null
STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_ss --output-format csv -- python3 tools/bench_step.py > gpurun_out/trace_ss.log 2>&1 < /dev/null
T=$(find gpurun_out/trace_ss -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && timeout -k 10 120 python3 tools/trace_gaps.py $T 0.5 --table --aten > gpurun_out/r03_steady_state.txt 2>&1 < /dev/null
rm -rf gpurun_out/trace_ss
A=$(find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1); [ -n "$A" ] && cp $A gpurun_out/r03_train_step_v5_kernel_stats.csv
B=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$B" ] && cp $B gpurun_out/r03_bench_v4_kernel_stats.csv
rm -rf gpurun_out/prof_step gpurun_out/prof_bench
timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_line_a.json 2> gpurun_out/r03_bench_line_a.err < /dev/null
head -8 gpurun_out/r03_steady_state.txt; cut -c1-200 gpurun_out/r03_bench_line_a.json
