# Round profiles: kernel-trace summary of a training-step run and of the bench command, PMC passes (MFMA utilisation, HBM traffic).
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh r02
R=${1:-r06}
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
export HD_COMMIT=${HD_COMMIT:-unknown}
STEPS=20 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_step --output-format csv -- python3 tools/bench_step.py > gpurun_out/prof_step.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
STEPS=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d gpurun_out/pmc_mfma --output-format csv -- python3 tools/bench_step.py > gpurun_out/pmc_mfma.log 2>&1
STEPS=3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmc_gui --output-format csv -- python3 tools/bench_step.py > gpurun_out/pmc_gui.log 2>&1
STEPS=3 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch --output-format csv -- python3 tools/bench_step.py > gpurun_out/pmc_fetch.log 2>&1
STEPS=3 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write --output-format csv -- python3 tools/bench_step.py > gpurun_out/pmc_write.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/pmc_gui gpurun_out/${R}_conv_mfma_util.json > /dev/null
F=$(ls gpurun_out/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls gpurun_out/pmc_write/*/*counter_collection.csv | head -1)
python3 tools/pmc_traffic.py $F $W conv_igemm_kernel,gemm_w8_kernel,conv3x3_small_kernel,conv3x3_w8_kernel,conv3x3_m160_kernel,conv3x3_c64_kernel,conv7x7s2_stem_kernel,conv3x3_c32to128_kernel,conv3x3_cat128to32_kernel gpurun_out/${R}_conv_traffic.json > /dev/null
STEPS=12 rocprofv3 --kernel-trace -d gpurun_out/trace_ss --output-format csv -- python3 tools/bench_step.py > gpurun_out/trace_ss.log 2>&1
python3 tools/trace_gaps.py $(ls gpurun_out/trace_ss/*/*kernel_trace.csv | head -1) 0.5 --table --aten > gpurun_out/${R}_steady_state.txt 2>&1
rm -rf gpurun_out/trace_ss
python3 tools/conv_table.py all > gpurun_out/${R}_conv_table.txt 2>&1
# the bench line quotes the PMC summaries from profiles/: put this run's there first (else the line marks them stale)
cp gpurun_out/${R}_conv_mfma_util.json gpurun_out/${R}_conv_traffic.json profiles/
python3 bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err
cp $(ls gpurun_out/prof_step/*/*kernel_stats.csv | head -1) gpurun_out/${R}_train_step_kernel_stats.csv
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_step gpurun_out/prof_bench gpurun_out/pmc_mfma gpurun_out/pmc_gui gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -2 gpurun_out/prof_step.log; tail -c 600 gpurun_out/prof_bench.log; cat gpurun_out/${R}_conv_mfma_util.json | head -40
