# Same-box A/B of two checkouts (round 5 in ab_r5/, this tree), alternating processes; then the data-parallel code path at world size 1.
#   bash tools/ab_round.sh   (on the GPU box, from the repo root)
#   Needs a built checkout of the other round in ab_r5/ (git-ignored):  git worktree add ab_r5 74ba14c && (cd ab_r5 && python -c "import __graft_entry__ as g; g.build()")
#   and remove it afterwards (git worktree remove ab_r5 --force): gpurun ships it to the box with the snapshot.
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/ab
line() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'images/s', d['ms_per_step'], 'ms/step', 'frac', d.get('roofline',{}).get('frac'), 'p50', d.get('ms_per_step_p50'))" $1 "$2"; }
for i in 1 2; do
  (cd ab_r5 && timeout 300 python3 bench.py --no-cpu-baseline > ../gpurun_out/ab/r5_$i.json 2> ../gpurun_out/ab/r5_$i.err); line gpurun_out/ab/r5_$i.json "round5 #$i"
  timeout 300 python3 bench.py --no-cpu-baseline > gpurun_out/ab/r6_$i.json 2> gpurun_out/ab/r6_$i.err; line gpurun_out/ab/r6_$i.json "round6 #$i"
done
(cd ab_r5 && HD_FORCE_DIST=1 timeout 300 python3 bench.py --no-cpu-baseline > ../gpurun_out/ab/r5_dist.json 2> ../gpurun_out/ab/r5_dist.err); line gpurun_out/ab/r5_dist.json "round5 HD_FORCE_DIST=1"
HD_FORCE_DIST=1 timeout 300 python3 bench.py --no-cpu-baseline > gpurun_out/ab/r6_dist.json 2> gpurun_out/ab/r6_dist.err; line gpurun_out/ab/r6_dist.json "round6 HD_FORCE_DIST=1"
HD_FORCE_DIST=1 STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_fd --output-format csv -- python3 tools/bench_step.py > gpurun_out/ab/trace_fd.log 2>&1 < /dev/null
T=$(find gpurun_out/trace_fd -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && timeout -k 10 120 python3 tools/trace_gaps.py $T 0.5 --table --aten > gpurun_out/ab/r06_forced_dist_steady_state.txt 2>&1 < /dev/null
rm -rf gpurun_out/trace_fd
head -30 gpurun_out/ab/r06_forced_dist_steady_state.txt
