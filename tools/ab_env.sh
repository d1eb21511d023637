# Same-box A/B of environment knobs on the whole step: every setting once per round, ROUNDS rounds (default 2).
#   bash tools/ab_env.sh "" "HD_W8_TS=2" "HD_M160_FIXED=7000 HD_M320_FIXED=8500"
cd /root/repo
R=${ROUNDS:-2}
for i in $(seq 1 $R); do
  for e in "$@"; do
    printf "%-60s " "[$e]"
    env $e timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['roofline']['groups']; print(d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'unet', g['unet_conv_blocks_total']['ms'], 'det', g['detector_conv']['ms'])"
  done
done
