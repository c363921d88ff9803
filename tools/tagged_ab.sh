cd /tmp && export TMPDIR=/tmp
for T in 0 1; do
  for rep in 1 2; do
    rm -rf /tmp/tg_$T; TAGGED=$T MODE=p2p SHARDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg_$T -- python3 $GRAFT_REPO_ROOT/tools/loopback_steps.py > /tmp/tg_$T.log 2>&1 || tail -5 /tmp/tg_$T.log
    echo "TAGGED=$T rep $rep: $(grep -E 'update_xr_p2p' $(find /tmp/tg_$T -name '*kernel_stats.csv') | awk -F, '{print $1, $2, $4}' | cut -c1-60,200-260)"
    grep -E "update_xr_p2p" $(find /tmp/tg_$T -name '*kernel_stats.csv') | awk -F'",' '{print $2}' | head -2
  done
done
