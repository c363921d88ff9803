cd $GRAFT_REPO_ROOT
for args in "--steps 500 --warmup 100" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --profile-every 2" "--steps 20 --warmup 5 --profile-every 4" "--steps 20 --warmup 5 --no-profile-gemv" "--steps 20 --warmup 5 --profile-update"; do
  for rep in 1 2; do
    python3 bench.py --matrix-size 12288 --no-cpu-baseline --no-solve-window --no-live-pmc $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-58s ms/step %.4f  dev_window %.4f  K1 median %s' % ('$args', d['ms_per_step'], d.get('device_window_ms_per_step',0), d['roofline'].get('median_launch_ms')))"
  done
done
