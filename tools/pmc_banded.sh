#!/bin/bash
# three separate counter passes over a short banded run (N = 2^24); summaries into gpurun_out/pmc_banded/
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -c1-5 | tr -d ' ')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $R/tools/banded_pmc.py 16777216 12 > /tmp/pmc_$tag.log 2>&1
  mkdir -p $R/gpurun_out/pmc_banded
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$R/gpurun_out/pmc_banded/$tag.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    k = (row["Kernel_Name"][:60], row["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
with open(sys.argv[2], "w") as o:
    for (kn, cn), (cnt, tot) in sorted(acc.items()):
        o.write("%-62s %-14s dispatches=%d mean=%.1f\n" % (kn, cn, cnt, tot / cnt))
PY
done
cat $R/gpurun_out/pmc_banded/*.txt
