#!/bin/bash
# PMC counters of the resident persistent kernel on one MI355X, separate passes with --kernel-trace only (as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE (KB per dispatch) and the LDS counters
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, for `cgsolver N out 400` with tol left at 1e-10 but 400 iterations at most, N = 2048
# (all rows in LDS) and 4096 (LDS + registers + 6 of 16 rows streamed every iteration).  One launch of the persistent kernel
# per run: counters / loop bodies = per iteration.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_resident_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in 2048 4096; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pmcr_${N}_$tag
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcr_${N}_$tag -- $R/conjugate-gradient_amd/cgsolver $N /tmp/pmcr_out.txt 200 > $OUT/n${N}_$tag.txt 2>&1 || { tail -5 $OUT/n${N}_$tag.txt; continue; }
    f="$(find /tmp/pmcr_${N}_$tag -name '*counter_collection.csv' | head -1)"
    [ -n "$f" ] && grep -i "k_cg_\|Counter_Name" "$f" > $OUT/n${N}_${tag}_counter_collection.csv
  done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        print(os.path.basename(f), r["Kernel_Name"][:60], r["Counter_Name"], r["Counter_Value"])
PY
