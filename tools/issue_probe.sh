#!/bin/bash
# tools/issue_probe.sh -- on the GPU box: VALU issue rate vs waves per SIMD, and what FETCH_SIZE / WRITE_SIZE count for the
# access shapes of k_intra_packed (tools/issue_probe.hip).  Results under gpurun_out/issue_probe/.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/issue_probe
mkdir -p $O
tools/bin/issue_probe > $O/issue.txt 2>&1
tools/bin/issue_probe fetch > $O/fetch_plain.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- tools/bin/issue_probe fetch > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- tools/bin/issue_probe fetch > $O/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/issue_probe/pmc_{name}/*/*counter_collection.csv")
    if not f:
        print(name, "no counter file"); continue
    tot = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        tot[k] = tot.get(k, 0) + float(r["Counter_Value"])
    with open(f"gpurun_out/issue_probe/{name}_counters.txt", "w") as o:
        for (k, c), v in tot.items():
            o.write(f"{k:60s} {c:12s} {v:16.1f} KB = {v * 1024:.0f} B\n")
PY
cat $O/issue.txt $O/fetch_plain.txt $O/fetch_counters.txt $O/write_counters.txt
