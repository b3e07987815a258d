#!/bin/bash
export TMPDIR=/tmp; R=$PWD; cd /tmp
for s in "65536 768 3072" "50432 768 2304" "65536 3072 768" "8192 8192 8192"; do
  rm -rf /tmp/kn; rocprofv3 --kernel-trace --stats -d /tmp/kn -o kn --output-format csv -- python $R/tools/probe/blaslt_kernel_name.py $s > /tmp/kn.log 2>&1
  echo "== $s"; python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/kn/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'Cijk' in r['Name'] or 'gemm' in r['Name'].lower():
            print(r['Calls'], r['AverageNs'], r['Name'][:400])
PY
done
