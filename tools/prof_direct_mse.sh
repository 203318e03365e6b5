#!/bin/bash
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_dm -- python3 $R/tools/time_direct_mse.py > /dev/null 2>&1
python3 - $R/gpurun_out/r04_dm <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    print('   %-52s calls %4s avg %9.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3))
PY
