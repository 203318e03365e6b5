#!/bin/bash
# rocprofv3 kernel-trace + stats of an arbitrary script: tools/prof_script.sh <out-subdir under gpurun_out> <script.py> [args]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; S=$R/$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $S "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python3 - $OUT/t <<'PY'
import csv,sys,glob,re
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    print('   %-56s calls %4s avg %9.1f us  min %9.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
