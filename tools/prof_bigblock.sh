#!/bin/bash
# rocprofv3 kernel-trace of the four big-block shapes (6144 / 3072 frames per launch: whole rounds of workgroups): tools/prof_bigblock.sh <out-subdir under gpurun_out>
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for spec in "8192 24" "8192 32" "16384 24" "16384 32"; do set -- $spec; fr=$((50331648 / $1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$1_p$2 -- python3 $R/tools/prof_config.py --n $1 --order $2 --frames $fr > /dev/null 2>&1
  python3 - $OUT/n$1_p$2 $1 $2 <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
tot=0
print('== n=%s order=%s'%(sys.argv[2],sys.argv[3]))
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    print('   %-46s calls %s avg %.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['TotalDurationNs'])/6
print('   total per call %.1f us -> %.1f G input samples/s'%(tot/1e3, 2*50331648/(tot/1e9)/1e9))
PY
done
