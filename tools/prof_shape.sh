#!/bin/bash
# rocprofv3 kernel-trace of stereo QLPC batch shapes: tools/prof_shape.sh <out-subdir> <flags> "<n> <order> [bps]" ...
R=$PWD; OUT=$R/gpurun_out/$1; FLAGS=$2; shift 2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do set -- $spec; n=$1; p=$2; bps=${3:-24}; fr=$((50331648 / n))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n${n}_p${p}_b${bps} -- python3 $R/tools/prof_config.py --n $n --order $p --bps $bps --frames $fr --flags $FLAGS > /dev/null 2>&1
  python3 - $OUT/n${n}_p${p}_b${bps} $n $p $bps $FLAGS <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
tot=0
print('== n=%s order=%s bps=%s flags=%s'%tuple(sys.argv[2:6]))
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    print('   %-46s calls %s avg %.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['TotalDurationNs'])/6
print('   total per call %.1f us -> %.1f G input samples/s'%(tot/1e3, 2*50331648/(tot/1e9)/1e9))
PY
done
