#!/bin/bash
# rocprofv3 kernel-trace of the frame-level pipeline (encode_stereo_frames + pack) for one shape:
#   tools/prof_frames.sh <out-subdir under gpurun_out> <n> <order> <bps> <frames> [--use-fixed]
R=$PWD; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
N=$1; P=$2; B=$3; F=$4; shift 4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/time_frames.py --n $N --order $P --bps $B --frames $F --reps 6 "$@" > $OUT/log.txt 2>&1
cat $OUT/log.txt | grep median
python3 - $OUT/t <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    print('   %-52s calls %4s avg %9.1f us  total %9.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
