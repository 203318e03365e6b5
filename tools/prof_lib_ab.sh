#!/bin/bash
# kernel-trace of one big-block candidate-batch shape for several library builds:
#   tools/prof_lib_ab.sh <out-subdir> "<n> <order>" lib1.so lib2.so ...   (kernels above 30 us are listed)
R=$PWD; OUT=$R/gpurun_out/$1; SPEC=$2; shift 2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
set -- $SPEC "$@"; N=$1; P=$2; shift 2
for lib in "$@"; do
  tag=$(basename $lib .so); fr=$((50331648 / N))
  FLACENC_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/tools/prof_config.py --n $N --order $P --frames $fr > /dev/null 2>&1
  python3 - $OUT/$tag $tag <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
print('==', sys.argv[2])
for r in csv.DictReader(open(f)):
    m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Name']); nm=(m.group(1)+(m.group(2) or '')) if m else r['Name'][:44]
    if float(r['AverageNs'])>30e3: print('   %-46s calls %s avg %.1f us'%(nm, r['Calls'], float(r['AverageNs'])/1e3))
PY
done
