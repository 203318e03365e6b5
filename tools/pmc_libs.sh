#!/bin/bash
# tools/pmc_libs.sh OUT lib1.so lib2.so ...: one counter pass (PMC="..." counters, default the instruction cache's) of the
# headline launch (tools/launch_headline.py) per library; per-dispatch means of the wave kernel, one line per library
R=$PWD; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
PMC=${PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU}
for lib in "$@"; do
  tag=$(basename $lib .so)
  FLACENC_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$tag -- python3 $R/tools/launch_headline.py > $OUT/$tag.log 2>&1
  python3 - $OUT/$tag $tag <<'PY' | tee -a $OUT/summary.txt
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'qlpc_wave4096' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('%-28s'%sys.argv[2], '  '.join('%s %.4g'%(k, sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
done
