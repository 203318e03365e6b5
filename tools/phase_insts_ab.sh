#!/bin/bash
# tools/phase_insts_ab.sh OUT lib1.so lib2.so ...: VALU / SALU / LDS instructions per wave of the headline launch for each library
# (diagnostic builds -DFLACENC_EXIT_AFTER=k end the program after phase k); FLACENC_FLAGS = flacenc_hip_qlpc_config.flags
R=$PWD; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT; rm -f $OUT/summary.txt; cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  FLACENC_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$tag -- python3 $R/tools/launch_headline.py > $OUT/$tag.log 2>&1
  python3 - $OUT/$tag $tag <<'PY' | tee -a $OUT/summary.txt
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'qlpc_wave4096' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
w=m.get('SQ_WAVES',1)
print('%-28s VALU/wave %7.1f  SALU/wave %6.1f  LDS/wave %6.1f  wave cycles %8.0f  wait_any %4.2f'%(sys.argv[2], m.get('SQ_INSTS_VALU',0)/w, m.get('SQ_INSTS_SALU',0)/w, m.get('SQ_INSTS_LDS',0)/w, 4*m.get('SQ_WAVE_CYCLES',0)/w, m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1)))
PY
done
