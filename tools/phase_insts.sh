#!/bin/bash
# Dynamic instruction counts per phase of the headline kernel: diagnostic builds that end the program after
# phase k (ab/libflacenc_exit<k>.so, -DFLACENC_EXIT_AFTER=k) under rocprofv3 --pmc; differences = per-phase cost.
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; rm -f $OUT/summary.txt
# the exit libraries must come from the sources of the library they are compared with (tools/build_exit_variants.sh)
want=$(cd $R && python3 -c "import bench; print(bench.kernel_source_sha())")
have=$(cat $R/ab/exit_variants.sha 2>/dev/null || echo none)
if [ "$want" != "$have" ]; then echo "phase_insts.sh: ab/libflacenc_exit*.so are stale ($have, sources are $want): run tools/build_exit_variants.sh" | tee $OUT/summary.txt; exit 1; fi
cd /tmp && export TMPDIR=/tmp
for lib in ab/libflacenc_exit0.so ab/libflacenc_exit1.so ab/libflacenc_exit2.so ab/libflacenc_exit3.so flacenc_rs_amd/libflacenc_hip.so; do
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
print('%-22s VALU/wave %7.1f  SALU/wave %6.1f  LDS/wave %6.1f  wave cycles %8.0f  wait_any %4.2f'%(sys.argv[2], m.get('SQ_INSTS_VALU',0)/w, m.get('SQ_INSTS_SALU',0)/w, m.get('SQ_INSTS_LDS',0)/w, 4*m.get('SQ_WAVE_CYCLES',0)/w, m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1)))
PY
done
