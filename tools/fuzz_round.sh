#!/bin/bash
# A fuzz campaign in one gpurun call: tools/fuzz_round.sh <out-subdir> <seed0> <frames> <channel> <extreme> <xcand>
O=gpurun_out/$1; S=$2; mkdir -p $O
python tools/fuzz_more.py $S $((S + $3)) > $O/frame.txt 2>&1
FUZZ_KIND=channel python tools/fuzz_more.py $S $((S + $4)) > $O/channel.txt 2>&1
FUZZ_KIND=extreme python tools/fuzz_more.py $S $((S + $5)) > $O/extreme.txt 2>&1
FUZZ_KIND=xcand python tools/fuzz_more.py $S $((S + $6)) > $O/xcand.txt 2>&1
FLACENC_FUZZ_ORDER=reference python tools/fuzz_more.py $S $((S + $3 / 2)) > $O/frame_reference.txt 2>&1
FLACENC_FUZZ_ORDER=nightly python tools/fuzz_more.py $S $((S + $3 / 3)) > $O/frame_nightly.txt 2>&1
FUZZ_FINEST=1 python tools/fuzz_more.py $S $((S + $3 / 3)) > $O/frame_finest.txt 2>&1
tail -n 2 $O/*.txt
