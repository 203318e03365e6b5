for spec in "8::o4_8_2" "8:--use-fixed:o4_8_3" "10::o4_10_2" "10:--use-fixed:o4_10_3" "12::o4_12_2" "12:--use-fixed:o4_12_3"; do
  IFS=: read ord fx lib <<< "$spec"
  echo "== order $ord $fx"
  AB_FLAGS="--lpc-order $ord $fx --frames 24576" bash tools/ab_bench.sh r05_ab15 2 flacenc_rs_amd/libflacenc_hip.so ab/libflacenc_hip_$lib.so
done
