for ord in 8 10 12; do for rep in 1 2; do
  python tools/time_config.py --n 4096 --order $ord --bps 16 --frames 24576
  FLACENC_HIP_LIB=$PWD/ab/libflacenc_hip_o4_${ord}_1.so python tools/time_config.py --n 4096 --order $ord --bps 16 --frames 24576
done; done
