mkdir -p gpurun_out/r5
run() { echo "TUNE=$1"; ZGML_QMV_KON_TUNE="$1" python tools/decode_run.py llama2-7b 128; }
base="1376x4096:4:2,768x4096:4:4,256x4096:4:4,256x11008:8:4,2000x4096:4:4"
run "$base"
for v in "1376x4096:4:4" "1376x4096:8:2" "768x4096:4:8" "768x4096:8:4" "256x4096:4:8" "256x11008:8:8" "256x11008:12:4" "256x11008:6:8" "2000x4096:4:8"; do
  key=${v%%:*}; t=$(echo "$base" | sed "s/$key:[0-9]*:[0-9]*/$v/"); run "$t"
done
