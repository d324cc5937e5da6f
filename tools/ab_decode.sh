#!/bin/bash
# A/B of one environment switch on ONE box: tools/ab_decode.sh <VAR=value> <model> <tokens> [rounds] -> alternating runs of tools/decode_run.py
set -e
var="$1"; model="${2:-smollm-135m}"; n="${3:-256}"; rounds="${4:-3}"
for r in $(seq 1 "$rounds"); do
  echo -n "default      : "; python tools/decode_run.py "$model" "$n"
  echo -n "$var : "; env "$var" python tools/decode_run.py "$model" "$n"
done
