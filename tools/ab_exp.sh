# A/B timing of experimental builds (wrong results, timing only) -- run on the GPU box
# usage: bash tools/ab_exp.sh [X ...]   with X in BASE NO_GLDS NO_STORE_A NO_BARRIER (env such as SNN_BF16X3_WN passes through)
set -e
R=$PWD
[ $# -eq 0 ] && set -- BASE NO_STORE_A NO_GLDS NO_BARRIER
for X in "$@"; do
  D=""; [ "$X" != "BASE" ] && D="-DSNN_EXPERIMENTS -DSNN_EXP_$X"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I$R/include -I$R/snn_automotive_object_detection_amd/csrc $D -o /tmp/libexp.so $R/snn_automotive_object_detection_amd/csrc/snn_kernels.hip
  echo "== $X"; SNN_HIP_LIB=/tmp/libexp.so python tools/time_bf16x3.py 2>&1 | grep -E "^fc6|FUSED" | sed 's/(un-fused.*//' | cut -c1-110
done
