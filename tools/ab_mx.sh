# A/B timing of k_gemm_mx builds -- run on the GPU box.
# usage: bash tools/ab_mx.sh [X ...]   X = BASE or SNN_EXP_ suffixes joined by '+' (MX_NOBAR, MX_NOREADB, MX_NOSTAGE, MX_NOA:
# timing only, wrong results)
set -e
R=$PWD
[ $# -eq 0 ] && set -- BASE
for X in "$@"; do
  D=""
  if [ "$X" != "BASE" ]; then D="-DSNN_EXPERIMENTS"; for Y in ${X//+/ }; do D="$D -DSNN_EXP_$Y"; done; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I$R/include -I$R/snn_automotive_object_detection_amd/csrc $D -o /tmp/libexp.so $R/snn_automotive_object_detection_amd/csrc/snn_kernels.hip
  echo "== $X"; SNN_HIP_LIB=/tmp/libexp.so python tools/time_bf16x3.py 2>&1 | grep -E "mxfp6" | cut -c1-120
done
