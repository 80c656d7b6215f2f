#!/bin/bash
# build A/B variants of libsnnhip.so into gpurun_out/ab/ (run HERE: hipcc cross-compiles; the .so files travel with the tree? no:
# gpurun_out/ is not sent - so build into tools/_ab/ which is git-ignored but travels)
# usage: bash tools/ab_build.sh NAME1:"-DFLAG1 -DFLAG2" NAME2:"" ...
R=$PWD; mkdir -p tools/_ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  case "$flags" in *SNN_EXP_*) flags="-DSNN_EXPERIMENTS $flags";; esac      # timing-only switches: wrong results by construction
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I$R/include -I$R/snn_automotive_object_detection_amd/csrc $flags \
      -o tools/_ab/lib_$name.so $R/snn_automotive_object_detection_amd/csrc/snn_kernels.hip && echo "built $name [$flags]" ) &
done
wait
