#!/bin/bash
# same-box A/B of library variants (tools/build_variant.py): sustained ms/launch, package power, sclk
# usage: tools/ab_variants.sh SECONDS name1 name2 ...   ("stock" = the in-tree library)
secs=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = stock ]; then unset AVLMAPS_HIP_LIB; else export AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_$v.so; fi
  echo "== $v"
  PP_KERNELS=${PP_KERNELS:-split_f16,prepared} timeout 120 python tools/power_probe.py $secs 2>&1 | grep "ms/launch"
done
done
