#!/bin/bash
# genome-like input (tools/genome_like.py) for the default library and every variant in caps-sa_amd/variants
out=gpurun_out/${OUT:-var_genome}.log
rm -f $out
for so in default caps-sa_amd/variants/libcaps_sa_hip_*.so; do
  if [ $so = default ]; then unset CAPS_SA_LIB; tag=default; else export CAPS_SA_LIB=$PWD/$so; tag=$(basename $so .so | sed 's/libcaps_sa_hip_//'); fi
  echo "== $tag" >> $out
  timeout -k 10 150 python tools/genome_like.py ${N:-268435456} 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
