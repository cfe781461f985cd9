#!/bin/bash
# equalised vs plain count split on the genome-like input (GPU box): N elements, both modes
out=gpurun_out/${OUT:-eq_compare}.log; rm -f $out
for mode in eq plain; do
  if [ $mode = plain ]; then export CAPS_SA_NO_EQUALISE=1; else unset CAPS_SA_NO_EQUALISE; fi
  echo "== $mode N=${N:-268435456} NBLOCKS=${NBLOCKS:-}" >> $out
  timeout -k 10 300 python tools/genome_like.py ${N:-268435456} 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
