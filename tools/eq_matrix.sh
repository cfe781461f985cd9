#!/bin/bash
# 2x2: equalised split on/off x equalised tile kernel on/off, genome-like input
out=gpurun_out/${OUT:-eq_matrix}.log; rm -f $out
for split in 1 0; do for tiles in 1 0; do
  unset CAPS_SA_NO_EQUALISE CAPS_SA_NO_EQ_TILES
  [ $split = 0 ] && export CAPS_SA_NO_EQUALISE=1
  [ $tiles = 0 ] && export CAPS_SA_NO_EQ_TILES=1
  echo "== split=$split tiles=$tiles N=${N:-268435456} NBLOCKS=${NBLOCKS:-}" >> $out
  timeout -k 10 300 python tools/genome_like.py ${N:-268435456} 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done; done
