for tag in default norun tie8 enter16; do
  if [ $tag = default ]; then unset CAPS_SA_LIB; else export CAPS_SA_LIB=$PWD/caps-sa_amd/variants/libcaps_sa_hip_$tag.so; fi
  echo "== $tag" >> gpurun_out/s2c_genome_variants.log
  timeout -k 10 150 python tools/genome_like.py 268435456 >> gpurun_out/s2c_genome_variants.log 2>&1 || exit 1
done
