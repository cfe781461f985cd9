#!/bin/bash
# Benchmarks every tuning variant in caps-sa_amd/variants on the C2 workload (GPU box).
mkdir -p gpurun_out
for so in caps-sa_amd/variants/libcaps_sa_hip_*.so; do
  tag=$(basename $so .so | sed 's/libcaps_sa_hip_//')
  CAPS_SA_LIB=$PWD/$so timeout 300 python bench.py --workload ${WL:-c2} --steps 3 --warmup 1 --no-cpu-baseline --verify > gpurun_out/var_$tag.log 2>&1
  python - "$tag" gpurun_out/var_$tag.log <<'PY'
import json,sys
tag,path=sys.argv[1:3]
line=[l for l in open(path) if l.startswith('{')]
if not line: print(tag,'FAILED'); sys.exit(0)
d=json.loads(line[-1]); p=d['phases_ms']
print(f"{tag}: {d['ms_per_step']:8.2f} ms  tile_sort={p['tile_sort_ms']:7.2f} merge={p['merge_pass_ms']:7.2f} passes={d['config']['merge_passes']} launches={d['roofline']['launches_per_step']} sortsub={p['ms_sort_subarrays']:.1f} part={p['ms_partition']:.1f} mergeparts={p['ms_merge_partitions']:.1f} verify={d.get('verify_errors')}")
PY
done
