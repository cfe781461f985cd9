"""One line per bench log (gpurun_out/<name>.json.log): ms per build, the kernel families' times, the count-free split's statistics and the
verifier's verdict -- the A/B runs behind DESIGN 5.3.  usage (on the GPU box): python3 tools/bench_compare.py <name> [<name> ...]"""
import json,sys
for f in sys.argv[1:]:
    d=None
    for l in open("gpurun_out/%s.json.log"%f):
        if l.startswith("{"): d=json.loads(l)
    if d is None: print(f,"no json"); continue
    print(f, round(d["ms_per_step"],2), {k:round(v["avg_launch_ms"],2) for k,v in d["roofline"]["kernels"].items()}, d["config"].get("knot_slot_splits"), "verify", d.get("verify_errors"))
