import json,sys
for f in sys.argv[1:]:
    d=None
    for l in open("gpurun_out/%s.json.log"%f):
        if l.startswith("{"): d=json.loads(l)
    if d is None: print(f,"no json"); continue
    print(f, round(d["ms_per_step"],2), {k:round(v["avg_launch_ms"],2) for k,v in d["roofline"]["kernels"].items()}, d["config"].get("knot_slot_splits"), "verify", d.get("verify_errors"))
