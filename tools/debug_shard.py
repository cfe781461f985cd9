import faulthandler, sys, runpy
faulthandler.dump_traceback_later(45, exit=True)
sys.argv = ["bench.py", "--workload", "c2", "--steps", "1", "--warmup", "1", "--verify", "--no-cpu-baseline"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
