import cProfile, pstats, sys, os
sys.argv = ["bench.py", "--config", "c2", "--steps", "400", "--warmup", "30", "--no-cpu-baseline", "--no-extras"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
