"""Where the decombine stage's host time goes (GPU box): tools/stage.py under cProfile, the library's and numpy's own time by function."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["stage.py", "--reads", "4000000"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "stage.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats("decombinator_amd|numpy|built-in", 40)
print(s.getvalue()[:9000])
