"""Dev: bench.py with a functional.py feature flag overridden:  bench_flag.py NAME=0|1 [bench flags]"""
import sys
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import functional as F
name, val = sys.argv[1].split("=")
setattr(F, name, bool(int(val)))
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
