"""Dev: run bench.py against another build of the library (A/B on one device):  bench_with_lib.py <lib.so> [bench flags]"""
import sys
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import _lib as L
L.LIB_PATH = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
