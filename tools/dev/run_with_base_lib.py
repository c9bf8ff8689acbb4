import sys
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import _lib as L
L.LIB_PATH = "/root/repo/tools/_bin/libdvt_hip_base.so"
import pytest
sys.exit(pytest.main(["tests/test_gpu_vivit.py", "-x", "-q", "-k", "trajectory"]))
