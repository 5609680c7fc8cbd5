"""bench.py on another build of the library (same-box A/B of two builds): python tools/bench_lib.py <file under hpfg_amd/lib/> [bench.py arguments]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
