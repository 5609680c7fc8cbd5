import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
sys.argv = ["thin_probe.py", "40"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "thin_probe.py")).read())
