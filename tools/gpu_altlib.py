"""Per-kernel times with an alternative build of the library: python tools/gpu_altlib.py <path.so> B..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(ROOT, 'tools', 'gpu_fixed_cost.py')).read())
