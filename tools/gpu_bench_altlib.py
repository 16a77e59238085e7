"""bench.py with an alternative build of the library: python tools/gpu_bench_altlib.py <path.so relative to the repo> [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
