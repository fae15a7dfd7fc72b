"""Run bench.py against another build of the library (same-box A/B):  python tools/bench_with_lib.py <lib.so> [bench.py flags]"""
import os, runpy, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
from packnet_sfm.hip import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
