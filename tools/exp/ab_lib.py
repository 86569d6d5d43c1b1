"""Run a tool against a VARIANT build of the library (experiments: one compile-time switch per variant .so under tools/exp/):
    python tools/exp/ab_lib.py tools/exp/libprio_variant.so tools/gemm_shapes_bench.py 2 4"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
