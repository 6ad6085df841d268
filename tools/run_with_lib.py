"""Run bench.py against an alternative build of the library (experiments): python tools/run_with_lib.py <lib.so> [bench args]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
