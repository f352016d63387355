"""Diagnostic: run bench.py's single-GPU measurement against another build of the library (timing experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench
bench.main()
