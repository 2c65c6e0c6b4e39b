#!/usr/bin/env python3
"""bench.py with another build of the library (same-box A/B of a kernel change end to end):
python tools/bench_lib.py build/ab/libcmr_<tag>.so [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
