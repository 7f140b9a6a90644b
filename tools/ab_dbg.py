"""A/B runs of bench.py under a kernel-variant selector: python tools/ab_dbg.py <key> <value> <bench.py args...>   (hs_debug_set(key, value))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsenet_amd import _lib as L
k, v = int(sys.argv[1]), int(sys.argv[2])
sys.argv = ["bench.py"] + sys.argv[3:]
L.load().hs_debug_set(k, v)
import bench
bench.main()
