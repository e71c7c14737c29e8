"""Tuning builds of liblrm_accel.so with a compile-time probe switched on (NOT the shipped library: results are
wrong by construction).  python tools/build_probe.py 1 2  ->  gpurun_out/probe_libs/liblrm_accel_vp1.so ...
Use with LRM_ACCEL_LIB=<path> python tools/seed_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import _build

out = os.path.join(_build.ROOT, "tools", "_probe_libs")
os.makedirs(out, exist_ok=True)
for v in sys.argv[1:]:
    if "=" in v:                       # any -D define, e.g. LRM_VOTE_T3_SLOTS=1024
        p = os.path.join(out, "liblrm_accel_%s.so" % v.replace("=", "_"))
        print(_build.build_accel(force=True, defines=[v], out=p))
    else:
        p = os.path.join(out, "liblrm_accel_vp%s.so" % v)
        print(_build.build_accel(force=True, defines=["LRM_VOTE_PROBE=" + v], out=p))
