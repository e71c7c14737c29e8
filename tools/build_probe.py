"""Tuning builds of liblrm_accel.so with compile-time knobs (-D defines), next to the shipped library:
  python tools/build_probe.py LRM_VOTE_T3_SLOTS=768 LRM_VOTE_WAVES_PER_EU=5  ->  tools/_probe_libs/liblrm_accel_<define>.so
Use with LRM_ACCEL_LIB=<path> python tools/seed_probe.py.  (Round 2 also used throw-away probe builds that skipped the
SA gathers / the table inserts to see where the vote kernel's time goes; their outputs are under profiles/r2/probes/,
the switches were removed from the product source afterwards.)"""
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
        raise SystemExit("expected NAME=VALUE")
