"""Generates tests/golden/histo_golden.json from the reference's histo.c compiled into
oracle/_ref/libref_histo.so (run in the build container: `make -C oracle && python
tests/golden/make_histo_golden.py`).  The key streams come from test_histo_ref._streams()."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
import test_histo_ref as t  # noqa: E402

lib = orc.ref_histo_lib()
assert lib is not None, "build oracle/_ref first (needs /root/reference)"
res = [t._run_ref(lib, keys) for keys in t._streams()]
with open(os.path.join(HERE, "histo_golden.json"), "w") as f:
    json.dump({"source": "reference histo/histo.c compiled unmodified (oracle/_ref/libref_histo.so)",
               "results": res}, f)
print("wrote", len(res), "cases")
