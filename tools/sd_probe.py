"""Seed-table probe: bench.py's serialized kernel times with the seed table on / off (and its geometry variants).
usage: python tools/sd_probe.py [extra bench.py args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("LRM_SD=1", {"LRM_SD": "1"}), ("LRM_SD=0", {"LRM_SD": "0"}), ("LRM_SD=1 LRM_SD_SHARE=2", {"LRM_SD": "1", "LRM_SD_SHARE": "2"})]
for tag, env in variants:
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-pcie", "--no-grch38"] + sys.argv[1:],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(tag, "rc", r.returncode, flush=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        iso = {k: round(v["avg_ms"], 3) for k, v in j["isolated"]["kernels"].items() if k in ("seed_search_kernel", "vote_kernel", "gact_bs_kernel")}
        print("  value", round(j["value"], 2), "ms/step", round(j["ms_per_step"], 2), iso, j["index"].get("tables"), flush=True)
    except Exception as ex:
        print("  no JSON:", ex, r.stderr[-2000:], flush=True)
