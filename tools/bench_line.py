"""Prints the few numbers of a bench.py JSON line (stdin) that A/B scripts compare."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d.get("stereo") or {}
print(d["ms_per_step"], d["roofline"]["launch_ms"], d["jac_passes"], d["err_passes"],
      {k: v["ms_per_step"] for k, v in st.items() if isinstance(v, dict)})
