"""Print the headline fields of a bench.py JSON line read from stdin (helper for shell loops)."""
import json
import sys

tag = " ".join(sys.argv[1:])
d = json.loads(sys.stdin.read())
r = d.get("roofline") or {}
print(tag, d["value"], "snippets/s", d["ms_per_step"], "ms", "gemm", r.get("achieved"), "TF", r.get("gemm_ms_per_step"), "ms",
      "pack", r.get("pack_ms_per_step"), "loss", round(d["loss_first_timed_step"], 4), round(d["loss_last_timed_step"], 4), flush=True)
