"""One-line summary of tools/kbench.py's JSON lines (stdin): mean microseconds per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import json, sys
r = {}
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        r[d["case"]] = d
keys = ("euler_t30", "euler_t59", "incoherent", "synth_frame_t1", "synth_frame_t30", "synth_frame_t59")
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join(f"{k}:{(r[k].get('splat_us') or r[k]['us'])[0]:.0f}" for k in keys if k in r))
