"""compact table of tools/probe_r04.py's JSON lines (stdin)"""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("##"):
        print(line)
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    keys = [k for k in d if k.endswith("_us")]
    head = " ".join(f"{k}={d[k]}" for k in d if not k.endswith("_us") and not k.endswith("TFs") and k != "gflop")
    print(head, "|", " ".join(f"{k[:-3]}={d[k]:.0f}" for k in keys))
