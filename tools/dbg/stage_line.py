import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]["stage_ms"]
print(sys.argv[1] if len(sys.argv) > 1 else "", d["ms_per_step"], {k: round(r[k] * 1e3) for k in ("scatter", "gate", "fuse", "post")})
