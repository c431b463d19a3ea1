import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]
out = {"cfg2": d["ms_per_step"]}
c3 = c.get("cfg3")
if c3:
    out["cfg3"] = c3["ms_per_step"]; out["cfg3_rays"] = round(c3["stage_ms"].get("rays", 0) * 1e3, 1); out["cold"] = c3.get("cold_start_ms")
    t = c3.get("terrain")
    if t: out["terrain"] = t["ms_per_step"]; out["terrain_rays"] = round(t["stage_ms"].get("rays", 0) * 1e3, 1); out["terrain_cold"] = t.get("cold_start_ms")
for k in ("cfg4", "cfg5"):
    if k in c: out[k] = c[k]["ms_per_step"]; out[k + "_rays"] = round(c[k]["stage_ms"].get("rays", 0) * 1e3, 1)
c1 = c.get("cfg1")
if c1: out["cfg1"] = {k: v["ms_per_step"] for k, v in c1.items() if isinstance(v, dict)}
print(sys.argv[1], json.dumps(out))
