#!/usr/bin/env python3
"""Dev helper: the fields of a bench line worth a glance."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = lambda *ks: (lambda x: x)(__import__("functools").reduce(lambda a, k: a.get(k, {}) if isinstance(a, dict) else {}, ks, d))
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "frac", round(g("roofline", "frac"), 3), "step_frac", round(g("roofline", "step_frac"), 3), "error:", d.get("error"))
print("value_verified", d.get("value_verified"))
for k, v in (d.get("shapes") or {}).items():
    if isinstance(v, dict):
        print(" shape", k, "ms", round(v["ms_per_step"], 4), "frac", v.get("frac") and round(v["frac"], 3), "step_frac", round(v["step_frac"], 3), "verified", (v.get("value_verified") or {}).get("bitexact"))
ft = d.get("full_tick") or {}
for kind in ("noise", "scene"):
    if kind in ft:
        print(" full_tick", kind, round(ft[kind]["value"]), "ticks/s", "one_plan", round((ft[kind].get("one_plan") or {}).get("value", 0)), ft[kind].get("stages_ms"), "verified", (ft[kind].get("value_verified") or {}).get("bitexact"))
print("icp_iter_ms", d.get("icp_iter_ms"), "config2", d.get("icp_iter_ms_config2"), "refine", g("refine", "total_ms"))
print("legs with errors:", [k for k, v in d.items() if isinstance(v, dict) and "error" in v])
