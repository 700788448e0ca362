"""Extract the Hydra surface of the sampling path from the reference's YAML tree into DATA:
tests/golden/hydra_targets.json = for each of cmd/conf/{sampler,score_model,score_model/noise_scheduler}/*.yaml
its ``_target_`` string, ``_partial_`` flag and keyword arguments (interpolations such as ``${fourier_transform}``
are recorded as the name they refer to, under "interpolated").  TEST INFRASTRUCTURE; run in the build container:
    python oracle/gen_hydra_targets.py
The reference pins this surface with tests/test_hydra_configs.py:21-51 (compose + instantiate every config)."""
import json
import os
import re

import yaml

REF = "/root/reference/cmd/conf"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hydra_targets.json")
GROUPS = ["sampler", "score_model", "score_model/noise_scheduler"]

out = {}
for grp in GROUPS:
    d = os.path.join(REF, grp)
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".yaml"):
            continue
        cfg = yaml.safe_load(open(os.path.join(d, fn)))
        entry = {"_target_": cfg.pop("_target_"), "_partial_": bool(cfg.pop("_partial_", False)), "kwargs": {},
                 "interpolated": {}, "defaults": cfg.pop("defaults", [])}
        for k, v in cfg.items():
            m = re.fullmatch(r"\$\{(.+)\}", v) if isinstance(v, str) else None
            if m:
                entry["interpolated"][k] = m.group(1)
            else:
                entry["kwargs"][k] = v
        out[f"{grp}/{fn[:-5]}"] = entry
sample = yaml.safe_load(open(os.path.join(REF, "sample.yaml")))
out["sample"] = {k: v for k, v in sample.items() if k not in ("defaults", "model_path")}
json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
