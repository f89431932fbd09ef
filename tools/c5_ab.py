#!/usr/bin/env python3
"""A/B of the c2c tuning switches over transform sizes (run once per environment setting)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_configs import bench_c5
out = {}
for lg in (14, 17, 19, 20, 22, 24, 27, 28, 29, 30):
    r = bench_c5(lg)
    out[lg] = round(r["ms"], 4)
    assert r["kat_impulse_err"] < 1e-13 and r["kat_tone_err"] < 1e-13, r
    torch.cuda.empty_cache()
print(os.environ.get("SA_C2C_FAR"), os.environ.get("SA_C2C_FINAL_PP16"), json.dumps(out))
