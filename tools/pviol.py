#!/usr/bin/env python3
"""Summarise a tools/smi_watch log: mean power / clock over the samples and PVIOL / TVIOL % between first and last sample.
    python3 tools/pviol.py LOG [skip_first_samples=1]"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{") and "error" not in l]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = rows[skip:] if len(rows) > skip + 1 else rows
if len(rows) < 2:
    print("    smi: too few samples")
    sys.exit(0)
a, b = rows[0], rows[-1]
acc = max(1, b["acc_counter"] - a["acc_counter"])
pw = sum(r["power_w"] for r in rows) / len(rows)
ck = sum(r["gfxclk_mhz"] for r in rows) / len(rows)
print("    smi: %d samples  power %.0f W (cap %.0f)  gfxclk %.0f MHz  hotspot %d C  PVIOL %.1f %%  TVIOL %.1f %%  HBM-thm %.1f %%  VR-thm %.1f %%  PROCHOT %.1f %%"
      % (len(rows), pw, a["power_cap_w"], ck, max(r["hotspot_c"] for r in rows), (b["ppt_acc"] - a["ppt_acc"]) * 100.0 / acc,
         (b["thm_acc"] - a["thm_acc"]) * 100.0 / acc, (b["hbm_thm_acc"] - a["hbm_thm_acc"]) * 100.0 / acc,
         (b["vr_thm_acc"] - a["vr_thm_acc"]) * 100.0 / acc, (b["prochot_acc"] - a["prochot_acc"]) * 100.0 / acc))
