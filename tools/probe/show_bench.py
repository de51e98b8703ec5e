#!/usr/bin/env python3
"""print the headline fields of a bench.py line (file with the JSON line as its last line)"""
import json
import sys

txt = open(sys.argv[1]).read().strip()
try:
    d = json.loads(txt)                       # (a pretty-printed profile copy)
except json.JSONDecodeError:
    d = json.loads(txt.splitlines()[-1])      # (bench.py's stdout: the line is the last one)
r, ex = d["roofline"], d.get("extras", {})
print("value %.4g %s  ms/step %.4f  frac %.3f  traffic %s  committed_profile %s" % (d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("traffic"), (r.get("committed_profile") or {}).get("file")))
for k in ("n32768_batch4096", "n32768_batch8192"):
    if k in ex:
        print("  %s: %.4g pairs/s" % (k, ex[k]["pairs_per_s"]))
ps = ex.get("power_sustained") or {}
if ps.get("smu"):
    print("  power: %.0f W, sclk %.0f MHz, PVIOL %.1f %%, TVIOL %.1f %%; model: %s" % (ps["package_power_w_mean"], ps["sclk_mhz_mean"], ps["smu"]["pviol_pct"], ps["smu"]["tviol_pct"],
                                                                               json.dumps((ps.get("model") or {}).get("pairs_per_s_the_same_power_admits_at_valu_utilisation"))))
for k in ("config4_bfv", "bfv_reference_demo_16_primes"):
    if k in ex and "complete_drivers_including_keystream_and_samplers" in ex[k]:
        c = ex[k]["complete_drivers_including_keystream_and_samplers"]
        print("  %s complete drivers: keygen %.1f us, encrypt %.1f us, decrypt %.1f us; batch64 encrypt %.0f us, decrypt %.0f us" % (
            k, c["keygen_rns_us"], c["encryption_rns_us"], c["decryption_rns_us"], ex[k]["batch64"]["encrypt_us_per_call"], ex[k]["batch64"]["decrypt_us_per_call"]))
if d.get("cpu_baseline"):
    print("  cpu_baseline: %.0f %s on %s cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"]))
