#!/usr/bin/env python3
"""Fixture for the REAL32 build: what the unmodified reference, compiled with -DREAL32
(src/lsqr_kinds.F90:16-17), returns on the two systems of lsqr_amd/fortran/tests/test_real32.f90
(driver: oracle/ref32_driver.f90, built by `make -C oracle` in the build container)."""
import json, os, re, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref32_driver")
out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
NUM = r"[-+]?\d\.\d+E[-+]\d+"
rec = {}
for l in out.splitlines():
    key = l.split("=")[0].strip().split(" istop")[0]
    nums = [float(t) for t in re.findall(NUM, l)]
    ints = [int(t) for t in re.findall(r"=\s*(\d+)(?!\.)", l)]
    rec[l.split("=")[0].strip()] = dict(ints=ints, nums=nums)
json.dump(rec, open(os.path.join(HERE, "real32_ref.json"), "w"))
print({k: (v["ints"], len(v["nums"])) for k, v in rec.items()})
