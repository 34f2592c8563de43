#!/usr/bin/env python3
"""Fixtures for the reference's own 18-problem suite (test/lsqrtest_module.f90:55-94).

Run in the build container (needs /root/reference and oracle/_ref/lsqrtest built by
`make -C oracle`):

  LSQR_ref_amdflang.LIS   the log the UNMODIFIED reference test program writes when compiled
                          here (amdflang -O2 -ffp-contract=off) -- output data of the reference
  LSQR_shipped_facts.json the numeric fields of the log the reference repository ships
                          (test/LSQR.LIS, another compiler): istop, itn, exit scalars, xcheck
                          inform, x(1:8), verdict and relative error per problem
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

exe = os.path.join(ROOT, "oracle", "_ref", "lsqrtest")
if not os.path.exists(exe):
    raise SystemExit("oracle/_ref/lsqrtest missing: run `make -C oracle` where /root/reference exists")
with tempfile.TemporaryDirectory() as d:
    subprocess.run([exe], cwd=d, check=True)
    text = open(os.path.join(d, "LSQR.LIS")).read()
open(os.path.join(HERE, "LSQR_ref_amdflang.LIS"), "w").write(text)
shipped = oracle.parse_lis(open("/root/reference/test/LSQR.LIS").read())
json.dump(shipped, open(os.path.join(HERE, "LSQR_shipped_facts.json"), "w"), indent=1)
print(len(oracle.parse_lis(text)), "problems (compiled here),", len(shipped), "problems (shipped log)")
