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


# ---- the reference's 18-problem suite under -DREAL32 (oracle/_ref/lsqrtest32, test/lsqrtest.f90 unchanged) ----------
# -> real32_lstp_ref.json: per problem (m, n, nduplc, npower, damp), LSQR's istop and itn, the relative error in x the
# test prints and its verdict.  The fixture of tests/test_gpu_operator.py's REAL32 suite on the device operator.
import tempfile
exe18 = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "lsqrtest32")
with tempfile.TemporaryDirectory() as d:
    subprocess.run([exe18], cwd=d, capture_output=True, text=True, check=True)
    lis = open(os.path.join(d, "LSQR.LIS")).read()
probs = []
for blk in lis.split("Least-Squares Test Problem")[1:]:
    head = re.search(r"P\(\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([-+0-9.E]+)\s*\)", blk)
    ex = re.search(r"istop\s*=\s*(\d+)\s+itn\s*=\s*(\d+)", blk)
    er = re.search(r"appears to (be successful|have failed)\.\s+Relative error in  x  =\s*([-+0-9.E]+)", blk)
    cn = re.search(r"Condition no\. =\s*([-+0-9.E]+)\s+Residual function =\s*([-+0-9.E]+)", blk)
    ac = re.search(r"aprod seems (OK|incorrect)", blk)
    xi = re.search(r"inform\s*=\s*(\d+)", blk)
    probs.append(dict(m=int(head.group(1)), n=int(head.group(2)), nduplc=int(head.group(3)), npower=int(head.group(4)),
                      damp=float(head.group(5)), istop=int(ex.group(1)), itn=int(ex.group(2)),
                      success=er.group(1) == "be successful", enorm=float(er.group(2)),
                      acond=float(cn.group(1)), rnorm=float(cn.group(2)),
                      acheck_ok=ac.group(1) == "OK", xcheck_inform=int(xi.group(1))))
assert len(probs) == 18
json.dump(probs, open(os.path.join(HERE, "real32_lstp_ref.json"), "w"), indent=0)
print("REAL32 suite:", [(p["istop"], p["itn"], p["enorm"]) for p in probs])
