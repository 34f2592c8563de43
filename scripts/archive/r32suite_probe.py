import sys, json, os
sys.path.insert(0, os.getcwd())
from lsqr_amd import operator as O
ref = json.load(open("tests/golden/real32_lstp_ref.json"))
res = O.run_suite(real32=True)
for r, g in zip(res, ref):
    print(f"P({r['m']},{r['n']},{r['npower']}) gpu istop {r['istop']} itn {r['itn']} enorm {r['enorm']:.3e} | ref istop {g['istop']} itn {g['itn']} enorm {g['enorm']:.3e}")
# a user operator in real32 through the Python callback path
