import io, oracle, numpy as np
from lsqr_amd.operator import SUITE, run_suite
res = run_suite(io.StringIO()); po = oracle.port()
for k,(c,r) in enumerate(zip(SUITE,res)):
    o = po.lstp_test(*c)
    print(k, c[:2], c[3], "itn gpu/oracle", r["itn"], o["itn"], "enorm %.2e %.2e"%(r["enorm"], o["enorm"]), "xinf", r["xcheck_inform"], o["xcheck_inform"], "|dx| %.2e"%np.linalg.norm(r["x"]-o["x"]), "anorm %.4f %.4f"%(r["anorm"],o["anorm"]))
