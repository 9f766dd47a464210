import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
prep = pkg.Optimizer.prepare(synth.ba_problem())
prep.solve()
pkg.abi.lib().dcs_dbg_ldlt_dump()
r = prep.solve()
print(r["n_trials"])
pkg.abi.lib().dcs_dbg_ldlt_dump()
