#!/usr/bin/env python3
"""Race stress for the resident-side epochs: MANY epochs of `k1_res` (static binding at cfg 2 shuffled, ticket binding at cfg 3 / 4
user-major, the burst variant on Zipf-0.8 degrees) and whole Funk runs on `k2_res`, each against the level schedule of the same plan
(itself pinned to the oracle by the test suite) -- factors and biases bit for bit, the online RMSE to 1e-12.  A memory-ordering slip in the
hand-off protocol would show as a difference in some epoch; one run = ~1.5 G hand-offs.  usage: stress_res.py [rounds]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan
S.init(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
HP = (0.01, 0.1, 0.1, 0.007)


def compare(tag, idx, r, nu, ni, k, epoch, epochs, mode=0):
    outs = {}
    for e in ("levels", epoch):
        u, v = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
        ib, ub = np.zeros(ni), np.zeros(nu)
        with SgdPlan(idx, r, nu, ni, k, epoch=e) as p:
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(mode, epochs, *HP, want_rmse=True)
            p.get_factors(u, v, ib, ub)
            name = p.kernel_name(mode)
        outs[e] = (u, v, ib, ub, np.asarray(rm[0] if isinstance(rm, tuple) else rm))
    same = all(np.array_equal(x, y) for x, y in zip(outs["levels"][:4], outs[epoch][:4]))
    # (the online RMSE adds the squared errors in another order on the resident path: equal to rounding, not to the bit)
    drm = float(np.max(np.abs(outs["levels"][4] - outs[epoch][4])))
    same = same and drm < 1e-12
    print(json.dumps({"case": tag, "kernel": name, "epochs": epochs, "ratings": int(r.shape[0]), "bit_equal": bool(same),
                      "max_rmse_diff": drm}), flush=True)
    return same


ok = True
t0 = time.time()
for rd in range(rounds):
    nu, ni, nnz, k = 100000, 50000, 10000000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, t0=rd * nnz)
    ok &= compare("cfg2 shuffled / static", idx, r, nu, ni, k, "static", 60)
    ok &= compare("cfg2 shuffled / static logistic", idx, r, nu, ni, k, "static", 10, mode=1)
    um = synth.user_major(idx, r)
    ok &= compare("cfg2 user-major / ticket", um[0], um[1], nu, ni, k, "ticket", 30)
    rs = np.random.RandomState(rd)
    pi = 1.0 / np.arange(1, ni + 1) ** 0.8
    zi = np.stack([rs.randint(0, nu, 2000000), rs.choice(ni, 2000000, p=pi / pi.sum())], 1).astype(np.int32)
    zr = rs.randint(1, 6, 2000000).astype(np.float64)
    ok &= compare("zipf-0.8 / static + bursts", zi, zr, nu, ni, k, "static", 4)
    nu3, ni3, nnz3 = 1000000, 200000, 40000000
    idx3, r3 = synth.make_ratings(nu3, ni3, nnz3, t0=rd * nnz3)
    um3 = synth.user_major(idx3, r3)
    ok &= compare("cfg3-shaped user-major (40 M) / ticket", um3[0], um3[1], nu3, ni3, k, "ticket", 6)
    for kk in (18, 32, 100, 128):
        i2, r2 = synth.make_ratings(30000, 12000, 2000000, t0=rd * 7 + kk)
        ok &= compare("k=%d / static" % kk, i2, r2, 30000, 12000, kk, "static", 8)
print(json.dumps({"all_bit_equal": bool(ok), "seconds": round(time.time() - t0, 1)}))
sys.exit(0 if ok else 1)
