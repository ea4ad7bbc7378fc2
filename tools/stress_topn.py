"""Diagnostic: the golden fixture's small top-N queries in a loop, every result compared; stops at the first mismatch."""
import os, sys, ctypes as C
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import numpy as np
from mfrec_amd import _shim as S
S.init(0)
g = np.load(os.path.join(root, "tests/golden/g12_topn.npz"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = {}
for tag in "ab":
    u, v = np.ascontiguousarray(g["in_%s_u" % tag]), np.ascontiguousarray(g["in_%s_v" % tag])
    ib, ub = np.ascontiguousarray(g["in_%s_ib" % tag]), np.ascontiguousarray(g["in_%s_ub" % tag])
    roff, rated, nu, ni = g["in_%s_rated_off" % tag], g["in_%s_rated" % tag], int(g["p_%s_nu" % tag]), int(g["p_%s_ni" % tag])
    users = np.asarray(g["in_%s_users" % tag], dtype=np.int32)
    v_sel = np.ascontiguousarray(v[:, users]); ub_sel = np.ascontiguousarray(ub[users])
    lists = [rated[roff[x]:roff[x + 1]] if x < nu else np.zeros(0, np.int32) for x in users]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    flat = np.ascontiguousarray(np.concatenate(lists + [np.zeros(1, np.int32)]), dtype=np.int32)
    L[tag] = (u, v_sel, ib, ub_sel, float(g["p_%s_mu" % tag]), ni, users, off, flat, g["p_%s_cases" % tag].tolist())
def run(tag, kind, ci):
    u, v_sel, ib, ub_sel, mu, ni, users, off, flat, cases = L[tag]
    nc, N = int(cases[ci][0]), int(cases[ci][1])
    items = np.full((users.shape[0], N), -1, dtype=np.int32); scores = np.zeros((users.shape[0], N))
    bias = kind in (1, 2, 3)
    S.check(S.lib().mfx_topn_batch(kind, u.shape[0], S.f64(u, "u", 2), S.f64(v_sel, "v", 2), S.f64(ib, "ib", 1) if bias else None,
                                   S.f64(ub_sel, "ub", 1) if bias else None, mu, ni, S.i32(users, "users", 1), users.shape[0], nc, N,
                                   off.ctypes.data_as(S._lp), S.i32(flat, "rated", 1), S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
    return items, scores
n = bad = 0
for rep in range(reps):
    for tag in "ab":
        for kind in (0, 1, 2, 3, 4):
            for ci in (0, 2):
                it, sc = run(tag, kind, ci)
                want = g["out_%s_items_kind%d_case%d" % (tag, kind, ci)]
                n += 1
                if not np.array_equal(it, want):
                    bad += 1
                    rows = np.where((it != want).any(1))[0]
                    print("MISMATCH call %d rep %d tag %s kind %d case %d rows %s got %s want %s" % (n, rep, tag, kind, ci, rows.tolist(), it[rows].tolist(), want[rows].tolist()), flush=True)
                    if bad >= 3:
                        print("calls", n, "mismatches", bad); sys.exit(1)
print("calls", n, "mismatches", bad)
