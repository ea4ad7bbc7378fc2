#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box, never imported by the product or the tests).

Pins the top-N / precision-recall RESTATEMENT (oracle/oracle.py: find_recommended_items, precision_recall) -- and with it the golden
tests/golden/g12_topn.npz that the GPU tests compare mfx_topn_batch / mfx_topn_plan_query with -- against the reference's OWN
source.  The reference's Python-2 harness cannot be imported (print statements, `from pymongo import Connection`, sparsesvd:
SURVEY 8(c)), but the functions on this path are self-contained methods.  This script

  1. reads the reference files as text where they lie and cuts out exactly these definitions:
       mf.py                find_recommended_items                       (:144-193)
       gradient_descent.py  find_user_top_match, predict_rating, predict_rating_with_bias   (:768-800, :621-631, :637-648)
       kmf.py               predict_logistic, predict_linear             (:79-94)
       wrmf.py              predict                                      (:67-69)
       base.py              class Error, get_items_subset                (:23, :1134-1145)
       metrics.py           precision_recall                             (:85-130)
  2. translates that text with the standard library's lib2to3 (print statement, dict.iteritems: nothing else applies), 
  3. executes the translated definitions as methods of a bare object whose attributes are the fixture's inputs (svd_u, svd_v, biases,
     the ratings matrix, neighborhood, ...) -- no stand-in for any library: numpy, scipy.sparse.find and operator.itemgetter are the
     real ones, `Error` is the reference's own class,
  4. runs every query of g12 (5 predictor families x 3 (candidates, N) cases x 9 users x 2 models, + precision / recall) and compares
     with the arrays stored in the golden file: item lists exactly, scores to 1e-12 (np.dot's association vs the sequential sum).

Nothing of the reference's text is written anywhere: the record tests/golden/g12_topn_pin.json holds the file names, line ranges, a
SHA-256 of the cut text and the counts.  A mismatch exits non-zero (oracle/make_golden.py calls this and refuses to go on)."""
import hashlib
import json
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("MFX_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
CUTS = (("base.py", "class Error", 0), ("base.py", "def get_items_subset", 4), ("mf.py", "def find_recommended_items", 4),
        ("gradient_descent.py", "def find_user_top_match", 4), ("gradient_descent.py", "def predict_rating", 4),
        ("gradient_descent.py", "def predict_rating_with_bias", 4), ("kmf.py", "def predict_logistic", 4),
        ("kmf.py", "def predict_linear", 4), ("wrmf.py", "def predict", 4), ("metrics.py", "def precision_recall", 0))


def cut(fname, head, indent):
    """The block that starts at the line `indent` spaces + head + '(' or ':' and runs to the next line of equal or smaller indentation."""
    lines = open(os.path.join(REF, "mfrec", "recommendation", fname)).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith(" " * indent + head) and l[indent + len(head):indent + len(head) + 1] in ("(", ":"):
            start = i
            break
    if start is None:
        raise SystemExit("pin_topn: %s not found in %s" % (head, fname))
    end = start + 1
    while end < len(lines):
        l = lines[end]
        if l.strip() and (len(l) - len(l.lstrip())) <= indent:
            break
        end += 1
    while not lines[end - 1].strip():
        end -= 1
    return "\n".join(lines[start:end]) + "\n", (start + 1, end)


def translate(src):
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    return str(tool.refactor_string(src, "<reference>"))


def _check_cut_text(pieces, record_name):
    """Nothing cut from the (untrusted) reference tree is executed unless the SHA-256 of ALL the cut text equals the one in the
    committed record -- i.e. unless it is the very text that was reviewed when the record was made.  `--repin` (after reading the
    text) accepts a new hash."""
    sha = hashlib.sha256()
    for src in pieces:
        sha.update(src.encode())
    path = os.path.join(GOLD, record_name)
    want = json.load(open(path)).get("sha256_of_the_cut_text") if os.path.exists(path) else None
    if want != sha.hexdigest() and "--repin" not in sys.argv:
        raise SystemExit("pin_topn: the text cut from %s hashes to %s, the committed record %s says %s -- nothing was executed; "
                         "read the definitions, then run with --repin" % (REF, sha.hexdigest(), record_name, want))
    return sha


def main():
    if not os.path.isdir(os.path.join(REF, "mfrec")):
        raise SystemExit("pin_topn: the reference is not present (build container only)")
    from operator import itemgetter
    from scipy.sparse import find, lil_matrix
    from oracle import oracle as O
    ns = {"np": np, "find": find, "itemgetter": itemgetter}
    record = []
    fns = {}
    cuts = [(fname, head) + cut(fname, head, indent) for fname, head, indent in CUTS]
    sha = _check_cut_text([c[2] for c in cuts], "g12_topn_pin.json")      # before any of it runs
    for fname, head, src, (a, b) in cuts:
        record.append({"file": "mfrec/recommendation/" + fname, "definition": head, "lines": "%d-%d" % (a, b)})
        code = translate(textwrap.dedent(src))
        before = set(ns)
        exec(compile(code, "<%s:%d>" % (fname, a), "exec"), ns)
        new = [n for n in ns if n not in before and n != "__builtins__"]
        fns[(fname, head)] = ns[new[0]] if new else None
        if fname == "wrmf.py":
            ns["wrmf_predict"] = ns.pop("predict")

    class Harness(object):        # a bare attribute holder: every METHOD below is the reference's own (translated) function
        pass
    for name in ("get_items_subset", "find_recommended_items", "find_user_top_match", "predict_rating", "predict_rating_with_bias",
                 "predict_logistic", "predict_linear"):
        setattr(Harness, name, ns[name])
    Harness.wrmf_predict = ns["wrmf_predict"]
    kinds = {O.RMSE_GD: "predict_rating", O.RMSE_KMF_LINEAR: "predict_linear", O.RMSE_KMF_LOGISTIC: "predict_logistic",
             O.RMSE_GD_BIAS: "predict_rating_with_bias", O.RMSE_WRMF: "wrmf_predict"}

    g = np.load(os.path.join(GOLD, "g12_topn.npz"))
    n_queries = n_pr = 0
    for tag in ("a", "b"):
        nu, ni = int(g["p_%s_nu" % tag]), int(g["p_%s_ni" % tag])
        off, flat, users = g["in_%s_rated_off" % tag], g["in_%s_rated" % tag], g["in_%s_users" % tag]
        M = lil_matrix((nu + 1, ni))                             # users x items (base.py:273); user `nu` = folded in, no ratings
        for usr in range(nu):
            for it in flat[off[usr]:off[usr + 1]]:
                M[usr, int(it)] = 1.0
        h = Harness()
        h.svd_u, h.svd_v = g["in_%s_u" % tag], g["in_%s_v" % tag]
        h.items_bias, h.users_bias, h.overall_bias = g["in_%s_ib" % tag], g["in_%s_ub" % tag], float(g["p_%s_mu" % tag])
        h.min_rating, h.max_rating = 1.0, 5.0                    # base.py:93-94
        h.nbr_items, h.nbr_users = ni, nu + 1
        h.relationship_matrix = M
        h.relationship_matrix_csc = M.T.tocsc()                  # what KMF / WRMF train() and find_user_top_match leave (kmf.py:207, wrmf.py:88, gradient_descent.py:777)
        h.users_index, h.items_label = {}, []
        for kind, pname in kinds.items():
            for ci, (n_cand, N) in enumerate(g["p_%s_cases" % tag]):
                want_i, want_s = g["out_%s_items_kind%d_case%d" % (tag, kind, ci)], g["out_%s_scores_kind%d_case%d" % (tag, kind, ci)]
                for j, usr in enumerate(users):
                    h.neighborhood = int(n_cand)
                    items, scores = h.find_recommended_items(user_index=int(usr), nbr_recommendations=int(N), predictor=pname)
                    n = len(items)
                    assert list(want_i[j, :n]) == items and (n == want_i.shape[1] or want_i[j, n] == -1), (tag, kind, ci, usr, items, want_i[j])
                    assert np.allclose(want_s[j, :n], scores, rtol=1e-12, atol=0), (tag, kind, ci, usr)
                    n_queries += 1
                    if kind == O.RMSE_GD and int(n_cand) == ni:
                        # GDRecommender.find_user_top_match (gradient_descent.py:768-800) is the same loop with every item a candidate
                        # and WITHOUT the NaN -> 0 step (mf.py:176): compared on a copy of the model whose NaN column is zeroed
                        keep = h.svd_u
                        h.svd_u = np.where(np.isnan(keep), 0.0, keep)
                        it1, sc1 = h.find_recommended_items(user_index=int(usr), nbr_recommendations=int(N), predictor=pname)
                        it2, sc2 = h.find_user_top_match(int(usr), nbr_recommendations=int(N))
                        h.relationship_matrix_csc = M.T.tocsc()
                        h.svd_u = keep
                        assert it2 == it1 and np.array_equal(sc2, sc1), (tag, usr)
        # precision / recall (metrics.py:85-130), GD predictor, every item a candidate
        h.neighborhood = ni
        h.relationship_matrix_csc = M.T.tocsc()
        p, r, f = ns["precision_recall"](h, g["in_%s_test" % tag], nbr_recommendations=5, predictor="predict_rating")
        assert np.allclose([p, r, f], g["out_%s_prf" % tag], rtol=1e-12), (tag, p, r, f, g["out_%s_prf" % tag])
        n_pr += 1
    out = {"pinned_by": "the reference's own source for these definitions, cut from the files where they lie, translated by lib2to3 "
                        "(print statement, dict.iteritems) and executed under Python %d.%d as methods of a bare attribute holder; "
                        "numpy / scipy.sparse.find / operator.itemgetter are the real ones" % sys.version_info[:2],
           "definitions": record, "sha256_of_the_cut_text": sha.hexdigest(),
           "queries_compared": n_queries, "precision_recall_runs_compared": n_pr,
           "tolerance": "item lists exact; scores and precision / recall / F to 1e-12 relative",
           "golden": "tests/golden/g12_topn.npz", "golden_sha256": hashlib.sha256(open(os.path.join(GOLD, "g12_topn.npz"), "rb").read()).hexdigest(),
           "script": "oracle/pin_topn.py"}
    json.dump(out, open(os.path.join(GOLD, "g12_topn_pin.json"), "w"), indent=1)
    print("top-N restatement == the reference's own (translated) source on %d queries + %d precision/recall runs: OK" % (n_queries, n_pr))


def pin_snapshots():
    """the model-snapshot and state FILES.  The reference's own statements that write and read them are executed here:
    save_model_snapshot / load_model_snapshot (base.py:805-812) whole, and the two statements of save_state that write the ratings
    matrix and the factor file (base.py:559-560: scipy.io.mmwrite + np.savez with svd_s / svd_u / svd_v) -- the five pickles behind them
    are written with `file(name, 'w')` + pickle.dump, which only Python 2 executes: their format stays restated (protocol-0 text
    pickles, tests/test_host_logic.py).  Both directions: files the reference's statements WRITE are committed as fixtures
    (tests/golden/refwritten_*) and read by mfrec_amd's load_model_snapshot / load_state in the CPU suite; files mfrec_amd writes
    are read back here by the reference's own load_model_snapshot and compared."""
    import scipy.io
    import tempfile
    from scipy.sparse import lil_matrix
    from mfrec_amd.recommendation.base import BaseRecommender
    ns = {"np": np, "scipy": __import__("scipy")}
    rec, pieces = [], []
    for head in ("def save_model_snapshot", "def load_model_snapshot", "def save_state"):
        src, (a, b) = cut("base.py", head, 4)
        if head == "def save_state":                     # the statements up to (not including) the first Python-2-only `file(` call
            keep = []
            for l in src.split("\n"):
                if "file(" in l:
                    break
                if "logger" in l:
                    continue
                keep.append(l)
            src = "\n".join(keep) + "\n"
        pieces.append((head, src, a, b))
    sha = _check_cut_text([p_[1] for p_ in pieces], "snapshot_pin.json")   # before any of it runs
    for head, src, a, b in pieces:
        rec.append({"file": "mfrec/recommendation/base.py", "definition": head, "lines": "%d-%d" % (a, b)})
        exec(compile(translate(textwrap.dedent(src)), "<base.py:%d>" % a, "exec"), ns)

    class H(object):
        pass
    H.save_model_snapshot, H.load_model_snapshot, H.save_state = ns["save_model_snapshot"], ns["load_model_snapshot"], ns["save_state"]
    rs = np.random.RandomState(5)
    h = H()
    h.svd_u, h.svd_v, h.svd_s = rs.normal(0, 0.1, (3, 6)), rs.normal(0, 0.1, (3, 4)), None
    M = lil_matrix((4, 6))
    for (a, b, c) in ((0, 1, 3.0), (3, 5, 5.0), (2, 2, 1.0), (2, 0, 4.0)):
        M[a, b] = c
    h.relationship_matrix = M
    # (1) written by the reference's statements -> committed fixtures
    h.save_model_snapshot(os.path.join(GOLD, "refwritten"))
    h.save_state(os.path.join(GOLD, "refwritten_state"))
    # (2) written by mfrec_amd -> read back by the reference's own load_model_snapshot
    with tempfile.TemporaryDirectory() as td:
        R = BaseRecommender(4, 6)
        R.svd_u, R.svd_v = h.svd_u.copy(), h.svd_v.copy()
        R.save_model_snapshot(os.path.join(td, "mine"))
        g = H()
        g.load_model_snapshot(os.path.join(td, "mine"))
        assert np.array_equal(g.svd_u, h.svd_u) and np.array_equal(g.svd_v, h.svd_v)
        # and the reference-written one through mfrec_amd's loader
        Q = BaseRecommender(1, 1)
        Q.load_model_snapshot(os.path.join(GOLD, "refwritten"))
        assert np.array_equal(Q.svd_u, h.svd_u) and np.array_equal(Q.svd_v, h.svd_v)
    out = {"pinned_by": "base.py's own save_model_snapshot / load_model_snapshot and the mmwrite + np.savez statements of save_state, "
                        "cut from the file, translated by lib2to3, executed under Python %d.%d" % sys.version_info[:2],
           "definitions": rec, "sha256_of_the_cut_text": sha.hexdigest(),
           "fixtures_written_by_the_reference_statements": ["refwritten_model_snapshot.npz", "refwritten_state_relations.mtx", "refwritten_state_svd.npz"],
           "not_covered": "the five label / index pickles of save_state (written with Python 2's file(name, 'w')): restated only",
           "script": "oracle/pin_topn.py (pin_snapshots)"}
    json.dump(out, open(os.path.join(GOLD, "snapshot_pin.json"), "w"), indent=1)
    print("snapshot files: the reference's own statements write what mfrec_amd reads and read what it writes: OK")


if __name__ == "__main__":
    main()
    pin_snapshots()
