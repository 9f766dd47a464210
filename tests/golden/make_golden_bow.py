#!/usr/bin/env python3
"""Golden fixture of the BoW front half, generated from the CPU oracle like the others (make_golden.py):
a small ragged synthetic vocabulary in the column form of the reference's text format, 400 descriptors, and the oracle's
transform (levelsup 2) + L1 scores. The reference ships neither a vocabulary nor vectors for this path (parity unpinned);
tests/test_oracle_bow.py ties the oracle to DBoW2's definitions by hand-computed cases."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "orb-slam2-dualcam_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)
O.build()

v = synth.vocabulary(k=5, L=4, seed=11, ragged=0.25, early_leaf=0.1, stop_frac=0.08, dup_frac=0.1)
feats = np.concatenate([synth.descriptors_near_words(v, 340, seed=5), synth.random_descriptors(60, seed=6)])
V = O.Vocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"])
r = V.transform(feats, 2)
r2 = V.transform(feats[::2], 2)
db_off = np.array([0, len(r["bow_word"]), len(r["bow_word"]) + len(r2["bow_word"])], np.int32)
score = O.bow_score_l1(r2["bow_word"], r2["bow_val"], db_off, np.concatenate([r["bow_word"], r2["bow_word"]]), np.concatenate([r["bow_val"], r2["bow_val"]]))
np.savez_compressed(os.path.join(HERE, "bow_small.npz"), k=np.array(v["k"]), L=np.array(v["L"]), parent=v["parent"], is_leaf=v["is_leaf"], voc_desc=v["desc"],
                    weight=v["weight"], feats=feats, levelsup=np.array(2), word=r["word"], node=r["node"], bow_word=r["bow_word"], bow_val=r["bow_val"],
                    fv_node=r["fv_node"], fv_off=r["fv_off"], fv_idx=r["fv_idx"], score_half_vs_full_and_self=score)
print("nodes", len(v["parent"]) + 1, "words", V.n_words(), "bow", len(r["bow_word"]), "fv nodes", len(r["fv_node"]), "stopped", int((r["word"] < 0).sum()), "scores", score)
