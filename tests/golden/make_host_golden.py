"""Generates tests/golden/host_golden.npz (EER cases as float32 / int32 arrays; the text-based cases as one JSON string in `text_cases`): host-side helpers of the training / validation loop pinned by the reference's OWN functions in
misc/utils.py - compute_cos_pairwise_eer (:273-312, the valid-set EER of Trainer.valid / train.py), load_lr (:193-200), load_valid_loss
(:203-214, + class ValidLoss :186-190), substring_in_list (:315-330), remove_params_prefix (:349-358), add_dict_prefix (:361-366);
and by the reference's misc/tools/sample_validset_spk2utt.py run as the script it is.

misc/utils.py cannot be imported here (its first lines import tensorflow, which this image does not have), but these functions do not
touch TensorFlow: each one's source is cut out of the file with `ast` at run time and executed against the numpy / scipy / sklearn this
image has - nothing of it is stored.  compute_cos_pairwise_eer was written for Python 2: its two integer divisions (`num_embeddings /
max_num_embeddings`, `num_embeddings * (num_embeddings - 1) / 2`) are floats under Python 3 and numpy refuses them, so exactly those
expressions get py2 semantics (`/` -> `//`), as make_attention_golden.py does for its reshape arguments.  It also normalises its input
IN PLACE and writes a scratch file `test.txt` into the working directory: it is run on copies inside a temporary directory.

Run in the build container only (needs /root/reference):  python tests/golden/make_host_golden.py
"""
import ast
import json
import os
import sys
import tempfile

import numpy as np
from scipy.interpolate import interp1d
from scipy.optimize import brentq
from six.moves import range
from sklearn import metrics

SRC_PATH = "/root/reference/misc/utils.py"
src = open(SRC_PATH).read()
tree = ast.parse(src)
WANT = ("ParamsPlain", "ValidLoss", "load_lr", "load_valid_loss", "compute_cos_pairwise_eer", "substring_in_list", "remove_params_prefix", "add_dict_prefix")
ns = {"np": np, "metrics": metrics, "brentq": brentq, "interp1d": interp1d, "range": range, "os": os, "sys": sys, "json": json}
for node in tree.body:
    if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in WANT:
        seg = ast.get_source_segment(src, node)
        if node.name == "compute_cos_pairwise_eer":
            for py2 in ("step = num_embeddings / max_num_embeddings", "(num_embeddings * (num_embeddings - 1) / 2)"):
                assert seg.count(py2) >= 1, py2
            seg = seg.replace("step = num_embeddings / max_num_embeddings", "step = num_embeddings // max_num_embeddings")
            seg = seg.replace("(num_embeddings * (num_embeddings - 1) / 2)", "(num_embeddings * (num_embeddings - 1) // 2)")
        exec(compile(seg, "<%s of misc/utils.py>" % node.name, "exec"), ns)
for name in WANT:
    assert name in ns, name

out = {"eer": [], "load_lr": [], "load_valid_loss": [], "substring_in_list": [], "remove_params_prefix": [], "add_dict_prefix": []}
rs = np.random.RandomState(20261004)
arrays, arrays_eer = {}, []
cwd = os.getcwd()
tmp = tempfile.mkdtemp(prefix="xv_host_golden_")
os.chdir(tmp)
try:
    # (embeddings, speakers, dimension, cluster spread, max_num_embeddings): separated and overlapping speakers, the subsampling branch with
    # integer strides 2 and 1 (n just above the cap), a two-speaker set
    for (n, spk, d, spread, cap) in ((40, 5, 16, 0.3, 1000), (300, 20, 16, 1.0, 1000), (300, 20, 16, 3.0, 1000), (2100, 30, 8, 1.5, 1000),
                                     (1100, 11, 8, 2.0, 1000), (64, 2, 4, 1.0, 1000), (150, 10, 16, 1.0, 60)):
        labels = rs.randint(0, spk, n)
        centres = rs.randn(spk, d)
        emb = (centres[labels] + spread * rs.randn(n, d)).astype(np.float32).astype(np.float64)
        eer = ns["compute_cos_pairwise_eer"](emb.copy(), labels.copy(), max_num_embeddings=cap)
        out["eer"].append({"n": n, "speakers": spk, "dim": d, "spread": spread, "max_num_embeddings": cap, "eer": float(eer)})
        arrays["emb_%d" % len(arrays_eer)] = emb.astype(np.float32)
        arrays["labels_%d" % len(arrays_eer)] = labels.astype(np.int32)
        arrays_eer.append(float(eer))
    for text in ("0 0.01\n1 0.01\n2 0.005\n3 0.0025\n", "0 0.001\n", "0 1e-2\n1 5e-3\n2 2.5e-3\n3 2.5e-3\n4 1.25e-3\n"):
        open("lr", "w").write(text)
        out["load_lr"].append({"text": text, "values": [float(v) for v in ns["load_lr"]("lr")]})
    for text in ("0 5.25 0.1234\n1 4.75 0.1111\n2 4.80 0.1000\n3 4.10 0.0990\n4 4.10 0.0980\n", "0 3.5 0.2\n", "0 9.0 0.3\n1 9.5 0.31\n"):
        open("vl", "w").write(text)
        v = ns["load_valid_loss"]("vl")
        out["load_valid_loss"].append({"text": text, "min_loss": float(v.min_loss), "min_loss_epoch": int(v.min_loss_epoch)})
finally:
    os.chdir(cwd)
for s, lst in (("tdnn/tdnn1_conv/kernel", ["tdnn1", "tdnn9"]), ("softmax/output/kernel", ["tdnn"]), ("tdnn/tdnn6_dense/bias", None), ("x", []),
               ("tdnn/attention/query", ["attention", "softmax"])):
    out["substring_in_list"].append({"s": s, "list": lst, "result": bool(ns["substring_in_list"](s, lst))})
for d, prefix in (({"att_key_input": "tdnn4_relu", "att_num_heads": 1, "pooling_type": "self_attention", "att": 3}, "att"),
                  ({"a_b": 1, "b": 2, "a_b_c": 3}, "a"), ({"x": 1}, "y")):
    p = ns["ParamsPlain"]()
    p.dict.update(d)
    q = ns["remove_params_prefix"](p, prefix)
    out["remove_params_prefix"].append({"dict": d, "prefix": prefix, "result": dict(q.dict), "input_after": dict(p.dict)})
    out["add_dict_prefix"].append({"dict": d, "prefix": prefix, "result": ns["add_dict_prefix"](dict(d), prefix)})
# misc/tools/sample_validset_spk2utt.py (run.sh:179) is a plain script: it is RUN as it is (runpy, argv as run.sh passes them) with the global
# `random` seeded first - the script itself never seeds, so its draw is whatever the interpreter's generator holds
import contextlib
import io
import random
import runpy
out["sample_validset"] = []
spk2utt = "".join("spk%02d %s\n" % (i, " ".join("spk%02d-u%d" % (i, j) for j in range(n))) for i, n in enumerate([12, 9, 7, 3, 2, 6, 15, 5, 8, 4]))
tmp2 = tempfile.mkdtemp(prefix="xv_host_golden_")
open(os.path.join(tmp2, "spk2utt"), "w").write(spk2utt)
for (nspk, nutt, seed) in ((4, 5, 7), (2, 3, 0), (8, 5, 11), (10, 6, 3), (3, 1, 5)):
    buf = io.StringIO()
    argv = sys.argv
    sys.argv = ["sample_validset_spk2utt.py", str(nspk), str(nutt), os.path.join(tmp2, "spk2utt")]
    random.seed(seed)
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path("/root/reference/misc/tools/sample_validset_spk2utt.py", run_name="__main__")
    finally:
        sys.argv = argv
    out["sample_validset"].append({"num_spks": nspk, "num_utts": nutt, "seed": seed, "spk2utt": spk2utt, "stdout": buf.getvalue()})
# dataset/data_loader.py:get_speaker_info (:14-55; the module imports tensorflow for its logging only): its source, with the one Python 2 idiom it
# holds (`line.decode()` on a str read in text mode) taken out at run time, on a small data directory of text files
dsrc = open("/root/reference/dataset/data_loader.py").read()
for node in ast.parse(dsrc).body:
    if isinstance(node, ast.FunctionDef) and node.name == "get_speaker_info":
        seg = ast.get_source_segment(dsrc, node)
        assert seg.count("line.decode().split(' ')") == 1
        exec(compile(seg.replace("line.decode().split(' ')", "line.split(' ')"), "<get_speaker_info of dataset/data_loader.py>", "exec"), ns)
files = {"spklist": "spkA 0\nspkB 1\nspkC 2\nspkD 3\n",      # spkD: in the list, not in this directory (a validation set holds a subset)
         "spk2utt": "spkB spkB-u0 spkB-u1\nspkA spkA-u0\nspkC spkC-u0 spkC-u1 spkC-u2\n",
         "feats.scp": "".join("%s /data/feats.%d.ark:%d\n" % (u, i % 2, 17 + 1000 * i) for i, u in enumerate(["spkA-u0", "spkB-u0", "spkB-u1", "spkC-u0", "spkC-u1", "spkC-u2"])),
         "utt2num_frames": "".join("%s %d\n" % (u, 300 + 10 * i) for i, u in enumerate(["spkA-u0", "spkB-u0", "spkB-u1", "spkC-u0", "spkC-u1", "spkC-u2"]))}
tmp3 = tempfile.mkdtemp(prefix="xv_host_golden_")
for name, text in files.items():
    open(os.path.join(tmp3, name), "w").write(text)
s2f, f2s, s2i = ns["get_speaker_info"](tmp3, os.path.join(tmp3, "spklist"))
out["speaker_info"] = {"files": files, "spk2features": {str(k): v for k, v in s2f.items()}, "features2spk": f2s, "spk2index": s2i}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_golden.npz")
arrays["eer"] = np.array(arrays_eer)
arrays["text_cases"] = np.array(json.dumps(out))
np.savez_compressed(path, **arrays)
print("wrote %s (%d EER cases: %s)" % (path, len(out["eer"]), " ".join("%.4f" % c["eer"] for c in out["eer"])))
