"""Builds a tiny synthetic Kaldi data directory (CM-compressed feats.ark, feats.scp, spk2utt,
utt2spk, utt2num_frames, spklist) for loader / Trainer tests."""
import os

import numpy as np

from tf_kaldi_speaker_amd.dataset import kaldi_io


def make_data_dir(root, num_spk=6, utts_per_spk=3, dim=30, min_frames=60, max_frames=120, seed=0):
    rs = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    ark = os.path.join(root, "feats.ark")
    scp, utt2spk, utt2nf, spk2utt, mats = [], [], [], {}, {}
    with open(ark, "wb") as f:
        for s in range(num_spk):
            spk = "spk%02d" % s
            centre = rs.randn(dim) * 2
            for u in range(utts_per_spk):
                utt = "%s-utt%d" % (spk, u)
                n = rs.randint(min_frames, max_frames + 1)
                m = (centre + rs.randn(n, dim)).astype(np.float32)
                f.write((utt + " ").encode())
                off = f.tell()
                kaldi_io.write_compressed_mat(f, m)
                scp.append("%s %s:%d" % (utt, ark, off))
                utt2spk.append("%s %s" % (utt, spk))
                utt2nf.append("%s %d" % (utt, n))
                spk2utt.setdefault(spk, []).append(utt)
                mats[utt] = m
    open(os.path.join(root, "feats.scp"), "w").write("\n".join(scp) + "\n")
    open(os.path.join(root, "utt2spk"), "w").write("\n".join(utt2spk) + "\n")
    open(os.path.join(root, "utt2num_frames"), "w").write("\n".join(utt2nf) + "\n")
    open(os.path.join(root, "spk2utt"), "w").write("\n".join("%s %s" % (k, " ".join(v)) for k, v in spk2utt.items()) + "\n")
    spklist = os.path.join(root, "spklist")
    open(spklist, "w").write("\n".join("%s %d" % (k, i) for i, k in enumerate(sorted(spk2utt))) + "\n")
    return root, spklist, mats
