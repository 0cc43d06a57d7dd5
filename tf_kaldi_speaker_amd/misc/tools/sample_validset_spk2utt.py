#!/usr/bin/env python3
"""Drop-in for misc/tools/sample_validset_spk2utt.py (called by egs/voxceleb/v1/run.sh:179):

    python sample_validset_spk2utt.py num_heldout_spk num_heldout_utts_per_spk input_spk2utt [--seed N] > valid/spk2utt

Draws the held-out validation speakers and, for each, the utterances moved to the validation set.  Rules of the
reference tool: speakers with at least `num_utts + 2` utterances are preferred (the pool is topped up with random
smaller speakers only when there are too few of them); a chosen speaker gives `num_utts` random utterances, or all
but one when it has no more than `num_utts` - one utterance of every speaker always stays in the training set.
--seed (an addition) makes the draw reproducible; without it the draw is seeded from the OS like the reference's.
"""
import argparse
import random
import sys


def read_spk2utt(path):
    speakers = []
    with open(path, "r") as f:
        for line in f:
            fields = line.split()
            if len(fields) < 2:
                continue
            speakers.append((fields[0], fields[1:]))
    return speakers


def sample_validset(speakers, num_spks, num_utts_per_spk, rng):
    """-> [(speaker, [utterances moved to the validation set])] for `num_spks` speakers."""
    rich = [s for s in speakers if len(s[1]) >= num_utts_per_spk + 2]
    poor = [s for s in speakers if len(s[1]) < num_utts_per_spk + 2]
    if len(rich) < num_spks:
        if len(rich) + len(poor) < num_spks:
            raise ValueError("only %d speakers, cannot hold out %d" % (len(rich) + len(poor), num_spks))
        rich = rich + rng.sample(poor, num_spks - len(rich))
    out = []
    for spk, utts in rng.sample(rich, num_spks):
        take = num_utts_per_spk if len(utts) > num_utts_per_spk else len(utts) - 1
        out.append((spk, rng.sample(utts, take)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(usage="%(prog)s num_heldout_spk num_heldout_utts_per_spk input_spk2utt [--seed N]")
    ap.add_argument("num_heldout_spk", type=int)
    ap.add_argument("num_heldout_utts_per_spk", type=int)
    ap.add_argument("input_spk2utt")
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    for spk, utts in sample_validset(read_spk2utt(args.input_spk2utt), args.num_heldout_spk, args.num_heldout_utts_per_spk, rng):
        sys.stdout.write(" ".join([spk] + utts) + "\n")


if __name__ == "__main__":
    main()
