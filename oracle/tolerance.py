"""TEST INFRASTRUCTURE (like everything under oracle/): the token-level statement of the tolerance-class (bf16) mode.

A greedy TDT decode is a chain of argmax decisions; under a numerical error of size eps a decision whose top-1 / top-2 margin is below eps may
legitimately go the other way, and everything after it follows a different path.  So two correct implementations of the tolerance-class
mode satisfy:  walking the reference decode's decisions in order, the other decode's tokens leave it only at a decision whose margin is
within the mode's error -- every token before the first such near-tie is identical.  first_divergence() evaluates that against the
per-decision labels and margins the oracle records (oracle.Model.tdt_greedy(margin=True); tests/golden/tdt600m_depth24_seed42.npz)."""
import numpy as np


def first_divergence(got, oracle_labels, oracle_margins, blank, got_frames=None, oracle_frames=None):
    """got: the token ids under test.  oracle_labels / oracle_margins: label chosen (blank included, -1 = unused slot) and the smallest
    top-1 / top-2 log-prob margin (label head, duration head) of every decision of the oracle's decode, in order.
    got_frames / oracle_frames (optional): (start, end) frame arrays of the tokens -- with them a token only counts as agreed when its id, its
    start frame AND its end frame agree, so a flipped DURATION decision (also of a blank step: it moves the frame pointer while the next ids may
    still coincide for a while) is located where it happened and not where the ids finally part.
    Returns (None, None) when the sequences are identical, else (index of the first differing token, smallest oracle margin among the
    decisions after the last agreed token's step up to and including the oracle's next token -- where the two decodes can have parted)."""
    want = [int(k) for k in oracle_labels if k >= 0 and k != blank]
    got = [int(k) for k in got]

    def same(i):
        if got[i] != want[i]:
            return False
        if got_frames is not None and oracle_frames is not None:
            return int(got_frames[0][i]) == int(oracle_frames[0][i]) and int(got_frames[1][i]) == int(oracle_frames[1][i])
        return True

    n_same = 0
    while n_same < min(len(got), len(want)) and same(n_same):
        n_same += 1
    if n_same == len(got) == len(want):
        return None, None
    seen, lo, hi = 0, 0, len(oracle_labels)
    for s, k in enumerate(oracle_labels):
        if k < 0:
            hi = s
            break
        if k != blank:
            seen += 1
            if seen == n_same:
                lo = s + 1
            if seen == n_same + 1:
                hi = s + 1
                break
    return n_same, (float(np.min(oracle_margins[lo:hi])) if hi > lo else float("inf"))
