"""A second, structurally different restatement of the reference's phrase boosting in plain Python (dict children, Python
sets of active states -- the containers the reference uses, src/phrase_boost.cpp:9-171), used only to cross-check the C oracle
on random inputs.  Only comparisons and one float32 add per candidate are involved, so agreement must be exact."""
import numpy as np


class PyTrie:
    def __init__(self, phrases=()):
        self.children = [dict()]
        for p in phrases:
            self.insert(p)

    def insert(self, ids):                      # :11-27
        if len(ids) == 0:
            return
        node = 0
        for t in ids:
            t = int(t)
            if t not in self.children[node]:
                self.children[node][t] = len(self.children)
                self.children.append(dict())
            node = self.children[node][t]

    def size(self):
        return len(self.children)

    def boosted(self, states):                  # :39-50
        out = set()
        for s in states:
            if 0 <= s < len(self.children):
                out |= set(self.children[s].keys())
        return out

    def advance(self, states, tok):             # :52-66
        nxt = {0}
        for s in states:
            if 0 <= s < len(self.children) and tok in self.children[s]:
                nxt.add(self.children[s][tok])
        return nxt


def ctc_boosted(logp, trie, boost, blank):
    """ctc_greedy_decode_with_timestamps_boosted (:114-171) for one utterance [T][V] -> (ids, start, end, raw log-probs)."""
    T, V = logp.shape
    boost = np.float32(boost)
    ids, start, end, lps = [], [], [], []
    prev, active = -1, {0}
    for t in range(T):
        val = logp[t].copy()
        for v in trie.boosted(active):
            if v < V:
                val[v] = np.float32(val[v] + boost)
        best = int(np.argmax(val))              # first maximum == strict '>' scan
        if best != prev:
            if prev != -1 and prev != blank and ids:
                end[-1] = t - 1
            if best != blank:
                ids.append(best); start.append(t); end.append(t); lps.append(logp[t, best])
                active = trie.advance(active, best)
        prev = best
    if ids:
        end[-1] = T - 1
    return ids, start, end, lps
