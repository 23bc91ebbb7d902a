"""A small hand-built corpus with a natural-language-like dictionary (mixed case, accents, near-duplicates,
transpositions, shared prefixes) for the fuzzy / prefix dictionary-scan parity tests."""
import random

import numpy as np

BASE = ["majestic", "majestät", "Majestic", "MAJESTY", "major", "majority", "magic", "magnet", "manner", "banner", "scanner", "planner",
        "straße", "strasse", "Straße", "über", "uber", "Über", "tree", "three", "there", "their", "thier", "theirs", "them", "theme",
        "привет", "Привет", "привед", "приветы", "search", "saerch", "serach", "searches", "searched", "research", "sea", "seat", "se",
        "a", "ab", "abc", "abcd", "abcde", "b", "ba", "東京", "東京都", "京都", "im", "immer", "imm", "nice", "niece", "Nice", "nIcE",
        "weather", "whether", "wether", "feather", "leather", "lather", "later", "latter", "letter", "litter", "glitter"]


def _mutations(rng, w, n):
    out = set()
    alpha = "abcdefghijklmnopqrstuvwxyzäöüéèßñ"
    for _ in range(n):
        s = list(w)
        for _ in range(rng.randint(1, 3)):
            op = rng.randint(0, 4)
            if op == 0 and s:
                del s[rng.randrange(len(s))]
            elif op == 1:
                s.insert(rng.randint(0, len(s)), rng.choice(alpha))
            elif op == 2 and s:
                s[rng.randrange(len(s))] = rng.choice(alpha)
            elif op == 3 and len(s) > 1:
                i = rng.randrange(len(s) - 1)
                s[i], s[i + 1] = s[i + 1], s[i]
            elif op == 4 and s:
                i = rng.randrange(len(s))
                s[i] = s[i].upper()
        if s:
            out.add("".join(s))
    return out


def build(num_docs=20_000, seed=7, per_word=25):
    from veloci_amd.index import IndexData
    rng = random.Random(seed)
    words = set(BASE)
    for w in BASE:
        words |= _mutations(rng, w, per_word)
    terms = sorted(w.encode("utf-8") for w in words)  # bytewise order == ordinal == term id
    nrng = np.random.default_rng(seed)
    lens = nrng.integers(1, 120, size=len(terms))
    offsets = np.zeros(len(terms) + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    anchors = np.zeros(int(offsets[-1]), np.uint32)
    scores = np.zeros(int(offsets[-1]), np.uint32)
    for t in range(len(terms)):
        o, n = int(offsets[t]), int(lens[t])
        anchors[o:o + n] = np.sort(nrng.choice(num_docs, size=n, replace=False))
        scores[o:o + n] = nrng.integers(1, 200, size=n)
    data = IndexData(num_docs)
    data.add_fst("body.textindex", terms)
    data.add_token_to_anchor_score("body.textindex.to_anchor_id_score", offsets, anchors, scores, None)
    data.add_key_value_store("body.textindex.text_id_to_anchor", offsets, anchors)
    return data, [t.decode("utf-8") for t in terms]
