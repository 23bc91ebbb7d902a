"""HBM footprint of a FULL-vocabulary index: 10 M docs, every one of the 1 M dictionary terms with a posting list (Zipf: rank r
has df = 0.1 N / r), plus the config-#4 facet columns.  Prints vq_index_device_bytes next to the raw array sizes, so the cost of the staged layout
(dense lists as bitmap + ids + scores, rank directories, per-list tables) is known.  usage: python tools/full_vocab_footprint.py [docs] [terms]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
terms = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
triples = 8
t0 = time.time()
spec = synth.SynthSpec(num_docs=docs, num_terms=terms, triples=triples, with_facets=True, with_t2t=False, with_phrase=False, with_boost=False,
                       background_terms=terms - 3 * triples)
data, meta = synth.generate(spec, device="cuda:0")
t_gen = time.time() - t0
offsets, anchors, scores, _ = data.token_to_anchor_score["body.textindex.to_anchor_id_score"]
lens = np.diff(offsets.astype(np.int64))
raw = {"postings": int(lens.sum()), "lists_with_postings": int((lens > 0).sum()), "raw_doc_ids_bytes": int(lens.sum()) * 4, "raw_f16_scores_bytes": int(lens.sum()) * 2,
       "dictionary_chars_bytes": int(sum(len(t) for t in synth.make_vocabulary(terms, spec.seed)))}
t0 = time.time()
index = veloci_amd.Index(data, device=0)
out = {"docs": docs, "terms": terms, "generate_s": round(t_gen, 1), "stage_s": round(time.time() - t0, 1), "device_bytes": index.device_bytes, **raw,
       "device_bytes_per_posting": round(index.device_bytes / max(raw["postings"], 1), 2),
       "lists_longer_than_docs_over_32": int((lens > docs // 32).sum())}
a = meta.triples[0]
res = veloci_amd.search({"search_req": {"search": {"terms": [a[0]], "path": "body", "levenshtein_distance": 2}}, "top": 10,
                         "facets": [{"field": "cat", "top": 10}, {"field": "tags[]", "top": 10}]}, index)
out["check_query_hits"] = int(res.num_hits)
print(json.dumps(out))
