"""BASELINE.json configs[0]: the reference's own benchmark request (benches/bench_jmdict.rs:115-235, `get_request(term, 0)`: an OR over
five field leaves with per-leaf anchor-level and 1:n boosts, prefix match on the kanji / kana fields) on a JMdict-like synthetic
corpus of 166 600 entries (bench_jmdict.rs:373; jmdict.json itself is a git-lfs pointer in the reference), index configuration of
veloci_bins/src/bin/create_test_index.rs:33-69.  Prints queries/s of batches through the C ABI, single-request p50, and the CPU
oracle (C++ restatement of the reference algorithm, not the Rust binary) on the same requests, one thread.
Lives under tests/ because it links the oracle (test infrastructure); run it as `python tests/bench_jmdict_shape.py`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def jmdict_request(term, lev):
    def leaf(path, boosts, starts_with):
        part = {"terms": [term], "path": path, "levenshtein_distance": lev, "options": {"boost": boosts}}
        if starts_with:
            part["starts_with"] = True
        return {"search": part}
    common = lambda p: {"path": "commonness", "boost_fun": "Log10", "param": p}
    return {"search_req": {"or": {"queries": [
        leaf("kanji[].text", [common(1), {"path": "kanji[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("kana[].text", [common(1), {"path": "kana[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("kana[].text", [common(1), {"path": "kana[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("meanings.ger[].text", [common(0), {"path": "meanings.ger[].rank", "expression": "10 / $SCORE"}], False),
        leaf("meanings.eng[]", [common(1)], False),
    ], "options": {"top": 10, "skip": 0}}}}


def corpus(n, seed=166600):
    rng = np.random.default_rng(seed)
    syll = np.array(["i", "yo", "ku", "u", "ji", "to", "ka", "na", "mi", "se", "ro", "ha", "ne", "so", "ta", "ki", "shi", "n"])
    def words(count, lo, hi):
        lens = rng.integers(lo, hi, count)
        picks = rng.integers(0, len(syll), int(lens.sum()))
        out, p = [], 0
        for l in lens:
            out.append("".join(syll[picks[p:p + l]]))
            p += l
        return out
    kan, eng, ger = words(20000, 1, 5), words(8000, 2, 5), [w.capitalize() for w in words(8000, 2, 5)]
    zipf = lambda size, count: np.minimum((rng.pareto(1.1, count) * 3).astype(np.int64), size - 1)  # a few very common words
    n_kanji, n_kana, n_eng, n_ger = rng.integers(0, 3, n), rng.integers(1, 3, n), rng.integers(1, 4, n), rng.integers(0, 3, n)
    ik, ia, ie, ig = zipf(len(kan), int(n_kanji.sum())), zipf(len(kan), int(n_kana.sum())), zipf(len(eng), 2 * int(n_eng.sum())), zipf(len(ger), 2 * int(n_ger.sum()))
    ck, ca, rk = rng.choice([0, 3, 40, 500], len(ik)), rng.choice([0, 7, 60], len(ia)), rng.integers(1, 6, int(n_ger.sum()))
    comm, pos = rng.choice([0, 5, 20, 350, 3000], n), rng.choice(["n", "v1", "adj-i"], n)
    docs, pk, pa, pe, pg = [], 0, 0, 0, 0
    for d in range(n):
        doc = {"commonness": int(comm[d]), "ent_seq": str(1000000 + d), "pos": [str(pos[d])]}
        if n_kanji[d]:
            doc["kanji"] = [{"text": kan[ik[pk + j]], "commonness": int(ck[pk + j])} for j in range(n_kanji[d])]
        pk += n_kanji[d]
        doc["kana"] = [{"text": kan[ia[pa + j]], "romaji": "iyoku", "commonness": int(ca[pa + j])} for j in range(n_kana[d])]
        pa += n_kana[d]
        m = {"eng": [eng[ie[2 * (pe + j)]] + " " + eng[ie[2 * (pe + j) + 1]] for j in range(n_eng[d])]}
        pe += n_eng[d]
        if n_ger[d]:
            m["ger"] = [{"text": ger[ig[2 * (pg + j)]] + " " + ger[ig[2 * (pg + j) + 1]], "rank": int(rk[pg + j])} for j in range(n_ger[d])]
        pg += n_ger[d]
        doc["meanings"] = m
        docs.append(doc)
    # common and rarer words; DISTINCT=1 (default): as many distinct terms as a batch has requests, so that nothing a batch resolves once (dictionary
    # scans, unions, 1:n boost lists) is shared between its requests; DISTINCT=0: the 100 terms of round 1, cycled
    if os.environ.get("DISTINCT", "1") == "1":
        def first_distinct(pool, count, seen):
            out = []
            for w in pool:
                if w not in seen:
                    seen.add(w)
                    out.append(w)
                    if len(out) == count:
                        break
            return out
        seen = set()
        terms = first_distinct(kan, 110, seen) + first_distinct(eng, 100, seen) + first_distinct(ger, 46, seen)
    else:
        terms = [kan[i] for i in range(0, 40)] + [eng[i] for i in range(0, 40)] + [ger[i] for i in range(0, 20)]
    return docs, terms


INDICES = {"commonness": {"boost": {"boost_type": "f32"}}, "meanings.ger[].rank": {"boost": {"boost_type": "f32"}},
           "kanji[].commonness": {"boost": {"boost_type": "f32"}}, "kana[].commonness": {"boost": {"boost_type": "f32"}},
           "kanji[].text": {"fulltext": {"tokenize": False}}, "kana[].text": {"fulltext": {"tokenize": False}},
           "kana[].romaji": {"fulltext": {"tokenize": True}}, "meanings.ger[].text": {"fulltext": {"tokenize": True}},
           "meanings.eng[]": {"fulltext": {"tokenize": True}}, "pos": {"fulltext": {"tokenize": False}}}


def main():
    import veloci_amd
    from veloci_amd import mini_indexer
    n = int(os.environ.get("DOCS", "166600")); lev = int(os.environ.get("LEV", "0")); batch = int(os.environ.get("BATCH", "256"))
    t0 = time.time()
    cache = os.environ.get("JM_CACHE")  # (parameter sweeps: build the corpus once per box)
    if cache and os.path.exists(cache):
        import pickle
        with open(cache, "rb") as f:
            data, terms = pickle.load(f)
    else:
        docs, terms = corpus(n)
        data, info = mini_indexer.build_index(docs, INDICES)
        if cache:
            import pickle
            with open(cache, "wb") as f:
                pickle.dump((data, terms), f, protocol=4)
    print(f"corpus of {n} entries generated and indexed in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    idx = veloci_amd.Index(data, device=0)
    reqs_json = [jmdict_request(terms[i % len(terms)], lev) for i in range(batch)]
    reqs = [veloci_amd.Request(r) for r in reqs_json]
    got = veloci_amd.search_batch(reqs, idx)
    for _ in range(2):
        veloci_amd.search_batch(reqs, idx)
    steps = int(os.environ.get("STEPS", "10"))
    t0 = time.perf_counter()
    for _ in range(steps):
        veloci_amd.search_batch(reqs, idx)
    dt_objects = time.perf_counter() - t0
    # the throughput entry point (vq_search_batch_flat: ids / scores as arrays, what bench.py's headline uses); same results as the objects above
    rb = veloci_amd.RequestBatch(reqs)
    num_hits, counts, ids, scores, status = veloci_amd.search_batch_flat(rb, idx, stride=10)
    assert not status.any()
    for i, g in enumerate(got):
        assert int(num_hits[i]) == g.num_hits and list(ids[i, :counts[i]]) == list(g.ids) and np.array_equal(scores[i, :counts[i]].view(np.uint32), np.asarray(g.scores, np.float32).view(np.uint32)), i
    t0 = time.perf_counter()
    for _ in range(steps):
        veloci_amd.search_batch_flat(rb, idx, stride=10)
    dt = time.perf_counter() - t0
    if os.environ.get("KERNELS") == "1":  # per-kernel device time of one more pass (event brackets on)
        idx.profile_enable(True)
        idx.profile_json(reset=True)
        for _ in range(steps):
            veloci_amd.search_batch(reqs, idx)
        prof = idx.profile_json(reset=True)
        idx.profile_enable(False)
        print(json.dumps({"per_step": steps, "kernels": prof}), file=sys.stderr, flush=True)
    lat = []
    for i in range(200):
        a = time.perf_counter()
        veloci_amd.search(reqs[i % batch], idx)
        lat.append(time.perf_counter() - a)
    out = {"workload": f"bench_jmdict get_request(term, {lev}) on {n} JMdict-like entries, batches of {batch} with {len(set(terms[:batch]))} distinct terms", "queries_per_s": round(batch * steps / dt, 1), "queries_per_s_result_objects": round(batch * steps / dt_objects, 1),
           "p50_latency_ms_single_request": round(float(np.percentile(lat, 50)) * 1e3, 3), "mean_hits": float(np.mean([g.num_hits for g in got]))}
    if os.environ.get("CPU", "1") == "1":
        from oracle import binding as O
        from parity import assert_same
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        js = [json.dumps(r) for r in reqs_json]
        want = [ora.search_json(j) for j in js[:min(len(terms), 100)]]
        for r, g, w in zip(reqs_json, got, want):
            assert_same(r, g, w, exact_scores=False)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 10.0:
            ora.search_json(js[k % batch])
            k += 1
        out["cpu_oracle_queries_per_s_one_thread"] = round(k / (time.perf_counter() - t0), 1)
        out["parity"] = f"{len(want)} distinct requests equal the oracle (ids exact, scores 1e-5)"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
