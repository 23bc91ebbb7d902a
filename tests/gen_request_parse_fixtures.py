"""Generates tests/golden/request_parse.json: request texts and what serde makes of them, derived with tests/reqparse.py (an independent restatement;
nothing of the product or the oracle runs here).  Inputs: every request of the committed reference fixtures, the bench's synthetic request shapes,
and hand-written edge cases of the wire format (SURVEY.md "Request wire format").  Run from the repo root: python tests/gen_request_parse_fixtures.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reqparse  # noqa: E402

EDGE = [
    '{"search_req":{"search":{"terms":["a"],"path":"f"}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","firstCharExactMatch":true,"unknown":{"nested":[1,2,{"x":null}]}}},"also_unknown":1}',
    '{"search_req":{"search":{"terms":["a","b"],"path":"f","levenshtein_distance":2,"starts_with":true,"is_regex":false,"boost":1.5,"ignore_case":false,"snippet":true,'
    '"snippet_info":{"num_words_around_snippet":3},"top":5,"skip":2,"token_value":{"path":"f","boost_fun":"Log10","param":1}}},"top":null,"skip":null}',
    '{"search_req":{"search":{"terms":[],"path":""}},"top":0,"skip":0,"why_found":true,"text_locality":true,"explain":true,"select":["a","b"]}',
    '{"search_req":{"or":{"queries":[{"search":{"terms":["\\u00e4\\ud83d\\ude00\\"\\\\\\/\\b\\f\\n\\r\\t"],"path":"meanings.ger[]"}},{"and":{"queries":[],"options":{"explain":true,"top":3,"skip":1,'
    '"boost":[{"path":"commonness","boost_fun":"Multiply","param":-0.0,"skip_when_score":[1e-7,3.4028236e38,1e39,0.1],"expression":"$SCORE * 2"}]}}}],"options":null}}}',
    '{"search_req":{"search":{"terms":["x"],"path":"p","boost":16777217,"options":{}}},"boost":[{"path":"c"},{"path":"d","boost_fun":null,"param":null,"skip_when_score":null,"expression":null}],'
    '"facets":[{"field":"tags[]"},{"field":"c","top":null},{"field":"d","top":3}],"boost_term":[{"terms":["t"],"path":"p","boost":2}],'
    '"phrase_boosts":[{"search1":{"terms":["a"],"path":"p"},"search2":{"terms":["b"],"path":"p"}}],"filter":{"search":{"terms":["nice"],"path":"tags[]"}}}',
    '{"suggest":[{"terms":["wi"],"path":"p","starts_with":true,"top":10,"skip":0}],"top":5}',
    '  {\n"search_req" : { "search" : { "terms" : [ "a" ] , "path" : "f" , "boost" : 2.5E+0 } } , "top" : 10 }  ',
    '{}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"num_words_around_snippet":-2,"snippet_start_tag":"<em>","snippet_end_tag":"</em>","snippet_connector":" .. ","max_snippets":4294967295,"other":1}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":null}}}',
    # --- errors
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"num_words_around_snippet":null}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"num_words_around_snippet":2.0}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"max_snippets":4294967296}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"max_snippets":-1}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"snippet_start_tag":1}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":{"snippet_connector":null}}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":[]}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","snippet_info":"x"}}}',
    '{"search_req":{"search":{"terms":["a"]}}}',
    '{"search_req":{"search":{"path":"f"}}}',
    '{"search_req":{"search":{"terms":"a","path":"f"}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"},"or":{"queries":[]}}}',
    '{"search_req":{"xor":{"queries":[]}}}',
    '{"search_req":{"or":{}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"top":-1}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"top":1.0}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"top":1e2}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"top":"10"}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"why_found":null}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"why_found":1}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","starts_with":null}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","levenshtein_distance":4294967296}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","boost":"2"}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"boost":[{"path":"c","boost_fun":"log10"}]}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"boost":{"path":"c"}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"facets":[{"top":3}]}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"select":"a"}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"select":[1]}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","path":"g"}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f",}}}',
    '{"search_req":{"search":{"terms":["\\ud83d"],"path":"f"}}}',
    '{"search_req":{"search":{"terms":["a\tb"],"path":"f"}}}',
    '{"search_req":{"search":{"terms":["a"],"path":"f"}},"top":01}',
    '{"search_req":{"search":{"terms":["a"],"path":"f","boost":NaN}}}',
    '[]',
    '',
]


def fixture_requests():
    out = []
    g = os.path.join(HERE, "golden")
    with open(os.path.join(g, "reference_integration.json"), encoding="utf-8") as f:
        out += [c["request"] for c in json.load(f)["cases"] if "request" in c]
    with open(os.path.join(g, "reference_query_generator.json"), encoding="utf-8") as f:
        out += [c["request"] for c in json.load(f)["cases"] if "request" in c]
    with open(os.path.join(g, "reference_explain.json"), encoding="utf-8") as f:
        out += [c["request"] for c in json.load(f)["cases"]]
    with open(os.path.join(g, "reference_suggest_regex.json"), encoding="utf-8") as f:
        d = json.load(f)
        out += [c["request"] for c in d["suggest"] if "suggest" in c["request"]] + [c["request"] for c in d["regex"]]
    leaf = lambda t: {"search": {"terms": [t], "path": "title"}}
    out += [  # the bench's shapes (bench.py make_requests)
        {"search_req": {"and": {"queries": [leaf("t1"), leaf("t2"), leaf("t3")]}}, "top": 10},
        {"search_req": {"search": {"terms": ["term"], "path": "title", "levenshtein_distance": 2}}, "top": 10, "facets": [{"field": "cat", "top": 10}, {"field": "tags[]", "top": 10}]},
    ]
    return out


def main():
    texts = [json.dumps(r, ensure_ascii=False) for r in fixture_requests()] + [json.dumps(r) for r in fixture_requests()[:40]] + EDGE
    seen, cases = set(), []
    for t in texts:
        if t in seen:
            continue
        seen.add(t)
        try:
            cases.append({"text": t, "parsed": reqparse.canonical(t)})
        except reqparse.ParseError:
            cases.append({"text": t, "error": True})
    out = {"_about": "Request texts and the canonical dump (vq_request_to_json layout) of what serde's derive(Deserialize) makes of them; `error`: serde_json returns an error. "
                     "Derived by tests/gen_request_parse_fixtures.py with tests/reqparse.py, an independent restatement over Python's json module.", "cases": cases}
    with open(os.path.join(HERE, "golden", "request_parse.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", len(cases), "cases,", sum(1 for c in cases if "error" in c), "errors")


if __name__ == "__main__":
    main()
