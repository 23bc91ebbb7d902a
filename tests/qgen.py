"""TEST INFRASTRUCTURE — restatement of the reference's query generator, used only to DERIVE the request fixtures of
tests/golden/reference_query_generator.json (tests/gen_query_generator_fixtures.py) and to check that they are reproducible.

Follows `query_generator::search_query` (src/query_generator.rs:175-246) and `ast_to_search_request`
(src/query_generator/query_parser_to_veloci_request.rs:11-109) for the parameters the reference's own tests use
(tests/all/test_query_generator.rs, tests/all/test_code_search.rs:73-150, tests/all/test_scores.rs:157-237): search_term, parser_options.no_attributes, top, skip,
ignore_case, levenshtein, levenshtein_auto_limit, facets, facetlimit, why_found, text_locality, fields, boost_fields, boost_terms,
explain, stopword_lists / stopwords (parsed and — as in the reference, :12 — without effect: filter_stopwords' result is dropped).
The query language is restated for the forms those tests use (query_parser/src/parser.rs:100-186): literals, `attr:literal`,
`AND` / `OR` (right-recursive), juxtaposition = OR, quoted literals, the parser options that turn `(` `)` `~` `:` into plain characters;
parenthesised groups and `~n` are not needed by them and are rejected here.
`metadata.get_all_fields()` iterates an FnvHashMap (src/metadata.rs:28-30): the field order is unspecified in the reference; sorted here.
"""
import re


class GeneratorError(Exception):
    pass


def parse(text, no_attributes=False, no_parentheses=False, no_levensthein=False):
    """-> AST: ("leaf", phrase) | ("attr", field, ast) | ("bin", ast, "and"|"or", ast)   (query_parser/src/parser.rs:139-186).
    Lexer (query_parser/src/lexer.rs:20-41,114-196): whitespace separates literals; a double quote opens a literal that runs to the next
    quote (whitespace and everything else inside it kept); '(' ')' are tokens unless `no_parentheses`, '~' unless `no_levensthein` — with
    the option set they are ordinary characters of a literal."""
    tokens, quoted = [], set()
    i, n = 0, len(text)
    while i < n:
        if text[i].isspace():
            i += 1
            continue
        if text[i] == '"':  # lexer.rs:132-159
            j = text.find('"', i + 1)
            j = n if j < 0 else j
            quoted.add(len(tokens))
            tokens.append(text[i + 1:j])
            i = j + 1
            continue
        j = i
        while j < n and not text[j].isspace():
            j += 1
        tokens.append(text[i:j])
        i = j
    for k, tok in enumerate(tokens):
        if k in quoted:
            continue
        if (not no_parentheses and any(c in tok for c in "()")) or (not no_levensthein and "~" in tok) or '"' in tok:
            raise GeneratorError("query form outside the restated subset: " + text)
    if not tokens:
        raise GeneratorError("empty query")

    def atom(pos):
        tok = tokens[pos]
        if pos in quoted:  # (a quoted literal directly followed by ':' would be an attribute, lexer.rs:149-154: not needed here)
            return ("leaf", tok)
        if not no_attributes and ":" in tok:
            field, phrase = tok.split(":", 1)
            if ":" in phrase or not phrase:
                raise GeneratorError("parse error: " + tok)  # parser.rs:205 `field:what:ok` is an error
            return ("attr", field, ("leaf", phrase))
        return ("leaf", tok)

    def expr(pos):
        cur = atom(pos)
        pos += 1
        if pos == len(tokens):
            return cur
        # AND / OR are operators only between two operands: first or last they are literals (query_parser/src/lexer.rs:265-275: "OR OR" is
        # two literals, "OR OR OR" is Literal Or Literal)
        if tokens[pos] == "AND" and pos not in quoted and pos + 1 < len(tokens):
            return ("bin", cur, "and", expr(pos + 1))
        if tokens[pos] == "OR" and pos not in quoted and pos + 1 < len(tokens):
            return ("bin", cur, "or", expr(pos + 1))
        return ("bin", cur, "or", expr(pos))  # two literals next to each other: OR (parser.rs:113-115)

    return expr(0)


def get_default_levenshtein(term, limit, wildcard):  # query_generator.rs:84-99
    n = len(term)
    if wildcard:
        return 0 if n <= 3 else min(1, limit) if n <= 5 else min(2, limit)
    return 0 if n <= 2 else min(1, limit) if n <= 5 else min(2, limit)


def get_levenshtein(term, levenshtein, auto_limit, wildcard):  # :129-132
    d = levenshtein if levenshtein is not None else get_default_levenshtein(term, auto_limit if auto_limit is not None else 1, wildcard)
    return min(d, len(term) - 1)


def regex_escape(s):
    """regex::escape: backslash before every regex meta character"""
    return re.sub(r"([\\.+*?()|\[\]{}^$#&\-~])", r"\\\1", s)


def expand_fields(ast, all_fields):  # query_parser_to_veloci_request.rs:84-109
    kind = ast[0]
    if kind == "bin":
        return ("bin", expand_fields(ast[1], all_fields), ast[2], expand_fields(ast[3], all_fields))
    if kind == "leaf":
        cur = ("attr", all_fields[0], ast)
        for f in all_fields[1:]:
            cur = ("bin", ("attr", f, ast), "or", cur)
        return cur
    if ast[1] not in all_fields:  # check_field :134-144
        raise GeneratorError(f"Field {ast[1]} not found in {all_fields}")
    return ast


def to_request(ast, opt, field=None):  # query_ast_to_request :23-82
    kind = ast[0]
    if kind == "bin":
        return {ast[2]: {"queries": [to_request(ast[1], opt, field), to_request(ast[3], opt, field)]}}
    if kind == "attr":
        return to_request(ast[2], opt, ast[1])
    term = ast[1]
    starts_with = term.endswith("*") and term.count("*") == 1
    if starts_with:
        term = term[:-1]
    is_regex = "*" in term
    part = {"path": field, "terms": [term]}
    if is_regex:
        part["terms"] = [".*".join(regex_escape(p) for p in term.split("*"))]
    else:
        part["levenshtein_distance"] = get_levenshtein(term, opt.get("levenshtein"), opt.get("levenshtein_auto_limit"), starts_with)
    boost = (opt.get("boost_fields") or {}).get(field)
    if boost is not None:
        part["boost"] = boost
    if starts_with:
        part["starts_with"] = True
    if is_regex:
        part["is_regex"] = True
    if opt.get("ignore_case") is not None:
        part["ignore_case"] = opt["ignore_case"]
    return {"search": part}


def simplify(req):  # search/request/search_request.rs:26-72: nested or-in-or / and-in-and (without options) are pulled up
    for kind in ("or", "and"):
        if kind in req:
            qs = [simplify(q) for q in req[kind]["queries"]]
            pulled = []
            for i in range(len(qs) - 1, -1, -1):
                if kind in qs[i] and "options" not in qs[i][kind]:
                    pulled.extend(qs.pop(i)[kind]["queries"])
            return {kind: {"queries": qs + pulled}}
    return req


def search_query(all_fields, search_fields, opt):
    """all_fields: every column of the index (sorted); search_fields: those with a posting index (`has_token_to_anchor`, :101-127).
    -> the search::Request as a JSON-able dict"""
    opt = dict(opt)
    if opt.get("fields") is not None:
        fields = [f for f in all_fields if f in opt["fields"]]
    else:
        fields = list(search_fields)
    if not fields:
        raise GeneratorError(f"All fields filtered all_fields: {all_fields} filter: {opt.get('fields')}")
    po = opt.get("parser_options") or {}
    ast = parse(opt["search_term"], po.get("no_attributes", False), po.get("no_parentheses", False), po.get("no_levensthein", False))
    request = {"search_req": simplify(to_request(expand_fields(ast, fields), opt))}
    if opt.get("facets") is not None:
        for f in opt["facets"]:
            if f not in all_fields:
                raise GeneratorError(f"Field {f} not found in {all_fields}")
        request["facets"] = [{"field": f, "top": opt.get("facetlimit", 5)} for f in opt["facets"]]
    if opt.get("boost_terms") is not None:  # handle_boost_term_query :146-172
        parts = []
        for bt, value in opt["boost_terms"].items():
            flt = None
            if ":" in bt:
                pieces = bt.split(":")
                bt = pieces.pop(1)
                flt = pieces
            bfields = [f for f in all_fields if f in flt] if flt is not None else list(search_fields)
            parts += [{"path": f, "terms": [bt], "boost": value} for f in bfields]
        request["boost_term"] = parts
    if opt.get("phrase_pairs"):
        raise GeneratorError("phrase_pairs: outside the restated subset")
    if opt.get("filter") is not None:  # :219-226: levenshtein 0, all columns
        fast = parse(opt["filter"], (opt.get("filter_parser_options") or {}).get("no_attributes", False))
        request["filter"] = simplify(to_request(expand_fields(fast, all_fields), {"levenshtein": 0}))
    if opt.get("top") is not None:
        request["top"] = opt["top"]
    if opt.get("skip") is not None:
        request["skip"] = opt["skip"]
    if opt.get("why_found"):
        request["why_found"] = True
    if opt.get("text_locality"):
        request["text_locality"] = True
    if opt.get("boost_queries") is not None:
        request["boost"] = opt["boost_queries"]
    if opt.get("explain"):
        request["explain"] = True
    return request
