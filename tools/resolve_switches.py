"""Partial preprocessor: resolve the conditionals of compile-time switches whose losing side is being deleted.

    python tools/resolve_switches.py file ... -- NAME=undef NAME=1 ...

Every `#if / #ifdef / #ifndef / #elif` whose condition mentions only the given names is decided and the dead side removed;
conditions that also mention other macros are simplified (False || X -> X, ...) and re-emitted.  `#ifndef NAME / #define
NAME v / #endif` default blocks of a NAME given a value disappear (the name is then "defined"); remaining uses of such a
name in code are reported so that they can be replaced by hand.
"""
import re
import sys

TOK = re.compile(r"\s*(defined\s*\(\s*\w+\s*\)|defined\s+\w+|\w+|\|\||&&|==|!=|!|\(|\))")


class Sym:
    def __init__(self, text):
        self.text = text


def parse(expr, known):
    toks = TOK.findall(expr)
    if "".join(toks).replace(" ", "") != expr.replace(" ", ""):
        raise ValueError("cannot parse: " + expr)
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else None

    def take():
        t = toks[pos[0]]
        pos[0] += 1
        return t

    def atom():
        t = take()
        if t == "(":
            v = p_or()
            assert take() == ")"
            return v
        if t == "!":
            v = atom()
            if isinstance(v, Sym):
                return Sym("!" + (v.text if re.fullmatch(r"[\w() ]+", v.text) and "||" not in v.text else "(" + v.text + ")"))
            return 0 if v else 1
        m = re.fullmatch(r"defined\s*\(?\s*(\w+)\s*\)?", t)
        if m:
            n = m.group(1)
            if n in known:
                return 0 if known[n] is None else 1
            return Sym("defined(%s)" % n)
        if re.fullmatch(r"\d+", t):
            return int(t)
        if t in known:
            return 0 if known[t] is None else int(known[t])
        return Sym(t)

    def p_cmp():
        a = atom()
        while peek() in ("==", "!="):
            op = take()
            b = atom()
            if isinstance(a, Sym) or isinstance(b, Sym):
                sa = a.text if isinstance(a, Sym) else str(a)
                sb = b.text if isinstance(b, Sym) else str(b)
                a = Sym("%s %s %s" % (sa, op, sb))
            else:
                a = int((a == b) if op == "==" else (a != b))
        return a

    def p_and():
        a = p_cmp()
        while peek() == "&&":
            take()
            b = p_cmp()
            if not isinstance(a, Sym) and not isinstance(b, Sym):
                a = int(bool(a) and bool(b))
            elif not isinstance(a, Sym):
                a = b if a else 0
            elif not isinstance(b, Sym):
                a = a if b else 0
            else:
                a = Sym("%s && %s" % (a.text, b.text))
        return a

    def p_or():
        a = p_and()
        while peek() == "||":
            take()
            b = p_and()
            if not isinstance(a, Sym) and not isinstance(b, Sym):
                a = int(bool(a) or bool(b))
            elif not isinstance(a, Sym):
                a = 1 if a else b
            elif not isinstance(b, Sym):
                a = 1 if b else a
            else:
                a = Sym("%s || %s" % (a.text, b.text))
        return a

    v = p_or()
    assert pos[0] == len(toks), expr
    return v


def emit_if(sym, kw, comment):
    m = re.fullmatch(r"defined\((\w+)\)", sym.text)
    if m and kw == "if":
        return "#ifdef %s%s" % (m.group(1), comment)
    m = re.fullmatch(r"!defined\((\w+)\)", sym.text)
    if m and kw == "if":
        return "#ifndef %s%s" % (m.group(1), comment)
    return "#%s %s%s" % (kw, sym.text, comment)


def resolve(lines, known):
    out = []
    # stack entries: dict(state=..., emitted=bool, taken=bool)
    #   state: 'keep' (symbolic: directives stay), 'on' (this branch is live, directives dropped), 'off' (dead),
    #          'done' (an earlier branch was decided true: the rest is dead)
    stack = []

    def live():
        return all(f["state"] in ("keep", "on") for f in stack)

    for line in lines:
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", line)
        if not m:
            if live():
                out.append(line)
            continue
        kw, rest = m.group(1), m.group(2)
        comment = ""
        cm = re.search(r"\s*(//.*|/\*.*\*/\s*)$", rest)
        if cm:
            comment, rest = "  " + cm.group(1).strip(), rest[:cm.start()]
        rest = rest.strip()
        if kw in ("ifdef", "ifndef", "if"):
            if not live():
                stack.append({"state": "off", "dead_parent": True})
                continue
            expr = rest if kw == "if" else ("defined(%s)" % rest if kw == "ifdef" else "!defined(%s)" % rest)
            v = parse(expr, known)
            if isinstance(v, Sym):
                untouched = not any(re.search(r"\b%s\b" % re.escape(n), expr) for n in known)
                out.append(line if untouched else emit_if(v, "if", comment))
                stack.append({"state": "keep"})
            else:
                stack.append({"state": "on" if v else "off", "decided": bool(v)})
        elif kw == "elif":
            f = stack[-1]
            if f.get("dead_parent"):
                continue
            if f["state"] == "keep":
                v = parse(rest, known)
                if isinstance(v, Sym):
                    out.append(emit_if(v, "elif", comment))
                elif v:
                    out.append("#else" + comment)
                    f["state"] = "keep"
                    f["else_done"] = True
                else:
                    f["state"] = "keep_skip"
                continue
            if f["state"] == "keep_skip":
                f["state"] = "keep"
                v = parse(rest, known)
                if isinstance(v, Sym):
                    out.append(emit_if(v, "elif", comment))
                elif v:
                    out.append("#else" + comment)
                else:
                    f["state"] = "keep_skip"
                continue
            if f["state"] == "on":
                f["state"] = "done"
            elif f["state"] == "off":
                v = parse(rest, known)
                if isinstance(v, Sym):
                    # becomes the head of a symbolic chain
                    out.append(emit_if(v, "if", comment))
                    f["state"] = "keep"
                elif v:
                    f["state"] = "on"
        elif kw == "else":
            f = stack[-1]
            if f.get("dead_parent"):
                continue
            if f["state"] in ("keep",):
                if f.get("else_done"):
                    f["state"] = "keep_skip"
                else:
                    out.append(line)
            elif f["state"] == "keep_skip":
                out.append(line.replace("#else", "#else"))
                f["state"] = "keep"
            elif f["state"] == "on":
                f["state"] = "done"
            elif f["state"] == "off":
                f["state"] = "on"
        else:  # endif
            f = stack.pop()
            if f.get("dead_parent"):
                continue
            if f["state"] in ("keep", "keep_skip"):
                out.append(line)
    assert not stack
    return out


def main():
    args = sys.argv[1:]
    i = args.index("--")
    files, defs = args[:i], args[i + 1:]
    known = {}
    for d in defs:
        n, v = d.split("=")
        known[n] = None if v == "undef" else v
    for path in files:
        src = open(path).read().split("\n")
        res = resolve(src, known)
        # a default block whose name now counts as defined has vanished; report what still mentions the names
        open(path, "w").write("\n".join(res))
        for n in known:
            for k, line in enumerate(res):
                if re.search(r"\b%s\b" % n, line):
                    print("%s:%d: %s" % (path, k + 1, line.strip()[:150]))


if __name__ == "__main__":
    main()
