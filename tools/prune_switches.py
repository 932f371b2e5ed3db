#!/usr/bin/env python3
"""Resolve compile-time A/B switches in place (round 4 prune): every `#if defined(X)` / `#if !defined(X)` /
`#elif defined(X)` / `... && !defined(X)` whose macro X is in the UNDEF list is evaluated as "X is not defined" and the
dead branch deleted.  Conditions on other macros are left alone.  Usage: prune_switches.py FILE... -- X Y Z"""
import re
import sys


def simplify(cond, undef):
    """cond with defined(X) -> 0 for X in undef; returns '0', '1' or the residual expression."""
    c = cond
    for x in undef:
        c = re.sub(r"!\s*defined\(\s*%s\s*\)" % x, "1", c)
        c = re.sub(r"defined\(\s*%s\s*\)" % x, "0", c)
    c = c.strip()
    # fold trivial conjunctions / disjunctions
    parts = [p.strip() for p in c.split("&&")]
    if len(parts) > 1:
        if any(p == "0" for p in parts):
            return "0"
        parts = [p for p in parts if p != "1"]
        return " && ".join(parts) if parts else "1"
    parts = [p.strip() for p in c.split("||")]
    if len(parts) > 1:
        if any(p == "1" for p in parts):
            return "1"
        parts = [p for p in parts if p != "0"]
        return " || ".join(parts) if parts else "0"
    return c


def process(text, undef):
    out = []
    # stack entries: dict(kind='plain'|'resolved', emitting=bool, taken=bool, parent_emit=bool)
    stack = []
    emit = True
    for line in text.split("\n"):
        st = line.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", st)
        if not m:
            if emit:
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2)
        comment = ""
        mc = re.search(r"(//.*|/\*.*)$", rest)
        expr = rest[:mc.start()] if mc else rest
        comment = rest[mc.start():] if mc else ""
        expr = expr.strip()
        if d == "ifdef":
            expr, d = "defined(%s)" % expr, "if"
        elif d == "ifndef":
            # keep tunable defaults (#ifndef X / #define X v / #endif) untouched unless X is to be undefined
            if expr in undef:
                expr, d = "!defined(%s)" % expr, "if"
            else:
                stack.append(dict(kind="plain", parent=emit))
                if emit:
                    out.append(line)
                continue
        if d == "if":
            touched = any(re.search(r"defined\(\s*%s\s*\)" % x, expr) for x in undef)
            if not touched:
                stack.append(dict(kind="plain", parent=emit))
                if emit:
                    out.append(line)
                continue
            r = simplify(expr, undef)
            if r in ("0", "1"):
                stack.append(dict(kind="resolved", parent=emit, taken=r == "1", open_if=False))
                emit = emit and r == "1"
            else:  # residual condition on other macros
                stack.append(dict(kind="plain", parent=emit))
                if emit:
                    out.append(re.sub(r"#\s*if.*", "#if " + r + ((" " + comment) if comment else ""), line))
            continue
        top = stack[-1]
        if d == "elif":
            if top["kind"] == "plain":
                if emit:
                    out.append(line)
                continue
            touched = any(re.search(r"defined\(\s*%s\s*\)" % x, expr) for x in undef)
            r = simplify(expr, undef) if touched else expr
            if top["taken"]:
                emit = False
            elif r == "1":
                top["taken"] = True
                emit = top["parent"]
            elif r == "0":
                emit = False
            else:
                raise SystemExit("unsupported: #elif with residual condition after a resolved #if: " + line)
            continue
        if d == "else":
            if top["kind"] == "plain":
                if emit:
                    out.append(line)
                continue
            emit = top["parent"] and not top["taken"]
            top["taken"] = True
            continue
        if d == "endif":
            stack.pop()
            if top["kind"] == "plain":
                if top["parent"]:
                    out.append(line)
            emit = top["parent"]
            continue
    assert not stack, "unbalanced conditionals"
    return "\n".join(out)


if __name__ == "__main__":
    i = sys.argv.index("--")
    files, undef = sys.argv[1:i], sys.argv[i + 1:]
    for f in files:
        src = open(f).read()
        dst = process(src, undef)
        if dst != src:
            open(f, "w").write(dst)
            print("%s: %d -> %d lines" % (f, len(src.splitlines()), len(dst.splitlines())))
