"""Wraps over-long Python lines after commas that sit inside brackets (where Python ignores line breaks) and checks that the
compiled code objects are unchanged apart from line numbers.  usage: wrap_long_py.py FILE [max_len]"""
import sys, types


def scan(code):
    i, n, depth = 0, len(code), 0
    while i < n:
        c = code[i]
        if c == "#": return
        if c in "\"'":
            q = code[i:i + 3] if code[i:i + 3] in ('"""', "'''") else c
            j = i + len(q)
            while j < n and code[j:j + len(q)] != q:
                j += 2 if code[j] == "\\" else 1
            if j >= n: raise ValueError("open string")
            i = j + len(q); continue
        if c in "([{": depth += 1
        elif c in ")]}": depth -= 1
        yield i, c, depth
        i += 1


def wrap(line, max_len):
    if len(line) <= max_len: return [line]
    try: cands = [i + 2 for i, c, d in scan(line) if c == "," and line[i:i + 2] == ", " and 1 <= d <= 2]
    except ValueError: return [line]
    indent = line[: len(line) - len(line.lstrip())]; cont = indent + "    "
    out, start = [], 0
    while len(line) - start + (len(cont) if out else 0) > max_len:
        limit = start + max_len - (len(cont) if out else 0)
        best = [p for p in cands if start < p <= limit] or [p for p in cands if p > limit][:1]
        if not best: break
        cut = best[-1]
        out.append((cont if out else "") + line[start:cut].rstrip()); start = cut
    out.append((cont if out else "") + line[start:])
    return out if len(out) > 1 else [line]


def strip(co):
    consts = tuple(strip(c) if isinstance(c, types.CodeType) else c for c in co.co_consts)
    return (co.co_code, consts, co.co_names, co.co_varnames, co.co_freevars, co.co_cellvars, co.co_argcount, co.co_kwonlyargcount, co.co_flags & ~0, co.co_name)


def main():
    path = sys.argv[1]; max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    src = open(path).read()
    before = strip(compile(src, path, "exec"))
    lines, out, in_triple = src.split("\n"), [], False
    for l in lines:
        triple = (l.count('"""') + l.count("'''")) % 2 == 1
        if in_triple or triple:
            out.append(l); in_triple = in_triple != triple; continue
        out.extend(wrap(l, max_len))
    new = "\n".join(out)
    after = strip(compile(new, path, "exec"))
    if before != after:
        print("bytecode differs: %s left untouched" % path); sys.exit(1)
    open(path, "w").write(new)
    print("%s: %d -> %d lines, bytecode identical" % (path, len(lines), len(out)))


if __name__ == "__main__":
    main()
