"""Re-lays out very long C++ lines (several statements / inline blocks on one line) as one statement per line.
Pure layout: it only inserts line breaks and indentation at statement boundaries; the result is accepted only if the compiler
output is unchanged (see the commit that used it: hipcc -S before/after).  usage: split_long_lines.py FILE [min_len]"""
import re, sys


def scan(code):
    """yield (index, char, paren_depth) for code positions outside string / char literals"""
    i, n, depth = 0, len(code), 0
    while i < n:
        c = code[i]
        if c == "/" and code[i:i + 2] == "/*":          # block comment: nothing inside it is a break point
            j = code.find("*/", i + 2)
            i = n if j < 0 else j + 2; continue
        if c in "\"'":
            q = c; j = i + 1
            while j < n and code[j] != q:
                j += 2 if code[j] == "\\" else 1
            i = j + 1; continue
        if c in "([": depth += 1
        elif c in ")]": depth -= 1
        yield i, c, depth
        i += 1


def split_comment(line):
    for i, c, _ in scan(line):
        if c == "/" and line[i:i + 2] == "//":
            return line[:i].rstrip(), line[i:]
    return line, ""


def is_block_open(code, i):
    """`{` at i opens a statement block (after `)`, else, do, `{`, `;`, `}` or at line start) rather than a braced initialiser"""
    prev = code[:i].rstrip()
    return prev == "" or prev.endswith((")", "else", "do", "{", ";", "}", "const", "noexcept")) or re.search(r"\]\s*$", prev) is None and prev.endswith("try")


def matching(code, i):
    depth = 0
    for j, c, _ in scan(code[i:]):
        if c == "{": depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0: return i + j
    return -1


def pieces(code):
    """split `code` (no comment) into (text, indent_delta_before, indent_delta_after) statements"""
    out, cur, i, n = [], "", 0, len(code)
    pos = {k: (c, d) for k, c, d in scan(code)}
    while i < n:
        c = code[i]
        if i not in pos:                      # inside a literal
            cur += c; i += 1; continue
        ch, depth = pos[i]
        if ch == "{" and depth == 0:
            if is_block_open(code, i):
                cur += "{"; out.append((cur.strip(), 0, 1)); cur = ""; i += 1; continue
            j = matching(code, i)             # braced initialiser: keep whole
            if j < 0: return None
            cur += code[i:j + 1]; i = j + 1; continue
        if ch == "}" and depth == 0:
            if cur.strip(): out.append((cur.strip(), 0, 0)); cur = ""
            # attach what follows up to the next statement boundary (`;`, ` else {`, `)` of a call …) to the brace
            k = i + 1; tail = "}"
            m = re.match(r"\s*(;|\)\s*;|,|\)\s*\)\s*;|else\b)", code[k:])
            if m and m.group(1) != "else":
                tail += m.group(0).strip() if m.group(1) != "," else ","; k += m.end()
            out.append((tail, -1, 0)); i = k; continue
        if ch == ";" and depth == 0:
            cur += ";"; out.append((cur.strip(), 0, 0)); cur = ""; i += 1; continue
        cur += c; i += 1
    if cur.strip(): out.append((cur.strip(), 0, 0))
    return out


def relayout(line, min_len):
    if len(line) <= min_len or line.lstrip().startswith("#") or line.rstrip().endswith("\\"): return [line]
    code, comment = split_comment(line.rstrip("\n"))
    indent = re.match(r"\s*", code).group(0)
    ps = pieces(code.strip())
    if not ps or len(ps) < 2: return [line]
    # merge `else` / `else if` heads with the closing brace before them, and `for (…)` headers are never split (paren depth)
    lines, level = [], 0
    for text, before, after in ps:
        level += before
        if level < 0: return [line]
        lines.append(indent + "  " * level + text)
        level += after
    if level != 0: return [line]
    merged = []
    for l in lines:
        if merged and merged[-1].strip() == "}" and re.match(r"\s*else\b", l):
            merged[-1] = merged[-1] + " " + l.strip()
        else: merged.append(l)
    if comment: merged.insert(0, indent + comment)
    return [m + "\n" for m in merged]


def wrap(line, max_len):
    """second pass: a single statement that is still too long is broken after top-level `, ` / ` || ` / ` && ` (never inside a
    literal or a comment), continuation lines indented by four more columns"""
    if len(line) <= max_len or line.lstrip().startswith("#") or line.rstrip().endswith("\\"): return [line]
    code, comment = split_comment(line)
    if len(code) <= max_len: return [line]
    indent = re.match(r"\s*", code).group(0)
    cands = []                                   # break positions (index after the separator)
    for i, c, depth in scan(code):
        if c == "," and code[i:i + 2] == ", " and depth <= 2: cands.append(i + 2)
        elif c in "|&" and code[i:i + 3] in ("|| ", "&& ") and code[i - 1] == " " and depth <= 1: cands.append(i + 3)
    out, start, cont = [], 0, indent + "    "
    while len(code) - start + (len(cont) if out else 0) > max_len:
        limit = start + max_len - (len(cont) if out else 0)
        best = [p for p in cands if start < p <= limit]
        if not best:
            later = [p for p in cands if p > limit]
            if not later: break
            cut = later[0]
        else: cut = best[-1]
        out.append((cont if out else "") + code[start:cut].rstrip())
        start = cut
    out.append((cont if out else "") + code[start:].rstrip())
    if len(out) == 1: return [line]
    if comment: out.insert(0, indent + comment)
    return out


def main():
    path = sys.argv[1]; min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 180
    src = open(path).read().split("\n")
    out = []
    for l in src:
        for x in relayout(l, min_len):
            out.extend(wrap(x.rstrip("\n"), min_len))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
