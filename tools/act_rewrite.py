"""One-off source rewrite used for the act_t refactor: turn float4 cast accesses into the overloaded ld4/st4 helpers.

  reinterpret_cast<const float4*>(E)[Q]      -> ld4q(E, Q)
  *reinterpret_cast<const float4*>(E)        -> ld4(E)
  reinterpret_cast<float4*>(E)[Q] = V;       -> st4q(E, Q, V);
  *reinterpret_cast<float4*>(E) = V;         -> st4(E, V);
Only simple statement forms are rewritten; everything else is reported for manual handling."""
import re
import sys


def match_paren(s, i):
    assert s[i] == "("
    d = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            d += 1
        elif s[j] == ")":
            d -= 1
            if d == 0:
                return j
    raise ValueError


def match_bracket(s, i):
    assert s[i] == "["
    d = 0
    for j in range(i, len(s)):
        if s[j] == "[":
            d += 1
        elif s[j] == "]":
            d -= 1
            if d == 0:
                return j
    raise ValueError


def rewrite(s):
    out, i, left = [], 0, []
    pat = re.compile(r"(\*?)reinterpret_cast<(const )?float4\*>\(")
    while True:
        m = pat.search(s, i)
        if not m:
            out.append(s[i:])
            break
        out.append(s[i:m.start()])
        star, const = m.group(1), m.group(2)
        e0 = m.end() - 1
        e1 = match_paren(s, e0)
        expr = s[e0 + 1:e1]
        rest = e1 + 1
        q = None
        if not star and rest < len(s) and s[rest] == "[":
            b1 = match_bracket(s, rest)
            q = s[rest + 1:b1]
            rest = b1 + 1
        if not star and q is None:
            out.append(s[m.start():rest])  # a float4* value (pointer variable): leave
            left.append(s[m.start():rest])
            i = rest
            continue
        if const:
            out.append(f"ld4q({expr}, {q})" if q is not None else f"ld4({expr})")
            i = rest
            continue
        # store or load through a non-const cast
        m2 = re.match(r"\s*=\s*(?!=)", s[rest:])
        if m2:
            semi = s.index(";", rest)
            val = s[rest + m2.end():semi]
            out.append(f"st4q({expr}, {q}, {val})" if q is not None else f"st4({expr}, {val})")
            i = semi
        else:
            out.append(f"ld4q({expr}, {q})" if q is not None else f"ld4({expr})")
            i = rest
    return "".join(out), left


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new, left = rewrite(src)
        open(path, "w").write(new)
        print(path, "left:", left)
