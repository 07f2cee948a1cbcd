#!/usr/bin/env python3
"""Removes the experiment branches (#ifdef / #ifndef / #if defined(DCRX_EXP_*) ... [#else ...] #endif) from a source file,
keeping what compiles when none of those macros is defined.  usage: strip_exp.py FILE ...  (in place)
The branches themselves live in tools/r05_experiments/experiment_branches.patch, which tools/build_variant.sh applies to a
copy of csrc/ when a variant is built with a -DDCRX_EXP_* macro."""
import re
import sys

HEAD = re.compile(r'^\s*#\s*(ifdef|ifndef|if)\b(.*)$')


def strip(text: str) -> str:
    out, stack = [], []          # stack entries: None (a foreign conditional) or [keep_now, seen_else]
    for line in text.split('\n'):
        m = HEAD.match(line)
        body = line.strip()
        if m:
            kind, rest = m.group(1), m.group(2)
            exp = re.search(r'\bDCRX_EXP_[A-Z0-9_]+', rest)
            ours = bool(exp) and (kind in ('ifdef', 'ifndef') or re.match(r'\s*defined\s*\(\s*DCRX_EXP_[A-Z0-9_]+\s*\)\s*(//.*|/\*.*)?$', rest))
            if ours:
                stack.append([kind == 'ifndef', False])      # the macro is undefined: #ifndef keeps its first branch
                continue
            stack.append(None)
        elif re.match(r'^\s*#\s*else\b', body) and stack and stack[-1] is not None:
            stack[-1][0] = not stack[-1][0]
            continue
        elif re.match(r'^\s*#\s*endif\b', body) and stack:
            top = stack.pop()
            if top is not None:
                continue
        if all(s is None or s[0] for s in stack):
            out.append(line)
    return '\n'.join(out)


for path in sys.argv[1:]:
    src = open(path).read()
    new = strip(src)
    if new != src:
        open(path, 'w').write(new)
        print(f"{path}: {src.count(chr(10)) - new.count(chr(10))} lines of experiment branches removed")
