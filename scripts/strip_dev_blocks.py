#!/usr/bin/env python
"""Removes the DEVELOPMENT preprocessor blocks (knock-outs GOM_KO_*, workgroup timelines GOM_PHASE_PROF, counters GOM_BLK_STATS / GOM_PAIR_STAT)
from a HIP source, as the preprocessor would with none of those macros defined: scripts/strip_dev_blocks.py IN OUT.
Round 6 housekeeping: the product translation units carry no development switches; csrc/lab/dev_switches.patch puts them back for
scripts/exp_build.py (knock-out ladders, timelines)."""
import re, sys

DEV = re.compile(r"GOM_KO_\w+|GOM_PHASE_PROF\w*|GOM_BLK_STATS")

def dev_condition(line):
    """None: not a development conditional.  True / False: its value with no development macro defined."""
    m = re.match(r"\s*#\s*(ifdef|ifndef|if)\b(.*)", line)
    if not m or not DEV.search(m.group(2).split("//")[0]):
        return None
    kind, cond = m.group(1), m.group(2).split("//")[0].strip()
    if kind == "ifdef":
        return False
    if kind == "ifndef":
        return True
    neg = cond.startswith("!(") and cond.endswith(")")
    return True if neg else False       # `defined(X) && X == n` -> false, `!(defined(X) && ...)` -> true

def strip(lines):
    out, stack = [], []                 # stack entries: ("dev", keep_now) | ("other",)
    for ln in lines:
        d = re.match(r"\s*#\s*(if|ifdef|ifndef|else|elif|endif)\b", ln)
        if d:
            kw = d.group(1)
            if kw in ("if", "ifdef", "ifndef"):
                c = dev_condition(ln)
                if c is None:
                    stack.append(["other"])
                else:
                    stack.append(["dev", c])
                    continue
            elif kw in ("else", "elif"):
                if stack[-1][0] == "dev":
                    assert kw == "else", ln
                    stack[-1][1] = not stack[-1][1]
                    continue
            else:
                top = stack.pop()
                if top[0] == "dev":
                    continue
        if all(e[0] != "dev" or e[1] for e in stack):
            out.append(ln)
    assert not stack
    return out

if __name__ == "__main__":
    src = open(sys.argv[1]).read().split("\n")
    res = [l for l in strip(src) if not re.match(r"\s*GOM_PAIR_STAT\(.*\);\s*(GOM_PAIR_STAT\(.*\);\s*)*$", l)]
    open(sys.argv[2], "w").write("\n".join(res))
