#!/usr/bin/env python3
"""Re-flow the prose of a markdown file to a column limit (default 120) without touching what cannot be wrapped: table rows, fenced code,
headings, reference-style link definitions.  Paragraphs and list items are re-flowed with their own indentation (a list item's
continuation lines hang under its text); a continuation line is never allowed to begin with something markdown would read as a new
block (a list marker, `#`, `>`, a table bar, a fence).

    python tools/wrap_md.py DESIGN.md [--width 120] [--check]      --check: exit 1 if a wrappable line is over the limit, change nothing
"""
import re
import sys
import textwrap

LIST = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")
BLOCK_START = re.compile(r"^(\s*)([-*+]\s|\d+[.)]\s|#|>|\||```|~~~)")


def unwrappable(line):
    s = line.lstrip()
    return s.startswith("|") or s.startswith("#") or s.startswith("```") or s.startswith("~~~") or re.match(r"^\[[^\]]+\]:\s", s) is not None


def flow(text, first_indent, rest_indent, width):
    words = text.split()
    lines, cur = [], first_indent
    fresh = True
    for w in words:
        if not fresh and len(cur) + 1 + len(w) > width:
            lines.append(cur)
            cur, fresh = rest_indent, True
        if fresh:
            if lines and BLOCK_START.match(w + " "):
                # this word would open a block at the start of a line: the previous line's last word comes down in front of it
                head, _, last = lines[-1].rstrip().rpartition(" ")
                if head.strip() and not BLOCK_START.match(last + " "):
                    lines[-1] = head
                    cur += last + " " + w
                else:
                    lines[-1] += " " + w        # (nothing to bring down: stay on the previous line, a word over the limit at worst)
                    continue
                fresh = False
                continue
            cur += w
            fresh = False
        else:
            cur += " " + w
    if not fresh:
        lines.append(cur)
    return lines


def wrap(src, width=120):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = LIST.match(first)
        if m:
            first_indent = first[:m.end()]
            rest_indent = " " * len(first_indent)
            body = first[m.end():] + " " + " ".join(l.strip() for l in para[1:])
        else:
            first_indent = rest_indent = re.match(r"^\s*", first).group(0)
            body = " ".join(l.strip() for l in para)
        out.extend(flow(body, first_indent, rest_indent, width))
        del para[:]

    for line in src.splitlines():
        if line.lstrip().startswith("```") or line.lstrip().startswith("~~~"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or unwrappable(line):
            flush()
            out.append(line)
            continue
        if not line.strip():
            flush()
            out.append("")
            continue
        if LIST.match(line):
            flush()
        para.append(line.rstrip())
    flush()
    return "\n".join(out) + "\n"


def too_long(src, width=120):
    bad, fence = [], False
    for k, line in enumerate(src.splitlines(), 1):
        if line.lstrip().startswith("```") or line.lstrip().startswith("~~~"):
            fence = not fence
            continue
        if not fence and not unwrappable(line) and len(line) > width:
            bad.append((k, len(line)))
    return bad


def main(argv):
    width = int(argv[argv.index("--width") + 1]) if "--width" in argv else 120
    files = [a for a in argv if not a.startswith("--") and not a.isdigit()]
    rc = 0
    for f in files:
        src = open(f).read()
        if "--check" in argv:
            bad = too_long(src, width)
            for k, n in bad[:20]:
                print("%s:%d: %d columns" % (f, k, n))
            rc |= 1 if bad else 0
        else:
            open(f, "w").write(wrap(src, width))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
