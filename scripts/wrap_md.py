#!/usr/bin/env python3
"""Re-wraps the prose of a Markdown file to <= 160 characters per line.  Table rows, headings, code fences and lines of a fenced block are left alone
(a table row is one line by construction).  Usage: wrap_md.py file.md [...]"""
import re
import sys
import textwrap

W = 160


def wrap_file(path):
    out, fence = [], False
    para = []

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[-*+]|\d+[.)])\s+)?", first)
        ind, bullet = m.group(1), m.group(2) or ""
        text = " ".join([first[len(ind) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
        out.extend(textwrap.wrap(text, W, initial_indent=ind + bullet, subsequent_indent=ind + " " * len(bullet), break_long_words=False, break_on_hyphens=False) or [""])
        para.clear()

    for line in open(path).read().split("\n"):
        s = line.strip()
        if s.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or s.startswith("|") or s.startswith("#") or s == "" or s.startswith("<") or re.match(r"^\s*[-*_]{3,}\s*$", line):
            flush(); out.append(line); continue
        if re.match(r"^\s*((?:[-*+]|\d+[.)])\s+)", line):          # a new list item starts a new paragraph
            flush()
        para.append(line)
    flush()
    open(path, "w").write("\n".join(out))


for p in sys.argv[1:]:
    wrap_file(p)
