#!/usr/bin/env python3
"""Re-wraps the prose of a Markdown file at WIDTH columns: paragraphs and list items are re-filled, tables, headings, code fences and
HTML stay as they are.  usage: python tools/reflow_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 132
out, para, fence = [], [], False


def flush():
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+|\s*>\s?)", first)
    lead = m.group(1) if m else re.match(r"^\s*", first).group(0)
    body = " ".join([first[len(lead):].strip()] + [ln.strip() for ln in para[1:]])
    sub = " " * len(lead) if not lead.lstrip().startswith(">") else lead
    out.extend(textwrap.wrap(body, W, initial_indent=lead, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
    para.clear()


for line in open(path).read().split("\n"):
    s = line.strip()
    if s.startswith("```"):
        flush(); fence = not fence; out.append(line); continue
    if fence or s.startswith(("|", "#", "<")) or s == "" or re.match(r"^-{3,}$", s):
        flush(); out.append(line); continue
    if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", line) or line.startswith(">"):
        flush()
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
