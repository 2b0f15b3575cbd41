#!/usr/bin/env python3
"""tools/sync_integration_md.py [--check] -- INTEGRATION.md quotes the binding headers under integration/ verbatim: the text between
`<!-- BEGIN integration/X.h -->` and `<!-- END integration/X.h -->` is regenerated from the file (as a ```c block).  --check: exit 1 when
the document is out of date (tests/test_integration_doc.py runs it)."""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "INTEGRATION.md")
doc = open(path).read()


def sub(m):
    name = m.group(1)
    body = open(os.path.join(root, name)).read().rstrip("\n")
    return f"<!-- BEGIN {name} -->\n```c\n{body}\n```\n<!-- END {name} -->"


new = re.sub(r"<!-- BEGIN (integration/[a-z_]+\.h) -->.*?<!-- END \1 -->", sub, doc, flags=re.S)
if "--check" in sys.argv:
    sys.exit(0 if new == doc else 1)
open(path, "w").write(new)
