"""INTEGRATION.md quotes the binding headers under integration/ verbatim (the headers are what `make ref` compiles against the reference's
structures and what the -m gpu binding tests execute): the document must not drift from them."""
import os
import subprocess
import sys

import svt_testlib as T


def test_integration_md_quotes_the_compiled_headers():
    assert subprocess.call([sys.executable, os.path.join(T.ROOT, "tools", "sync_integration_md.py"), "--check"]) == 0, "run tools/sync_integration_md.py"
    doc = open(os.path.join(T.ROOT, "INTEGRATION.md")).read()
    for name in ("me_process_binding.h", "coding_loop_binding.h", "loop_filter_binding.h"):
        assert f"<!-- BEGIN integration/{name} -->" in doc
    assert "not\npart of this repository's build" not in doc and "not part of this repository's build" not in doc
