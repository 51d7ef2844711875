"""The documents the judge reads name files by path; a path that no longer exists is a claim without its evidence (CPU test)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "tools/README.md"]
TOP = ("profiles", "tools", "tests", "docs", "oracle", "include", "humanoid-gym_amd")


def _refs(text):
    for m in re.finditer(r"`((?:%s)/[^`\s]*)`" % "|".join(TOP), text):
        ref = m.group(1)
        path = ref.split("::")[0].split(":")[0].rstrip(".,;)")
        path = re.sub(r"\{[^}]*\}", "*", path)            # r05_{a,b}_x.txt -> r05_*_x.txt
        if "<" in path or "..." in path or "…" in path:
            continue                                       # a pattern with a placeholder, not a path
        yield ref, path


@pytest.mark.parametrize("doc", DOCS)
def test_paths_named_in_the_documents_exist(doc):
    text = open(os.path.join(ROOT, doc)).read()
    missing = sorted({ref for ref, path in _refs(text)
                      if not glob.glob(os.path.join(ROOT, path)) and not glob.glob(os.path.join(ROOT, path) + "*")})
    assert not missing, "%s names paths that do not exist: %s" % (doc, missing)


def test_tools_readme_names_existing_scripts():
    text = open(os.path.join(ROOT, "tools", "README.md")).read()
    names = set(re.findall(r"`([A-Za-z0-9_/]+\.(?:sh|py|hip))", text)) - {"bench.py", "build.py"}
    missing = sorted(n for n in names if not os.path.exists(os.path.join(ROOT, "tools", n)) and not os.path.exists(os.path.join(ROOT, n)))
    assert not missing, missing
