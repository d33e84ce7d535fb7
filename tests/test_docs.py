"""DESIGN.md / README.md / INTEGRATION.md cite files and tests as evidence: every `profiles/...`, `tests/...::test`, `tools/...`, `scannet_amd/...`, `oracle/...`,
`include/...` path they name in backticks must exist (wildcards must match something; a test name may be a prefix written with a trailing `_` or `*`)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILT = ("bin/", "oracle/_ref/", "scannet_amd/libscanfuse.so", "scannet_amd/_build", "oracle/liboracle.so")     # made by build(), not tracked


def test_cited_paths_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for tok in set(re.findall(r"`([^`\n]+)`", text)):
            for m in re.finditer(r"((?:profiles|tests|tools|scannet_amd|oracle|include|conformance|bin)/[A-Za-z0-9_./*+-]+)(::[A-Za-z0-9_]+)?", tok):
                path, node = m.group(1).rstrip(".,"), m.group(2)
                if path.startswith(BUILT) or path.endswith("/"):
                    continue
                full = os.path.join(ROOT, path)
                if not (glob.glob(full) if "*" in path else os.path.exists(full)):
                    missing.append((doc, path))
                elif node and path.endswith(".py") and os.path.isfile(full):
                    name = node[2:]
                    if not re.search(r"def %s" % re.escape(name), open(full).read()):        # a prefix of a test's name is accepted
                        missing.append((doc, path + node))
    assert sorted(set(missing)) == []
