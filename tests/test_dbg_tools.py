"""The developer tools that profiles/README.md and DESIGN.md cite (tools/dbg/, tools/*.py) are scripts that run GPU work
when executed, so they cannot be imported here; what CAN rot silently is checked statically: every script compiles, every
`_lib.options(...)` keyword it uses is one the binding accepts, and every attribute it reads from the package's modules
exists (an option or function rename breaks this test, not the next profiling session)."""
import ast
import glob
import importlib
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (tools/gen_golden.py is left out: the utils_* modules it names are the REFERENCE's, imported in the build container only)
SCRIPTS = sorted(glob.glob(os.path.join(REPO, "tools", "dbg", "*.py")) +
                 [f for f in glob.glob(os.path.join(REPO, "tools", "*.py")) if os.path.basename(f) != "gen_golden.py"])
MODULES = ("_lib", "synthetic", "utils_match", "utils_hist", "utils_icp", "utils_helper", "utils_flow", "utils_track",
           "utils_cluster", "utils_check", "frame_pairs", "utils_eval", "utils_icp_pytorch3d", "sharding", "hist")


def test_there_are_tools_to_check():
    assert len(SCRIPTS) >= 20


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(s, REPO) for s in SCRIPTS])
def test_tool_is_in_step_with_the_package(path):
    from icp_flow_amd import _lib
    src = open(path).read()
    tree = ast.parse(src, filename=path)          # compiles
    valid = set(_lib.OPT_FLAGS) | {"search", "arith", "profile", "vote_bins", "icp_history", "icp_scale", "icp_init",
                                   "allow_reflection", "estimate_scale"}
    mods = {m: importlib.import_module("icp_flow_amd." + m) for m in MODULES}
    for node in ast.walk(tree):
        # _lib.options(keyword=...)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "options" and \
                isinstance(node.func.value, ast.Name) and node.func.value.id == "_lib":
            for kw in node.keywords:
                assert kw.arg is None or kw.arg in valid, f"{os.path.basename(path)}: _lib.options({kw.arg}=...) is not an option"
        # module.attribute of the package's modules (only where the name is bound by an import of the package)
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in mods:
            if node.value.id == "_lib" and node.attr == "_L":
                continue
            assert hasattr(mods[node.value.id], node.attr), f"{os.path.basename(path)}: icp_flow_amd.{node.value.id}.{node.attr} does not exist"
    # the debug exports a tool calls are the ones the instrumented builds define (icp.hip / nn.hip, #ifdef ICPFLOW_*)
    csrc = "".join(open(f).read() for f in glob.glob(os.path.join(REPO, "icp_flow_amd", "csrc", "*.hip")))
    for name in set(re.findall(r"_L\.(icpflow_debug_\w+)", src)):
        assert re.search(r"\b" + name + r"\s*\(", csrc), f"{os.path.basename(path)}: {name} is not defined by any debug build"
