"""The helper scripts under tools/ are what the profiles were made with: each must at least parse (python) / pass `bash -n` (shell), and the
ones that document a usage must say it when asked (`-h`).  Nothing here needs a GPU."""
import glob
import os
import py_compile
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))), ids=os.path.basename)
def test_python_tool_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))), ids=os.path.basename)
def test_shell_tool_parses(path):
    r = subprocess.run(["bash", "-n", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("tool", ["sweep.py", "mailbox_rate.py", "mailbox_soak.py"])
def test_tool_prints_its_usage(tool):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "usage" in r.stdout.lower()


def test_markdown_wrapper_keeps_what_cannot_be_wrapped():
    """tools/wrap_md.py (what keeps DESIGN.md's prose within 120 columns): paragraphs and list items are re-flowed with hanging indents,
    tables / headings / fenced code are left alone, no continuation line starts like a new block, and wrapping is idempotent"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wrap_md", os.path.join(ROOT, "tools", "wrap_md.py"))
    wm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wm)
    long_cell = "| a | " + "x" * 200 + " |"
    src = ("# Heading " + "h" * 150 + "\n\n" + "word " * 60 + "\n\n- item " + "alpha " * 40 + "\n  continuation " + "beta " * 30 + "\n"
           "1. first " + "gamma " * 40 + "\n\n" + long_cell + "\n|---|---|\n\n```\n" + "c" * 200 + "\n```\n\nsum a " + "+ b " * 60 + "\n")
    out = wm.wrap(src, 80)
    lines = out.splitlines()
    assert lines[0].startswith("# Heading") and len(lines[0]) > 150 and long_cell in lines and "c" * 200 in lines
    assert wm.too_long(out, 84) == []                        # (a word that would open a block stays on the previous line: a few columns over at most)
    body = [l for l in lines if l.startswith("  ") and "alpha" in l or "beta" in l]
    assert body and all(l.startswith("  ") for l in body[1:])                           # hanging indent under the list item's text
    assert not any(l.lstrip().startswith("+ ") for l in lines)                          # "+ b" never opens a bullet
    assert wm.wrap(out, 80) == out
