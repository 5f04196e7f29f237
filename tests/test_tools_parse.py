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
