"""bench.py's stdout carries ONE line, the JSON (the driver reads it); whatever a library prints on stdout while the legs run -- this image's RCCL prints a banner
from C when a communicator is created -- must end up on stderr."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_only_the_line_reaches_stdout():
    code = textwrap.dedent(f'''
        import sys, ctypes
        sys.path.insert(0, {ROOT!r})
        import bench
        t = bench._OnlyTheLineOnStdout()
        print("noise from python")
        ctypes.CDLL(None).puts(b"noise from C")
        t.line('{{"metric": "correlators/s"}}')
        ctypes.CDLL(None).puts(b"late noise from C")
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": "correlators/s"}\n'
    for noise in ("noise from python", "noise from C", "late noise from C"):
        assert noise in r.stderr
