"""Run tools/fuzz_parity.py against one build of the library.

usage: fuzz_case_with_lib.py <path to libmirec.so> <arguments of fuzz_parity.py ...>
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movie_recommender_system_amd import _lib  # noqa: E402

_lib.load(sys.argv[1])
sys.argv = [os.path.join(ROOT, "tools", "fuzz_parity.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
