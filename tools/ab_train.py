"""A/B of two builds of the library on ONE box (box-to-box variance is larger than most kernel tweaks):
build the other version to acav100m_amd/<name>.so and run   ACAV_LIB=<name>.so python tools/ab_train.py [b ...]
next to a plain run.  Only this tool redirects the library path; the product always loads libacav_hip.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acav100m_amd import _lib

if os.environ.get("ACAV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), os.environ["ACAV_LIB"])
sys.argv = ["bench_train_b.py"] + (sys.argv[1:] or ["32"])
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train_b.py")).read())
