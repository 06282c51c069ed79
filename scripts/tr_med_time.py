"""translate / remove_motion / median timings in one go (development aid).  python scripts/tr_med_time.py"""
import os
import runpy
import sys

here = os.path.dirname(os.path.abspath(__file__))
for s in ("tr_time.py", "median_time.py"):
    runpy.run_path(os.path.join(here, s), run_name="__main__")
