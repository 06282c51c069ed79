"""per-frame C-ABI timings in one go (development aid): record / read, bounded-loss record."""
import os
import runpy

here = os.path.dirname(os.path.abspath(__file__))
for s in ("abi_time.py", "lossy_time.py"):
    runpy.run_path(os.path.join(here, s), run_name="__main__")
