"""bench.py with a tuning flag set first (A/B of fsg_set_tuning switches): python tools/bench_ab.py FLAGS [bench args]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from fetalsyngen_amd import _lib  # noqa: E402

flags = int(sys.argv[1])
_lib.load().fsg_set_tuning(flags)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
