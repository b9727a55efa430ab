"""Run bench.py against an alternative build of the library (development A/B tool).
    python tools/bench_with_lib.py moldiff_amd/libmoldiff_hip_et2.so --steps 200 --no-cpu-baseline
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
