"""tools/pmc_summary.py against an alternative build of the library (development A/B tool): the three PMC passes run
`tools/bench_with_lib.py <lib>` instead of bench.py.   python tools/pmc_with_lib.py moldiff_amd/libmoldiff_hip_x.so out.json [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pmc_summary  # noqa: E402

lib = os.path.abspath(sys.argv[1])
_run = pmc_summary.subprocess.run


def run(cmd, **kw):
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    cmd = cmd[:i] + [os.path.join(ROOT, 'tools', 'bench_with_lib.py'), lib] + cmd[i + 1:]
    return _run(cmd, **kw)


pmc_summary.subprocess.run = run
sys.argv = ['pmc_summary.py'] + sys.argv[2:]
pmc_summary.main()
