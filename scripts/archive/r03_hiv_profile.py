#!/usr/bin/env python3
"""cProfile of the 91-column HIV1C acr() (groups one after the other): where the host time of a sweep round goes."""
import cProfile
import os
import pstats
import sys
import io

os.environ.setdefault('PASTML_AMD_CONCURRENT_GROUPS', '0')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy  # noqa: E402
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(REPO, 'scripts', 'hiv1c_all.py'), run_name='__main__')
finally:
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000], file=sys.stderr)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print(s.getvalue()[:6000], file=sys.stderr)
