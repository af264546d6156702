#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
timeout -k 10 300 python scripts/r05_tune_ab.py hiv12 default= old=NARROW_UNITS:36,NO_TD_TAIL:1 noheight=NO_HEIGHT_ORDER:1
timeout -k 10 300 python scripts/r05_tune_ab.py hiv2 default= old=NARROW_UNITS:8,NO_TD_TAIL:1 noheight=NO_HEIGHT_ORDER:1
timeout -k 10 300 python scripts/r05_tune_ab.py hiv67 default= old=NARROW_UNITS:8,NO_TD_TAIL:1 noheight=NO_HEIGHT_ORDER:1
timeout -k 10 300 python scripts/r05_tune_ab.py cfg2 default= old=NARROW_UNITS:512,NO_TD_TAIL:1
} 2>&1 | grep -v Warn | tee gpurun_out/r05r_small_forests_ab.txt
