timeout 900 python deep-tracking-control_amd/tools/debug/comp_split_diff.py 2>&1 | tail -60
