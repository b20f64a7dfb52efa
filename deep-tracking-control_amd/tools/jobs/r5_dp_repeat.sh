for i in 1 2 3; do timeout 900 python -m pytest tests/test_hip_dp.py -x -q -m gpu 2>&1 | tail -40 | cut -c1-250; done
