# last call of the round: the DP test that flaked, five times on the defaults; then the whole suite + smoke + driver command
T="tests/test_hip_dp.py::test_bucketed_exchange_on_the_side_stream_equals_one_exchange_after_the_join"
for i in 1 2 3 4 5; do echo "default $i: $(timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | tail -1)"; done
bash deep-tracking-control_amd/tools/jobs/r5_suite.sh
