# which switch does the intermittent mismatch of test_bucketed_exchange_on_the_side_stream_equals_one_exchange_after_the_join follow?
T="tests/test_hip_dp.py::test_bucketed_exchange_on_the_side_stream_equals_one_exchange_after_the_join"
for i in 1 2 3 4 5; do echo "default $i: $(timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | tail -1)"; done
for i in 1 2 3 4 5; do echo "DTC_HEADS_UNROLL=0 $i: $(DTC_HEADS_UNROLL=0 timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | tail -1)"; done
