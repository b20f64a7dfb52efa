# final check of the round: full GPU suite, smoke(), the driver's bench command, then the profile set
O=gpurun_out/r5f
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like.json 2> $O/driver_like.err ) 2>&1 | grep real
cut -c1-260 $O/driver_like.json
bash deep-tracking-control_amd/tools/jobs/r5_refresh_profiles.sh > $O/refresh.log 2>&1
tail -3 $O/refresh.log | cut -c1-200
find gpurun_out -type f -size +4M -delete
