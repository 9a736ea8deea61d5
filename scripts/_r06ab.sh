cd /root/repo
O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_event_order.py tests/test_gpu_corpora.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/summary.txt; tail -3 $O/pytest.log >> $O/summary.txt
for i in 1 2 3; do timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt; done
cat $O/summary.txt
