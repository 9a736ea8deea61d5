cd /root/repo
O=gpurun_out/r06s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_event_order.py tests/test_gpu_enqueue.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/summary.txt; tail -5 $O/pytest.log >> $O/summary.txt
for i in 1 2; do timeout 300 python scripts/bench_nat.py 20 >> $O/nat.jsonl 2>> $O/nat.err; done; cat $O/nat.jsonl >> $O/summary.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/tr -o t -- python /root/repo/scripts/bench_nat.py 4 sherlock > /root/repo/$O/nat_prof.txt 2>&1)
python scripts/step_timeline.py $O/tr "k_pfx_count<" > $O/nat_step_timeline.txt 2>&1; rm -rf $O/tr
tail -22 $O/nat_step_timeline.txt >> $O/summary.txt
