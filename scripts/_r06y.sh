cd /root/repo
O=gpurun_out/r06y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_event_order.py tests/test_gpu_corpora.py tests/test_gpu_parity.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/summary.txt; tail -5 $O/pytest.log >> $O/summary.txt
for i in 1 2; do timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt; done
scripts/pmc_traffic.sh $O/nat_sherlock_pmc.json "k_pfx_count<true" 1 "sherlock 1 GiB / words-5000" -- python /root/repo/scripts/bench_nat.py 4 sherlock >> $O/summary.txt 2>&1
scripts/pmc_traffic.sh $O/nat_enhuge_pmc.json "k_pfx_count<true" 1 "en-huge 1 GiB / words-15000" -- python /root/repo/scripts/bench_nat.py 4 en-huge >> $O/summary.txt 2>&1
cat $O/summary.txt
