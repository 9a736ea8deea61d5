cd /root/repo
O=gpurun_out/r06aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_event_order.py tests/test_gpu_corpora.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/summary.txt; tail -3 $O/pytest.log >> $O/summary.txt
for i in 1 2; do timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt; done
ACGPU_LIB=/root/repo/aho-corasick_amd/lib/exp/libacgpu_pfx_prof.so timeout 300 python scripts/pfx_prof.py >> $O/summary.txt 2>> $O/nat.err
scripts/pmc_traffic.sh $O/nat_sherlock_pmc.json "k_pfx_count<true" 1 "sherlock 1 GiB / words-5000" -- python /root/repo/scripts/bench_nat.py 4 sherlock >> $O/summary.txt 2>&1
cat $O/summary.txt
