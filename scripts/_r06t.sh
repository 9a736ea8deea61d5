cd /root/repo
O=gpurun_out/r06t; mkdir -p $O
E=/root/repo/aho-corasick_amd/lib/exp/libacgpu_pfx_12_4x8.so
for i in 1 2; do
  echo "default" >> $O/summary.txt; timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | tee -a $O/nat.jsonl >> $O/summary.txt
  echo "temporal" >> $O/summary.txt; ACGPU_LIB=$E timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | tee -a $O/nat_t.jsonl >> $O/summary.txt
done
ACGPU_LIB=$E scripts/pmc_traffic.sh $O/nat_sherlock_temporal_pmc.json "k_pfx_count<true" 1 "sherlock 1 GiB / words-5000, temporal row loads (PFX_EXP=8)" -- python /root/repo/scripts/bench_nat.py 4 sherlock >> $O/summary.txt 2>&1
ACGPU_LIB=$E scripts/pmc_traffic.sh $O/nat_enhuge_temporal_pmc.json "k_pfx_count<true" 1 "en-huge 1 GiB / words-15000, temporal row loads (PFX_EXP=8)" -- python /root/repo/scripts/bench_nat.py 4 en-huge >> $O/summary.txt 2>&1
