cd /root/repo
O=gpurun_out/r06x; mkdir -p $O
E=/root/repo/aho-corasick_amd/lib/exp
for X in 1 16 2; do
  echo "EXP=$X" >> $O/summary.txt
  ACGPU_LIB=$E/libacgpu_pfx_exp$X.so timeout 300 python scripts/bench_nat.py 10 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt
  ACGPU_LIB=$E/libacgpu_pfx_exp$X.so scripts/pmc_traffic.sh $O/nat_sherlock_exp${X}_pmc.json "k_pfx_count<true" 1 "sherlock, PFX_EXP=$X" -- python /root/repo/scripts/bench_nat.py 4 sherlock >> $O/summary.txt 2>&1
  ACGPU_LIB=$E/libacgpu_pfx_exp$X.so scripts/pmc_traffic.sh $O/nat_enhuge_exp${X}_pmc.json "k_pfx_count<true" 1 "en-huge, PFX_EXP=$X" -- python /root/repo/scripts/bench_nat.py 4 en-huge >> $O/summary.txt 2>&1
done
cat $O/summary.txt
